"""CPU tests of the estimation-quality metrics: OSPA of python/ospa.py:220-274
against an independent solution of the assignment problem (scipy), and the per-log evaluation of
python/batch_analyze.py:16-37."""
import ctypes as C

import numpy as np
from scipy.optimize import linear_sum_assignment

from parity_utils import pkg


def ospa(X, Y, p=1.0, c=5.0):
    L = pkg()._lib.lib()
    X = np.ascontiguousarray(X, np.float32).reshape(-1, 2)
    Y = np.ascontiguousarray(Y, np.float32).reshape(-1, 2)
    out = np.zeros(3)
    rc = L.phd_ospa(X.ctypes.data_as(C.c_void_p), len(X), Y.ctypes.data_as(C.c_void_p), len(Y), p, c,
                    out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    return out


def ospa_ref(X, Y, p, c):
    X = np.asarray(X, np.float32).reshape(-1, 2).astype(np.float64)
    Y = np.asarray(Y, np.float32).reshape(-1, 2).astype(np.float64)
    if len(X) == 0 and len(Y) == 0:
        return np.zeros(3)
    if len(X) == 0 or len(Y) == 0:
        return np.array([c, 0, c])
    if len(X) > len(Y):
        X, Y = Y, X
    m, n = len(X), len(Y)
    d = np.minimum(np.linalg.norm(X[:, None, :] - Y[None, :, :], axis=2), c)
    r, k = linear_sum_assignment(d)              # the reference assigns on the cut-off distances (p = 1 cost)
    tot = (d[r, k] ** p).sum()
    return np.array([((tot + (n - m) * c ** p) / n) ** (1 / p), (tot / n) ** (1 / p), (c ** p * (n - m) / n) ** (1 / p)])


def test_ospa_against_scipy():
    rng = np.random.default_rng(0)
    for m, n, p, c in [(5, 5, 1, 5), (3, 9, 1, 5), (12, 4, 1, 5), (20, 31, 2, 3), (1, 1, 1, 5), (40, 40, 1, 1.5)]:
        X = rng.uniform(-10, 10, (m, 2))
        Y = np.concatenate([X[:min(m, n)] + rng.normal(0, 0.3, (min(m, n), 2)), rng.uniform(-10, 10, (max(n - m, 0), 2))])[:n]
        assert np.allclose(ospa(X, Y, p, c), ospa_ref(X, Y, p, c), rtol=1e-9, atol=1e-12), (m, n, p, c)


def test_ospa_edge_cases():
    assert np.array_equal(ospa(np.zeros((0, 2)), np.zeros((0, 2))), [0, 0, 0])       # ospa.py:224-225
    assert np.array_equal(ospa(np.zeros((0, 2)), np.ones((3, 2))), [5, 0, 5])         # :226-227
    assert np.array_equal(ospa(np.ones((3, 2)), np.zeros((0, 2))), [5, 0, 5])
    # identical sets: zero; one extra point: pure cardinality error c/n
    X = np.array([[0, 0], [3, 4.0]])
    assert np.allclose(ospa(X, X), 0)
    o = ospa(X, np.concatenate([X, [[50, 50]]]))
    assert np.allclose(o, [5 / 3, 0, 5 / 3])
    # far apart points saturate at the cut-off
    assert np.allclose(ospa([[0, 0]], [[100, 0]])[0], 5)


def test_evaluate_state_log(tmp_path):
    P = pkg()
    L = P._lib.lib()
    truth = np.array([[1, 2], [5, 5], [-3, 4]], np.float32)
    e = np.zeros(1, P.POSE); e["px"], e["py"] = 0.3, -0.4
    g = np.zeros(5, P.GAUSSIAN)
    g["weight"] = [0.9, 0.05, 1.1, 0.02, 0.95]          # sum 3.02 -> the 3 heaviest features are the estimate
    g["mean"] = [[1.1, 2.0], [20, 20], [5.0, 5.2], [-20, 3], [-3.1, 4.1]]
    g["cov"] = 0.01
    lw = np.log(np.array([0.5, 0.25, 0.25], np.float32))
    poses = np.zeros(3, P.POSE)
    P.write_state_log(str(tmp_path), 0, e, g, lw, poses, max_cardinality=2)
    out = np.zeros(5)
    tp = np.array([0.0, 0.0], np.float32)
    rc = L.phd_evaluate_state_log(str(tmp_path / "state_estimate00000.log").encode(), tp.ctypes.data_as(C.c_void_p),
                                  truth.ctypes.data_as(C.c_void_p), 3, 1.0, 5.0, out.ctypes.data_as(C.c_void_p))
    assert rc == 0
    assert abs(out[0] - 0.5) < 1e-6                                                   # pose error
    est = g["mean"][[2, 4, 0]]
    assert np.allclose(out[1:4], ospa_ref(truth, est, 1.0, 5.0), atol=1e-6)
    assert abs(out[4] - 1 / (0.25 + 0.0625 + 0.0625)) < 1e-5                          # nEff = 1/sum w^2
