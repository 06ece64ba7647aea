"""The C++ host side on hardware: the run_synth-style driver (cuda-phdslam_amd/bin/phdslam, C-ABI with
device-resident state) and the phdfilter.h-compatible adapter (libphdfilter_compat.so) must produce
what the same sequence of C-ABI calls produces from Python, on a small synthetic data directory in the
reference's file formats (config.cfg / measurements.txt / controls.txt -> state_estimate%05d.log)."""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "cuda-phdslam_amd")


def write_dataset(d, n_steps=6, n_particles=48, seed=5, cphd=False):
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    w = S.make_workload(1, 20, 12, seed=seed, n_meas_sets=n_steps)
    with open(os.path.join(d, "measurements.txt"), "w") as f:
        f.write("% range bearing pairs, one step per line\n")
        for k in range(n_steps):
            z = w["z"][k]
            f.write(" ".join("%.6f %.6f" % (z["range"][i], z["bearing"][i]) for i in range(len(z))) + " \n")
    with open(os.path.join(d, "controls.txt"), "w") as f:
        f.write("% velocity\tsteering angle\n")
        for k in range(n_steps):
            f.write("%.5f %.6f\n" % (2.0 + 0.1 * k, 0.05 - 0.01 * k))
    cfg = open(os.path.join(ROOT, "tests", "golden", "config_sample.cfg")).read()
    cfg = cfg.replace("n_particles = 200", "n_particles = %d" % n_particles)
    cfg = cfg.replace("resample_threshold = 0.5", "resample_threshold = 0.97")  # make the nEff trigger fire in a short run
    cfg = cfg.replace("data_directory = /data/synth_bowtie/", "data_directory = %s/" % d)
    if cphd:
        assert "filter_type = 0" in cfg and "max_cardinality = 255" in cfg
        cfg = cfg.replace("filter_type = 0", "filter_type = 1").replace("max_cardinality = 255", "max_cardinality = 63")
    with open(os.path.join(d, "config.cfg"), "w") as f:
        f.write(cfg)
    return os.path.join(d, "config.cfg")


def parse_log(path):
    lines = open(path).read().split("\n")
    pose = np.array(lines[0].split(), float)
    gmap = np.array(lines[1].split(), float).reshape(-1, 7)
    lw = np.array(lines[2].split(), float)
    poses = np.array(lines[3].split(), float).reshape(-1, 6)
    return pose, gmap, lw, poses


def test_driver_matches_python_host(tmp_path):
    P = importlib.import_module("cuda-phdslam_amd")
    d = str(tmp_path)
    cfg_path = write_dataset(d)
    out_c = os.path.join(d, "out_c"); os.makedirs(out_c)
    r = subprocess.run([os.path.join(PKG, "bin", "phdslam"), cfg_path, "synth", "--out", out_c, "--seed", "9",
                        "--capacity", "256"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    # the same run from Python: same RNG (the adapter's randn/randu01), same C-ABI calls
    rng = C.CDLL(os.path.join(PKG, "libphdfilter_compat.so"))
    rng.randn.restype = C.c_double
    rng.randu01.restype = C.c_double
    rng.phd_compat_seed_rng(C.c_uint64(9))
    cfg, ddir, _ = P.load_config(cfg_path)
    Z = P.load_measurements(os.path.join(ddir, "measurements.txt"))
    U = P.load_controls(os.path.join(ddir, "controls.txt"))
    N = cfg.n_particles
    resampled = 0
    with P.PhdFilter(cfg, n_particles=N, map_capacity=256, max_measurements=max(len(z) for z in Z)) as f:
        for n in range(len(Z)):
            if n > 0:
                noise = np.zeros((N, 2), np.float32)
                for i in range(N):
                    noise[i, 0] = cfg.stdAlpha * rng.randn()
                    noise[i, 1] = cfg.stdEncoder * rng.randn()
                f.predict((U["v_encoder"][n - 1], U["alpha"][n - 1]), noise)
            f.update(Z[n])
            e = f.expected_pose()
            m, who = f.map_estimate()
            poses, lw = f.get_particles()
            pose_c, map_c, lw_c, poses_c = parse_log(os.path.join(out_c, "state_estimate%05d.log" % n))
            # the log has 6 significant digits (default operator<< formatting)
            assert np.allclose(pose_c[:3], [e["px"], e["py"], e["ptheta"]], rtol=2e-5, atol=1e-6), n
            assert len(map_c) == len(m), n
            assert np.allclose(map_c[:, 0], m["weight"], rtol=2e-5) and np.allclose(map_c[:, 1:3], m["mean"], rtol=2e-5, atol=1e-6)
            assert np.allclose(map_c[:, 3:], m["cov"], rtol=2e-5, atol=1e-9)
            assert np.allclose(lw_c, lw, rtol=2e-5) and np.allclose(poses_c[:, :3], np.stack([poses["px"], poses["py"], poses["ptheta"]], 1), rtol=2e-5, atol=1e-6)
            did, _ = f.resample_if_needed(rng.randu01(), had_measurements=len(Z[n]) > 0)
            resampled += int(did)
    assert os.path.exists(os.path.join(out_c, "loopTime.log"))
    assert resampled >= 1  # the trigger fired at least once, so the resampled state was compared too


def test_phdfilter_h_adapter(tmp_path):
    """phdPredict / phdUpdateSynth / resampleParticles / recoverSlamState of the adapter, called from a
    small C++ program written against the reference-style header, equal the Python host's results."""
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    src = tmp_path / "adapter_test.cpp"
    src.write_text(r'''
#include <cstdio>
#include "phdfilter_compat.h"
int main(int argc, char** argv) {
    char ddir[1024]; int32_t ns;
    if (phd_config_load(argv[1], &config, ddir, sizeof(ddir), &ns)) { printf("cfg: %s\n", phd_last_error()); return 1; }
    phd_compat_seed_rng(3);
    setDeviceConfig(config);
    initRandomNumberGenerators();
    const int N = config.n_particles;
    SynthSLAM particles(N);
    for (int i = 0; i < N; ++i) {
        particles.states[i] = ConstantVelocityState{0.01f * i, -0.02f * i, 0.001f * i, 0, 0, 0};
        particles.weights[i] = -2.7725887f; // -log(16)
        for (int g = 0; g < 5; ++g) {
            Gaussian2D f; f.cov[0] = 0.04f; f.cov[1] = f.cov[2] = 0.001f * g; f.cov[3] = 0.09f;
            f.mean[0] = 2.0f + 1.5f * g; f.mean[1] = -3.0f + 1.1f * g; f.weight = 0.5f + 0.1f * g;
            particles.maps_static[i].push_back(f);
        }
    }
    AckermanControl u; u.alpha = 0.05f; u.v_encoder = 2.0f;
    phdPredict(particles, u);
    measurementSet Z;
    for (int m = 0; m < 6; ++m) { RangeBearingMeasurement z; z.range = 3.0f + m; z.bearing = -1.0f + 0.4f * m; z.label = 0; Z.push_back(z); }
    SynthSLAM pre = phdUpdateSynth(particles, Z);
    ConstantVelocityState e; std::vector<REAL> cn;
    recoverSlamState(particles, e, cn);
    printf("pre %zu %.9g\n", pre.maps_static[0].size(), pre.weights[0]);
    printf("pose %.9g %.9g %.9g\n", e.px, e.py, e.ptheta);
    printf("neff %.9g\n", computeNeff(particles));
    for (int i = 0; i < N; ++i) printf("p %d %.9g %zu %.9g\n", i, particles.weights[i], particles.maps_static[i].size(), particles.maps_static[i].size() ? particles.maps_static[i][0].weight : 0.f);
    SynthSLAM rs = resampleParticles(particles, N);
    for (int i = 0; i < N; ++i) printf("r %d %d\n", i, rs.resample_idx[i]);
    return 0;
}
''')
    exe = tmp_path / "adapter_test"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "csrc", "compat"), "-Wno-varargs", str(src), "-o", str(exe),
                           "-L" + PKG, "-lphdfilter_compat", "-lphdslam", "-Wl,-rpath," + PKG])
    d = str(tmp_path)
    cfg_path = write_dataset(d, n_particles=16)
    r = subprocess.run([str(exe), cfg_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.split("\n")
    # the same sequence from Python
    rng = C.CDLL(os.path.join(PKG, "libphdfilter_compat.so"))
    rng.randn.restype = C.c_double
    rng.randu01.restype = C.c_double
    rng.phd_compat_seed_rng(C.c_uint64(3))
    cfg, _, _ = P.load_config(cfg_path)
    N = 16
    poses = np.zeros(N, P.POSE)
    poses["px"] = np.float32(0.01) * np.arange(N, dtype=np.float32)
    poses["py"] = np.float32(-0.02) * np.arange(N, dtype=np.float32)
    poses["ptheta"] = np.float32(0.001) * np.arange(N, dtype=np.float32)
    g = np.zeros(5, P.GAUSSIAN)
    for k in range(5):
        g[k]["cov"] = [0.04, np.float32(0.001) * np.float32(k), np.float32(0.001) * np.float32(k), 0.09]
        g[k]["mean"] = [2.0 + np.float32(1.5) * k, -3.0 + np.float32(1.1) * k]
        g[k]["weight"] = np.float32(0.5) + np.float32(0.1) * np.float32(k)
    z = np.zeros(6, P.MEAS)
    z["range"] = 3.0 + np.arange(6)
    z["bearing"] = np.float32(-1.0) + np.float32(0.4) * np.arange(6, dtype=np.float32)
    with P.PhdFilter(cfg, n_particles=N, map_capacity=256) as f:
        f.set_particles(poses, np.full(N, -2.7725887, np.float32))
        f.set_maps([g] * N)
        noise = np.zeros((N, 2), np.float32)
        for i in range(N):
            noise[i, 0] = cfg.stdAlpha * rng.randn()
            noise[i, 1] = cfg.stdEncoder * rng.randn()
        f.predict((2.0, 0.05), noise)
        f.update(z)
        e = f.expected_pose()
        _, lw = f.get_particles()
        maps = f.get_maps()
        ne = f.neff()
        idx = f.resample(rng.randu01())
    assert out[0].split()[1] == "5"
    assert np.allclose([float(v) for v in out[1].split()[1:]], [e["px"], e["py"], e["ptheta"]], rtol=1e-6, atol=1e-7)
    assert abs(float(out[2].split()[1]) - ne) < 1e-6
    for i in range(N):
        t = out[3 + i].split()
        assert abs(float(t[2]) - lw[i]) <= 1e-6 * abs(lw[i]) and int(t[3]) == len(maps[i])
        assert abs(float(t[4]) - maps[i]["weight"][0]) <= 1e-6 * maps[i]["weight"][0]
        assert int(out[3 + N + i].split()[2]) == idx[i]


def test_phdfilter_h_adapter_particle_shotgun(tmp_path):
    """n_predict_particles = 2 through the adapter (the old adapter read past its noise vector and wrote n k poses
    into a vector of n): phdPredict grows the caller's SynthSLAM like the reference (src/phdfilter.cu:1182-1234 — maps,
    cardinalities and resample indices k times, weights - log k, n_particles = n k), phdUpdateSynth works on the grown
    set, resampleParticles(particles, config.n_particles) brings it back (src/main.cpp:1289).  Same numbers as the
    C-ABI driven from Python."""
    P = importlib.import_module("cuda-phdslam_amd")
    src = tmp_path / "adapter_shotgun.cpp"
    src.write_text(r'''
#include <cstdio>
#include "phdfilter_compat.h"
int main(int argc, char** argv) {
    char ddir[1024]; int32_t ns;
    if (phd_config_load(argv[1], &config, ddir, sizeof(ddir), &ns)) { printf("cfg: %s\n", phd_last_error()); return 1; }
    config.nPredictParticles = 2;
    phd_compat_seed_rng(4);
    setDeviceConfig(config);
    const int N = config.n_particles;
    SynthSLAM particles(N);
    for (int i = 0; i < N; ++i) {
        particles.states[i] = ConstantVelocityState{0.01f * i, -0.02f * i, 0.001f * i, 0, 0, 0};
        particles.weights[i] = -2.7725887f - 0.01f * i;
        particles.resample_idx[i] = i;
        for (int g = 0; g < 5; ++g) {
            Gaussian2D f; f.cov[0] = 0.04f; f.cov[1] = f.cov[2] = 0.001f * g; f.cov[3] = 0.09f;
            f.mean[0] = 2.0f + 1.5f * g + 0.01f * i; f.mean[1] = -3.0f + 1.1f * g; f.weight = 0.5f + 0.1f * g;
            particles.maps_static[i].push_back(f);
        }
    }
    AckermanControl u; u.alpha = 0.05f; u.v_encoder = 2.0f;
    phdPredict(particles, u);
    printf("grown %d %zu %zu %zu %zu\n", particles.n_particles, particles.states.size(), particles.weights.size(),
           particles.maps_static.size(), particles.resample_idx.size());
    for (int i = 0; i < particles.n_particles; ++i)
        printf("g %d %.9g %.9g %.9g %d\n", i, particles.weights[i], particles.states[i].px, particles.maps_static[i][0].mean[0], particles.resample_idx[i]);
    measurementSet Z;
    for (int m = 0; m < 6; ++m) { RangeBearingMeasurement z; z.range = 3.0f + m; z.bearing = -1.0f + 0.4f * m; z.label = 0; Z.push_back(z); }
    phdUpdateSynth(particles, Z);
    for (int i = 0; i < particles.n_particles; ++i) printf("p %d %.9g %zu\n", i, particles.weights[i], particles.maps_static[i].size());
    SynthSLAM rs = resampleParticles(particles, config.n_particles);
    printf("back %d\n", rs.n_particles);
    for (int i = 0; i < rs.n_particles; ++i) printf("r %d %d %zu\n", i, rs.resample_idx[i], rs.maps_static[i].size());
    return 0;
}
''')
    exe = tmp_path / "adapter_shotgun"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(PKG, "csrc", "compat"), "-Wno-varargs", str(src), "-o", str(exe),
                           "-L" + PKG, "-lphdfilter_compat", "-lphdslam", "-Wl,-rpath," + PKG])
    d = str(tmp_path)
    cfg_path = write_dataset(d, n_particles=16)
    r = subprocess.run([str(exe), cfg_path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    out = r.stdout.split("\n")
    rng = C.CDLL(os.path.join(PKG, "libphdfilter_compat.so"))
    rng.randn.restype = C.c_double
    rng.randu01.restype = C.c_double
    rng.phd_compat_seed_rng(C.c_uint64(4))
    cfg, _, _ = P.load_config(cfg_path)
    cfg.nPredictParticles = 2
    N, k = 16, 2
    poses = np.zeros(N, P.POSE)
    poses["px"] = np.float32(0.01) * np.arange(N, dtype=np.float32)
    poses["py"] = np.float32(-0.02) * np.arange(N, dtype=np.float32)
    poses["ptheta"] = np.float32(0.001) * np.arange(N, dtype=np.float32)
    maps = []
    for i in range(N):
        g = np.zeros(5, P.GAUSSIAN)
        for j in range(5):
            g[j]["cov"] = [0.04, np.float32(0.001) * np.float32(j), np.float32(0.001) * np.float32(j), 0.09]
            g[j]["mean"] = [np.float32(2.0) + np.float32(1.5) * np.float32(j) + np.float32(0.01) * np.float32(i),
                            np.float32(-3.0) + np.float32(1.1) * np.float32(j)]
            g[j]["weight"] = np.float32(0.5) + np.float32(0.1) * np.float32(j)
        maps.append(g)
    z = np.zeros(6, P.MEAS)
    z["range"] = 3.0 + np.arange(6)
    z["bearing"] = np.float32(-1.0) + np.float32(0.4) * np.arange(6, dtype=np.float32)
    lw0 = np.float32(-2.7725887) - np.float32(0.01) * np.arange(N, dtype=np.float32)
    with P.PhdFilter(cfg, n_particles=N, map_capacity=256) as f:
        f.set_particles(poses, lw0)
        f.set_maps(maps)
        noise = np.zeros((N * k, 2), np.float32)
        for i in range(N * k):
            noise[i, 0] = cfg.stdAlpha * rng.randn()
            noise[i, 1] = cfg.stdEncoder * rng.randn()
        f.predict((2.0, 0.05), noise)
        pg, lg = f.get_particles()
        f.update(z)
        _, lw = f.get_particles()
        sizes = f.map_sizes()
        idx = f.resample(rng.randu01())
        sizes_after = f.map_sizes()
    assert out[0].split() == ["grown", "32", "32", "32", "32", "32"], out[0]
    for i in range(N * k):
        t = out[1 + i].split()
        assert abs(float(t[2]) - lg[i]) <= 2e-6 * abs(lg[i]) and abs(float(t[3]) - pg["px"][i]) <= 1e-6
        assert abs(float(t[4]) - maps[i // k]["mean"][0][0]) <= 1e-6 and int(t[5]) == i // k
        t = out[1 + N * k + i].split()
        assert abs(float(t[2]) - lw[i]) <= 2e-6 * abs(lw[i]) and int(t[3]) == sizes[i]
    assert out[1 + 2 * N * k].split() == ["back", "16"]
    for i in range(N):
        t = out[2 + 2 * N * k + i].split()
        assert int(t[2]) == idx[i] and int(t[3]) == sizes_after[i]


@pytest.mark.parametrize("common", [[], ["--device-noise"]])
@pytest.mark.parametrize("extra", [["--devices", "1"], ["--devices", "1", "--shards", "3"]])
def test_driver_sharded_equals_single_device(tmp_path, extra, common):
    """phdslam --devices N [--shards S] (the C++ multi-device host: RCCL on a one-rank communicator / three shards sharing the
    GPU through device copies) writes the logs the single-device run writes: maps, weights and poses character for
    character; the expected pose (summed on the host in double by the sharded host) to the log's 6 digits"""
    d = str(tmp_path)
    cfg_path = write_dataset(d, n_steps=6, n_particles=48)
    outs = []
    for name, args in (("one", []), ("multi", extra)):
        o = os.path.join(d, name)
        os.makedirs(o)
        # (--device-noise: the control noise from the device generator, which draws by GLOBAL particle index — the sharded run
        #  still equals the single-device one; the host's randn() stream is then used for the resampling uniform only)
        cmd = [os.path.join(PKG, "bin", "phdslam"), cfg_path, "synth", "--out", o, "--seed", "9", "--capacity", "256"] + args + common
        # Round 2 retried a start that stalled once before its first output line (a one-rank RCCL communicator).  200 consecutive
        # starts with NCCL_DEBUG=INFO have not reproduced it (profiles/r03_stress_start.txt: median 1.99 s to the first line, max
        # 2.24 s), and communicator creation now runs under a watchdog (phd_multi_create, PHD_RCCL_INIT_TIMEOUT): a bootstrap
        # that does not return is an error with a message, not a hang.  No retry: a stall fails this test with that message.
        env = dict(os.environ, PHD_RCCL_INIT_TIMEOUT="60")
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=150, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append((o, r.stdout))
    assert "sharded filter: 48 particles" in outs[1][1]
    resampled = 0
    for n in range(6):
        a = open(os.path.join(outs[0][0], "state_estimate%05d.log" % n)).read().split("\n")
        b = open(os.path.join(outs[1][0], "state_estimate%05d.log" % n)).read().split("\n")
        assert a[1:] == b[1:], n                                   # map, weights, poses, cardinality line
        assert np.allclose([float(x) for x in a[0].split()], [float(x) for x in b[0].split()], rtol=2e-5, atol=1e-6), n
    for line_a, line_b in zip(outs[0][1].split("\n"), outs[1][1].split("\n")[1:]):
        if "resampled=1" in line_a:
            resampled += 1
    assert resampled >= 1 or common      # (the device generator's stream happens not to trigger a resample within these six steps)


@pytest.mark.parametrize("extra", [[], ["--log7"], ["--device-noise"], ["cphd"], ["cphd", "--log7"]], ids=["log5", "log7", "device_noise", "cphd", "cphd_log7"])
def test_pipelined_loop_writes_the_files_of_the_synchronous_loop(tmp_path, extra):
    """Round 6: the driver's default loop no longer waits for the device inside a step (noise drawn ahead by a helper thread in the
    same order from the same stream, state captured on the device between update and resample, resample decided on the device,
    the log of step n written while step n + 1 runs).  Every state_estimate file must be byte for byte what the step-synchronous
    loop (PHD_DRIVER_SYNC=1: run_synth's own structure, src/main.cpp:1178-1312) writes — resampled steps included."""
    d = str(tmp_path)
    cphd = "cphd" in extra                       # filter_type = 1: the log's last line is cn_estimate, which rides in the snapshot too
    extra = [e for e in extra if e != "cphd"]
    cfg_path = write_dataset(d, n_steps=12, n_particles=300, seed=11, cphd=cphd)
    outs = {}
    for mode in ("pipelined", "sync"):
        out = os.path.join(d, "out_" + mode); os.makedirs(out)
        env = dict(os.environ)
        env.pop("PHD_DRIVER_PROFILE", None)
        if mode == "sync":
            env["PHD_DRIVER_SYNC"] = "1"
        else:
            env.pop("PHD_DRIVER_SYNC", None)
        r = subprocess.run([os.path.join(PKG, "bin", "phdslam"), cfg_path, "synth", "--out", out, "--seed", "21", "--capacity", "256"] + extra,
                           env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[mode] = (out, r.stdout)
    names = sorted(n for n in os.listdir(outs["sync"][0]) if n.startswith("state_estimate"))
    assert len(names) == 12 and names == sorted(n for n in os.listdir(outs["pipelined"][0]) if n.startswith("state_estimate"))
    for nm in names:
        a = open(os.path.join(outs["sync"][0], nm), "rb").read()
        b = open(os.path.join(outs["pipelined"][0], nm), "rb").read()
        assert a == b, nm
    # the per-step lines (M, particle count, map size, resample decision) agree too, and some step resampled
    strip = lambda t: [ln.rsplit(" ", 2)[0] for ln in t.splitlines() if ln.startswith("****** Time Step")]
    assert strip(outs["sync"][1]) == strip(outs["pipelined"][1]) and len(strip(outs["sync"][1])) == 12
    assert any("resampled=1" in ln for ln in strip(outs["pipelined"][1]))
    assert len(open(os.path.join(outs["pipelined"][0], "loopTime.log")).read().split()) == 12


def test_snapshot_slots_equal_the_blocking_snapshot():
    """phd_snapshot_capture / _send / _wait (the pipelined loop's state extraction) against phd_state_snapshot on the same filter:
    the captured block holds the state BETWEEN the update and the resample although it is read after the resample and after the
    next step was enqueued; report and resample indices ride along."""
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    N, G, M = 600, 24, 10
    w = S.make_workload(N, G, M, seed=77, n_meas_sets=3, clustered=True)
    cfg = P.default_config(n_particles=N, resampleThresh=2.0)           # nEff / N <= 2: every step with a scan resamples
    ref = []
    with P.PhdFilter(cfg, n_particles=N, map_capacity=128, max_measurements=M) as f:
        f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
        for s in range(3):
            f.predict((2.0, 0.05), w["noise"][s]); f.update(w["z"][s])
            ref.append(f.state_snapshot())
            did, idx = f.resample_if_needed(w["uniform"][s], had_measurements=True)
            assert did
            ref[-1] = ref[-1] + (idx,)
    with P.PhdFilter(cfg, n_particles=N, map_capacity=128, max_measurements=M) as f:
        f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
        got = []
        for s in range(3):
            f.predict((2.0, 0.05), w["noise"][s]); f.update(w["z"][s])
            f.snapshot_capture(s & 1)
            P._lib.check(P._lib.lib().phd_resample_if_needed(f._h, float(w["uniform"][s]), 1, None, None), "phd_resample_if_needed")
            f.snapshot_send(s & 1, want_resample_idx=True)
            if s > 0:
                got.append(f.snapshot_wait((s - 1) & 1))                # read one step late, as the driver does
        got.append(f.snapshot_wait(2 & 1))
    for s in range(3):
        e, m, who, poses, lw, idx = ref[s]
        e2, m2, who2, poses2, lw2, idx2, rep = got[s]
        assert e.tobytes() == e2.tobytes() and who == who2 and m.tobytes() == m2.tobytes(), s
        assert poses.tobytes() == poses2.tobytes() and lw.tobytes() == lw2.tobytes(), s
        assert rep.did_resample == 1 and np.array_equal(idx, idx2) and rep.neff > 0, s


def test_snapshot_slots_refuse_misuse():
    """the slot protocol of the pipelined snapshot fails loudly: a slot is captured, then sent, then waited for — anything else is
    PHD_ERR_INVALID_ARG with a message, never a silent overwrite of a block that is still being downloaded"""
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    w = S.make_workload(64, 8, 4, seed=3)
    with P.PhdFilter(P.default_config(n_particles=64), n_particles=64, map_capacity=32, max_measurements=8) as f:
        f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"])
        f.update(w["z"][0])
        for bad in (lambda: f.snapshot_send(0), lambda: f.snapshot_wait(0), lambda: f.snapshot_capture(2)):
            with pytest.raises(P._lib.PhdError):
                bad()
        f.snapshot_capture(0)
        with pytest.raises(P._lib.PhdError):
            f.snapshot_capture(0)                              # captured, not yet waited for
        with pytest.raises(P._lib.PhdError):
            f.snapshot_wait(0)                                 # captured, not yet sent
        f.snapshot_send(0)
        with pytest.raises(P._lib.PhdError):
            f.snapshot_send(0)
        e, m, who, poses, lw, idx, rep = f.snapshot_wait(0)
        assert idx is None and len(poses) == 64 and rep.status == 0
        e2, m2, who2, poses2, lw2 = f.state_snapshot()
        assert e.tobytes() == e2.tobytes() and m.tobytes() == m2.tobytes() and lw.tobytes() == lw2.tobytes()
        f.snapshot_capture(0)                                  # the slot is free again
        f.snapshot_send(0)
        f.snapshot_wait(0)


@pytest.mark.parametrize("ft,N", [(0, 300), (1, 200), (0, 5000)])
def test_predict_update_in_one_launch_equals_the_two_calls(ft, N):
    """phd_predict_update (the pipelined loop's step: vehicle predict fused in front of the update kernel, normalisation as its tail)
    against phd_predict_ackerman + phd_update: particles, weights, maps, the step report — bit for bit, over three chained steps
    with host noise and with the device generator, PHD and CPHD, below and above the block-form tail's 4096 particles."""
    P = importlib.import_module("cuda-phdslam_amd")
    S = importlib.import_module("cuda-phdslam_amd.synthetic")
    G, M = 24, 10
    w = S.make_workload(N, G, M, seed=91 + N, n_meas_sets=3, clustered=True)
    cfg = P.default_config(n_particles=N, filterType=ft, maxCardinality=63)
    outs = []
    for fused in (False, True):
        with P.PhdFilter(cfg, n_particles=N, map_capacity=128, max_measurements=16) as f:
            f.set_particles(w["poses"], w["logw"]); f.set_maps(w["maps"], w["sizes"]); f.seed(5)
            for s in range(3):
                noise = w["noise"][s] if s != 1 else None                       # step 1: the device generator
                z = w["z"][s] if s != 2 else w["z"][s][:7]                      # step 2: a shorter scan
                if fused:
                    f.predict_update((2.0, 0.05), noise, z)
                else:
                    f.predict((2.0, 0.05), noise); f.update(z)
                f.resample_if_needed(w["uniform"][s], had_measurements=True)
            poses, lw = f.get_particles()
            rep = f.step_report()
            outs.append((poses.tobytes(), lw.tobytes(), [m.tobytes() for m in f.get_maps()], rep.neff, rep.did_resample))
    assert outs[0][0] == outs[1][0] and outs[0][1] == outs[1][1] and outs[0][2] == outs[1][2]
    assert outs[0][3] == outs[1][3] and outs[0][4] == outs[1][4]
