// phd_eval.cpp — estimation-quality metrics of the reference's offline tooling, host side:
// the OSPA distance of python/ospa.py:220-274 (Hungarian assignment, cut-off c, order p) and the
// per-step evaluation of python/batch_analyze.py:16-37 (pose error, OSPA of the top-round(sum w)
// map features against the true map, nEff).  The reference's own ospa.py cannot be built here
// (pyximport of munkres_step4.pyx); this is the replacement a batch analysis links against.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <fstream>
#include <limits>
#include <sstream>
#include <string>
#include <vector>

#include "phdslam.h"

extern "C" int phd_internal_set_error(int code, const char* msg);

namespace {
// minimum-cost assignment of every row (m rows <= n columns): shortest augmenting paths with
// potentials (Kuhn-Munkres in O(m^2 n)); returns the column assigned to each row
std::vector<int> assign_rows(const std::vector<double>& cost, int m, int n)
{
    const double INF = std::numeric_limits<double>::infinity();
    std::vector<double> u(m + 1, 0.0), v(n + 1, 0.0);
    std::vector<int> p(n + 1, 0), way(n + 1, 0);
    for (int i = 1; i <= m; ++i) {
        p[0] = i;
        int j0 = 0;
        std::vector<double> minv(n + 1, INF);
        std::vector<char> used(n + 1, 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = INF;
            int j1 = 0;
            for (int j = 1; j <= n; ++j) {
                if (used[j]) continue;
                const double cur = cost[(size_t)(i0 - 1) * n + (j - 1)] - u[i0] - v[j];
                if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                if (minv[j] < delta) { delta = minv[j]; j1 = j; }
            }
            for (int j = 0; j <= n; ++j) {
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            }
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    std::vector<int> col(m, -1);
    for (int j = 1; j <= n; ++j)
        if (p[j] > 0) col[p[j] - 1] = j - 1;
    return col;
}
} // namespace

// ospa_distance(X, Y, p, c) of python/ospa.py:220-274.  X: m points, Y: n points, row-major (x, y).
// out = (ospa, localisation part, cardinality part).
extern "C" int phd_ospa(const float* X, int m, const float* Y, int n, double p, double c, double* out)
{
    if (!out || m < 0 || n < 0 || p <= 0 || c <= 0) return phd_internal_set_error(PHD_ERR_INVALID_ARG, "phd_ospa: bad argument");
    if (m == 0 && n == 0) { out[0] = out[1] = out[2] = 0; return PHD_OK; }                    // :224-225
    if (m == 0 || n == 0) { out[0] = c; out[1] = 0; out[2] = c; return PHD_OK; }              // :226-227
    if (m > n) { std::swap(X, Y); std::swap(m, n); }                                           // :231-235: Y is the larger set
    std::vector<double> cost((size_t)m * n);
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            const double dx = (double)X[2 * i] - Y[2 * j], dy = (double)X[2 * i + 1] - Y[2 * j + 1];
            const double d = sqrt(dx * dx + dy * dy);
            cost[(size_t)i * n + j] = d > c ? c : d;                                           // cut-off (:246)
        }
    const std::vector<int> col = assign_rows(cost, m, n);                                      // Munkres (:255-256)
    double total_loc = 0;
    // like the reference: the assignment minimises the sum of cut-off distances, the metric sums d^p (:263)
    for (int i = 0; i < m; ++i) total_loc += pow(cost[(size_t)i * n + col[i]], p);
    const double cp = pow(c, p);
    out[2] = pow(cp * (n - m) / n, 1.0 / p);                                                   // :264
    out[1] = pow(total_loc / n, 1.0 / p);                                                      // :265
    out[0] = pow((total_loc + (n - m) * cp) / n, 1.0 / p);                                     // :266
    return PHD_OK;
}

// compute_error_k of python/batch_analyze.py:16-37 on one state_estimate log:
// out = (pose error, OSPA, OSPA localisation, OSPA cardinality, nEff)
extern "C" int phd_evaluate_state_log(const char* path, const float* true_pose_xy, const float* true_map, int n_true,
                                      double p, double c, double* out)
{
    if (!path || !true_pose_xy || !out) return phd_internal_set_error(PHD_ERR_INVALID_ARG, "phd_evaluate_state_log: null argument");
    std::ifstream f(path);
    if (!f) return phd_internal_set_error(PHD_ERR_IO, (std::string("cannot open ") + path).c_str());
    std::string l1, l2, l3;
    std::getline(f, l1); std::getline(f, l2); std::getline(f, l3);
    auto numbers = [](const std::string& s) {
        std::vector<double> v;
        std::istringstream is(s);
        double x;
        while (is >> x) v.push_back(x);
        return v;
    };
    const std::vector<double> pose = numbers(l1), mapv = numbers(l2), lw = numbers(l3);
    if (pose.size() < 3 || mapv.size() % 7 != 0)
        return phd_internal_set_error(PHD_ERR_PARSE, (std::string(path) + ": not a state_estimate log").c_str());
    const int nf = (int)mapv.size() / 7;
    // the round(sum of weights) highest-weighted features are the map estimate (:23-27)
    std::vector<int> order(nf);
    double wsum = 0;
    for (int i = 0; i < nf; ++i) { order[i] = i; wsum += mapv[7 * i]; }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return mapv[7 * a] > mapv[7 * b]; });
    int keep = (int)llround(wsum);
    keep = std::max(0, std::min(keep, nf));
    std::vector<float> est(2 * (size_t)keep + 1);
    for (int k = 0; k < keep; ++k) { est[2 * k] = (float)mapv[7 * order[k] + 1]; est[2 * k + 1] = (float)mapv[7 * order[k] + 2]; }
    const double ex = true_pose_xy[0] - pose[0], ey = true_pose_xy[1] - pose[1];
    out[0] = sqrt(ex * ex + ey * ey);                                                          // :28
    int rc = phd_ospa(true_map, n_true, est.data(), keep, p, c, out + 1);                      // :29 (p = 1, c = 5 there)
    if (rc) return rc;
    double s2 = 0;
    for (double w : lw) s2 += exp(w) * exp(w);
    out[4] = s2 > 0 ? 1.0 / s2 : 0.0;                                                          // :34
    return PHD_OK;
}
