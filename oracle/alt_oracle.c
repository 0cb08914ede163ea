/*
 * alt_oracle.c -- CPU restatement of the altitude optimiser's two quadratic solves
 * (/root/reference/uavPathPlanning.cpp:1575-1713 optimizeHeights, :1715-1827
 * optimizeHeightsGlobalSmooth; parameters uavPathPlanning.hpp:415-421).  These are the reference's
 * only Eigen::SimplicialLDLT call sites (:1670, :1796).
 * TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: no Eigen here, no goldens in the reference; pinned
 * against an independent numpy restatement in tests/test_alt.py.
 *
 * The Hessian is assembled DENSE from the same triplet lists and solved by dense Cholesky, on
 * purpose unlike the HIP kernel (banded LDL^T): two different algorithms, one set of equations.
 * The elevation lookups (cost map / GeoTIFF) are outside the path: the caller passes the terrain
 * elevation per sample, NaN where the reference's lookups would both fail.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int chol_solve(double *H, double *b, int n) { /* in-place dense Cholesky, b <- H^-1 b */
    for (int j = 0; j < n; ++j) {
        double d = H[(size_t)j * n + j];
        for (int k = 0; k < j; ++k) d -= H[(size_t)j * n + k] * H[(size_t)j * n + k];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        H[(size_t)j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = H[(size_t)i * n + j];
            for (int k = 0; k < j; ++k) v -= H[(size_t)i * n + k] * H[(size_t)j * n + k];
            H[(size_t)i * n + j] = v / d;
        }
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= H[(size_t)i * n + k] * b[k];
        b[i] = s / H[(size_t)i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= H[(size_t)k * n + i] * b[k];
        b[i] = s / H[(size_t)i * n + i];
    }
    return 0;
}

static void add_smooth_and_climb(double *H, int n, const double *xyz, double lambda_smooth, double max_climb_rate) {
    if (n >= 3 && lambda_smooth > 0.0)
        for (int i = 1; i + 1 < n; ++i) {            /* lambda * (L^T L), L z = z[i-1] - 2 z[i] + z[i+1] */
            const double c[3] = {1.0, -2.0, 1.0};
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb < 3; ++bb) H[(size_t)(i - 1 + a) * n + (i - 1 + bb)] += lambda_smooth * (c[a] * c[bb]);
        }
    if (max_climb_rate > 0.0)
        for (int i = 0; i + 1 < n; ++i) {            /* ((z[i+1]-z[i]) / (dist * rate))^2 */
            double dist = hypot(xyz[(i + 1) * 3] - xyz[i * 3], xyz[(i + 1) * 3 + 1] - xyz[i * 3 + 1]);
            if (dist <= 1e-9) continue;
            double denom = dist * max_climb_rate;
            if (denom <= 1e-12) continue;
            double w = 1.0 / (denom * denom);
            H[(size_t)i * n + i] += w;
            H[(size_t)i * n + i + 1] += -w;
            H[(size_t)(i + 1) * n + i] += -w;
            H[(size_t)(i + 1) * n + i + 1] += w;
        }
}

/* optimizeHeights (:1575-1713) */
int csp_oracle_alt_optimize(int n, const double *xyz, const double *elev, double lambda_smooth,
                            double lambda_follow, double safe_distance, double max_climb_rate, double *out_z) {
    if (n <= 0) return -1;
    double *H = (double *)calloc((size_t)n * n, sizeof(double)), *b = (double *)calloc((size_t)n, sizeof(double));
    add_smooth_and_climb(H, n, xyz, lambda_smooth, max_climb_rate);
    for (int i = 0; i < n; ++i) {
        if (!isnan(elev[i])) {
            double safe_h = elev[i] + safe_distance;
            double target = xyz[i * 3 + 2] > safe_h ? xyz[i * 3 + 2] : safe_h;
            H[(size_t)i * n + i] += lambda_follow;
            b[i] += lambda_follow * target;
        }
        H[(size_t)i * n + i] += 1e-8;
    }
    int rc = chol_solve(H, b, n);
    if (rc == 0)
        for (int i = 0; i < n; ++i) {
            out_z[i] = b[i];
            if (!isnan(elev[i]) && out_z[i] < elev[i] + safe_distance) out_z[i] = elev[i] + safe_distance;
        }
    free(H); free(b);
    return rc;
}

/* optimizeHeightsGlobalSmooth (:1715-1827); returns the number of solves, <0 on failure */
int csp_oracle_alt_global_smooth(int n, const double *input_z, const double *xyz, double lambda_smooth,
                                 double max_climb_rate, double *out_z) {
    if (n <= 0) return -1;
    double *H = (double *)malloc((size_t)n * n * sizeof(double)), *b = (double *)malloc((size_t)n * sizeof(double));
    char *active = (char *)calloc((size_t)n, 1);
    memcpy(out_z, input_z, (size_t)n * sizeof(double));
    int solves = 0;
    for (int iter = 0; iter < 10; ++iter) {
        memset(H, 0, (size_t)n * n * sizeof(double));
        memset(b, 0, (size_t)n * sizeof(double));
        add_smooth_and_climb(H, n, xyz, lambda_smooth, max_climb_rate);
        H[0] += 1e10; b[0] += 1e10 * input_z[0];
        H[(size_t)(n - 1) * n + n - 1] += 1e10; b[n - 1] += 1e10 * input_z[n - 1];
        for (int i = 1; i + 1 < n; ++i)
            if (active[i]) { H[(size_t)i * n + i] += 1e8; b[i] += 1e8 * input_z[i]; }
        for (int i = 0; i < n; ++i) H[(size_t)i * n + i] += 1e-8;
        if (chol_solve(H, b, n)) { solves = -1; break; }
        ++solves;
        int violation = 0;
        for (int i = 0; i < n; ++i) {
            out_z[i] = b[i];
            if (out_z[i] < input_z[i] - 1e-3 && !active[i]) { active[i] = 1; violation = 1; }
        }
        if (!violation) break;
    }
    if (solves > 0)
        for (int i = 0; i < n; ++i) if (out_z[i] < input_z[i]) out_z[i] = input_z[i];
    free(H); free(b); free(active);
    return solves;
}


/* ---- the same two solves with the Hessian kept in its THREE BANDS and a banded Cholesky -------------------------------------
 * (round 3).  Not a third restatement of the equations: the bands are filled by the same triplet rules as the dense matrix
 * above (and checked against it, tests/test_alt.py); what differs is the cost -- O(n) instead of O(n^3) -- which makes this
 * the one-core CPU time that stands beside the GPU's for ONE long problem (bench.py `single_altitude`), the reference's own
 * call pattern (one pentadiagonal problem per plan, uavPathPlanning.cpp:1670-1676, :1796-1799). */
static void band_fill(double *d, double *e, double *f, int n, const double *xyz, double lambda_smooth, double max_climb_rate) {
    /* d[i] = H[i][i], e[i] = H[i][i-1], f[i] = H[i][i-2] */
    memset(d, 0, (size_t)n * sizeof(double)); memset(e, 0, (size_t)n * sizeof(double)); memset(f, 0, (size_t)n * sizeof(double));
    if (n >= 3 && lambda_smooth > 0.0)
        for (int i = 1; i + 1 < n; ++i) {
            const double c[3] = {1.0, -2.0, 1.0};
            for (int a = 0; a < 3; ++a)
                for (int bb = 0; bb <= a; ++bb) {
                    const int r = i - 1 + a, cc = i - 1 + bb;   /* r >= cc */
                    const double v = lambda_smooth * (c[a] * c[bb]);
                    if (r == cc) d[r] += v; else if (r == cc + 1) e[r] += v; else f[r] += v;
                }
        }
    if (max_climb_rate > 0.0)
        for (int i = 0; i + 1 < n; ++i) {
            double dist = hypot(xyz[(i + 1) * 3] - xyz[i * 3], xyz[(i + 1) * 3 + 1] - xyz[i * 3 + 1]);
            if (dist <= 1e-9) continue;
            double denom = dist * max_climb_rate;
            if (denom <= 1e-12) continue;
            double w = 1.0 / (denom * denom);
            d[i] += w; d[i + 1] += w; e[i + 1] += -w;
        }
}

static int band_chol_solve(double *d, double *e, double *f, double *b, int n) {   /* in place: L L^T, b <- H^-1 b */
    for (int i = 0; i < n; ++i) {
        /* L[i][i-2] = f[i] / L[i-2][i-2];  L[i][i-1] = (e[i] - L[i][i-2] L[i-1][i-2]) / L[i-1][i-1] */
        double l2 = i >= 2 ? f[i] / d[i - 2] : 0.0;
        double l1 = i >= 1 ? (e[i] - l2 * (i >= 2 ? e[i - 1] : 0.0)) / d[i - 1] : 0.0;
        double dd = d[i] - l1 * l1 - l2 * l2;
        if (!(dd > 0.0)) return -1;
        d[i] = sqrt(dd); e[i] = l1; f[i] = l2;
    }
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        if (i >= 1) s -= e[i] * b[i - 1];
        if (i >= 2) s -= f[i] * b[i - 2];
        b[i] = s / d[i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        if (i + 1 < n) s -= e[i + 1] * b[i + 1];
        if (i + 2 < n) s -= f[i + 2] * b[i + 2];
        b[i] = s / d[i];
    }
    return 0;
}

int csp_oracle_alt_optimize_banded(int n, const double *xyz, const double *elev, double lambda_smooth,
                                   double lambda_follow, double safe_distance, double max_climb_rate, double *out_z) {
    if (n <= 0) return -1;
    double *d = (double *)malloc((size_t)n * 4 * sizeof(double)), *e = d + n, *f = e + n, *b = f + n;
    band_fill(d, e, f, n, xyz, lambda_smooth, max_climb_rate);
    for (int i = 0; i < n; ++i) {
        b[i] = 0.0;
        if (!isnan(elev[i])) {
            double safe_h = elev[i] + safe_distance;
            double target = xyz[i * 3 + 2] > safe_h ? xyz[i * 3 + 2] : safe_h;
            d[i] += lambda_follow;
            b[i] += lambda_follow * target;
        }
        d[i] += 1e-8;
    }
    int rc = band_chol_solve(d, e, f, b, n);
    if (rc == 0)
        for (int i = 0; i < n; ++i) {
            out_z[i] = b[i];
            if (!isnan(elev[i]) && out_z[i] < elev[i] + safe_distance) out_z[i] = elev[i] + safe_distance;
        }
    free(d);
    return rc;
}

int csp_oracle_alt_global_smooth_banded(int n, const double *input_z, const double *xyz, double lambda_smooth,
                                        double max_climb_rate, double *out_z) {
    if (n <= 0) return -1;
    double *d = (double *)malloc((size_t)n * 4 * sizeof(double)), *e = d + n, *f = e + n, *b = f + n;
    char *active = (char *)calloc((size_t)n, 1);
    memcpy(out_z, input_z, (size_t)n * sizeof(double));
    int solves = 0;
    for (int iter = 0; iter < 10; ++iter) {
        band_fill(d, e, f, n, xyz, lambda_smooth, max_climb_rate);
        memset(b, 0, (size_t)n * sizeof(double));
        d[0] += 1e10; b[0] += 1e10 * input_z[0];
        d[n - 1] += 1e10; b[n - 1] += 1e10 * input_z[n - 1];
        for (int i = 1; i + 1 < n; ++i)
            if (active[i]) { d[i] += 1e8; b[i] += 1e8 * input_z[i]; }
        for (int i = 0; i < n; ++i) d[i] += 1e-8;
        if (band_chol_solve(d, e, f, b, n)) { solves = -1; break; }
        ++solves;
        int violation = 0;
        for (int i = 0; i < n; ++i) {
            out_z[i] = b[i];
            if (out_z[i] < input_z[i] - 1e-3 && !active[i]) { active[i] = 1; violation = 1; }
        }
        if (!violation) break;
    }
    if (solves > 0)
        for (int i = 0; i < n; ++i) if (out_z[i] < input_z[i]) out_z[i] = input_z[i];
    free(d); free(active);
    return solves;
}
