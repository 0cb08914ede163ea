// structured_oracle.cpp -- TEST INFRASTRUCTURE / CPU BASELINE ONLY (never linked by the product).
//
// A structured CPU solver for the unpenalised / zero-velocity-penalised minimum-snap problem: the same
// problem dense_oracle.c solves by literal restatement of minimum_snap.cpp:227-649, solved here the
// way a CPU implementation that knows the structure would ("also time the
// structured CPU solver as the honest CPU line"): constant tables for M(1)^-1 and
// M(1)^-T Q(1) M(1)^-1 scaled by powers of T, a block-tridiagonal LDL^T over the free derivatives
// shared by the three axes, per-segment Hermite -> monomial recovery.  Plain loops, OpenMP over the
// batch.  It is NOT the oracle of record (parity is judged against dense_oracle.c and the golden
// fixtures); tests/test_oracle.py checks it against the dense oracle so the timing line is known to
// time a correct solve.  The tables come from cs-pathplan_amd/tablegen.py (generated header).
#include "../cs-pathplan_amd/csrc/minsnap_tables.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

template <int O> struct Tb;
#define TB(o)                                                                  \
    template <> struct Tb<o> {                                                 \
        static double G(int i, int a) { return csp::tables::G##o[i][a]; }      \
        static double QT(int a, int b) { return csp::tables::QT##o[a][b]; }    \
    };
TB(1) TB(2) TB(3) TB(4) TB(5)
#undef TB

// in-place Cholesky solve of the small SPD system A X = B (A n x n, B n x m, row-major); false if not SPD
static bool chol_solve(int n, int m, double *A, double *B) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return false;
        d = std::sqrt(d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = A[i * n + j];
            for (int k = 0; k < j; ++k) v -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = v / d;
        }
    }
    for (int c = 0; c < m; ++c) {
        for (int i = 0; i < n; ++i) {
            double v = B[i * m + c];
            for (int k = 0; k < i; ++k) v -= A[i * n + k] * B[k * m + c];
            B[i * m + c] = v / A[i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double v = B[i * m + c];
            for (int k = i + 1; k < n; ++k) v -= A[k * n + i] * B[k * m + c];
            B[i * m + c] = v / A[i * n + i];
        }
    }
    return true;
}

// One trajectory.  work: (S+1) * (N*N + 3N) + S * 3o doubles.  bc rows: v0, v1, a0, a1.
template <int O>
static int solve_one(int S, const double *wp, const double *tm, const double *bc, double vw, double *coeff, double *work) {
    constexpr int N = O - 1, M = 2 * O, NN = (N > 0 ? N : 1);
    double *Wst = work, *zst = work + (size_t)(S + 1) * NN * NN;
    // powers of each segment time, once: ipw[k][e] = T_k^-e (e < 2o), tpw[k][e] = T_k^e (e < o)
    double *ipw = work + (size_t)(S + 1) * (NN * NN + 3 * NN), *tpw = ipw + (size_t)S * M;
    for (int k = 0; k < S; ++k) {
        const double inv = 1.0 / tm[k];
        ipw[(size_t)k * M] = 1.0;
        for (int e = 1; e < M; ++e) ipw[(size_t)k * M + e] = ipw[(size_t)k * M + e - 1] * inv;
        tpw[(size_t)k * O] = 1.0;
        for (int e = 1; e < O; ++e) tpw[(size_t)k * O + e] = tpw[(size_t)k * O + e - 1] * tm[k];
    }
    auto qt = [&](int a, int b, int k) {   // entry (a, b) of M(T_k)^-T Q(T_k) M(T_k)^-1 = Qt1[a][b] T^(1 - 2o + d_a + d_b)
        return Tb<O>::QT(a, b) * ipw[(size_t)k * M + (2 * O - 1 - (a % O) - (b % O))];
    };
    double x[2][NN][3];   // boundary free derivatives: v, a, then zeros
    for (int e = 0; e < 2; ++e)
        for (int r = 0; r < N; ++r)
            for (int ax = 0; ax < 3; ++ax) x[e][r][ax] = r == 0 ? bc[e * 3 + ax] : r == 1 ? bc[(2 + e) * 3 + ax] : 0.0;
    int rc = 0;
    // forward elimination over interior waypoints k = 1..S-1 (x_k = z_k - W_k x_{k+1})
    double Wp[NN * NN], zp[NN * 3];
    std::memset(Wp, 0, sizeof Wp);
    for (int r = 0; r < N; ++r)
        for (int ax = 0; ax < 3; ++ax) zp[r * 3 + ax] = x[0][r][ax];
    for (int k = 1; k < S && N > 0; ++k) {
        const int Tl = k - 1, Tr = k;   // segment indices of the left / right neighbour
        double A[NN * NN], R[NN * (NN + 3)];
        for (int r = 0; r < N; ++r) {
            for (int c = 0; c < N; ++c) {
                double v = qt(O + r + 1, O + c + 1, Tl) + qt(r + 1, c + 1, Tr);
                if (r == 0 && c == 0) v += 2.0 * vw;   // +w on the velocity diagonal of both neighbours
                for (int j = 0; j < N; ++j) v -= qt(j + 1, O + r + 1, Tl) * Wp[j * N + c];
                A[r * N + c] = v;
                R[r * (N + 3) + c] = qt(r + 1, O + c + 1, Tr);
            }
            for (int ax = 0; ax < 3; ++ax) {
                const double *p0 = wp + (size_t)(k - 1) * 3, *p1 = wp + (size_t)k * 3, *p2 = wp + (size_t)(k + 1) * 3;
                double v = -(qt(O + r + 1, 0, Tl) * p0[ax] + qt(O + r + 1, O, Tl) * p1[ax] + qt(r + 1, 0, Tr) * p1[ax] + qt(r + 1, O, Tr) * p2[ax]);
                for (int j = 0; j < N; ++j) v -= qt(j + 1, O + r + 1, Tl) * zp[j * 3 + ax];
                R[r * (N + 3) + N + ax] = v;
            }
        }
        if (!chol_solve(N, N + 3, A, R)) rc |= 2;
        for (int r = 0; r < N; ++r) {
            for (int c = 0; c < N; ++c) Wp[r * N + c] = Wst[(size_t)k * N * N + r * N + c] = R[r * (N + 3) + c];
            for (int ax = 0; ax < 3; ++ax) zp[r * 3 + ax] = zst[(size_t)k * N * 3 + r * 3 + ax] = R[r * (N + 3) + N + ax];
        }
    }
    // back-substitution and recovery, segments S-1 .. 0
    double xn[NN][3], xk[NN][3];
    for (int r = 0; r < N; ++r)
        for (int ax = 0; ax < 3; ++ax) xn[r][ax] = x[1][r][ax];
    for (int k = S - 1; k >= 0; --k) {
        for (int r = 0; r < N; ++r)
            for (int ax = 0; ax < 3; ++ax) {
                if (k == 0) { xk[r][ax] = x[0][r][ax]; continue; }
                double v = zst[(size_t)k * N * 3 + r * 3 + ax];
                for (int c = 0; c < N; ++c) v -= Wst[(size_t)k * N * N + r * N + c] * xn[c][ax];
                xk[r][ax] = v;
            }
        for (int ax = 0; ax < 3; ++ax) {
            double d[M];
            d[0] = wp[(size_t)k * 3 + ax];
            d[O] = wp[(size_t)(k + 1) * 3 + ax];
            for (int r = 0; r < N; ++r) { d[r + 1] = xk[r][ax]; d[O + r + 1] = xn[r][ax]; }
            for (int i = 0; i < M; ++i) {   // coefficient of t^(M-1-i)
                double acc = 0.0;
                for (int a = 0; a < M; ++a) acc += Tb<O>::G(i, a) * d[a] * tpw[(size_t)k * O + (a % O)];
                const double c = acc * ipw[(size_t)k * M + (M - 1 - i)];
                coeff[((size_t)k * 3 + ax) * M + i] = c;
                if (!std::isfinite(c)) rc |= 1;
            }
        }
        for (int r = 0; r < N; ++r)
            for (int ax = 0; ax < 3; ++ax) xn[r][ax] = xk[r][ax];
    }
    return rc;
}

}  // namespace

// waypoints [B][S+1][3], times [B][S], bc [B or 1][4][3], coeff [B][S][3][2o].  Returns 0, or -1 for a
// bad order.  Per-trajectory problems (non-SPD pivot, non-finite output) are ignored like the reference does.
extern "C" int csp_struct_solve_batch(int order, int S, long B, const double *wp, const double *tm, const double *bc,
                                      int bc_bcast, double vel_zero_weight, double *coeff, int nthreads) {
    if (order < 1 || order > 5 || S < 1 || B < 0) return -1;
    const int n = order - 1 > 0 ? order - 1 : 1;
    const size_t work_n = (size_t)(S + 1) * (size_t)(n * n + 3 * n) + (size_t)S * 3 * order;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        double *work = (double *)std::malloc(work_n * sizeof(double));
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (long b = 0; b < B; ++b) {
            const double *w = wp + (size_t)b * (S + 1) * 3, *t = tm + (size_t)b * S, *c0 = bc + (bc_bcast ? 0 : (size_t)b * 12);
            double *co = coeff + (size_t)b * S * 3 * 2 * order;
            switch (order) {
                case 1: solve_one<1>(S, w, t, c0, vel_zero_weight, co, work); break;
                case 2: solve_one<2>(S, w, t, c0, vel_zero_weight, co, work); break;
                case 3: solve_one<3>(S, w, t, c0, vel_zero_weight, co, work); break;
                case 4: solve_one<4>(S, w, t, c0, vel_zero_weight, co, work); break;
                case 5: solve_one<5>(S, w, t, c0, vel_zero_weight, co, work); break;
            }
        }
        std::free(work);
    }
    return 0;
}
