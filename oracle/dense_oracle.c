/*
 * dense_oracle.c -- CPU restatement of the reference's closed-form minimum-snap solve.
 *
 * TEST INFRASTRUCTURE ONLY.  Linked/loaded by tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py -- never by the product path (cs-pathplan_amd/), which is
 * HIP-only and fails loudly without its extension.
 *
 * PARITY UNPINNED.  The reference (/root/reference/math_util/minimum_snap.cpp) needs
 * <Eigen/Dense>, which is not installed in the build image, and the reference ships no golden
 * coefficient vectors.  This file therefore restates the ALGORITHM (same dense matrices, same
 * places where an inverse is taken, same int arithmetic for factorial ratios) with a
 * hand-written partial-pivot LU standing where Eigen's dynamic-size inverse() stands; it is
 * pinned against an independent numpy/LAPACK restatement (oracle/numpy_ref.py), against
 * analytic known answers (tests/golden/) and against an 80-bit long-double build of itself,
 * not against an output of the real Eigen build.
 *
 * Function -> reference map (file: /root/reference/math_util/minimum_snap.cpp)
 *   fact_i32            :15-20   Factorial (C int)
 *   assemble_M          :247-266 derivative map M
 *   selection_column    :268-310 selection matrix C_T (one 1 per row; column index only)
 *   assemble_Q          :312-330 snap-integral Hessian Q (int prefactor)
 *   fill_fixed          :526-562 fixed part of d_selected
 *   solve_axes          :524-592 (and :357-405 for the pre-solve)
 *   path penalty        :347-469, vel-zero penalty :473-509, R :511, f_valid :517-522
 *   deviation metric    :594-624, output pack :626-648
 *   csp_oracle_time_alloc            :63-72
 *   csp_oracle_generate_trajectory   :22-206
 *
 * Built twice from this one source: REAL=double (symbols csp_oracle_*) and REAL=long double
 * (symbols csp_oracle_ld_*), see oracle/Makefile.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifdef CSP_ORACLE_LONG_DOUBLE
typedef long double real;
#define SYM(name) csp_oracle_ld_##name
#define R_POW powl
#define R_SQRT sqrtl
#define R_FABS fabsl
#else
typedef double real;
#define SYM(name) csp_oracle_##name
#define R_POW pow
#define R_SQRT sqrt
#define R_FABS fabs
#endif

/* ---- tiny dense row-major matrix kit --------------------------------------------------- */

static real *mat_new(int r, int c) { return (real *)calloc((size_t)r * (size_t)c + 1, sizeof(real)); }

/* C[r x c] = A[r x k] * B[k x c] */
static void mat_mul(const real *A, const real *B, real *C, int r, int k, int c) {
    for (int i = 0; i < r; ++i) {
        real *Ci = C + (size_t)i * c;
        for (int j = 0; j < c; ++j) Ci[j] = 0;
        for (int p = 0; p < k; ++p) {
            real a = A[(size_t)i * k + p];
            if (a == 0) continue; /* exact: skipping a zero multiplier adds +0 */
            const real *Bp = B + (size_t)p * c;
            for (int j = 0; j < c; ++j) Ci[j] += a * Bp[j];
        }
    }
}

static void mat_transpose(const real *A, real *At, int r, int c) {
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < c; ++j) At[(size_t)j * r + i] = A[(size_t)i * c + j];
}

/* Inverse by LU with partial (row) pivoting, then n triangular solves against the permuted
 * identity -- the algorithm class of Eigen's PartialPivLU::inverse() for dynamic sizes.
 * No singularity check, like the reference (SURVEY.md §8b "Errors"). */
static void mat_inverse(const real *A, real *Ainv, int n) {
    real *LU = mat_new(n, n);
    int *perm = (int *)malloc(sizeof(int) * (size_t)(n + 1));
    memcpy(LU, A, sizeof(real) * (size_t)n * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int c = 0; c < n; ++c) {
        int piv = c;
        real best = R_FABS(LU[(size_t)c * n + c]);
        for (int r = c + 1; r < n; ++r) {
            real v = R_FABS(LU[(size_t)r * n + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != c) {
            for (int j = 0; j < n; ++j) {
                real t = LU[(size_t)c * n + j];
                LU[(size_t)c * n + j] = LU[(size_t)piv * n + j];
                LU[(size_t)piv * n + j] = t;
            }
            int t = perm[c]; perm[c] = perm[piv]; perm[piv] = t;
        }
        real d = LU[(size_t)c * n + c];
        for (int r = c + 1; r < n; ++r) {
            real f = LU[(size_t)r * n + c] / d;
            LU[(size_t)r * n + c] = f;
            if (f == 0) continue;
            for (int j = c + 1; j < n; ++j) LU[(size_t)r * n + j] -= f * LU[(size_t)c * n + j];
        }
    }
    real *col = (real *)malloc(sizeof(real) * (size_t)(n + 1));
    for (int e = 0; e < n; ++e) {
        for (int i = 0; i < n; ++i) col[i] = (perm[i] == e) ? 1 : 0;
        for (int i = 0; i < n; ++i) {
            real s = col[i];
            for (int j = 0; j < i; ++j) s -= LU[(size_t)i * n + j] * col[j];
            col[i] = s;
        }
        for (int i = n - 1; i >= 0; --i) {
            real s = col[i];
            for (int j = i + 1; j < n; ++j) s -= LU[(size_t)i * n + j] * col[j];
            col[i] = s / LU[(size_t)i * n + i];
        }
        for (int i = 0; i < n; ++i) Ainv[(size_t)i * n + e] = col[i];
    }
    free(col); free(perm); free(LU);
}

/* ---- assembly -------------------------------------------------------------------------- */

static int fact_i32(int x) {
    int f = 1;
    for (int i = x; i > 0; --i) f = f * i;
    return f;
}

static real pow_int(real base, int e) { return R_POW(base, (real)e); } /* pow(0,0)=1 */

static void assemble_M(int o, int S, const real *T, real *M) {
    const int m = 2 * o, N = m * S;
    for (int s = 0; s < S; ++s)
        for (int j = 0; j < o; ++j)
            for (int k = j; k < m; ++k) {
                int ratio = fact_i32(k) / fact_i32(k - j);
                size_t colidx = (size_t)s * m + (size_t)(m - 1 - k);
                M[((size_t)s * m + j) * N + colidx] = (real)ratio * pow_int(0, k - j);
                M[((size_t)s * m + j + o) * N + colidx] = (real)ratio * pow_int(T[s], k - j);
            }
}

/* Column of the single 1 in row i of C_T.  Written from the waypoint picture (SURVEY.md §8a
 * A6): row i is derivative r = i % o at the start (even block) or end (odd block) of segment
 * i / (2o).  tests/test_oracle.py checks it against the reference's literal branch cascade
 * as transcribed in oracle/numpy_ref.py:build_CT. */
static int selection_column(int o, int S, int i) {
    const int N = 2 * o * S, F = 2 * o + (S - 1);
    if (i < o) return i;                       /* trajectory start block */
    if (i >= N - o) return F - o + (i - (N - o)); /* trajectory end block   */
    int seg = i / (2 * o), at_end = (i / o) % 2, r = i % o;
    int waypoint = seg + at_end;               /* interior waypoint 1..S-1 */
    if (r == 0) return o + waypoint - 1;       /* its (fixed) position     */
    return F + (waypoint - 1) * (o - 1) + (r - 1); /* its free derivative r */
}

static void assemble_Q(int o, int S, const real *T, real *Q) {
    const int m = 2 * o, N = m * S, p = m - 1;
    for (int s = 0; s < S; ++s)
        for (int i = 0; i < m; ++i)
            for (int l = 0; l < m; ++l) {
                if (m - i <= o || m - l <= o) continue;
                int e = p - i + p - l - (2 * o - 1);
                int pre = (fact_i32(p - i) / fact_i32(p - o - i)) *
                          (fact_i32(p - l) / fact_i32(p - o - l)) / e;
                Q[((size_t)s * m + i) * N + (size_t)s * m + l] = (real)pre * pow_int(T[s], e);
            }
}

static void fill_fixed(int o, int S, const real *path, const real *vel, const real *acc, int axis,
                       real *d, int V) {
    const int F = 2 * o + (S - 1);
    for (int i = 0; i < V; ++i) d[i] = 0;
    d[0] = path[axis];
    if (o >= 2) d[1] = vel[axis];
    if (o >= 3) d[2] = acc[axis];
    if (o >= 3) d[F - o + 2] = acc[3 + axis];
    if (o >= 2) d[F - o + 1] = vel[3 + axis];
    d[F - o] = path[(size_t)S * 3 + axis];
    for (int k = 1; k < S; ++k) d[o + k - 1] = path[(size_t)k * 3 + axis];
}

/* R (V x V), sel (N), Minv (N x N), optional f_valid[3] (V each) -> P[3] (N each). */
static void solve_axes(int o, int S, const real *R, const int *sel, const real *Minv,
                       const real *path, const real *vel, const real *acc,
                       real *const f_valid[3], real *const P[3]) {
    const int m = 2 * o, N = m * S, V = (S + 1) * o, F = 2 * o + (S - 1), NP = V - F;
    real *d = mat_new(V, 1), *dfull = mat_new(N, 1), *rhs = mat_new(NP + 1, 1);
    real *RPP = mat_new(NP + 1, NP + 1), *RPPinv = mat_new(NP + 1, NP + 1);
    for (int axis = 0; axis < 3; ++axis) {
        fill_fixed(o, S, path, vel, acc, axis, d, V);
        if (NP > 0) {
            /* inverse recomputed per axis, as the reference does (:579) */
            for (int i = 0; i < NP; ++i)
                for (int j = 0; j < NP; ++j) RPP[(size_t)i * NP + j] = R[(size_t)(F + i) * V + F + j];
            mat_inverse(RPP, RPPinv, NP);
            for (int j = 0; j < NP; ++j) {
                real s = 0;
                for (int i = 0; i < F; ++i) s += R[(size_t)i * V + F + j] * d[i]; /* R_FP^T d_F */
                if (f_valid) s += f_valid[axis][F + j];
                rhs[j] = s;
            }
            for (int i = 0; i < NP; ++i) {
                real s = 0;
                for (int j = 0; j < NP; ++j) s += (-RPPinv[(size_t)i * NP + j]) * rhs[j];
                d[F + i] = s;
            }
        }
        for (int i = 0; i < N; ++i) dfull[i] = d[sel[i]]; /* C_T * d_selected */
        mat_mul(Minv, dfull, P[axis], N, N, 1);
    }
    free(d); free(dfull); free(rhs); free(RPP); free(RPPinv);
}

/* R = C_T^T * (M^T)^-1 * Q * M^-1 * C_T, evaluated left to right (:511). */
static void form_R(int N, int V, const int *sel, const real *MTinv, const real *Q, const real *Minv,
                   real *R) {
    real *CT = mat_new(N, V), *CTt = mat_new(V, N);
    for (int i = 0; i < N; ++i) CT[(size_t)i * V + sel[i]] = 1;
    mat_transpose(CT, CTt, N, V);
    real *t1 = mat_new(V, N), *t2 = mat_new(V, N);
    mat_mul(CTt, MTinv, t1, V, N, N);
    mat_mul(t1, Q, t2, V, N, N);
    mat_mul(t2, Minv, t1, V, N, N);
    mat_mul(t1, CT, R, V, N, V);
    free(CT); free(CTt); free(t1); free(t2);
}

static void monomials(real t, int m, real *phi) {
    for (int i = 0; i < m; ++i) phi[i] = pow_int(t, m - 1 - i);
}

int SYM(solve)(int order, int S, const double *path_in, const double *vel_in, const double *acc_in,
               const double *time_in, double path_weight, double vel_zero_weight,
               double *coeff_out, double *max_dev_out) {
    const int o = order;
    if (o < 1 || o > 5 || S < 1) return -1; /* o>=6 overflows the reference's int prefactor */
    const int m = 2 * o, N = m * S, V = (S + 1) * o;
    real *path = mat_new(S + 1, 3), *T = mat_new(S, 1), vel[6], acc[6];
    for (int i = 0; i < (S + 1) * 3; ++i) path[i] = path_in[i];
    for (int i = 0; i < S; ++i) T[i] = time_in[i];
    for (int i = 0; i < 6; ++i) { vel[i] = vel_in[i]; acc[i] = acc_in[i]; }

    real *M = mat_new(N, N), *Mt = mat_new(N, N), *Q = mat_new(N, N);
    real *Minv = mat_new(N, N), *MTinv = mat_new(N, N), *R = mat_new(V, V);
    int *sel = (int *)malloc(sizeof(int) * (size_t)N);
    assemble_M(o, S, T, M);
    assemble_Q(o, S, T, Q);
    for (int i = 0; i < N; ++i) sel[i] = selection_column(o, S, i);
    mat_transpose(M, Mt, N, N);
    mat_inverse(M, Minv, N);
    mat_inverse(Mt, MTinv, N);

    real *P[3], *fc[3], *fv[3];
    for (int a = 0; a < 3; ++a) { P[a] = mat_new(N, 1); fc[a] = mat_new(N, 1); fv[a] = mat_new(V, 1); }
    real *best_t = mat_new(S, 1);
    real phi[16], Lb[3];

    if (path_weight > 0.0) {
        form_R(N, V, sel, MTinv, Q, Minv, R);
        solve_axes(o, S, R, sel, Minv, path, vel, acc, NULL, P);
        for (int k = 0; k < S; ++k) {
            real bt = 0, bd = -1;
            for (int s = 0; s <= 16; ++s) {
                real tt = T[k] * (real)s / (real)16;
                monomials(tt, m, phi);
                real d2 = 0;
                for (int a = 0; a < 3; ++a) {
                    real v = 0;
                    for (int i = 0; i < m; ++i) v += phi[i] * P[a][(size_t)k * m + i];
                    real L = path[k * 3 + a] + (tt / T[k]) * (path[(k + 1) * 3 + a] - path[k * 3 + a]);
                    d2 += (v - L) * (v - L);
                }
                if (d2 > bd) { bd = d2; bt = tt; }
            }
            monomials(bt, m, phi);
            for (int i = 0; i < m; ++i)
                for (int l = 0; l < m; ++l)
                    Q[((size_t)k * m + i) * N + (size_t)k * m + l] += (real)path_weight * (phi[i] * phi[l]);
            for (int a = 0; a < 3; ++a) {
                Lb[a] = path[k * 3 + a] + (bt / T[k]) * (path[(k + 1) * 3 + a] - path[k * 3 + a]);
                for (int i = 0; i < m; ++i)
                    fc[a][(size_t)k * m + i] = (real)-2.0 * (phi[i] * Lb[a]) * (real)path_weight;
            }
            best_t[k] = bt;
        }
    }
    if (vel_zero_weight > 0.0) {
        real pd[16];
        for (int k = 0; k < S; ++k)
            for (int side = 0; side < 2; ++side) {
                real t = side ? T[k] : 0;
                for (int i = 0; i < m; ++i) {
                    int power = m - 1 - i - 1;
                    pd[i] = (power < 0) ? 0 : (power == 0 ? (real)(m - 1 - i)
                                                          : (real)(m - 1 - i) * pow_int(t, power));
                }
                for (int i = 0; i < m; ++i)
                    for (int l = 0; l < m; ++l)
                        Q[((size_t)k * m + i) * N + (size_t)k * m + l] += (real)vel_zero_weight * (pd[i] * pd[l]);
            }
    }
    form_R(N, V, sel, MTinv, Q, Minv, R);
    if (path_weight > 0.0) {
        real *tmp = mat_new(N, 1);
        for (int a = 0; a < 3; ++a) {
            mat_mul(MTinv, fc[a], tmp, N, N, 1);
            for (int j = 0; j < V; ++j) fv[a][j] = 0;
            for (int i = 0; i < N; ++i) fv[a][sel[i]] += tmp[i];
        }
        free(tmp);
    }
    solve_axes(o, S, R, sel, Minv, path, vel, acc, path_weight > 0.0 ? fv : NULL, P);

    real max_dev = 0;
    for (int k = 0; k < S; ++k) {
        monomials(best_t[k], m, phi);
        real d2 = 0, len2 = 0;
        for (int a = 0; a < 3; ++a) {
            real v = 0;
            for (int i = 0; i < m; ++i) v += phi[i] * P[a][(size_t)k * m + i];
            real dp = path[(k + 1) * 3 + a] - path[k * 3 + a];
            real L = path[k * 3 + a] + (best_t[k] / T[k]) * dp;
            d2 += (v - L) * (v - L);
            len2 += dp * dp;
        }
        real seg_len = R_SQRT(len2), ratio = 0;
        if (seg_len > (real)1e-6) ratio = R_SQRT(d2) / seg_len;
        if (ratio > max_dev) max_dev = ratio;
    }
    if (max_dev_out) *max_dev_out = (double)max_dev;
    for (int k = 0; k < S; ++k)
        for (int a = 0; a < 3; ++a)
            for (int i = 0; i < m; ++i)
                coeff_out[((size_t)k * 3 + a) * m + i] = (double)P[a][(size_t)k * m + i];

    for (int a = 0; a < 3; ++a) { free(P[a]); free(fc[a]); free(fv[a]); }
    free(best_t); free(sel); free(M); free(Mt); free(Q); free(Minv); free(MTinv); free(R);
    free(path); free(T);
    return 0;
}

/* Batch driver (uniform S).  bc = [B or 1][4][3] rows v0, v1, a0, a1.  OpenMP over the batch. */
int SYM(solve_batch)(int order, int S, long B, const double *path, const double *time,
                     const double *bc, int bc_broadcast, double path_weight, double vel_zero_weight,
                     double *coeff, double *max_dev, int nthreads) {
    const int m = 2 * order;
    int rc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long b = 0; b < B; ++b) {
        const double *c = bc + (bc_broadcast ? 0 : (size_t)b * 12);
        double vel[6] = {c[0], c[1], c[2], c[3], c[4], c[5]};
        double acc[6] = {c[6], c[7], c[8], c[9], c[10], c[11]};
        double md = 0;
        int r = SYM(solve)(order, S, path + (size_t)b * (S + 1) * 3, vel, acc, time + (size_t)b * S,
                           path_weight, vel_zero_weight, coeff + (size_t)b * S * 3 * m, &md);
        if (max_dev) max_dev[b] = md;
        if (r) rc = r;
    }
    return rc;
}

#ifndef CSP_ORACLE_LONG_DOUBLE
/* T_i = max(len_i / V_avg, min_time_s)  (:63-72) */
int csp_oracle_time_alloc(int W, const double *path, double v_avg, double min_time_s, double *time) {
    for (int i = 0; i + 1 < W; ++i) {
        double dx = path[(i + 1) * 3] - path[i * 3], dy = path[(i + 1) * 3 + 1] - path[i * 3 + 1],
               dz = path[(i + 1) * 3 + 2] - path[i * 3 + 2];
        double len = sqrt(dx * dx + dy * dy + dz * dz);
        double t = (v_avg > 1e-6) ? (len / v_avg) : min_time_s;
        if (t < min_time_s) t = min_time_s;
        time[i] = t;
    }
    return 0;
}

static void eval_poly(const double *coeff, int m, int seg, double t, double out[3]) {
    for (int a = 0; a < 3; ++a) {
        double v = 0.0;
        for (int k = 0; k < m; ++k) v += coeff[((size_t)seg * 3 + a) * m + k] * pow(t, (double)(m - 1 - k));
        out[a] = v;
    }
}

/* The sampling loop of GenerateTrajectoryMatrix (:97-161) on GIVEN coefficients: candidates at dt = min(0.1, T/10), evaluated
 * term by term with pow() like the reference's eval lambda (:104-117), a candidate kept when its distance to the previously
 * kept one is >= sample_distance (:142-150), the end point appended unless it repeats the last sample (:157-160). */
static long sample_polynomials(int S, int m, const double *coeff, const double *T, double sample_distance, double *samples, long cap) {
    long n = 0;
    double prev[3] = {0, 0, 0}, cur[3], last[3] = {0, 0, 0};
    for (int seg = 0; seg < S; ++seg) {
        double Ts = T[seg], dt = 0.1;
        if (dt > Ts / 10.0) dt = Ts / 10.0;
        double t0[3];
        eval_poly(coeff, m, seg, 0.0, t0);
        if (n == 0) { if (n < cap) memcpy(samples + n * 3, t0, 24); memcpy(last, t0, 24); ++n; }
        memcpy(prev, t0, 24);
        for (double t = dt; t <= Ts + 1e-12; t += dt) {
            double tt = t < Ts ? t : Ts;
            eval_poly(coeff, m, seg, tt, cur);
            double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
            if (sqrt(dx * dx + dy * dy + dz * dz) >= sample_distance) {
                memcpy(prev, cur, 24);
                if (n < cap) memcpy(samples + n * 3, cur, 24);
                memcpy(last, cur, 24);
                ++n;
            }
        }
        if (seg == S - 1) {
            eval_poly(coeff, m, seg, Ts, cur);
            double dx = last[0] - cur[0], dy = last[1] - cur[1], dz = last[2] - cur[2];
            if (n == 0 || sqrt(dx * dx + dy * dy + dz * dz) > 1e-6) {
                if (n < cap) memcpy(samples + n * 3, cur, 24);
                ++n;
            }
        }
    }
    return n;
}

/* The same loop as an entry point of its own (round 3): lets a test feed the oracle's pow()-based evaluation and the HIP
 * sampler's power ladder THE SAME coefficients, so that what differs is the evaluation alone (the 1-ulp band of the keep test). */
long csp_oracle_sample(int S, int order, const double *coeff, const double *times, double sample_distance, double *samples, long cap) {
    return sample_polynomials(S, 2 * order, coeff, times, sample_distance, samples, cap);
}

/* GenerateTrajectoryMatrix (:22-206): time allocation, re-solve loop, fixed-distance thinning.
 * cfg = {order, path_weight, vel_zero_weight, V_avg, min_time_s, sample_distance};
 * bc rows v0, v1, a0, a1.  Writes at most cap samples; returns the sample count (or -1 on the
 * reference's bad-shape path, :54-57).  stats = {final vel_zero_weight, iterations, max_dev,
 * max climb rate, min turn radius}. */
long csp_oracle_generate_trajectory(int W, const double *path, int order, double path_weight,
                                    double vel_zero_weight, double v_avg, double min_time_s,
                                    double sample_distance, const double *bc, double *samples,
                                    long cap, double *coeff_out, double *time_out, double *stats) {
    if (W < 2) return -1;
    const int S = W - 1, m = 2 * order;
    double *T = (double *)malloc(sizeof(double) * (size_t)S);
    double *coeff = (double *)malloc(sizeof(double) * (size_t)S * 3 * m);
    csp_oracle_time_alloc(W, path, v_avg, min_time_s, T);
    double vel[6] = {bc[0], bc[1], bc[2], bc[3], bc[4], bc[5]};
    double acc[6] = {bc[6], bc[7], bc[8], bc[9], bc[10], bc[11]};
    double max_dev = 0.0;
    int iter = 0;
    for (;;) {
        if (csp_oracle_solve(order, S, path, vel, acc, T, path_weight, vel_zero_weight, coeff, &max_dev)) {
            free(T); free(coeff); return -1;
        }
        if (max_dev > 0.2 && iter < 10) {
            vel_zero_weight = (vel_zero_weight < 1e-6) ? 0.01 : vel_zero_weight * 2.0;
            ++iter;
        } else break;
    }
    long n = sample_polynomials(S, m, coeff, T, sample_distance, samples, cap);
    if (stats) {
        double max_climb = 0.0, min_r = 1.0e12;
        long lim = n < cap ? n : cap;
        for (long i = 0; i + 1 < lim; ++i) {
            const double *p1 = samples + i * 3, *p2 = samples + (i + 1) * 3;
            double dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = fabs(p2[2] - p1[2]);
            double hd = sqrt(dx * dx + dy * dy);
            if (hd > 1e-6 && dz / hd > max_climb) max_climb = dz / hd;
            if (i > 0) {
                const double *p0 = samples + (i - 1) * 3;
                double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                double w[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                double v[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
                double a = sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                double b = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                double c = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
                double area = 0.5 * sqrt(cx * cx + cy * cy + cz * cz);
                if (area > 1e-8) { double Rr = a * b * c / (4.0 * area); if (Rr < min_r) min_r = Rr; }
            }
        }
        stats[0] = vel_zero_weight; stats[1] = (double)iter; stats[2] = max_dev;
        stats[3] = max_climb; stats[4] = min_r;
    }
    if (coeff_out) memcpy(coeff_out, coeff, sizeof(double) * (size_t)S * 3 * m);
    if (time_out) memcpy(time_out, T, sizeof(double) * (size_t)S);
    free(T); free(coeff);
    return n;
}

int csp_oracle_max_threads(void) {
#ifdef _OPENMP
    extern int omp_get_max_threads(void);
    return omp_get_max_threads();
#else
    return 1;
#endif
}
#endif
