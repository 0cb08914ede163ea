/* sanitize_driver.c -- TEST INFRASTRUCTURE.  Runs the CPU oracle (dense_oracle.c, geo_oracle.c, alt_oracle.c) under
 * AddressSanitizer + UndefinedBehaviorSanitizer (host-side sanitizers on the CPU build; the GPU pool
 * allows none).  Built and run by `make -C oracle sanitize` / tests/test_sanitizers.py.  Covers every order, single- and
 * multi-segment shapes, both penalties, the long-double build, the batch entry with OpenMP threads, the trajectory
 * generator with a capacity smaller than the sample count, the coordinate transforms and the altitude solves. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int csp_oracle_solve(int, int, const double *, const double *, const double *, const double *, double, double, double *, double *);
int csp_oracle_ld_solve(int, int, const double *, const double *, const double *, const double *, double, double, double *, double *);
int csp_oracle_solve_batch(int, int, long, const double *, const double *, const double *, int, double, double, double *, double *, int);
int csp_oracle_time_alloc(int, const double *, double, double, double *);
long csp_oracle_generate_trajectory(int, const double *, int, double, double, double, double, double, const double *, double *, long,
                                    double *, double *, double *);
int csp_oracle_wgs84_to_enu(const double *, const double *, double *, long);
int csp_oracle_enu_to_wgs84(const double *, const double *, double *, long);
int csp_oracle_alt_optimize(int, const double *, const double *, double, double, double, double, double *);
int csp_oracle_alt_global_smooth(int, const double *, const double *, double, double, double *);

static unsigned long long s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }

int main(void) {
    int bad = 0;
    for (int order = 1; order <= 5; ++order)
        for (int S = 1; S <= 9; S += (S < 3 ? 1 : 3)) {
            const int m = 2 * order;
            double *path = malloc(sizeof(double) * (size_t)(S + 1) * 3), *tm = malloc(sizeof(double) * (size_t)S);
            double *co = malloc(sizeof(double) * (size_t)S * 3 * m), vel[6], acc[6], md = 0.0;
            for (int i = 0; i < (S + 1) * 3; ++i) path[i] = (i >= 3 ? path[i - 3] : 0.0) + rnd() * 2.0 - 1.0;
            for (int i = 0; i < S; ++i) tm[i] = 0.5 + 1.5 * rnd();
            for (int i = 0; i < 6; ++i) { vel[i] = rnd() - 0.5; acc[i] = rnd() - 0.5; }
            for (int pen = 0; pen < 3; ++pen) {
                const double pw = pen == 2 ? 0.3 : 0.0, vw = pen >= 1 ? 0.05 : 0.0;
                bad |= csp_oracle_solve(order, S, path, vel, acc, tm, pw, vw, co, &md);
                bad |= csp_oracle_ld_solve(order, S, path, vel, acc, tm, pw, vw, co, &md);
                for (int i = 0; i < S * 3 * m; ++i) bad |= !isfinite(co[i]);
            }
            free(path); free(tm); free(co);
        }
    {   /* batch entry, shared and per-trajectory boundary conditions, 4 threads */
        const int S = 6, order = 4, B = 23, m = 8;
        double *path = malloc(sizeof(double) * B * (S + 1) * 3), *tm = malloc(sizeof(double) * B * S), *bc = calloc((size_t)B * 12, sizeof(double));
        double *co = malloc(sizeof(double) * B * S * 3 * m), *md = malloc(sizeof(double) * B);
        for (int i = 0; i < B * (S + 1) * 3; ++i) path[i] = rnd() * 10.0;
        for (int i = 0; i < B * S; ++i) tm[i] = 0.5 + rnd();
        for (int i = 0; i < B * 12; ++i) bc[i] = rnd() - 0.5;
        bad |= csp_oracle_solve_batch(order, S, B, path, tm, bc, 1, 0.0, 0.0, co, md, 4);
        bad |= csp_oracle_solve_batch(order, S, B, path, tm, bc, 0, 0.2, 0.01, co, md, 4);
        free(path); free(tm); free(bc); free(co); free(md);
    }
    {   /* GenerateTrajectoryMatrix restatement: capacity both ample and too small */
        const int W = 7, order = 3;
        double path[21], bc[12] = {0}, coeff[6 * 3 * 6], T[6], stats[5];
        for (int i = 0; i < 21; ++i) path[i] = (i >= 3 ? path[i - 3] : 0.0) + (rnd() - 0.3) * 40.0;
        double *samples = malloc(sizeof(double) * 3 * 100000);
        long n = csp_oracle_generate_trajectory(W, path, order, 0.4, 0.0, 5.0, 0.1, 0.7, bc, samples, 100000, coeff, T, stats);
        bad |= n < 2;
        double small[3 * 5];
        long n2 = csp_oracle_generate_trajectory(W, path, order, 0.4, 0.0, 5.0, 0.1, 0.7, bc, small, 5, coeff, T, stats);
        bad |= n2 != n;
        bad |= csp_oracle_generate_trajectory(1, path, order, 0.0, 0.0, 5.0, 0.1, 0.7, bc, small, 5, coeff, T, stats) != -1;
        free(samples);
    }
    {   /* coordinate transforms and the altitude optimiser's pentadiagonal solves */
        double ref[3] = {111.5, 40.8, 0.0}, lla[30], enu[30], back[30];
        for (int i = 0; i < 10; ++i) { lla[i * 3] = 111.5 + rnd() * 0.2; lla[i * 3 + 1] = 40.8 + rnd() * 0.2; lla[i * 3 + 2] = 1000.0 + rnd() * 500.0; }
        bad |= csp_oracle_wgs84_to_enu(lla, ref, enu, 10);
        bad |= csp_oracle_enu_to_wgs84(enu, ref, back, 10);
        for (int i = 0; i < 30; ++i) bad |= fabs(back[i] - lla[i]) > 1e-6;
        for (int n = 1; n <= 40; n += 13) {
            double *xyz = malloc(sizeof(double) * 3 * (size_t)n), *el = malloc(sizeof(double) * (size_t)n), *out = malloc(sizeof(double) * (size_t)n);
            for (int i = 0; i < n; ++i) { xyz[i * 3] = i * 30.0; xyz[i * 3 + 1] = rnd() * 5.0; xyz[i * 3 + 2] = 1500.0 + rnd() * 100.0; el[i] = (i % 5) ? 1300.0 + rnd() * 50.0 : NAN; }
            bad |= csp_oracle_alt_optimize(n, xyz, el, 1.0, 0.2, 50.0, 2.0, out);
            bad |= csp_oracle_alt_global_smooth(n, out, xyz, 1.0, 2.0, out) < 0;
            free(xyz); free(el); free(out);
        }
    }
    printf("sanitize_driver: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
