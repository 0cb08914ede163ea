// shim_selftest.cpp -- compiles the class shim exactly the way the reference's planner uses it
// and doubles as the marshalling harness of SURVEY.md §8a A14: the bodies of minisnap_3d /
// minisnap_en below reproduce UavPathPlanner::Minisnap_3D / Minisnap_EN
// (uavPathPlanning.cpp:4401-4474) against the shim.
//
//   shim_selftest kat                      -> K1 known answers through SolveQPClosedForm
//   shim_selftest plan3d|planen <file>     -> reads "order pw vw V_avg min_t sample_dist n" then n
//                                             rows "e n u"; prints the sampled ENU points
//   shim_selftest bezier <file>            -> same input, Bezier::GenerateTrajectoryMatrix
//   shim_selftest time3d <file> <reps>     -> wall time of one Minisnap_3D call (one flight: the reference's own
//                                             call pattern), averaged over <reps> after 5 warm-up calls; one JSON line
// Exit code 3 = no usable device (the shim has no CPU fallback).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>

#include "math_util/bezier.hpp"
#include "math_util/minimum_snap.hpp"

struct ENUPoint { double east = 0, north = 0, up = 0; };

struct Planner {
    TrajectoryGeneratorTool generator_;   // by value, like uavPathPlanning.hpp:294
    MinimumSnapConfig minimum_snap;

    std::vector<ENUPoint> minisnap_3d(std::vector<ENUPoint> wps, double distance_, double v_avg_override) {
        std::vector<ENUPoint> result;
        const int n = (int)wps.size();
        if (n < 2) return result;
        csp_host::MatrixXd route(n, 3);
        for (int i = 0; i < n; ++i) { route(i, 0) = wps[i].east; route(i, 1) = wps[i].north; route(i, 2) = wps[i].up; }
        csp_host::MatrixXd s = generator_.GenerateTrajectoryMatrix(route, minimum_snap, distance_, v_avg_override);
        for (long i = 0; i < s.rows(); ++i) { ENUPoint p; p.east = s(i, 0); p.north = s(i, 1); p.up = s(i, 2); result.push_back(p); }
        return result;
    }
    std::vector<ENUPoint> minisnap_en(std::vector<ENUPoint> wps, double distance_, double v_avg_override) {
        std::vector<ENUPoint> result;
        const int n = (int)wps.size();
        if (n < 2) return result;
        csp_host::MatrixXd route(n, 3);
        for (int i = 0; i < n; ++i) { route(i, 0) = wps[i].east; route(i, 1) = wps[i].north; route(i, 2) = 0.0; }
        csp_host::MatrixXd s = generator_.GenerateTrajectoryMatrix(route, minimum_snap, distance_, v_avg_override);
        for (long i = 0; i < s.rows(); ++i) { ENUPoint p; p.east = s(i, 0); p.north = s(i, 1); p.up = wps[0].up; result.push_back(p); }
        return result;
    }
};

static int run_kat() {
    TrajectoryGeneratorTool g;
    const double kat4[8] = {-20, 70, -84, 35, 0, 0, 0, 0};
    csp_host::MatrixXd path(2, 3), vel = csp_host::MatrixXd::Zero(2, 3), acc = csp_host::MatrixXd::Zero(2, 3);
    for (int a = 0; a < 3; ++a) { path(0, a) = 0.0; path(1, a) = 1.0; }
    csp_host::VectorXd T(1);
    T(0) = 1.0;
    double md = -1.0;
    csp_host::MatrixXd c = g.SolveQPClosedForm(4, path, vel, acc, T, 0.0, 0.0, &md);
    if (c.size() == 0) return g.last_status == CSP_ERR_NO_DEVICE ? 3 : 2;
    double worst = 0.0;
    for (int a = 0; a < 3; ++a)
        for (int i = 0; i < 8; ++i) worst = std::fmax(worst, std::fabs(c(0, a * 8 + i) - kat4[i]));
    std::printf("K1 order 4: max abs err %.3e, max_dev %.3e\n", worst, md);
    // bad shapes come back empty, like the reference (minimum_snap.cpp:54-57)
    MinimumSnapConfig cfg;
    csp_host::MatrixXd bad(1, 3);
    if (g.GenerateTrajectoryMatrix(bad, cfg).size() != 0) return 4;
    return worst < 1e-12 ? 0 : 1;
}

int main(int argc, char **argv) {
    const std::string mode = argc > 1 ? argv[1] : "kat";
    if (mode == "kat") return run_kat();
    if (argc < 3) return 64;
    std::ifstream in(argv[2]);
    Planner pl;
    int n = 0;
    in >> pl.minimum_snap.order >> pl.minimum_snap.path_weight >> pl.minimum_snap.vel_zero_weight >>
        pl.minimum_snap.V_avg >> pl.minimum_snap.min_time_s >> pl.minimum_snap.sample_distance >> n;
    std::vector<ENUPoint> wps((size_t)n);
    for (auto &p : wps) in >> p.east >> p.north >> p.up;
    if (mode == "bezier") {
        math_util::Bezier bz;
        math_util::BezierConfig bc;
        bc.min_radius = pl.minimum_snap.V_avg;  // harness convention: the V_avg slot carries min_radius
        bz.SetConfig(bc);
        csp_host::MatrixXd route(n, 3);
        for (int i = 0; i < n; ++i) { route(i, 0) = wps[i].east; route(i, 1) = wps[i].north; route(i, 2) = wps[i].up; }
        csp_host::MatrixXd s = bz.GenerateTrajectoryMatrix(route, "", pl.minimum_snap.sample_distance, -1.0);
        for (long i = 0; i < s.rows(); ++i) std::printf("%.17g %.17g %.17g\n", s(i, 0), s(i, 1), s(i, 2));
        return 0;
    }
    if (mode == "time3d") {
        const int reps = argc > 3 ? std::atoi(argv[3]) : 100;
        std::vector<ENUPoint> o;
        for (int i = 0; i < 5; ++i) o = pl.minisnap_3d(wps, -1.0, -1.0);
        if (o.empty()) return pl.generator_.last_status == CSP_ERR_NO_DEVICE ? 3 : 2;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i) o = pl.minisnap_3d(wps, -1.0, -1.0);
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        std::printf("{\"plan_plus_sample_us\": %.1f, \"samples\": %zu, \"resolve_iterations\": %d, \"reps\": %d}\n", us, o.size(),
                    pl.generator_.last_iterations, reps);
        return 0;
    }
    std::vector<ENUPoint> out = (mode == "planen") ? pl.minisnap_en(wps, -1.0, -1.0) : pl.minisnap_3d(wps, -1.0, -1.0);
    if (out.empty() && pl.generator_.last_status == CSP_ERR_NO_DEVICE) return 3;
    for (const auto &p : out) std::printf("%.17g %.17g %.17g\n", p.east, p.north, p.up);
    std::fprintf(stderr, "climb %.17g radius %.17g\n", pl.generator_.last_max_climb_rate, pl.generator_.last_min_turn_radius);
    return 0;
}
