// minimum_snap.hpp -- host-side class shim with the surface of the reference's
// math_util/minimum_snap.hpp (MinimumSnapConfig :9-33, TrajectoryGeneratorTool :36-63), so that
// UavPathPlanner::Minisnap_3D / Minisnap_EN (uavPathPlanning.cpp:4401-4474), which hold a
// `TrajectoryGeneratorTool generator_` by value (uavPathPlanning.hpp:294), compile unchanged.
//
// Every solve goes through the C-ABI (include/csp_minsnap.h) to the HIP kernels: there is no
// host solver in here.  What stays on the host is the wrapper logic of GenerateTrajectoryMatrix
// (time allocation, the <=10x re-solve loop, polynomial sampling / distance thinning / stats,
// minimum_snap.cpp:59-205) -- scalar control flow around the solve; its batched GPU form is the
// "next" row N1 of SURVEY.md §8f.
//
// Matrix types: real Eigen when <Eigen/Dense> exists (then the signatures are the reference's,
// token for token), otherwise the bundled csp_host mini types (same member names).
// Unlike the reference the solver prints nothing unless TrajectoryGeneratorTool::verbose is set.
#ifndef CSP_HOST_MINIMUM_SNAP_HPP_
#define CSP_HOST_MINIMUM_SNAP_HPP_

#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <vector>

#include "csp_minsnap.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define CSP_HOST_HAVE_EIGEN 1
#endif
#endif

#ifdef CSP_HOST_HAVE_EIGEN
namespace csp_host {
using MatrixXd = Eigen::MatrixXd;
using VectorXd = Eigen::VectorXd;
using Vector3d = Eigen::Vector3d;
}
#else
#include "mini_matrix.hpp"
#endif

// Same fields, defaults and order as the reference struct (minimum_snap.hpp:9-33).
struct MinimumSnapConfig {
    int order = 3;
    double path_weight = 0.0;
    double vel_zero_weight = 0.0;
    double V_avg = 5.0;
    double min_time_s = 0.1;
    double sample_distance = 1.0;
    csp_host::Vector3d start_vel = csp_host::Vector3d::Zero();
    csp_host::Vector3d end_vel = csp_host::Vector3d::Zero();
    csp_host::Vector3d start_acc = csp_host::Vector3d::Zero();
    csp_host::Vector3d end_acc = csp_host::Vector3d::Zero();
};

class TrajectoryGeneratorTool {
public:
    using MatrixXd = csp_host::MatrixXd;
    using VectorXd = csp_host::VectorXd;

    TrajectoryGeneratorTool() = default;
    ~TrajectoryGeneratorTool() = default;

    // opt-in chatter (the reference prints unconditionally, minimum_snap.cpp:239,472,621)
    bool verbose = false;
    // filled by GenerateTrajectoryMatrix (the reference only prints them, :194-195)
    double last_max_climb_rate = 0.0;
    double last_min_turn_radius = 1.0e12;
    int last_status = CSP_OK;

    // Replaces minimum_snap.cpp:227-649.  Path W x 3, Vel/Acc 2 x 3 (row 0 start, row 1 end),
    // Time S.  Returns PolyCoeff S x 3*2*order, highest power first (:220-223); an EMPTY matrix
    // when the device call fails (the reference has no such failure mode; see last_status).
    MatrixXd SolveQPClosedForm(int order, const MatrixXd &Path, const MatrixXd &Vel, const MatrixXd &Acc,
                               const VectorXd &Time, double path_weight = 0.0, double vel_zero_weight = 0.0,
                               double *max_deviation = nullptr) {
        const int S = (int)Time.size();
        const int m = 2 * order;
        if (S < 1 || Path.rows() != S + 1 || Path.cols() < 3) { last_status = CSP_ERR_INVALID_ARG; return MatrixXd(); }
        std::vector<double> wp((size_t)(S + 1) * 3), tm((size_t)S), bc(12), co((size_t)S * 3 * m);
        for (int i = 0; i <= S; ++i)
            for (int a = 0; a < 3; ++a) wp[(size_t)i * 3 + a] = Path(i, a);
        for (int i = 0; i < S; ++i) tm[(size_t)i] = Time(i);
        for (int a = 0; a < 3; ++a) {
            bc[0 + a] = Vel(0, a); bc[3 + a] = Vel(1, a);
            bc[6 + a] = Acc(0, a); bc[9 + a] = Acc(1, a);
        }
        csp_minsnap_desc d;
        std::memset(&d, 0, sizeof d);
        d.abi_version = CSP_MINSNAP_ABI_VERSION;
        d.dtype = CSP_DTYPE_F64;
        d.order = order;
        d.num_segments = S;
        d.batch = 1;
        d.path_weight = path_weight;
        d.vel_zero_weight = vel_zero_weight;
        d.mem_space = CSP_MEM_HOST;
        d.device_id = -1;
        double md = 0.0;
        last_status = csp_minsnap_solve_batch(&d, wp.data(), tm.data(), bc.data(), co.data(), &md, nullptr,
                                              nullptr, 0, nullptr);
        if (last_status != CSP_OK) {
            std::cerr << "TrajectoryGeneratorTool::SolveQPClosedForm: " << csp_minsnap_strerror(last_status)
                      << " " << csp_minsnap_last_hip_error() << std::endl;
            return MatrixXd();
        }
        if (max_deviation) *max_deviation = md;
        MatrixXd PolyCoeff = MatrixXd::Zero(S, 3 * m);
        for (int k = 0; k < S; ++k)
            for (int j = 0; j < 3 * m; ++j) PolyCoeff(k, j) = co[(size_t)k * 3 * m + j];
        if (verbose) std::cout << "input points number : " << Path.rows() << "  path_weight: " << path_weight << std::endl;
        return PolyCoeff;
    }

    // Replaces minimum_snap.cpp:22-206.
    MatrixXd GenerateTrajectoryMatrix(const MatrixXd &Path, const MinimumSnapConfig &cfg,
                                      double sample_distance_override = -1.0, double v_avg_override = -1.0) {
        const int order = cfg.order;
        double V_avg = cfg.V_avg;
        const double min_time_s = cfg.min_time_s;
        double sample_distance = cfg.sample_distance;
        MatrixXd Vel = MatrixXd::Zero(2, 3), Acc = MatrixXd::Zero(2, 3);
        for (int a = 0; a < 3; ++a) {
            Vel(0, a) = cfg.start_vel(a); Vel(1, a) = cfg.end_vel(a);
            Acc(0, a) = cfg.start_acc(a); Acc(1, a) = cfg.end_acc(a);
        }
        this->path_weight = cfg.path_weight;  // the reference mutates its member too (:38)
        double vel_zero_weight = cfg.vel_zero_weight;
        if (sample_distance_override > 0.0) sample_distance = sample_distance_override;
        if (v_avg_override > 0.0) V_avg = v_avg_override;
        if (Path.rows() < 2 || Path.cols() < 3) {
            std::cerr << "TrajectoryGeneratorTool::GenerateTrajectoryMatrix: Path must be (N>=2 x 3)" << std::endl;
            return MatrixXd();
        }
        const int S = (int)Path.rows() - 1;
        VectorXd Time(S);
        for (int i = 0; i < S; ++i) {  // :63-72
            const double dx = Path(i + 1, 0) - Path(i, 0), dy = Path(i + 1, 1) - Path(i, 1), dz = Path(i + 1, 2) - Path(i, 2);
            double t = (V_avg > 1e-6) ? std::sqrt(dx * dx + dy * dy + dz * dz) / V_avg : min_time_s;
            Time(i) = t < min_time_s ? min_time_s : t;
        }
        MatrixXd polyCoeff;
        double max_dev = 0.0;
        for (int iter = 0;; ++iter) {  // :80-90
            polyCoeff = SolveQPClosedForm(order, Path, Vel, Acc, Time, path_weight, vel_zero_weight, &max_dev);
            if (polyCoeff.size() == 0) return MatrixXd();
            if (max_dev > 0.2 && iter < 10) {
                vel_zero_weight = (vel_zero_weight < 1e-6) ? 0.01 : vel_zero_weight * 2.0;
                if (verbose) std::cout << "Iteration " << iter + 1 << ": max_dev=" << max_dev << " > 0.2. Increasing vel_zero_weight to " << vel_zero_weight << std::endl;
            } else {
                break;
            }
        }
        const int m = 2 * order;
        auto eval = [&](int seg, double t, double out[3]) {  // :104-117 (std::pow per term, same order)
            for (int dim = 0; dim < 3; ++dim) {
                double val = 0.0;
                for (int k = 0; k < m; ++k) val += polyCoeff(seg, dim * m + k) * std::pow(t, m - 1 - k);
                out[dim] = val;
            }
        };
        std::vector<double> samples;
        samples.reserve(3000);
        auto push = [&](const double p[3]) { samples.push_back(p[0]); samples.push_back(p[1]); samples.push_back(p[2]); };
        double prev[3] = {0, 0, 0}, cur[3];
        for (int seg = 0; seg < S; ++seg) {  // :123-161
            const double T = Time(seg);
            double dt = 0.1;
            if (dt > T / 10.0) dt = T / 10.0;
            double t0[3];
            eval(seg, 0.0, t0);
            if (samples.empty()) push(t0);
            std::memcpy(prev, t0, sizeof prev);
            for (double t = dt; t <= T + 1e-12; t += dt) {
                eval(seg, t < T ? t : T, cur);
                const double dx = cur[0] - prev[0], dy = cur[1] - prev[1], dz = cur[2] - prev[2];
                if (std::sqrt(dx * dx + dy * dy + dz * dz) >= sample_distance) { std::memcpy(prev, cur, sizeof prev); push(cur); }
            }
            if (seg == S - 1) {
                eval(seg, T, cur);
                const size_t n = samples.size();
                const double dx = samples[n - 3] - cur[0], dy = samples[n - 2] - cur[1], dz = samples[n - 1] - cur[2];
                if (std::sqrt(dx * dx + dy * dy + dz * dz) > 1e-6) push(cur);
            }
        }
        const long n = (long)(samples.size() / 3);
        last_max_climb_rate = 0.0;  // :163-195
        last_min_turn_radius = 1.0e12;
        for (long i = 0; i + 1 < n; ++i) {
            const double *p1 = &samples[(size_t)i * 3], *p2 = &samples[(size_t)(i + 1) * 3];
            const double dx = p2[0] - p1[0], dy = p2[1] - p1[1], dz = std::fabs(p2[2] - p1[2]);
            const double hd = std::sqrt(dx * dx + dy * dy);
            if (hd > 1e-6 && dz / hd > last_max_climb_rate) last_max_climb_rate = dz / hd;
            if (i > 0) {
                const double *p0 = &samples[(size_t)(i - 1) * 3];
                const double u[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
                const double w[3] = {p2[0] - p0[0], p2[1] - p0[1], p2[2] - p0[2]};
                const double a = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
                const double b = std::sqrt((p2[0] - p1[0]) * (p2[0] - p1[0]) + (p2[1] - p1[1]) * (p2[1] - p1[1]) + (p2[2] - p1[2]) * (p2[2] - p1[2]));
                const double c = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                const double cx = u[1] * w[2] - u[2] * w[1], cy = u[2] * w[0] - u[0] * w[2], cz = u[0] * w[1] - u[1] * w[0];
                const double area = 0.5 * std::sqrt(cx * cx + cy * cy + cz * cz);
                if (area > 1e-8) { const double R = a * b * c / (4.0 * area); if (R < last_min_turn_radius) last_min_turn_radius = R; }
            }
        }
        if (verbose) {
            std::cout << "Trajectory Max Climb/Descent Rate: " << last_max_climb_rate << std::endl;
            std::cout << "Trajectory Min Turn Radius: " << last_min_turn_radius << std::endl;
        }
        MatrixXd out(n, 3);
        for (long i = 0; i < n; ++i)
            for (int a = 0; a < 3; ++a) out(i, a) = samples[(size_t)i * 3 + a];
        return out;
    }

private:
    double path_weight = 0.0;
};

#endif  // CSP_HOST_MINIMUM_SNAP_HPP_
