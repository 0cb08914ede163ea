// minimum_snap.hpp -- host-side class shim with the surface of the reference's
// math_util/minimum_snap.hpp (MinimumSnapConfig :9-33, TrajectoryGeneratorTool :36-63), so that
// UavPathPlanner::Minisnap_3D / Minisnap_EN (uavPathPlanning.cpp:4401-4474), which hold a
// `TrajectoryGeneratorTool generator_` by value (uavPathPlanning.hpp:294), compile unchanged.
//
// Everything numeric goes through the C-ABI (include/csp_minsnap.h) to the HIP kernels: there is
// no host solver, sampler or time allocation in here.  SolveQPClosedForm -> csp_minsnap_solve_batch;
// GenerateTrajectoryMatrix -> csp_minsnap_generate_batch (time allocation + the <=10x re-solve loop,
// minimum_snap.cpp:59-90, then sampling / distance thinning / statistics, :97-205, in one call).
// The host only marshals matrices.
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
    double last_vel_zero_weight = 0.0;   // weight the final solve used (after the :80-90 loop)
    int last_iterations = 0;

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
        std::vector<double> wp((size_t)(S + 1) * 3), bc(12);
        for (int i = 0; i <= S; ++i)
            for (int a = 0; a < 3; ++a) wp[(size_t)i * 3 + a] = Path(i, a);
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
        // the whole routine in ONE call: time allocation (:63-72), re-solve loop (:80-90), sampling, thinning and statistics
        // (:97-195); capacity = every evaluation point (degenerate / absurd segment times get no candidates on the device)
        const int64_t cap = csp_minsnap_sample_capacity(&d, wp.data(), V_avg, min_time_s);
        if (cap < 1) { last_status = CSP_ERR_INVALID_ARG; return MatrixXd(); }
        std::vector<double> samples((size_t)cap * 3);
        int32_t count = 0, iters = 0;
        double stats[2] = {0.0, 1.0e12};
        double max_dev = 0.0, vw_final = vel_zero_weight;
        last_status = csp_minsnap_generate_batch(&d, wp.data(), V_avg, min_time_s, bc.data(), sample_distance, cap, samples.data(),
                                                 &count, stats, nullptr, nullptr, &max_dev, &vw_final, &iters, nullptr, nullptr, 0,
                                                 nullptr);
        if (last_status != CSP_OK || count > cap) {
            std::cerr << "TrajectoryGeneratorTool::GenerateTrajectoryMatrix: " << csp_minsnap_strerror(last_status)
                      << " " << csp_minsnap_last_hip_error() << std::endl;
            return MatrixXd();
        }
        last_vel_zero_weight = vw_final;
        last_iterations = iters;
        if (verbose && iters > 0) std::cout << "vel_zero_weight increased " << iters << "x to " << vw_final << " (max_dev=" << max_dev << ")" << std::endl;
        last_max_climb_rate = stats[0];
        last_min_turn_radius = stats[1];
        const long n = (long)count;
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
