// bezier.hpp -- class-surface mirror of the reference's math_util::Bezier
// (math_util/bezier.hpp:98-119, math_util/bezier.cpp:6-190) so that UavPathPlanner::Bezier_3D
// (uavPathPlanning.cpp:4477-4510, marked unused there) keeps compiling against this package.
// It is NOT part of the accelerated path: no linear solve, pure per-segment evaluation on the
// host.  Behaviour follows the reference: control-arm factor k grows
// from 1/3 in steps of 0.02 (at most 10 tries, capped at 0.45) until the curvature at
// t in {0, 0.5, 1} respects min_radius; float-accumulated parameter stepping; failed segments
// contribute their end point.
#ifndef CSP_HOST_BEZIER_HPP_
#define CSP_HOST_BEZIER_HPP_

#include <cmath>
#include <string>
#include <vector>

#include "minimum_snap.hpp"  // matrix types

namespace math_util {

struct C_Point {
    double x, y, z, heading;
    C_Point() : x(0.0), y(0.0), z(0.0), heading(0.0) {}
    C_Point(const double &_x, const double &_y, const double &_z, const double &_heading) : x(_x), y(_y), z(_z), heading(_heading) {}
    C_Point(const double &_x, const double &_y, const double &_heading) : x(_x), y(_y), z(0.0), heading(_heading) {}
    C_Point(const double &_x, const double &_y) : x(_x), y(_y), z(0.0), heading(0.0) {}
    void Set(const double &_x, const double &_y, const double &_z, const double &_heading) { x = _x; y = _y; z = _z; heading = _heading; }
    void SetX(const double &_x) { x = _x; }
    void SetY(const double &_y) { y = _y; }
    void SetZ(const double &_z) { z = _z; }
    void SetHeading(const double &_heading) { heading = _heading; }
    double GetX() { return x; }
    double GetY() { return y; }
    double GetZ() { return z; }
    double GetHeading() { return heading; }
    C_Point operator-(const C_Point &p) const { return C_Point(x - p.x, y - p.y, z - p.z, heading - p.heading); }
    C_Point operator+(const C_Point &p) const { return C_Point(x + p.x, y + p.y, z + p.z, heading + p.heading); }
};

struct BezierConfig {
    double min_radius;
    BezierConfig() : min_radius(1.0) {}
};

class Bezier {
public:
    using MatrixXd = csp_host::MatrixXd;

    Bezier() : is_init_(false), path_resolution_(1.0) {}
    void SetConfig(const BezierConfig &config) { config_ = config; }
    void Init(const C_Point &p1, const C_Point &p2, const double path_resolution) {
        start_pt_ = p1;
        end_pt_ = p2;
        result_path_.clear();
        path_resolution_ = path_resolution;
        is_init_ = true;
    }

    // 0 on success, -1 when not initialised or the end points are closer than 0.1 in the plane.
    int GeneratePath() {
        if (!is_init_) return -1;
        const C_Point a = start_pt_, d = end_pt_;
        const double chord = std::hypot(a.x - d.x, a.y - d.y);
        if (chord < 1e-1) return -1;
        C_Point b, c;
        auto place = [&](double k) {  // inner control points for arm factor k
            b.x = a.x + std::cos(a.heading) * chord * k;
            b.y = a.y + std::sin(a.heading) * chord * k;
            b.z = a.z + (d.z - a.z) * 1.0 / 3.0;
            c.x = d.x - std::cos(d.heading) * chord * k;
            c.y = d.y - std::sin(d.heading) * chord * k;
            c.z = a.z + (d.z - a.z) * 2.0 / 3.0;
        };
        auto too_tight = [&](double t) {  // curvature |v x acc| / |v|^3 against 1/min_radius
            const double u = 1.0 - t;
            const double v[3] = {3 * u * u * (b.x - a.x) + 6 * u * t * (c.x - b.x) + 3 * t * t * (d.x - c.x),
                                 3 * u * u * (b.y - a.y) + 6 * u * t * (c.y - b.y) + 3 * t * t * (d.y - c.y),
                                 3 * u * u * (b.z - a.z) + 6 * u * t * (c.z - b.z) + 3 * t * t * (d.z - c.z)};
            const double w[3] = {6 * u * (c.x - 2 * b.x + a.x) + 6 * t * (d.x - 2 * c.x + b.x),
                                 6 * u * (c.y - 2 * b.y + a.y) + 6 * t * (d.y - 2 * c.y + b.y),
                                 6 * u * (c.z - 2 * b.z + a.z) + 6 * t * (d.z - 2 * c.z + b.z)};
            const double cx = v[1] * w[2] - v[2] * w[1], cy = v[2] * w[0] - v[0] * w[2], cz = v[0] * w[1] - v[1] * w[0];
            const double speed = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const double speed3 = speed * speed * speed;
            if (!(speed3 > 1e-6)) return false;
            return std::sqrt(cx * cx + cy * cy + cz * cz) / speed3 > 1.0 / config_.min_radius;
        };
        double k = 1.0 / 3.0;
        for (int attempt = 0; attempt < 10; ++attempt) {
            place(k);
            if (config_.min_radius <= 1.0) break;
            if (!(too_tight(0.0) || too_tight(0.5) || too_tight(1.0))) break;
            k += 0.02;
            if (k > 0.45) { k = 0.45; break; }
        }
        place(k);
        const double length_estimate = std::hypot(c.x - b.x, c.y - b.y) + chord * 2.0 / 3.0;
        const double step = path_resolution_ / length_estimate;
        for (double t = 0.0; t <= 1.0; t += step) {  // accumulated like the reference (bezier.cpp:110)
            const double u = 1.0 - t;
            const double w0 = u * u * u, w1 = 3 * u * u * t, w2 = 3 * u * t * t, w3 = t * t * t;
            C_Point p;
            p.x = w0 * a.x + w1 * b.x + w2 * c.x + w3 * d.x;
            p.y = w0 * a.y + w1 * b.y + w2 * c.y + w3 * d.y;
            p.z = w0 * a.z + w1 * b.z + w2 * c.z + w3 * d.z;
            result_path_.push_back(p);
        }
        return 0;
    }

    bool GetResult() { return is_init_ && !result_path_.empty(); }
    std::vector<C_Point> GetResultPath() { return result_path_; }

    // Path N x 3 -> sampled M x 3.  yaml_path and v_avg_override are unused, as in the reference.
    MatrixXd GenerateTrajectoryMatrix(const MatrixXd &Path, const std::string & /*yaml_path*/,
                                      double sample_distance_override = -1.0, double /*v_avg_override*/ = -1.0) {
        if (Path.rows() < 2) return MatrixXd(0, 3);
        const double resolution = sample_distance_override > 0 ? sample_distance_override : 1.0;
        result_path_.clear();
        const int n = (int)Path.rows();
        std::vector<double> heading((size_t)n);
        for (int i = 0; i < n; ++i) {  // one-sided at the ends, central differences inside
            const int lo = (i == 0) ? 0 : i - 1, hi = (i == n - 1) ? n - 1 : i + 1;
            heading[(size_t)i] = std::atan2(Path(hi, 1) - Path(lo, 1), Path(hi, 0) - Path(lo, 0));
        }
        std::vector<C_Point> all;
        for (int i = 0; i + 1 < n; ++i) {
            Init(C_Point(Path(i, 0), Path(i, 1), Path(i, 2), heading[(size_t)i]),
                 C_Point(Path(i + 1, 0), Path(i + 1, 1), Path(i + 1, 2), heading[(size_t)i + 1]), resolution);
            if (GeneratePath() == 0) {
                for (size_t q = (i == 0) ? 0 : 1; q < result_path_.size(); ++q) all.push_back(result_path_[q]);
            } else {
                all.push_back(C_Point(Path(i + 1, 0), Path(i + 1, 1), Path(i + 1, 2), 0));
            }
        }
        MatrixXd out((long)all.size(), 3);
        for (size_t i = 0; i < all.size(); ++i) { out((long)i, 0) = all[i].x; out((long)i, 1) = all[i].y; out((long)i, 2) = all[i].z; }
        return out;
    }

private:
    BezierConfig config_;
    bool is_init_;
    C_Point start_pt_, end_pt_;
    std::vector<C_Point> result_path_;
    double path_resolution_;
};

}  // namespace math_util

#endif  // CSP_HOST_BEZIER_HPP_
