// mini_matrix.hpp -- the small dense matrix type the class shim falls back to when <Eigen/Dense>
// is not installed (it is not, in the build image).  It implements only the members the shim and
// the reference's callers of TrajectoryGeneratorTool touch: rows(), cols(), size(), operator(),
// Zero(), and construction by shape.  Column-major like Eigen's default.
#pragma once
#include <cstddef>
#include <vector>

namespace csp_host {

class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(long rows, long cols) : r_(rows), c_(cols), d_((size_t)(rows * cols), 0.0) {}
    static MatrixXd Zero(long rows, long cols) { return MatrixXd(rows, cols); }
    long rows() const { return r_; }
    long cols() const { return c_; }
    long size() const { return r_ * c_; }
    double &operator()(long i, long j) { return d_[(size_t)(j * r_ + i)]; }
    double operator()(long i, long j) const { return d_[(size_t)(j * r_ + i)]; }
    const double *data() const { return d_.data(); }
private:
    long r_, c_;
    std::vector<double> d_;
};

class VectorXd {
public:
    VectorXd() {}
    explicit VectorXd(long n) : d_((size_t)n, 0.0) {}
    static VectorXd Zero(long n) { return VectorXd(n); }
    long size() const { return (long)d_.size(); }
    long rows() const { return (long)d_.size(); }
    double &operator()(long i) { return d_[(size_t)i]; }
    double operator()(long i) const { return d_[(size_t)i]; }
private:
    std::vector<double> d_;
};

class Vector3d {
public:
    Vector3d() { v_[0] = v_[1] = v_[2] = 0.0; }
    Vector3d(double x, double y, double z) { v_[0] = x; v_[1] = y; v_[2] = z; }
    static Vector3d Zero() { return Vector3d(); }
    double &operator()(int i) { return v_[i]; }
    double operator()(int i) const { return v_[i]; }
private:
    double v_[3];
};

}  // namespace csp_host
