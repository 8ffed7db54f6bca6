// TEST INFRASTRUCTURE -- minimal stand-ins for the parts of Eigen 3 that include/orbslam3_shim.hpp touches, so that the
// reference-typed half of the shim can be compiled (and its glue run on a toy map) in an image that has no Eigen.
// Not a linear-algebra library: fixed-size dense matrices with the handful of members the shim calls, written for this
// test only.  Nothing under orb_slam3-1_amd/ includes it.
#pragma once

#include <array>
#include <cmath>
#include <cstddef>
#include <vector>

namespace Eigen {

template <typename T, int R, int C>
class Matrix {
public:
    Matrix() { a_.fill(T(0)); }
    Matrix(T x, T y) { static_assert(R * C == 2, "2-vector"); a_[0] = x; a_[1] = y; }
    Matrix(T x, T y, T z) { static_assert(R * C == 3, "3-vector"); a_[0] = x; a_[1] = y; a_[2] = z; }
    T& operator()(int r, int c) { return a_[(size_t)r * C + c]; }
    const T& operator()(int r, int c) const { return a_[(size_t)r * C + c]; }
    T& operator()(int i) { return a_[(size_t)i]; }
    const T& operator()(int i) const { return a_[(size_t)i]; }
    T& operator[](int i) { return a_[(size_t)i]; }
    const T& operator[](int i) const { return a_[(size_t)i]; }
    T x() const { return a_[0]; }
    T y() const { return a_[1]; }
    T z() const { return a_[2]; }
    template <typename U> Matrix<U, R, C> cast() const { Matrix<U, R, C> m; for (int i = 0; i < R * C; i++) m[i] = (U)a_[(size_t)i]; return m; }
    Matrix<T, C, R> transpose() const { Matrix<T, C, R> m; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) m(c, r) = (*this)(r, c); return m; }
    T dot(const Matrix& o) const { T s = 0; for (int i = 0; i < R * C; i++) s += a_[(size_t)i] * o[i]; return s; }
    T norm() const { return std::sqrt(dot(*this)); }
    template <int BR, int BC> Matrix<T, BR, BC> block(int r0, int c0) const
    { Matrix<T, BR, BC> m; for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) m(r, c) = (*this)(r0 + r, c0 + c); return m; }
    Matrix<T, R, R> asDiagonal() const { static_assert(C == 1, "vector"); Matrix<T, R, R> m; for (int i = 0; i < R; i++) m(i, i) = a_[(size_t)i]; return m; }
    Matrix inverse() const     // Gauss-Jordan with partial pivoting
    {
        static_assert(R == C, "square");
        Matrix a = *this, inv;
        for (int i = 0; i < R; i++) inv(i, i) = T(1);
        for (int k = 0; k < R; k++) {
            int p = k;
            for (int r = k + 1; r < R; r++) if (std::fabs(a(r, k)) > std::fabs(a(p, k))) p = r;
            for (int c = 0; c < R; c++) { std::swap(a(k, c), a(p, c)); std::swap(inv(k, c), inv(p, c)); }
            const T d = T(1) / a(k, k);
            for (int c = 0; c < R; c++) { a(k, c) *= d; inv(k, c) *= d; }
            for (int r = 0; r < R; r++) if (r != k) { const T f = a(r, k); for (int c = 0; c < R; c++) { a(r, c) -= f * a(k, c); inv(r, c) -= f * inv(k, c); } }
        }
        return inv;
    }
    struct Comma { Matrix* m; int i; Comma operator,(T v) { (*m)[i] = v; return Comma{m, i + 1}; } };
    Comma operator<<(T v) { a_[0] = v; return Comma{this, 1}; }
    Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; i++) m[i] = -a_[(size_t)i]; return m; }
    Matrix& operator*=(T s) { for (auto& v : a_) v *= s; return *this; }

private:
    std::array<T, (size_t)R * C> a_;
};

template <typename T, int R, int C> Matrix<T, R, C> operator+(const Matrix<T, R, C>& a, const Matrix<T, R, C>& b) { Matrix<T, R, C> m; for (int i = 0; i < R * C; i++) m[i] = a[i] + b[i]; return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator-(const Matrix<T, R, C>& a, const Matrix<T, R, C>& b) { Matrix<T, R, C> m; for (int i = 0; i < R * C; i++) m[i] = a[i] - b[i]; return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator/(const Matrix<T, R, C>& a, double s) { Matrix<T, R, C> m; for (int i = 0; i < R * C; i++) m[i] = (T)(a[i] / s); return m; }
template <typename T, int R, int C> Matrix<T, R, C> operator*(const Matrix<T, R, C>& a, double s) { Matrix<T, R, C> m; for (int i = 0; i < R * C; i++) m[i] = (T)(a[i] * s); return m; }
template <typename T, int R, int K, int C> Matrix<T, R, C> operator*(const Matrix<T, R, K>& a, const Matrix<T, K, C>& b)
{ Matrix<T, R, C> m; for (int r = 0; r < R; r++) for (int c = 0; c < C; c++) { T s = 0; for (int k = 0; k < K; k++) s += a(r, k) * b(k, c); m(r, c) = s; } return m; }

typedef Matrix<float, 2, 1> Vector2f;
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<float, 3, 3> Matrix3f;
typedef Matrix<double, 3, 3> Matrix3d;

class MatrixXd {
public:
    MatrixXd() : r_(0), c_(0) {}
    MatrixXd(int r, int c) : r_(r), c_(c), a_((size_t)r * c, 0.0) {}
    double& operator()(int r, int c) { return a_[(size_t)r * c_ + c]; }
    const double& operator()(int r, int c) const { return a_[(size_t)r * c_ + c]; }
    int rows() const { return r_; }
    int cols() const { return c_; }
    template <int BR, int BC> Matrix<double, BR, BC> block(int r0, int c0) const
    { Matrix<double, BR, BC> m; for (int r = 0; r < BR; r++) for (int c = 0; c < BC; c++) m(r, c) = (*this)(r0 + r, c0 + c); return m; }

private:
    int r_, c_;
    std::vector<double> a_;
};

template <typename T>
class Quaternion {
public:
    Quaternion() : w_(1), x_(0), y_(0), z_(0) {}
    Quaternion(T w, T x, T y, T z) : w_(w), x_(x), y_(y), z_(z) {}
    explicit Quaternion(const Matrix<T, 3, 3>& R)      // Shepperd's method
    {
        const T tr = R(0, 0) + R(1, 1) + R(2, 2);
        if (tr > 0) { T s = std::sqrt(tr + 1) * 2; w_ = s / 4; x_ = (R(2, 1) - R(1, 2)) / s; y_ = (R(0, 2) - R(2, 0)) / s; z_ = (R(1, 0) - R(0, 1)) / s; }
        else if (R(0, 0) > R(1, 1) && R(0, 0) > R(2, 2)) { T s = std::sqrt(1 + R(0, 0) - R(1, 1) - R(2, 2)) * 2; w_ = (R(2, 1) - R(1, 2)) / s; x_ = s / 4; y_ = (R(0, 1) + R(1, 0)) / s; z_ = (R(0, 2) + R(2, 0)) / s; }
        else if (R(1, 1) > R(2, 2)) { T s = std::sqrt(1 + R(1, 1) - R(0, 0) - R(2, 2)) * 2; w_ = (R(0, 2) - R(2, 0)) / s; x_ = (R(0, 1) + R(1, 0)) / s; y_ = s / 4; z_ = (R(1, 2) + R(2, 1)) / s; }
        else { T s = std::sqrt(1 + R(2, 2) - R(0, 0) - R(1, 1)) * 2; w_ = (R(1, 0) - R(0, 1)) / s; x_ = (R(0, 2) + R(2, 0)) / s; y_ = (R(1, 2) + R(2, 1)) / s; z_ = s / 4; }
    }
    T w() const { return w_; }
    T x() const { return x_; }
    T y() const { return y_; }
    T z() const { return z_; }
    template <typename U> Quaternion<U> cast() const { return Quaternion<U>((U)w_, (U)x_, (U)y_, (U)z_); }
    Matrix<T, 3, 3> toRotationMatrix() const
    {
        Matrix<T, 3, 3> R;
        R(0, 0) = 1 - 2 * (y_ * y_ + z_ * z_); R(0, 1) = 2 * (x_ * y_ - z_ * w_); R(0, 2) = 2 * (x_ * z_ + y_ * w_);
        R(1, 0) = 2 * (x_ * y_ + z_ * w_); R(1, 1) = 1 - 2 * (x_ * x_ + z_ * z_); R(1, 2) = 2 * (y_ * z_ - x_ * w_);
        R(2, 0) = 2 * (x_ * z_ - y_ * w_); R(2, 1) = 2 * (y_ * z_ + x_ * w_); R(2, 2) = 1 - 2 * (x_ * x_ + y_ * y_);
        return R;
    }

private:
    T w_, x_, y_, z_;
};
typedef Quaternion<float> Quaternionf;
typedef Quaternion<double> Quaterniond;

// cyclic Jacobi: enough for the 9x9 / 15x15 symmetric information matrices of the inertial adapters
template <typename M> class SelfAdjointEigenSolver;
template <typename T, int N>
class SelfAdjointEigenSolver<Matrix<T, N, N> > {
public:
    explicit SelfAdjointEigenSolver(const Matrix<T, N, N>& A)
    {
        Matrix<T, N, N> a = A;
        for (int i = 0; i < N; i++) v_(i, i) = T(1);
        for (int sweep = 0; sweep < 60; sweep++) {
            T off = 0;
            for (int p = 0; p < N; p++) for (int q = p + 1; q < N; q++) off += a(p, q) * a(p, q);
            if (off < T(1e-300)) break;
            for (int p = 0; p < N; p++)
                for (int q = p + 1; q < N; q++) {
                    if (a(p, q) == T(0)) continue;
                    const T theta = (a(q, q) - a(p, p)) / (2 * a(p, q));
                    const T t = (theta >= 0 ? T(1) : T(-1)) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
                    const T c = 1 / std::sqrt(t * t + 1), s = t * c;
                    for (int k = 0; k < N; k++) { const T akp = a(k, p), akq = a(k, q); a(k, p) = c * akp - s * akq; a(k, q) = s * akp + c * akq; }
                    for (int k = 0; k < N; k++) { const T apk = a(p, k), aqk = a(q, k); a(p, k) = c * apk - s * aqk; a(q, k) = s * apk + c * aqk; }
                    for (int k = 0; k < N; k++) { const T vkp = v_(k, p), vkq = v_(k, q); v_(k, p) = c * vkp - s * vkq; v_(k, q) = s * vkp + c * vkq; }
                }
        }
        for (int i = 0; i < N; i++) e_[i] = a(i, i);
    }
    const Matrix<T, N, 1>& eigenvalues() const { return e_; }
    const Matrix<T, N, N>& eigenvectors() const { return v_; }

private:
    Matrix<T, N, 1> e_;
    Matrix<T, N, N> v_;
};

}  // namespace Eigen

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW
