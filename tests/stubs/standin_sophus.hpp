// TEST INFRASTRUCTURE -- stand-ins for the Sophus types the shim touches (SE3<T>, SO3f::hat, Sim3f); see standin_eigen.hpp.
#pragma once

#include "standin_eigen.hpp"

namespace Sophus {

template <typename T>
class SE3 {
public:
    SE3() { for (int i = 0; i < 3; i++) R_(i, i) = T(1); }
    SE3(const Eigen::Matrix<T, 3, 3>& R, const Eigen::Matrix<T, 3, 1>& t) : R_(R), t_(t) {}
    SE3(const Eigen::Quaternion<T>& q, const Eigen::Matrix<T, 3, 1>& t) : R_(q.toRotationMatrix()), t_(t) {}
    Eigen::Quaternion<T> unit_quaternion() const { return Eigen::Quaternion<T>(R_); }
    const Eigen::Matrix<T, 3, 1>& translation() const { return t_; }
    Eigen::Matrix<T, 3, 3> rotationMatrix() const { return R_; }
    SE3 inverse() const { const Eigen::Matrix<T, 3, 3> Rt = R_.transpose(); return SE3(Rt, -(Rt * t_)); }
    Eigen::Matrix<T, 3, 1> operator*(const Eigen::Matrix<T, 3, 1>& p) const { return R_ * p + t_; }
    SE3 operator*(const SE3& o) const { return SE3(R_ * o.R_, R_ * o.t_ + t_); }
    template <typename U> SE3<U> cast() const { return SE3<U>(R_.template cast<U>(), t_.template cast<U>()); }

private:
    Eigen::Matrix<T, 3, 3> R_;
    Eigen::Matrix<T, 3, 1> t_;
};
typedef SE3<float> SE3f;
typedef SE3<double> SE3d;

struct SO3f {
    static Eigen::Matrix3f hat(const Eigen::Vector3f& v)
    {
        Eigen::Matrix3f m;
        m(0, 1) = -v(2); m(0, 2) = v(1); m(1, 0) = v(2); m(1, 2) = -v(0); m(2, 0) = -v(1); m(2, 1) = v(0);
        return m;
    }
};

class Sim3f {
public:
    Sim3f() : s_(1.f) {}
    Sim3f(float s, const SE3f& T) : s_(s), T_(T) {}
    Sim3f inverse() const
    {
        const Eigen::Matrix3f Rt = T_.rotationMatrix().transpose();
        return Sim3f(1.f / s_, SE3f(Rt, -((Rt * T_.translation()) * (1.0 / s_))));
    }
    Eigen::Vector3f operator*(const Eigen::Vector3f& p) const { return (T_.rotationMatrix() * p) * (double)s_ + T_.translation(); }

private:
    float s_;
    SE3f T_;
};

}  // namespace Sophus
