// TEST INFRASTRUCTURE ONLY -- stand-in for <gtsam/geometry/Pose3.h> (GTSAM 4.2a9 is not installed):
// gtsam::Vector6 and Pose3::Expmap(xi).matrix(), xi = [omega; v], the full SE(3) exponential as GTSAM
// documents it: R = Rodrigues(omega), t = V(omega) v with
// V = I + (1 - cos th)/th^2 [omega]x + (th - sin th)/th^3 [omega]x^2  (small-angle series below 1e-10).
#pragma once
#include <Eigen/Core>

namespace gtsam {

typedef Eigen::Matrix<double, 6, 1> Vector6;

class Pose3 {
public:
  static Pose3 Expmap(const Vector6& xi) {
    Pose3 P;
    const double wx = xi[0], wy = xi[1], wz = xi[2];
    const double th2 = wx * wx + wy * wy + wz * wz, th = std::sqrt(th2);
    double a, b, c;  // sin th / th, (1 - cos th) / th^2, (th - sin th) / th^3
    if (th < 1e-10) {
      a = 1.0 - th2 / 6.0;
      b = 0.5 - th2 / 24.0;
      c = 1.0 / 6.0 - th2 / 120.0;
    } else {
      a = std::sin(th) / th;
      b = (1.0 - std::cos(th)) / th2;
      c = (th - std::sin(th)) / (th2 * th);
    }
    const double K[3][3] = {{0, -wz, wy}, {wz, 0, -wx}, {-wy, wx, 0}};
    double K2[3][3];
    for (int i = 0; i < 3; i++)
      for (int j = 0; j < 3; j++) K2[i][j] = K[i][0] * K[0][j] + K[i][1] * K[1][j] + K[i][2] * K[2][j];
    P.m = Eigen::Matrix4d::Zero();
    const double v[3] = {xi[3], xi[4], xi[5]};
    for (int i = 0; i < 3; i++) {
      double t = 0.0;
      for (int j = 0; j < 3; j++) {
        const double I = i == j ? 1.0 : 0.0;
        P.m(i, j) = I + a * K[i][j] + b * K2[i][j];
        t += (I + b * K[i][j] + c * K2[i][j]) * v[j];
      }
      P.m(i, 3) = t;
    }
    P.m(3, 3) = 1.0;
    return P;
  }
  const Eigen::Matrix4d& matrix() const { return m; }

private:
  Eigen::Matrix4d m;
};

}  // namespace gtsam
