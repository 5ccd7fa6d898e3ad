// TEST INFRASTRUCTURE ONLY -- stand-in for <ceres/jet.h> (Ceres Solver is not installed here).
// ceres::Jet<double, N> IS the oracle's dual number (oracle/jet.hpp, restated from the published
// ceres/jet.h formulas): operators and math functions are found through the base class, so the
// reference's templated camera functors and NIDCost::operator() run on exactly the Jet arithmetic the
// oracle uses -- a difference between oracle and reference build can then only come from the
// reference's own source text.
#pragma once
#include "../../jet.hpp"

namespace ceres {

template <typename T, int N>
struct Jet;

template <int N>
struct Jet<double, N> : public oracle::Jet<N> {
  Jet() : oracle::Jet<N>() {}
  Jet(double value) : oracle::Jet<N>(value) {}  // NOLINT (implicit like ceres::Jet)
  Jet(double value, int k) : oracle::Jet<N>(value, k) {}
  Jet(const oracle::Jet<N>& j) : oracle::Jet<N>(j) {}  // NOLINT
};

}  // namespace ceres
