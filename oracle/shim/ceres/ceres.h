// TEST INFRASTRUCTURE ONLY -- stand-in for the slice of Ceres Solver's front end that
// src/vlcal/calib/visual_camera_calibration.cpp:180-238 touches (Ceres is not installed).  It is NOT a solver:
// ceres::Solve() here only EVALUATES the problem -- at the starting point and at every probe point the test
// driver registered -- and records (ok, cost, gradient), which is how tests/test_reference_build.py reaches the
// reference's MultiNIDCost functor (a struct private to that .cpp) through the reference's own
// AutoDiffFirstOrderFunction wiring.  The parameters are left unchanged.
#pragma once
#include <functional>
#include <memory>
#include <vector>

#include <ceres/jet.h>

namespace ceres {

enum CallbackReturnType { SOLVER_CONTINUE, SOLVER_ABORT, SOLVER_TERMINATE_SUCCESSFULLY };
enum LineSearchDirectionType { STEEPEST_DESCENT, NONLINEAR_CONJUGATE_GRADIENT, LBFGS, BFGS };
struct IterationSummary {};
class IterationCallback {
public:
  virtual ~IterationCallback() {}
  virtual CallbackReturnType operator()(const IterationSummary& summary) = 0;
};
class Manifold {
public:
  virtual ~Manifold() {}
};
class FirstOrderFunction {
public:
  virtual ~FirstOrderFunction() {}
  virtual bool Evaluate(const double* parameters, double* cost, double* gradient) const = 0;
  virtual int NumParameters() const = 0;
};

// published behaviour of ceres::AutoDiffFirstOrderFunction<F, N>::Evaluate: gradient == nullptr -> F with
// doubles; otherwise Jets seeded with unit partials, cost = residual.a, gradient = residual.v
template <typename Functor, int kNumParameters>
class AutoDiffFirstOrderFunction : public FirstOrderFunction {
public:
  explicit AutoDiffFirstOrderFunction(Functor* functor) : functor_(functor) {}
  bool Evaluate(const double* parameters, double* cost, double* gradient) const override {
    if (gradient == nullptr) return (*functor_)(parameters, cost);
    typedef Jet<double, kNumParameters> JetT;
    JetT x[kNumParameters], out;
    for (int i = 0; i < kNumParameters; i++) x[i] = JetT(parameters[i], i);
    if (!(*functor_)(x, &out)) return false;
    *cost = out.a;
    for (int i = 0; i < kNumParameters; i++) gradient[i] = out.v[i];
    return true;
  }
  int NumParameters() const override { return kNumParameters; }

private:
  std::unique_ptr<Functor> functor_;
};

class GradientProblem {
public:
  GradientProblem(FirstOrderFunction* f, Manifold* m) : function(f), manifold(m) {}
  std::unique_ptr<FirstOrderFunction> function;
  std::unique_ptr<Manifold> manifold;
};

class GradientProblemSolver {
public:
  struct Options {
    bool minimizer_progress_to_stdout = false;
    bool update_state_every_iteration = false;
    LineSearchDirectionType line_search_direction_type = LBFGS;
    std::vector<IterationCallback*> callbacks;
    ~Options() {
      for (auto* c : callbacks) delete c;
    }
  };
  struct Summary {
    std::vector<IterationSummary> iterations;
    double final_cost = 0.0;
  };
};

// evaluation record + probe list shared with the driver (oracle/ref_driver.cpp)
struct ProbeLog {
  std::vector<std::vector<double>> probes;  // parameter vectors to evaluate (besides the starting point)
  struct Entry {
    bool ok_value, ok_grad;
    double cost_value, cost_grad;
    std::vector<double> grad;
  };
  std::vector<Entry> entries;
};
ProbeLog& probe_log();

void Solve(const GradientProblemSolver::Options& options, const GradientProblem& problem, double* parameters, GradientProblemSolver::Summary* summary);

}  // namespace ceres
