// double-precision NEAREST (CostCalculatorNID) kernels.  Build with -ffp-contract=off: the arithmetic
// follows the reference's expression tree, so +,-,*,/,sqrt are bit-identical to the CPU's and the
// integer histogram is exactly reproducible.
#include "nid_launch_impl.hpp"

namespace nidreg {

template <> hipError_t launch_nearest_hist<double>(const PassArgs& a) {
  if (a.nchunks == 0) return hipSuccess;
  return a.rec64 ? launch_nearest_hist_rec<double, Rec64>(a) : launch_nearest_hist_rec<double, Rec32>(a);
}
}  // namespace nidreg
