// Accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of the Newton-refined reciprocals the SPLINE kernels use
// (nid_device.hpp fast_rcp / fast_rsq, kNewtonSteps): max relative error over a sweep of depths, measured on the GPU
// against the correctly rounded 1/z and 1/sqrt(z).  Build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_rcp.bin tools/ubench_rcp.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void k_err(const double* z, int n, double* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = z[i];
  const double ex = 1.0 / x, es = 1.0 / sqrt(x);  // IEEE division / sqrt sequences
  double r = __builtin_amdgcn_rcp(x);
  out[0 * n + i] = fabs(r - ex) / ex;
  r = fma(fma(-x, r, 1.0), r, r);
  out[1 * n + i] = fabs(r - ex) / ex;
  r = fma(fma(-x, r, 1.0), r, r);
  out[2 * n + i] = fabs(r - ex) / ex;
  double q = __builtin_amdgcn_rsq(x);
  out[3 * n + i] = fabs(q - es) / es;
  q = fma(0.5 * q, fma(-x * q, q, 1.0), q);
  out[4 * n + i] = fabs(q - es) / es;
  q = fma(0.5 * q, fma(-x * q, q, 1.0), q);
  out[5 * n + i] = fabs(q - es) / es;
}

int main() {
  const int n = 1 << 22;
  std::vector<double> z(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    const double u = double(s >> 11) * (1.0 / 9007199254740992.0);
    z[i] = std::exp(std::log(1e-3) + u * (std::log(1e4) - std::log(1e-3)));  // log-uniform in [1e-3, 1e4]
  }
  double *dz, *dout;
  hipMalloc(&dz, n * sizeof(double));
  hipMalloc(&dout, 6 * size_t(n) * sizeof(double));
  hipMemcpy(dz, z.data(), n * sizeof(double), hipMemcpyHostToDevice);
  k_err<<<n / 256, 256>>>(dz, n, dout);
  std::vector<double> out(6 * size_t(n));
  if (hipMemcpy(out.data(), dout, out.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  const char* names[6] = {"v_rcp_f64 seed", "rcp + 1 Newton step", "rcp + 2 Newton steps", "v_rsq_f64 seed", "rsq + 1 Newton step", "rsq + 2 Newton steps"};
  for (int k = 0; k < 6; k++) {
    double m = 0, a = 0;
    for (int i = 0; i < n; i++) m = std::fmax(m, out[size_t(k) * n + i]), a += out[size_t(k) * n + i];
    std::printf("%-22s max rel err %.3e (2^%.1f)  mean %.3e\n", names[k], m, m > 0 ? std::log2(m) : -99.0, a / n);
  }
  return 0;
}
