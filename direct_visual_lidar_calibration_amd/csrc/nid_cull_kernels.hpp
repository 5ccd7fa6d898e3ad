// nid_cull_kernels.hpp -- ViewCulling::cull on the GPU (reference: src/vlcal/calib/view_culling.cpp:21-92),
// SURVEY.md "next" row N1.  Two streaming passes over the un-culled cloud:
//   k_cull_zbuf  transform (Eigen 4x4 * xyz1 order), FoV gate on the normalised 4-VECTOR (x,y,z,1)
//                (view_culling.cpp:45 quirk), projection, truncating cast, in-image test; per-pixel
//                atomicMin of float(dist).  The reference's sequential z-buffer ends with
//                dist_map[p] = float(min dist) whatever the visiting order (float rounding is monotone),
//                so an atomic min reproduces it exactly.
//   k_cull_keep  keep = candidate && !(dist > double(dist_map[p]) + 0.1)   (view_culling.cpp:81)
// Compiled in the -ffp-contract=off translation unit: every +,-,*,/,sqrt matches the CPU bit for bit,
// so the surviving index list is identical to the reference's.
#pragma once
#include "nid_device.hpp"

namespace nidreg {

template <int MODEL>
__global__ __launch_bounds__(256) void k_cull_zbuf(
  const double* __restrict__ pts, long long stride_d, long long n, IsoParams<double> iso, CamParams<double> cam, int W, int H, double min_z, int depth, int* __restrict__ pix,
  unsigned int* __restrict__ zbuf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* p = pts + i * stride_d;
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  // Eigen 4x4 * 4x1, summed left to right; row 3 of an isometry is (0,0,0,1)
  const double cx = ((iso.m[0] * x + iso.m[1] * y) + iso.m[2] * z) + iso.m[3] * w;
  const double cy = ((iso.m[4] * x + iso.m[5] * y) + iso.m[6] * z) + iso.m[7] * w;
  const double cz = ((iso.m[8] * x + iso.m[9] * y) + iso.m[10] * z) + iso.m[11] * w;
  const double cw = ((0.0 * x + 0.0 * y) + 0.0 * z) + 1.0 * w;
  int out = -1;
  const double n4 = sqrt(((cx * cx + cy * cy) + cz * cz) + cw * cw);
  if (!(cz / n4 < min_z)) {
    double u, v;
    project<MODEL, double, double, false>(cam, cx, cy, cz, u, v);
    // .cast<int>() truncates; x86 sends NaN / overflow to INT_MIN (rejected)
    if ((u > -1.0) && (u < double(W)) && (v > -1.0) && (v < double(H))) {
      const int px = int(u), py = int(v);
      out = py * W + px;
      if (depth) {
        const double dist = sqrt((cx * cx + cy * cy) + cz * cz);
        atomicMin(&zbuf[out], __float_as_uint(float(dist)));  // dist >= 0: float bits order like unsigned
      }
    }
  }
  pix[i] = out;
}

__global__ __launch_bounds__(256) void k_cull_keep(
  const double* __restrict__ pts, long long stride_d, long long n, IsoParams<double> iso, int depth, const int* __restrict__ pix, const unsigned int* __restrict__ zbuf,
  unsigned char* __restrict__ keep) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int q = pix[i];
  unsigned char k = q >= 0 ? 1 : 0;
  if (k && depth) {
    const double* p = pts + i * stride_d;
    const double x = p[0], y = p[1], z = p[2], w = p[3];
    const double cx = ((iso.m[0] * x + iso.m[1] * y) + iso.m[2] * z) + iso.m[3] * w;
    const double cy = ((iso.m[4] * x + iso.m[5] * y) + iso.m[6] * z) + iso.m[7] * w;
    const double cz = ((iso.m[8] * x + iso.m[9] * y) + iso.m[10] * z) + iso.m[11] * w;
    const double dist = sqrt((cx * cx + cy * cy) + cz * cz);
    if (dist > double(__uint_as_float(zbuf[q])) + 0.1) k = 0;
  }
  keep[i] = k;
}

}  // namespace nidreg
