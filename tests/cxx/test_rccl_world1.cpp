// The in-library RCCL route of one pair whose points are split over ranks (include/nidreg.h: nidreg_shard_comm_init /
// nidreg_shard_attach_rccl), driven from C through the C ABI alone.  One GPU is all a test box has, so the communicators
// have ONE rank: the all-reduce of one rank is the identity, and the chain
//     histogram -> ncclAllReduce(int64) -> entropy tail -> gradient -> ncclAllReduce(f64 x 7)
// must return the plain handle's bits.  Two slices of the cloud on the same device, one communicator each, are evaluated
// as well: their integer histograms are added by the caller here (what the all-reduce of two ranks would do is covered by
// the world-size-2 gloo test of the same protocol, tests/test_parallel_gloo.py).
// Reads the scene file of tests/test_cxx_dropin.py; prints "plain: cost g0..g6 | owned comm: cost g0..g6 | attached comm: cost g0..g6".
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <rccl/rccl.h>

#include "nidreg.h"

#define CHECK(expr)                                                                  \
  do {                                                                               \
    const int _rc = (expr);                                                          \
    if (_rc < 0) {                                                                   \
      std::fprintf(stderr, "%s failed (%d): %s\n", #expr, _rc, nidreg_last_error()); \
      return 10;                                                                     \
    }                                                                                \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int W, H, N, bins, nintr, ndist;
  char model[64] = {0};
  double intr[5], dist[8], se3[7], max_fov, T[16];
  if (fread(model, 1, 64, f) != 64) return 4;
  if (fread(&W, 4, 1, f) != 1 || fread(&H, 4, 1, f) != 1 || fread(&N, 4, 1, f) != 1 || fread(&bins, 4, 1, f) != 1 || fread(&nintr, 4, 1, f) != 1 || fread(&ndist, 4, 1, f) != 1) return 4;
  if (fread(intr, 8, 5, f) != 5 || fread(dist, 8, 8, f) != 8 || fread(se3, 8, 7, f) != 7 || fread(&max_fov, 8, 1, f) != 1 || fread(T, 8, 16, f) != 16) return 4;
  std::vector<unsigned char> img8(size_t(W) * H);
  if (fread(img8.data(), 1, img8.size(), f) != img8.size()) return 4;
  std::vector<double> img64(img8.size());
  for (size_t k = 0; k < img8.size(); k++) img64[k] = img8[k] * (1.0 / 255.0);
  std::vector<double> pts(size_t(N) * 4), ints(static_cast<size_t>(N));
  if (fread(pts.data(), 32, N, f) != size_t(N) || fread(ints.data(), 8, N, f) != size_t(N)) return 4;
  fclose(f);

  nidreg_desc d;
  std::memset(&d, 0, sizeof(d));
  d.struct_size = sizeof(d);
  d.device_id = 0;
  d.model_id = nidreg_model_from_name(model, nullptr, nullptr);
  d.mode = NIDREG_MODE_SPLINE;
  d.precision = NIDREG_PREC_FP64;
  d.bins = bins;
  std::memcpy(d.intrinsics, intr, sizeof(intr));
  std::memcpy(d.distortion, dist, sizeof(dist));
  d.width = W;
  d.height = H;
  d.image_dtype = NIDREG_IMAGE_F64;
  d.image = img64.data();
  d.image_row_stride = int64_t(W) * 8;
  d.num_points = N;
  d.points = pts.data();
  d.point_stride = 32;
  d.intensities = ints.data();
  d.scale_points = N;

  double out[3][8];
  nidreg_handle* plain = nullptr;
  CHECK(nidreg_create(&d, &plain));
  CHECK(nidreg_eval(plain, se3, &out[0][0], &out[0][1]));

  // (1) a communicator the library creates and owns
  nidreg_handle* owned = nullptr;
  CHECK(nidreg_create(&d, &owned));
  unsigned char id[NIDREG_RCCL_ID_BYTES];
  CHECK(nidreg_rccl_unique_id(id));
  CHECK(nidreg_shard_comm_init(owned, 1, 0, id));
  CHECK(nidreg_eval(owned, se3, &out[1][0], &out[1][1]));
  double c_only = 0.0;
  CHECK(nidreg_eval(owned, se3, &c_only, nullptr));
  if (c_only != out[1][0]) return 11;
  double batch_c[3], batch_g[21], poses[21];
  for (int k = 0; k < 3; k++) std::memcpy(poses + 7 * k, se3, sizeof(se3));
  CHECK(nidreg_eval_batch(owned, poses, 3, batch_c, batch_g));  // back to back: both histogram buffers
  for (int k = 0; k < 3; k++)
    if (batch_c[k] != out[1][0] || std::memcmp(batch_g + 7 * k, &out[1][1], 7 * sizeof(double)) != 0) return 12;

  // (2) a communicator of the caller's (linked against librccl here; the library only sees the pointer)
  ncclComm_t comm = nullptr;
  int dev0 = 0;
  if (ncclCommInitAll(&comm, 1, &dev0) != ncclSuccess) return 13;
  nidreg_handle* attached = nullptr;
  CHECK(nidreg_create(&d, &attached));
  CHECK(nidreg_shard_attach_rccl(attached, comm));
  CHECK(nidreg_eval(attached, se3, &out[2][0], &out[2][1]));
  // refused where a collective per handle makes no sense
  nidreg_handle* two[2] = {attached, plain};
  double cm = 0.0;
  if (nidreg_eval_multi(two, 2, nullptr, se3, &cm, nullptr) != NIDREG_ERR_INVALID) return 14;
  CHECK(nidreg_shard_attach_rccl(attached, nullptr));  // detached: the plain route again
  double c_det = 0.0, g_det[7];
  CHECK(nidreg_eval(attached, se3, &c_det, g_det));
  if (c_det != out[0][0] || std::memcmp(g_det, &out[0][1], sizeof(g_det)) != 0) return 15;

  nidreg_destroy(attached);
  ncclCommDestroy(comm);
  nidreg_destroy(owned);
  nidreg_destroy(plain);
  for (int r = 0; r < 3; r++)
    for (int k = 0; k < 8; k++) std::printf("%.17g ", out[r][k]);
  std::printf("\n");
  return 0;
}
