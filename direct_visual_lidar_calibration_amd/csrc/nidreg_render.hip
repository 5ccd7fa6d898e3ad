// nidreg_render.hip -- C ABI of the two point-to-image consumers next to the NID path
// (include/nidreg.h: nidreg_colorizer_*, nidreg_generate_lidar_image).  Host side only: device
// residency and launches; the kernels are in nid_render_kernels.hpp.  No CPU compute path.
#include "nid_device.hpp"
#include "nid_launch.hpp"

#include <cstring>
#include <string>

#include "../../include/nidreg.h"

using namespace nidreg;

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t _e = (expr);                                                                             \
    if (_e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
  } while (0)

struct nidreg_colorizer {
  int device = 0, model = 0, W = 0, H = 0;
  double intr[5] = {0, 0, 0, 0, 0}, dist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  double min_nz = 0.0;
  long long n = 0, stride_d = 4;
  double* d_pts = nullptr;
  float* d_icolor = nullptr;  // nullable: (1,1,1,1)
  float* d_out = nullptr;
  uint8_t* d_img = nullptr;
  hipStream_t stream = nullptr;
};

namespace {

void colorizer_free(nidreg_colorizer* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->d_pts) (void)hipFree(c->d_pts);
  if (c->d_icolor) (void)hipFree(c->d_icolor);
  if (c->d_out) (void)hipFree(c->d_out);
  if (c->d_img) (void)hipFree(c->d_img);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // namespace

extern "C" {

int nidreg_colorizer_create(int device_id, int model_id, const double* intrinsics, const double* distortion, int width, int height, const uint8_t* image, int64_t image_row_stride,
                            int64_t num_points, const double* points, int64_t point_stride, const float* intensity_colors, double min_nz, nidreg_colorizer** out) {
  if (!out) return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_create: null out");
  *out = nullptr;
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || width < 1 || height < 1 || !image || num_points < 0 || (num_points > 0 && !points))
    return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_create: bad argument");
  const int64_t stride = point_stride > 0 ? point_stride : 32;
  if (stride % 8 != 0 || stride < 32) return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_create: point_stride must be a multiple of 8, at least 32");
  const int64_t rs = image_row_stride > 0 ? image_row_stride : width;
  if (rs < width) return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_create: image_row_stride < width");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_colorizer_create: no HIP device");
  if (device_id < 0 || device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_create: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  nidreg_colorizer* c = new nidreg_colorizer();
  c->device = device_id;
  c->model = model_id;
  c->W = width;
  c->H = height;
  c->min_nz = min_nz;
  c->n = num_points;
  c->stride_d = stride / 8;
  std::memcpy(c->intr, intrinsics, sizeof(c->intr));
  std::memcpy(c->dist, distortion, sizeof(c->dist));
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  const size_t n = size_t(num_points);
  if (e == hipSuccess) e = hipMalloc(&c->d_img, size_t(width) * height);
  if (e == hipSuccess) e = hipMemcpy2D(c->d_img, size_t(width), image, size_t(rs), size_t(width), size_t(height), hipMemcpyHostToDevice);
  if (n > 0) {
    if (e == hipSuccess) e = hipMalloc(&c->d_pts, n * size_t(stride));
    if (e == hipSuccess) e = hipMemcpy(c->d_pts, points, n * size_t(stride), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&c->d_out, n * 4 * sizeof(float));
    if (intensity_colors) {
      if (e == hipSuccess) e = hipMalloc(&c->d_icolor, n * 4 * sizeof(float));
      if (e == hipSuccess) e = hipMemcpy(c->d_icolor, intensity_colors, n * 4 * sizeof(float), hipMemcpyHostToDevice);
    }
  }
  if (e != hipSuccess) {
    colorizer_free(c);
    return fail(NIDREG_ERR_HIP, std::string("nidreg_colorizer_create: ") + hipGetErrorString(e));
  }
  *out = c;
  return NIDREG_OK;
}

int nidreg_colorizer_update(nidreg_colorizer* c, const double* T_camera_lidar, double blend_weight, float* colors_out) {
  if (!c || !T_camera_lidar) return fail(NIDREG_ERR_INVALID, "nidreg_colorizer_update: null argument");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(launch_colorize(c->model, c->intr, c->dist, c->d_pts, c->stride_d, c->n, T_camera_lidar, c->d_img, c->W, c->H, c->min_nz, c->d_icolor, blend_weight, c->d_out, c->stream));
  if (colors_out && c->n > 0) HIP_TRY(hipMemcpyAsync(colors_out, c->d_out, size_t(c->n) * 4 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  return NIDREG_OK;
}

const float* nidreg_colorizer_device_colors(nidreg_colorizer* c) { return c ? c->d_out : nullptr; }

void nidreg_colorizer_destroy(nidreg_colorizer* c) { colorizer_free(c); }

int nidreg_generate_lidar_image(int model_id, const double* intrinsics, const double* distortion, int device_id, int width, int height, double min_nz, const double* points,
                                int64_t point_stride, const double* intensities, int64_t num_points, const double* T_camera_lidar, double* intensity_image, int32_t* index_image) {
  if (model_id < 0 || model_id > 5 || !intrinsics || !distortion || width < 1 || height < 1 || num_points < 0 || num_points > 2147483647LL || !T_camera_lidar ||
      (num_points > 0 && (!points || !intensities)) || (!intensity_image && !index_image))
    return fail(NIDREG_ERR_INVALID, "nidreg_generate_lidar_image: bad argument");
  const int64_t stride = point_stride > 0 ? point_stride : 32;
  if (stride % 8 != 0 || stride < 32) return fail(NIDREG_ERR_INVALID, "nidreg_generate_lidar_image: point_stride must be a multiple of 8, at least 32");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_generate_lidar_image: no HIP device");
  if (device_id < 0 || device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_generate_lidar_image: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  const size_t n = size_t(num_points), npix = size_t(width) * size_t(height);
  double *d_pts = nullptr, *d_int = nullptr, *d_iimg = nullptr;
  int *d_pix = nullptr, *d_idx = nullptr;
  u64* d_zmin = nullptr;
  hipError_t e = hipMalloc(&d_zmin, npix * sizeof(u64));
  if (e == hipSuccess) e = hipMalloc(&d_idx, npix * sizeof(int));
  if (e == hipSuccess) e = hipMalloc(&d_iimg, npix * sizeof(double));
  if (n > 0) {
    if (e == hipSuccess) e = hipMalloc(&d_pts, n * size_t(stride));
    if (e == hipSuccess) e = hipMalloc(&d_int, n * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_pix, n * sizeof(int));
    if (e == hipSuccess) e = hipMemcpy(d_pts, points, n * size_t(stride), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_int, intensities, n * sizeof(double), hipMemcpyHostToDevice);
  }
  if (e == hipSuccess)
    e = launch_lidar_image(model_id, intrinsics, distortion, d_pts, stride / 8, d_int, num_points, T_camera_lidar, width, height, min_nz, d_pix, d_zmin, d_idx, d_iimg, nullptr);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess && intensity_image) e = hipMemcpy(intensity_image, d_iimg, npix * sizeof(double), hipMemcpyDeviceToHost);
  if (e == hipSuccess && index_image) e = hipMemcpy(index_image, d_idx, npix * sizeof(int), hipMemcpyDeviceToHost);
  if (d_pts) (void)hipFree(d_pts);
  if (d_int) (void)hipFree(d_int);
  if (d_pix) (void)hipFree(d_pix);
  if (d_idx) (void)hipFree(d_idx);
  if (d_iimg) (void)hipFree(d_iimg);
  if (d_zmin) (void)hipFree(d_zmin);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_generate_lidar_image: ") + hipGetErrorString(e));
  return NIDREG_OK;
}

int nidreg_equalize_intensities(int device_id, double* intensities, int64_t num_points) {
  if (num_points < 0 || num_points > 4294967295LL || (num_points > 0 && !intensities)) return fail(NIDREG_ERR_INVALID, "nidreg_equalize_intensities: bad argument");
  if (num_points == 0) return NIDREG_OK;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_equalize_intensities: no HIP device");
  if (device_id < 0 || device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_equalize_intensities: device_id out of range");
  HIP_TRY(hipSetDevice(device_id));
  double* d = nullptr;
  hipError_t e = hipMalloc(&d, size_t(num_points) * sizeof(double));
  if (e == hipSuccess) e = hipMemcpy(d, intensities, size_t(num_points) * sizeof(double), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = equalize_intensities_device(d, num_points, nullptr);
  if (e == hipSuccess) e = hipMemcpy(intensities, d, size_t(num_points) * sizeof(double), hipMemcpyDeviceToHost);
  if (d) (void)hipFree(d);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string("nidreg_equalize_intensities: ") + hipGetErrorString(e));
  return NIDREG_OK;
}

}  // extern "C"
