// The INTEGRATION.md recipe executed for real: this translation unit is compiled with
// -DNIDREG_WITH_REFERENCE_DEPS, next to the REFERENCE'S OWN src/camera/create_camera.cpp, against
//   * a scratch copy of include/camera/generic_camera{,_base}.hpp with integration/reference_camera.patch applied,
//   * the rest of the reference's include tree, unmodified (frame.hpp, camera models, create_camera.hpp),
//   * Eigen / ceres::Jet / cv::Mat as the build image knows them (the stand-ins of oracle/shim/),
// and linked with libnidreg.so.  The camera objects are the reference's camera::GenericCamera<Projection>;
// vlcal::NIDCost / vlcal::CostCalculatorNID are this repository's drop-in classes evaluating on the GPU.
//   test_integration.bin --describe            print model id + parameters of every camera model (no GPU)
//   test_integration.bin <scene.bin>           same scene file and output line as test_dropin.cpp (GPU)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "vlcal_amd/cost_calculator_nid.hpp"
#include "vlcal_amd/nid_cost.hpp"

static int describe() {
  struct Case {
    const char* name;
    std::vector<double> intr, dist;
  };
  const std::vector<Case> cases = {
    {"plumb_bob", {210, 205, 160, 120}, {-0.04, 0.08, 1e-4, -3e-4, -0.04}}, {"fisheye", {140, 140, 160, 120}, {-0.01, 0.002}}, {"equidistant", {140, 140, 160, 120}, {}},
    {"omnidir", {110, 110, 160, 160, 1.0}, {-0.02, 0.003, 1e-4, -2e-4}}, {"equirectangular", {384, 256}, {}}, {"atan", {210, 205, 160, 120}, {0.6}},
    {"rational_polynomial", {210, 205, 160, 120}, {0.05, -0.02, 1e-4, -2e-4, 0.01, 0.03, -0.01, 0.002, 99.0}}};
  for (const auto& c : cases) {
    auto proj = camera::create_camera(c.name, c.intr, c.dist);  // the reference's factory
    if (!proj) return 1;
    const camera::NidregCameraParams cp = camera::nidreg_camera_params(*proj);
    std::printf("%s %d", c.name, cp.model_id);
    for (int i = 0; i < 5; i++) std::printf(" %.17g", cp.intrinsics[i]);
    for (int i = 0; i < 8; i++) std::printf(" %.17g", cp.distortion[i]);
    const Eigen::Vector2d uv = proj->project(Eigen::Vector3d(0.3, -0.2, 2.0));  // the reference's CPU projection still works
    std::printf(" %.17g %.17g\n", uv[0], uv[1]);
  }
  if (camera::create_camera("pinhole", {1, 2, 3, 4}, {}) || camera::create_camera("plumb_bob", {1, 2, 3}, {})) return 2;
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  if (std::strcmp(argv[1], "--describe") == 0) return describe();
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  int W, H, N, bins, nintr, ndist;
  char model[64] = {0};
  double intr[5], dist[8], se3[7], max_fov, T[16];
  if (fread(model, 1, 64, f) != 64) return 4;
  if (fread(&W, 4, 1, f) != 1 || fread(&H, 4, 1, f) != 1 || fread(&N, 4, 1, f) != 1 || fread(&bins, 4, 1, f) != 1 || fread(&nintr, 4, 1, f) != 1 || fread(&ndist, 4, 1, f) != 1) return 4;
  if (fread(intr, 8, 5, f) != 5 || fread(dist, 8, 8, f) != 8 || fread(se3, 8, 7, f) != 7 || fread(&max_fov, 8, 1, f) != 1 || fread(T, 8, 16, f) != 16) return 4;
  cv::Mat img8(H, W, CV_8UC1), img64(H, W, CV_64FC1);
  if (fread(img8.data, 1, size_t(W) * H, f) != size_t(W) * H) return 4;
  for (int y = 0; y < H; y++)
    for (int x = 0; x < W; x++) img64.at<double>(y, x) = img8.at<unsigned char>(y, x) * (1.0 / 255.0);
  std::vector<Eigen::Vector4d> pts(N);
  std::vector<double> ints(N);
  if (fread(pts.data(), 32, N, f) != size_t(N) || fread(ints.data(), 8, N, f) != size_t(N)) return 4;
  fclose(f);

  auto proj = camera::create_camera(model, std::vector<double>(intr, intr + nintr), std::vector<double>(dist, dist + ndist));
  if (!proj) return 6;
  auto frame = std::make_shared<vlcal::Frame>();  // the reference's Frame (include/vlcal/common/frame.hpp)
  frame->num_points = N;
  frame->points = pts.data();
  frame->intensities = ints.data();

  const vlcal::NIDCost cost(proj, img64, frame, bins);
  typedef ceres::Jet<double, 7> J;
  J params[7], res;
  for (int k = 0; k < 7; k++) params[k] = J(se3[k], k);
  if (!cost(params, &res)) return 7;
  double c2 = 0.0;
  if (!cost(se3, &c2)) return 8;
  std::printf("%.17g", res.a);
  for (int k = 0; k < 7; k++) std::printf(" %.17g", res.v[k]);
  std::printf(" %.17g\n", c2);
  return 0;
}
