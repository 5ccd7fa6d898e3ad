/*
 * nidreg.h -- C ABI of the MI355X-native NID direct LiDAR-camera registration core.
 *
 * This is the drop-in boundary for the hot path of koide3/direct_visual_lidar_calibration's
 * `calibrate`: everything behind these entry points runs as hand-written HIP kernels on gfx950
 * (direct_visual_lidar_calibration_amd/csrc/).  Plain pointers and sizes only; no C++/torch types.
 *
 * Reference interfaces each entry point replaces (paths relative to the reference tree):
 *   nidreg_model_from_name   camera::create_camera model strings + parameter counts
 *                            (src/camera/create_camera.cpp:17-51)
 *   nidreg_create            vlcal::NIDCost::NIDCost           (include/vlcal/costs/nid_cost.hpp:23-34)
 *                            vlcal::CostCalculatorNID ctor     (src/vlcal/calib/cost_calculator_nid.cpp:13-17)
 *   nidreg_eval              NIDCost::operator()<double | ceres::Jet<double,7>>
 *                                                              (include/vlcal/costs/nid_cost.hpp:36-107)
 *   nidreg_eval_iso          CostCalculatorNID::calculate      (src/vlcal/calib/cost_calculator_nid.cpp:21-67)
 *   nidreg_eval_multi        MultiNIDCost::operator()          (src/vlcal/calib/visual_camera_calibration.cpp:147-173)
 *   nidreg_eval_iso_multi    the Nelder-Mead objective's sum over pairs
 *                                                              (src/vlcal/calib/visual_camera_calibration.cpp:103-119)
 *   nidreg_project           GenericCameraBase::project        (include/camera/generic_camera_base.hpp:29)
 *   nidreg_view_culling      ViewCulling::cull                 (src/vlcal/calib/view_culling.cpp:21-92)
 *   nidreg_colorizer_*       vlcal::PointsColorUpdater         (src/vlcal/common/points_color_updater.cpp:26-61)
 *   nidreg_generate_lidar_image  vlcal::generate_lidar_image   (src/vlcal/preprocess/generate_lidar_image.cpp:7-41)
 *   nidreg_equalize_intensities  the rank equalisation loop    (src/vlcal/preprocess/preprocess.cpp:464-473)
 *   nidreg_shard_*           (no reference counterpart) split-phase evaluation of one pair whose
 *                            points are sharded over several GPUs, ONE PROCESS PER GPU; the caller
 *                            all-reduces the fixed-point histogram (RCCL) between the phases.
 *   desc.num_devices /       (no reference counterpart) one pair over several GPUs inside ONE process: nidreg_create /
 *   NIDREG_DEVICES           nidreg_create_from_cloud cut the cloud along the histogram column (every GPU owns a range of
 *                            column groups and the points that fall into them) and every GPU stores its columns of the
 *                            integer histogram into every other GPU's replica, once per evaluation (the reference's calibrate is a
 *                            single process, src/calibrate.cpp:117-120)
 *   nidreg_destroy           ~NIDCost / ~CostCalculatorNID
 *
 * Conventions
 *   - return value: 0 = ok; NIDREG_FALSE (1) = the functor's `return false` (non-finite NID,
 *     nid_cost.hpp:98-102, or the 0.2 m / 2 deg trust gate, visual_camera_calibration.cpp:154);
 *     < 0 = error (nidreg_last_error() gives the text for the calling thread).
 *   - pose for nidreg_eval*: Sophus SE3d::data() order [qx qy qz qw tx ty tz] = T_camera_lidar,
 *     quaternion NOT re-normalised (the reference differentiates the un-normalised formula).
 *     The 7-gradient returned is the ambient Jet gradient d NID / d [qx qy qz qw tx ty tz].
 *   - pose for nidreg_eval_iso*: row-major 4x4 T_camera_lidar (Eigen::Isometry3d::matrix()).
 *   - histograms handed back are row-major [bin_image][bin_points] like `hist(bin_image, bin_points)`.
 *   - threading: concurrent calls on DIFFERENT handles from different host threads are safe (the
 *     reference evaluates one NIDCost per OpenMP thread, visual_camera_calibration.cpp:161);
 *     concurrent calls on the same handle are not supported (never happens in the reference).
 *     Handles sharded over several GPUs are evaluated one after the other on every device they share
 *     (a per-device lock inside nidreg_eval*: each of them uses all of its GPUs anyway).
 *   - an evaluation that has its device to itself runs with issue priority by progress in the two streaming kernels;
 *     evaluations that share a device (concurrent callers) run without it.  The cost and the histograms do not depend
 *     on it, nor on the chunk tables or the GPU count (fixed-point histogram, fixed-point entropy sums: bit-identical);
 *     the gradient is equal up to the order of the workgroup partials, which is a fixed function of the handle.
 *   - NEAREST handles (nidreg_eval_iso*): the integer histogram is the reference's, count for count, for every point whose
 *     decisions (inside the FoV cone, inside the image, which pixel) the arithmetic fixes -- +, -, *, /, sqrt follow the
 *     reference's expression order without contraction.  Three camera models put a libm function between the point and its
 *     pixel (fisheye, atan: atan2 / atan; equirectangular: asin, atan2): there the pixel is exact up to the last place in which
 *     the device's libm and the host's agree, i.e. a point within ~3e-14 px of a pixel boundary may fall on either side
 *     (two CPU builds with different libms disagree in the same way); none of the 10^7 points of the full-size test clouds does.
 *   - the caller keeps ownership of every host buffer; nidreg_create copies what it needs.
 */
#ifndef NIDREG_H
#define NIDREG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NIDREG_OK 0
#define NIDREG_FALSE 1
#define NIDREG_ERR_INVALID (-1)     /* bad argument / unsupported configuration */
#define NIDREG_ERR_HIP (-2)         /* HIP runtime error */
#define NIDREG_ERR_NO_DEVICE (-3)   /* no usable gfx950 device */

/* camera models (create_camera.cpp:34-51) */
#define NIDREG_MODEL_PLUMB_BOB 0            /* "plumb_bob": 4 intrinsics, 5 distortion (k1 k2 p1 p2 k3) */
#define NIDREG_MODEL_FISHEYE 1              /* "fisheye" | "equidistant": 4 + 4 */
#define NIDREG_MODEL_OMNIDIR 2              /* "omnidir": 5 (fx fy cx cy xi) + 4 */
#define NIDREG_MODEL_EQUIRECTANGULAR 3      /* "equirectangular": 2 (W H) + 0 */
#define NIDREG_MODEL_ATAN 4                 /* "atan": 4 + 1 */
#define NIDREG_MODEL_RATIONAL_POLYNOMIAL 5  /* "rational_polynomial": 4 + 8 (k1 k2 p1 p2 k3 k4 k5 k6) */

/* which reference functor the handle implements */
#define NIDREG_MODE_SPLINE 0   /* NIDCost: B-spline soft assignment, value + Jacobian */
#define NIDREG_MODE_NEAREST 1  /* CostCalculatorNID: FoV gate + nearest pixel, integer histogram */

/* per-point arithmetic */
#define NIDREG_PREC_FP64 0  /* transform / projection / weights in double (parity mode) */
#define NIDREG_PREC_FP32 1  /* REMOVED (round 5): float transform / projection bought 8 % for |dNID| <= 2e-5; refused with NIDREG_ERR_INVALID */

#define NIDREG_IMAGE_F64 0  /* CV_64FC1 normalised to [0,1] (what NIDCost receives) */
#define NIDREG_IMAGE_U8 1   /* CV_8UC1 (what CostCalculatorNID receives) */

/* nidreg_desc.bins: 2 .. NIDREG_MAX_BINS_WIDE, like the reference's `const int bins` (nid_cost.hpp:23, --nid_bins of
 * src/calibrate.cpp:175).  The kernels hold NIDREG_MAX_BINS bins per axis; more are accepted while at most NIDREG_MAX_BINS of
 * them are OCCUPIED on each axis -- always the case for the reference's own data path (8-bit camera images,
 * visual_camera_calibration.cpp:204; intensities rank-equalised to 256 levels, preprocess.cpp:464-473).  The NID and its
 * gradient depend on the multiset of histogram cells and on the marginals only, so the handle runs on the occupied bins,
 * relabelled 0, 1, 2, ... (same cost, same gradient), and nidreg_get_hist* expand them back to the caller's layout.  An input
 * that occupies more than 256 bins on an axis is refused with NIDREG_ERR_INVALID (never truncated). */
#define NIDREG_MAX_BINS 256
#define NIDREG_MAX_BINS_WIDE 4096
#define NIDREG_MAX_DEVICES 16

/* nidreg_desc.flags */
#define NIDREG_FLAG_EXT_STREAM 2  /* launch on desc.ext_stream even when it is NULL (= the legacy default
                                     stream, e.g. torch's current stream); otherwise NULL means "own stream" */
#define NIDREG_FLAG_NEAREST_EXACT 4 /* NEAREST handles: every point through the exact decision tier (the reference's expression
                                     order), none through the fast one -- for A/B runs and bisecting; the environment variable
                                     NIDREG_NEAREST_EXACT=1 does the same for every handle of the process */
#define NIDREG_FLAG_INPUT_ORDER 1 /* keep the caller's point order inside each column group instead of the
                                     default Morton order of the LiDAR-frame bearing (results are
                                     bit-identical either way; the default gathers ~2x faster) */

typedef struct nidreg_handle nidreg_handle;

typedef struct nidreg_desc {
  int32_t struct_size;      /* sizeof(nidreg_desc), for forward compatibility */
  int32_t device_id;        /* HIP device ordinal */
  int32_t model_id;         /* NIDREG_MODEL_* */
  int32_t mode;             /* NIDREG_MODE_* */
  int32_t precision;        /* NIDREG_PREC_* */
  int32_t bins;             /* nid_bins, 2..NIDREG_MAX_BINS_WIDE (4096), see above: beyond NIDREG_MAX_BINS (256) the handle runs on the
                               OCCUPIED bins (the reference's data path quantises both inputs to 256 levels: 8-bit images,
                               visual_camera_calibration.cpp:204; intensities equalised to floor(256 i / n) / 256, preprocess.cpp:464-473);
                               an input that occupies more than 256 bins on an axis, or bins outside the range, is REFUSED with
                               NIDREG_ERR_INVALID (tests/test_abi.py), never truncated. */
  double intrinsics[5];     /* exactly the model's count is read */
  double distortion[8];     /* already zero-padded / truncated like create_camera.cpp:24-27 */
  int32_t width, height;    /* image cols, rows */
  int32_t image_dtype;      /* NIDREG_IMAGE_* */
  int32_t flags;            /* NIDREG_FLAG_* */
  const void* image;        /* host pointer, rows x cols */
  int64_t image_row_stride; /* bytes between rows (cv::Mat::step) */
  int64_t num_points;       /* Frame::size() */
  const double* points;     /* host; Eigen::Vector4d (x y z 1) per point */
  int64_t point_stride;     /* bytes between points (32 for Frame::points) */
  const double* intensities;/* host; Frame::intensities */
  double max_fov;           /* NEAREST only: estimate_camera_fov() result [rad] */
  /* tuning (0 = default) */
  int32_t columns_per_group;/* histogram columns (bin_points values) a workgroup owns in LDS */
  int32_t target_blocks;    /* approximate number of point chunks = workgroups per pass */
  int32_t lds_copies;       /* lane-private copies of each histogram cell in LDS (1,2,4,8,16) */
  int32_t reserved1;
  int64_t scale_points;     /* points the fixed-point scale must hold (0 = num_points); shards of one
                               pair pass the TOTAL so that every rank uses the same 2^-frac unit */
  /* optional externally owned device resources (sharded multi-GPU use); NULL = internal */
  void* ext_stream;         /* hipStream_t the handle launches on */
  void* ext_hist;           /* device buffer of nidreg_hist_words(bins) 64-bit words */
  void* ext_out;            /* device buffer of NIDREG_OUT_DOUBLES doubles */
  /* one pair spread over several GPUs inside the library (single process): num_devices > 1 cuts the cloud along the
     pose-independent histogram column -- device_ids[k] owns a contiguous range of column groups, balanced by their point
     counts, and holds the points that fall into them -- so the GPUs' joint histograms have disjoint support: every GPU keeps
     a replica of the whole integer histogram and stores its own columns into every other GPU's replica, directly between the
     GPUs, once per evaluation (plain stores: no two GPUs write the same word).  Cost and histograms are bit-identical to the unsharded handle's, the gradient differs by
     summation order only.  The same device may be listed several times (a 1-GPU box exercising the protocol).
     0 = device_id alone, unless the environment variable NIDREG_DEVICES="0,1,..." is set -- which shards every handle
     (nidreg_create and nidreg_create_from_cloud), so a caller that knows nothing about GPUs (the reference's
     `new NIDCost(proj, image, points, bins)`) uses the whole node. */
  int32_t num_devices;
  int32_t device_ids[NIDREG_MAX_DEVICES];
} nidreg_desc;

#define NIDREG_OUT_DOUBLES 16  /* [0]=cost [1..7]=grad7 [8]=status [9]=inliers [10..15] reserved */

/* model string -> NIDREG_MODEL_* (or -1); optionally returns the parameter counts */
int nidreg_model_from_name(const char* name, int* num_intrinsics, int* num_distortion);

int nidreg_device_count(void);

int nidreg_create(const nidreg_desc* desc, nidreg_handle** out);
void nidreg_destroy(nidreg_handle* h);

/* Device-resident cloud (one upload per LiDAR-camera pair).  nidreg_create_from_cloud builds a handle from
 * it entirely on the GPU: optional ViewCulling::cull at T_camera_lidar (row-major 4x4; NULL = no culling;
 * min_z / enable_depth_buffer_culling as in nidreg_view_culling), then bucketing, Morton sort and record
 * gather -- the reference's per-outer-iteration `cull -> new NIDCost` (visual_camera_calibration.cpp:
 * 201-206) without a host round trip.  desc->points / intensities / num_points are ignored.  desc->num_devices /
 * NIDREG_DEVICES are honoured: the cull and the sort run on the cloud's GPU, every shard then takes the records of its
 * column groups device to device. */
typedef struct nidreg_cloud nidreg_cloud;
int nidreg_cloud_create(int device_id, const double* points, int64_t point_stride, const double* intensities, int64_t num_points, nidreg_cloud** out);
void nidreg_cloud_destroy(nidreg_cloud* cloud);
int nidreg_create_from_cloud(const nidreg_desc* desc, const nidreg_cloud* cloud, const double* T_camera_lidar, double min_z, int enable_depth_buffer_culling, nidreg_handle** out);

/* NIDCost::operator(): cost (+ gradient when grad7 != NULL) at se3 = [qx qy qz qw tx ty tz] */
int nidreg_eval(nidreg_handle* h, const double* se3, double* cost, double* grad7);

/* n synchronous nidreg_eval calls back to back, as an optimiser's inner loop issues them (each evaluation completes --
 * host sync included -- before the next starts): se3s n x 7, costs n (nullable), grads7 n x 7 (NULL = cost only).
 * Returns < 0 on the first error, else NIDREG_FALSE if any evaluation was rejected, else NIDREG_OK. */
int nidreg_eval_batch(nidreg_handle* h, const double* se3s, int n, double* costs, double* grads7);

/* CostCalculatorNID::calculate at a row-major 4x4 T_camera_lidar */
int nidreg_eval_iso(nidreg_handle* h, const double* T_camera_lidar, double* cost);

/* Asynchronous evaluation, for callers that hold several INDEPENDENT poses -- Nelder-Mead's initial simplex
 * (include/dfo/nelder_mead.hpp:32-57 evaluates its n + 1 vertices before the first step), multi-start, finite-difference
 * checks; a line search (BFGS, visual_camera_calibration.cpp:210-229) has nothing to overlap and keeps nidreg_eval.
 * nidreg_submit queues the kernels of NIDCost::operator() (want_grad != 0: with the Jacobian) and returns at once with a
 * ticket; nidreg_submit_iso does the same for CostCalculatorNID::calculate.  nidreg_wait blocks until that evaluation is
 * complete and returns what nidreg_eval / nidreg_eval_iso would have (cost, grad7 when it was asked for; NIDREG_FALSE for a
 * non-finite NID).  The evaluations of one handle run back to back in submission order -- no host round trip between
 * them --, each into its own host-mapped result block; tickets may be collected in any order.  At most 8 evaluations of a
 * handle may be in flight (NIDREG_ERR_INVALID beyond that: collect first); a handle must not be destroyed, and its
 * histograms not read, while tickets are outstanding.  Handles spread over several GPUs, handles with caller-provided
 * result buffers and handles with timing enabled evaluate inside nidreg_submit (the ticket then only carries the results).
 * Results are bit-identical to the synchronous calls. */
int nidreg_submit(nidreg_handle* h, const double* se3, int want_grad, int64_t* ticket);
int nidreg_submit_iso(nidreg_handle* h, const double* T_camera_lidar, int64_t* ticket);
int nidreg_wait(nidreg_handle* h, int64_t ticket, double* cost, double* grad7);

/* nidreg_eval_batch's arguments and results for n INDEPENDENT poses, through the submit / wait pair: up to 7 evaluations
 * queued ahead of the one being collected (the handle must have no outstanding tickets). */
int nidreg_eval_pipelined(nidreg_handle* h, const double* se3s, int n, double* costs, double* grads7);

/* MultiNIDCost::operator(): trust gate against init_se3 (NULL = no gate), all pairs in flight at once, plain sum
 * of costs / gradients, NIDREG_FALSE if the gate rejects or any pair is non-finite.  Handles on different GPUs run
 * concurrently (every GPU's histogram pass is queued before the rest).  2..16 compatible SPLINE handles on ONE GPU
 * (same camera, image size, bins, precision) are evaluated by ONE grid per pass over the chunks of all pairs -- three
 * launches in all instead of three per pair; each pair keeps its own histogram, fixed-point unit and result block, so
 * every pair's cost and histogram are bit-identical to evaluating the handles one by one; the gradient's workgroup
 * partials follow the group's chunk table, i.e. it is equal up to summation order (NIDREG_NO_MULTI_GRID=1 in the
 * environment forces the per-pair launches). */
int nidreg_eval_multi(nidreg_handle* const* handles, int n, const double* init_se3, const double* se3, double* cost, double* grad7);

/* sum_i CostCalculatorNID_i::calculate(T); compatible NEAREST handles on one GPU share one grid per pass like nidreg_eval_multi */
int nidreg_eval_iso_multi(nidreg_handle* const* handles, int n, const double* T_camera_lidar, double* cost);

/* raw histograms of the most recent evaluation (any pointer may be NULL).
 * joint: bins*bins doubles [bin_image][bin_points]; hist_image, hist_points: bins doubles. */
int nidreg_get_hist(nidreg_handle* h, double* joint, double* hist_image, double* hist_points);
/* the same joint histogram as exact integers: fixed-point words (value * 2^frac_bits) in SPLINE
 * mode, counts (frac_bits = 0) in NEAREST mode. */
int nidreg_get_hist_fixed(nidreg_handle* h, int64_t* joint, int64_t* inliers, int* frac_bits);

/* GenericCameraBase::project for n points (host in, host out), evaluated on the device with the
 * same projection code the cost kernels use.  p3: n x 3 doubles, uv: n x 2 doubles,
 * jac (nullable): n x 6 doubles = d(u,v)/d(x,y,z) row-major 2x3. */
int nidreg_project(nidreg_handle* h, const double* p3, int64_t n, double* uv, double* jac);
/* the same without a handle (used for one-time host set-up such as estimate_camera_fov,
 * src/vlcal/common/estimate_fov.cpp:17-51); intrinsics[5] / distortion[8] as in nidreg_desc.
 * device_id = NIDREG_DEVICE_HOST: the device's scalar projection code compiled for the host (fp64; equal to the device's
 * result up to the reciprocal seeds, ~1e-14 relative) -- for callers that probe ONE point at a time, like the ~240 probes of
 * estimate_camera_fov's NelderMead<2>, which is host work in the reference too. */
#define NIDREG_DEVICE_HOST (-1)
/* vlcal::estimate_camera_fov (src/vlcal/common/estimate_fov.cpp:36-51; estimate_direction :17-34 with NelderMead<2>,
 * include/dfo/nelder_mead.hpp): the largest view angle over the pixels (0,0), (W/2,0), (0,H/2) [rad] -- what
 * CostCalculatorNID's max_fov, ViewCulling's min_z and generate_lidar_image take.  Host only (no GPU involved). */
int nidreg_estimate_camera_fov(int model_id, const double* intrinsics, const double* distortion, int width, int height, double* max_fov);
int nidreg_project_model(int model_id, const double* intrinsics, const double* distortion, int device_id, int precision, const double* p3, int64_t n, double* uv, double* jac);

/* ViewCulling::cull (src/vlcal/calib/view_culling.cpp:21-92) on the device: FoV gate against
 * min_z = cos(estimate_camera_fov), in-image test, per-pixel depth buffer (float distance) and the
 * +0.1 m keep threshold.  points: host, (x y z 1) doubles with the given byte stride; T: row-major 4x4
 * T_camera_lidar.  Writes the surviving indices (ascending; identical to the in-repo oracle's and to the reference's
 * source compiled against the stand-in Eigen of oracle/shim -- a build against real Eigen, whose fixed-size reductions
 * and 4x4 products may associate differently, can differ by 1 ulp in a borderline FoV / pixel / depth decision) into
 * indices_out (capacity num_points) and returns their number, or a negative error code. */
int64_t nidreg_view_culling(int model_id, const double* intrinsics, const double* distortion, int device_id, int width, int height, double min_z, int enable_depth_buffer_culling,
                            const double* points, int64_t point_stride, int64_t num_points, const double* T_camera_lidar, int32_t* indices_out);

/* PointsColorUpdater (src/vlcal/common/points_color_updater.cpp): the per-frame recolouring of the
 * whole cloud the viewer runs every 50 ms while the optimiser works (visual_lidar_visualizer.cpp:89-100).
 * create = the constructor (:26-35): uploads the 8-bit image, the points ((x y z 1) doubles, byte
 * stride >= 32) and the per-point intensity colours (n x RGBA float; the reference fills them with
 * glk::colormapf(TURBO, intensity), a table that lives in Iridescence, so the caller passes them;
 * NULL = (1,1,1,1) like the icosahedron constructor, :24).  min_nz = cos(estimate_camera_fov + 0.5 deg) (:12).
 * update = PointsColorUpdater::update (:37-61): colours[i] = (pix/255, pix/255, pix/255, 1) * float(w)
 * + intensity_colors[i] * float(1 - w) in float arithmetic, or (0,0,0,0) when the point is outside the
 * FoV / image.  colors_out: host n x 4 floats, or NULL to leave the result on the device
 * (nidreg_colorizer_device_colors; valid until the next update / destroy). */
typedef struct nidreg_colorizer nidreg_colorizer;
int nidreg_colorizer_create(int device_id, int model_id, const double* intrinsics, const double* distortion, int width, int height, const uint8_t* image, int64_t image_row_stride,
                            int64_t num_points, const double* points, int64_t point_stride, const float* intensity_colors, double min_nz, nidreg_colorizer** out);
int nidreg_colorizer_update(nidreg_colorizer* c, const double* T_camera_lidar, double blend_weight, float* colors_out);
const float* nidreg_colorizer_device_colors(nidreg_colorizer* c);
void nidreg_colorizer_destroy(nidreg_colorizer* c);

/* generate_lidar_image (src/vlcal/preprocess/generate_lidar_image.cpp:7-41): z-buffered rendering of
 * the cloud through the camera model.  min_nz = cos(estimate_camera_fov) (:10-11).  Per pixel the point
 * with the smallest squared camera-frame distance wins; on exact ties the largest index (the sequential
 * reference overwrites on `!(stored < sq_dist)`, :31).  intensity_image: height x width doubles (0 where
 * nothing landed), index_image: height x width int32 (-1 where nothing landed); either may be NULL.
 * Output is independent of the thread order and identical to the in-repo oracle's sequential loop (same 1-ulp caveat
 * about real Eigen's association order as nidreg_view_culling). */
int nidreg_generate_lidar_image(int model_id, const double* intrinsics, const double* distortion, int device_id, int width, int height, double min_nz, const double* points,
                                int64_t point_stride, const double* intensities, int64_t num_points, const double* T_camera_lidar, double* intensity_image, int32_t* index_image);

/* Intensity rank equalisation of an integrated cloud (src/vlcal/preprocess/preprocess.cpp:464-473,
 * src/preprocess_map.cpp): in place, intensities[rank order i] = floor(256 * i / n) / 256, the values the
 * NID path then bins.  Device radix sort; stable, i.e. equal intensities keep their index order (the
 * reference's std::sort leaves that order unspecified). */
int nidreg_equalize_intensities(int device_id, double* intensities, int64_t num_points);

/* ---- split-phase evaluation for a pair whose points are sharded across GPUs --------------
 * rank r:  nidreg_shard_hist(h, se3)      zero + accumulate this shard's fixed-point histogram
 *          <caller: all-reduce(sum, int64) of ext_hist over ranks, ordered on ext_stream>
 *          nidreg_shard_entropy(h)        entropy tail on the full histogram -> cost
 *          nidreg_shard_grad(h)           this shard's gradient partial -> out[1..7]
 *          <caller: all-reduce(sum, f64) of out[1..7]>
 *          nidreg_shard_finish(h, ...)    stream sync + read back
 * All shard calls are asynchronous on the handle's stream except nidreg_shard_finish. */
/* 64-bit words in a histogram buffer (desc.ext_hist must hold this many): bins*bins joint cells [bin_points][bin_image], 8 tail
 * words (0: inlier count, 1: fixed-point joint entropy), roundup8(bins) column sums (added by the histogram kernels' flush) and
 * roundup8(bins) row sums (added by k_entropy) -- bins*bins + 8 + 2*roundup8(bins) */
int64_t nidreg_hist_words(int bins);
int nidreg_shard_hist(nidreg_handle* h, const double* se3);
int nidreg_shard_entropy(nidreg_handle* h);
int nidreg_shard_grad(nidreg_handle* h);
int nidreg_shard_finish(nidreg_handle* h, double* cost, double* grad7);

/* ---- the same protocol with RCCL inside the library (one process per GPU; BASELINE north_star: "disjoint point slices with
 * a final RCCL all-reduce of the 2D histogram over xGMI") -----------------------------------------------------------------
 * Every rank creates a plain handle over ITS slice of the pair's cloud (desc.scale_points = the pair's total point count, so
 * that every rank uses the same fixed-point unit; no ext_hist / ext_out / device_ids) and gives it a communicator; from then on
 * nidreg_eval / nidreg_eval_iso / nidreg_eval_batch on that handle are COLLECTIVE calls (every rank, same pose) that run
 *     histogram kernel -> ncclAllReduce(int64 sum, nidreg_hist_words(bins) words) -> entropy tail ->
 *     gradient kernel  -> ncclAllReduce(float64 sum, 7 words)
 * on the handle's stream.  Replaces nothing in the reference (its only parallelism is the OpenMP loop over pairs,
 * src/vlcal/calib/visual_camera_calibration.cpp:161); the cost is bit-identical on every rank and to the unsharded handle's,
 * the gradient equal up to the order of the partial sums.  librccl.so is opened with dlopen at the first of these calls
 * (NIDREG_RCCL_LIB overrides the name); libnidreg.so does not link it.
 *   nidreg_rccl_unique_id     rank 0: ncclGetUniqueId into 128 bytes, which the caller hands to the other ranks (MPI, a file,
 *                             torch.distributed -- any way it likes)
 *   nidreg_shard_comm_init    every rank: ncclCommInitRank(world_size, id, rank) on the handle's device; the handle owns the
 *                             communicator and destroys it with itself
 *   nidreg_shard_attach_rccl  a communicator the caller owns (an ncclComm_t, passed as void*); NULL detaches
 * Both run ONE collective themselves before they accept the communicator: the ranks' table parameters (fixed-point fraction bits --
 * i.e. desc.scale_points --, bins, nidreg_hist_words, mode) are max-/min-reduced and every rank refuses alike
 * (NIDREG_ERR_INVALID, the handle stays detached) when they differ; a rank created with another unit would otherwise add
 * incompatible integers into a cost that is wrong yet identical on every rank.
 * FAILURE SEMANTICS: an evaluation is a sequence of collectives.  A rank that returns early with a negative status (a launch error,
 * a lost device) leaves its peers inside ncclAllReduce; the caller must then abort the communicator on the other ranks
 * (ncclCommAbort) -- RCCL has no timeout of its own -- and rebuild it before evaluating again. */
#define NIDREG_RCCL_ID_BYTES 128
int nidreg_rccl_unique_id(unsigned char* id128);
int nidreg_shard_comm_init(nidreg_handle* h, int world_size, int rank, const unsigned char* id128);
int nidreg_shard_attach_rccl(nidreg_handle* h, void* nccl_comm);

/* number of GPUs a handle is sharded over (1 = plain handle) and their device ordinals */
int nidreg_num_shards(nidreg_handle* h);
int nidreg_shard_devices(nidreg_handle* h, int* device_ids, int capacity);

/* release what the library keeps per device between handles: the scratch arenas of handle construction (upload staging, sort
 * keys) and the streams / host-mapped result blocks of destroyed handles (the next handle takes them: the reference builds new
 * cost objects in every outer iteration, visual_camera_calibration.cpp:199-208) */
void nidreg_trim(void);

/* device-side timing of the most recent nidreg_eval / nidreg_eval_iso in milliseconds (HIP events
 * recorded on the handle's stream, i.e. the stream the kernels run on):
 * [0]=whole launch sequence [1]=histogram memset [2]=histogram kernel [3]=entropy kernels
 * [4]=gradient kernel (0 if not run) [5]=gradient finalisation.
 * nidreg_set_timing(h, 1): an event after every kernel; (h, 2): only [0], around the whole evaluation; (h, 0): off. */
int nidreg_set_timing(nidreg_handle* h, int enable);
int nidreg_get_timing(nidreg_handle* h, float* ms6);

/* layout facts for DESIGN.md / bench: [0]=record bytes per point on device, [1]=number of chunks,
 * [2]=columns per group, [3]=fixed-point fraction bits, [4]=LDS bytes per workgroup,
 * [5]=padded image pitch, [6]=points stored, [7]=bit0: float32 records, bit1: the gradient-pass chunk table has chunks that run
 * across column groups (the looped kernel instantiations, csrc/nid_kernels.hpp Segments), bit2: so has the WIDE histogram
 * kernel's table, bit3: NEAREST handle whose evaluations use the fast decision tier (plumb_bob with a FoV cone below ~84 degrees,
 * fisheye, omnidir with xi >= 0, equirectangular; the other models and degenerate cones run the exact tier only;
 * csrc/nid_kernels.hpp NearestFast), bit4: cost+Jacobian evaluations launch no entropy kernel (bins <= 32: the gradient workgroups
 * sum the table themselves, csrc/nid_kernels.hpp kSelfEntropyCells), bit5: a synchronous cost+Jacobian evaluation that has its device to
 * itself runs as ONE launch (csrc/nid_fused.hpp: bins <= 32 and a cloud whose chunks fit the LDS stash), bits 8..15: LDS copies per
 * histogram cell, bits 16..27: workgroups of that launch, bit 28: its stash holds the projection context too (not only u, v) */
int nidreg_get_info(nidreg_handle* h, int64_t* info8);

const char* nidreg_last_error(void);
const char* nidreg_version(void);
/* 16 hex digits: sha256 over the kernel sources (csrc/nid_*.hpp, nid_kernels_*.hip, Makefile) this library was BUILT from.  The
 * measurement tooling stamps rocprofv3 summaries with it and refuses to stamp when it differs from the sources next to the library
 * (tools/kernel_stats_json.py, tools/traffic_from_pmc.py): a summary can then only ever describe the kernels that produced it. */
const char* nidreg_kernel_build(void);

#ifdef __cplusplus
}
#endif
#endif /* NIDREG_H */
