#include "nidreg_internal.hpp"

namespace nidreg_detail {

// ---- chunk tables: each chunk = one workgroup's slice of the bucketed record array.  A pass should be ONE round of
// co-resident workgroups (`target` of them; measured on cfg 2: the per-workgroup prologue / flush is amortised over more
// points and no partial last round is left -- 2048 chunks +4 %, 4096 +12 %), and the round ends with its LONGEST chunk.
// Rounds 1-3 gave every chunk the points of one column group only (a group split evenly into an integer number of chunks):
// exact on a cloud whose columns are equally full -- the rank equalisation of preprocess.cpp:464-473 makes them so for a
// WHOLE cloud --, but the clouds `calibrate` evaluates are view-culled (visual_camera_calibration.cpp:201-206) and a pair of
// a multi-pair set is a subset: with columns of 0.5 ... 1.5 x the mean an integer split leaves chunks of 3/4 ... 3/2 of the
// mean (measured: +20 % per point, profiles/archive/r03r_culled_cloud_ab*.json).  Since round 4 a chunk is a CONTIGUOUS RANGE of
// records that may run across group boundaries (nid_kernels.hpp Segments): the workgroup flushes / rebuilds its tile at
// every boundary, which costs about `overhead` records' worth of time (pipeline drain, 64 KB of LDS traffic, the first
// load latency of the next segment).  The table minimises the longest chunk under that cost model:
//     cost(chunk) = sum over its segments of (overhead + records),
// smallest bound C for which a greedy left-to-right fill needs <= target chunks (binary search; the fill never opens a
// segment shorter than `overhead` at the end of a chunk, and cuts inside a group at multiples of 64 records).  A pure function
// of (group offsets, target, overhead, max_segs): the gradient's partial sums keep a fixed order, run to run.
// max_segs = segments a chunk may hold: kMaxSegs for the single-column kernels (B > 128) and every NEAREST table, 1 where a
// workgroup's tile spans several columns (B <= 128: few groups, large tiles -- a chunk then ends at the group boundary as it
// always did).  Tables without a multi-segment chunk run the straight-line kernels of rounds 1-3, the others the looped
// (SEG) instantiations; NIDREG_MAX_SEGS=1 keeps every table on one segment per chunk (A/B runs).
// Chunk::pad = pair | slot << 8: pair = index into the multi-pair table (0 in a single-pair table, pair < 0), slot = number of
// segments in the table's earlier chunks -- the gradient pass stores one 12-double partial per SEGMENT, in this order.
// *nslots_out (nullable) = segments in the whole table.
int64_t fill_chunks(const int64_t* gcount, int NG, int64_t C, int64_t overhead, int max_segs, int pair, std::vector<Chunk>* out, int64_t* nslots_out) {
  int64_t nchunks = 0, cur = 0;  // cur = cost already in the open chunk (0: none open)
  int64_t nslots = 0, first_slot = 0;
  Chunk c{0, 0, 0, 0};
  auto close = [&]() {
    if (cur > 0 && out) {
      c.pad = uint32_t(pair < 0 ? 0 : pair) | (uint32_t(first_slot) << 8);
      out->push_back(c);
    }
    cur = 0;
  };
  for (int g = 0; g < NG; g++) {
    int64_t pos = gcount[g];
    const int64_t hi = gcount[g + 1];
    while (pos < hi) {
      const int64_t rem = hi - pos;
      int64_t room = C - cur - overhead;  // records of this group the open chunk can still take
      if (cur > 0 && (room < std::min(rem, std::max<int64_t>(overhead, 64)) || nslots - first_slot >= max_segs)) {  // not worth a segment (or the chunk is at its segment limit): next chunk
        close();
        continue;
      }
      if (cur == 0) {
        nchunks++;
        c.start = uint32_t(pos);
        c.count = 0;
        c.group = uint32_t(g);
        first_slot = nslots;
        room = std::max<int64_t>(room, 64);  // an empty chunk always makes progress
      }
      int64_t take = std::min(rem, room);
      if (take < rem) take = std::max<int64_t>(64, take / 64 * 64);  // cut inside a group: whole waves
      take = std::min(take, rem);
      c.count += uint32_t(take);
      cur += overhead + take;
      nslots++;
      pos += take;
      if (pos < hi) close();  // the group goes on in the next chunk
    }
  }
  close();
  if (nslots_out) *nslots_out = nslots;
  return nchunks;
}
// returns the number of segments (= gradient partial slots) of the table appended to `chunks`
// smallest cost bound for which the fill needs <= target chunks (or, when even one chunk per group is too many, the bound that
// gives one chunk per group) -- and the number of chunks at that bound
int64_t best_bound(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int64_t N, int64_t nonempty, int64_t* nchunks_out) {
  // fill_chunks(C) is non-increasing in C; C = everything in one chunk always fits (as far as max_segs allows)
  int64_t lo = overhead + 63, hi = N + nonempty * overhead + 64;  // lo: too small (or just feasible -- checked first), hi: feasible
  if (fill_chunks(gcount, NG, lo + 1, overhead, max_segs, -1, nullptr) <= target) {
    hi = lo + 1;
  } else {
    while (hi - lo > 1) {
      const int64_t mid = lo + (hi - lo) / 2;
      if (fill_chunks(gcount, NG, mid, overhead, max_segs, -1, nullptr) <= target) {
        hi = mid;
      } else {
        lo = mid;
      }
    }
  }
  *nchunks_out = fill_chunks(gcount, NG, hi, overhead, max_segs, -1, nullptr);
  return hi;
}
int64_t split_groups(const int64_t* gcount, int NG, int64_t target, int64_t overhead, int max_segs, int pair, std::vector<Chunk>& chunks) {
  const int64_t N = gcount[NG] - gcount[0];
  if (N <= 0) return 0;
  target = std::max<int64_t>(target, 1);
  overhead = std::max<int64_t>(overhead, 0);
  max_segs = std::max(1, std::min(max_segs, kMaxSegs));
  int64_t nonempty = 0;
  for (int g = 0; g < NG; g++) nonempty += gcount[g + 1] > gcount[g] ? 1 : 0;
  int64_t n_one = 0;
  const int64_t c_one = best_bound(gcount, NG, target, overhead, 1, N, nonempty, &n_one);
  int64_t bound = c_one;
  int segs = 1;
  if (max_segs > 1) {
    // Chunks across groups only where they PAY: the looped kernel instantiations run 2-6 % slower per point than the
    // straight-line ones (measured, profiles/archive/r04c_culled_cloud_ab.jsonl: on clouds whose columns are nearly equally full the
    // better balance did not make up for it), so the segmented table must beat the best one-group-per-chunk table by
    // NIDREG_SEG_MIN_GAIN (default 10 %) in the cost model -- rounds x longest chunk -- to be chosen.
    const char* mg = std::getenv("NIDREG_SEG_MIN_GAIN");
    const double min_gain = mg ? std::max(0.0, std::strtod(mg, nullptr)) : 0.10;
    int64_t n_seg = 0;
    const int64_t c_seg = best_bound(gcount, NG, target, overhead, max_segs, N, nonempty, &n_seg);
    const double t_one = double(c_one) * double((n_one + target - 1) / target), t_seg = double(c_seg) * double((n_seg + target - 1) / target);
    if (t_seg * (1.0 + min_gain) < t_one) {
      bound = c_seg;
      segs = max_segs;
    }
  }
  int64_t nslots = 0;
  fill_chunks(gcount, NG, bound, overhead, segs, pair, &chunks, &nslots);
  return nslots;
}
// records' worth of time one more segment costs a workgroup of the given kernel (measured orders of magnitude: a WIDE
// histogram workgroup streams ~400 records/us and a boundary costs it ~2.5 us; a gradient / generic workgroup ~150-200
// records/us and ~2 us).  NIDREG_SEG_OVERHEAD=<records> overrides both (A/B runs).
int max_segments(int mode, int GW) {
  int m = (mode == NIDREG_MODE_NEAREST || GW == 1) ? kMaxSegs : 1;
  if (const char* e = std::getenv("NIDREG_MAX_SEGS")) m = std::max(1, std::min(m, int(std::strtol(e, nullptr, 10))));
  return m;
}
int64_t segment_overhead(bool wide_hist) {
  if (const char* e = std::getenv("NIDREG_SEG_OVERHEAD")) return std::max<int64_t>(0, std::strtoll(e, nullptr, 10));
  return wide_hist ? 1024 : 384;
}

// How many chunks (= workgroups) a pass over `points` records gets.  A FULL round -- every co-resident slot of the GPU, per_cu
// workgroups on each CU -- is right for the 10M-point headline and wrong for the clouds `calibrate` usually sees: a workgroup
// costs a fixed prologue + epilogue (tile zeroing, the G columns' logarithms, the flush's atomics on the cells every other
// workgroup flushes too, the partial reduction), and the workgroups of a CU share its issue slots, so a pass costs about
//     a x chunks / CUs  +  b x points / chunks          (prologues on the busiest CU + the sweeps of one workgroup's slice),
// smallest at chunks ~ sqrt(points).  Measured (profiles/archive/r04i_small_cloud_sweep.jsonl, synchronous cost+Jacobian evaluation,
// best chunk count against the full round's): 30k points 64-128 chunks, 28 us against 37; 100k 128, 29 against 48; 300k
// 192-256, 35 against 51; 1M 384-512, 45 against 54; 3M 512-768, 71 against 76; 10M 1024 (the full round).  The rule is the
// square root through those points -- CUs/2 chunks at 100k points --, capped by the full round (reached at 6.4M points).
// NIDREG_FULL_ROUND=1 restores the full round at every size (A/B runs).
int64_t round_chunks(int per_cu, int num_cus, int64_t points) {
  const int64_t full = std::max<int64_t>(1, int64_t(per_cu) * num_cus);
  static const bool always_full = [] {
    const char* e = std::getenv("NIDREG_FULL_ROUND");
    return e && *e && *e != '0';
  }();
  if (always_full) return full;
  double t = 0.5 * double(num_cus) * std::sqrt(double(std::max<int64_t>(points, 1)) / 1.0e5);
  // Round 5: from 0.6 CUs on, whole multiples of the CU count (the nearest one on a logarithmic scale).  With the per-workgroup
  // overhead of this round's kernels (fast_log, the reduction through LDS) a count between two multiples -- 569 chunks on 256 CUs
  // -- leaves some CUs a workgroup more than the others and loses 3-8 % against the multiple next to it: 500k points 280 -> 256
  // chunks 35.5 -> 32.7 us, 2M 569 -> 512 57.3 -> 53.6, 3M 700 -> 768 70.9 -> 67.1, 4M 802 -> 768 80.3 -> 77.1
  // (profiles/r05z_small_cloud_sweep_b16.jsonl; 256-bin tables were multiples of their 256 column groups already).
  if (t >= 0.6 * double(num_cus)) {
    int k = 1;
    while (t > double(num_cus) * std::sqrt(double(k) * double(k + 1))) k++;
    t = double(k) * double(num_cus);
  }
  return std::max<int64_t>(1, std::min<int64_t>(full, int64_t(t + 0.5)));
}
// ... for a handle's own tables, in whole multiples of its non-empty column groups where it has several: a target between
// two multiples splits SOME groups once more and leaves the longest chunk as it was (B = 256, 1M points: 384 chunks 53 us,
// 256 chunks 47 us, 512 chunks 47 us), and a target below the group count would put several groups into every chunk --
// the looped kernels, 60 us against 34 us for 256 one-group chunks at 30k points.
int64_t snap_to_groups(int64_t target, const int64_t* gcount, int NG, int64_t cap) {
  int64_t nonempty = 0;
  for (int g = 0; g < NG; g++) nonempty += gcount[g + 1] > gcount[g] ? 1 : 0;
  if (nonempty <= 1) return target;
  int64_t per_group = std::max<int64_t>(1, (target + nonempty / 2) / nonempty);
  while (per_group > 1 && per_group * nonempty > cap) per_group--;
  return per_group * nonempty;
}

int resolve_wide_bins(const nidreg_desc* d, const nidreg_cloud* cloud, WideBins& wb) {
  const int B = d->bins;
  if (d->ext_hist) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins > 256 with a caller-provided histogram buffer (ext_hist) is not supported");
  if (!d->image || d->width < 1 || d->height < 1) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image");
  if (d->image_row_stride < int64_t(d->width) * (d->image_dtype == NIDREG_IMAGE_F64 ? 8 : 1)) return fail(NIDREG_ERR_INVALID, "nidreg_create: image_row_stride smaller than a row");
  std::vector<unsigned char> used_img(size_t(B), 0), used_pts(size_t(B), 0);
  const bool f64 = d->image_dtype == NIDREG_IMAGE_F64;
  for (int y = 0; y < d->height; y++) {
    const unsigned char* row = static_cast<const unsigned char*>(d->image) + size_t(y) * size_t(d->image_row_stride);
    for (int x = 0; x < d->width; x++) {
      int b;
      if (f64) {
        double v;
        std::memcpy(&v, row + size_t(x) * 8, 8);
        b = std::max(0, std::min(cast_int(v * double(B)), B - 1));  // nid_cost.hpp:78-79 (k_build_bin_image)
      } else {
        b = std::max(0, std::min(B - 1, cast_int(double(row[x]) / 255.0 * double(B))));  // cost_calculator_nid.cpp:43-46
      }
      used_img[size_t(b)] = 1;
    }
  }
  if (cloud) {
    HIP_TRY(hipSetDevice(cloud->device));
    HIP_TRY(mark_bins_device(cloud->d_int, cloud->n, B, used_pts.data()));
  } else {
    for (int64_t i = 0; i < d->num_points; i++) used_pts[size_t(std::max(0, std::min(B - 1, cast_int(d->intensities[i] * double(B)))))] = 1;  // nid_cost.hpp:49
  }
  wb.user_bins = B;
  wb.lut_img.assign(size_t(B), 0);
  wb.lut_pts.assign(size_t(B), 0);
  wb.inv_img.clear();
  wb.inv_pts.clear();
  for (int b = 0; b < B; b++) {
    if (used_img[size_t(b)]) {
      wb.lut_img[size_t(b)] = uint16_t(wb.inv_img.size() & 0xffff);
      wb.inv_img.push_back(uint16_t(b));
    }
    if (used_pts[size_t(b)]) {
      wb.lut_pts[size_t(b)] = uint16_t(wb.inv_pts.size() & 0xffff);
      wb.inv_pts.push_back(uint16_t(b));
    }
  }
  if (wb.inv_img.size() > size_t(NIDREG_MAX_BINS) || wb.inv_pts.size() > size_t(NIDREG_MAX_BINS))
    return fail(NIDREG_ERR_INVALID, "nidreg_create: bins = " + std::to_string(B) + " with " + std::to_string(wb.inv_img.size()) + " occupied image bins and " + std::to_string(wb.inv_pts.size()) +
                                      " occupied intensity bins: more than 256 bins per axis are supported only while at most 256 of them are occupied (8-bit images and 256-level "
                                      "equalised intensities, what the reference's own pipeline produces, always are); refused, not truncated");
  wb.compact_bins = int(std::max<size_t>(2, std::max(wb.inv_img.size(), wb.inv_pts.size())));
  return NIDREG_OK;
}

int create_impl(const nidreg_desc* d, const nidreg_cloud* cloud, const double* T_cull, double min_z, int enable_depth, const CreateOpts& opts, nidreg_handle** out) {
  if (!d || !out) return fail(NIDREG_ERR_INVALID, "nidreg_create: null argument");
  *out = nullptr;
  if (d->struct_size != int32_t(sizeof(nidreg_desc))) return fail(NIDREG_ERR_INVALID, "nidreg_create: struct_size mismatch");
  if (d->model_id < 0 || d->model_id > 5) return fail(NIDREG_ERR_INVALID, "nidreg_create: unknown camera model");
  // The reference takes any int (src/calibrate.cpp:175 --nid_bins, nid_cost.hpp:23); its own data path quantises BOTH inputs
  // to 256 levels before the cost sees them -- the camera image is 8-bit (pix = k / 255, visual_camera_calibration.cpp:204),
  // the LiDAR intensities are rank-equalised to floor(256 i / n) / 256 (preprocess.cpp:464-473) -- so more than 256 bins
  // only adds rows and columns that stay empty.  The kernels' layouts (8-bit bin image, one histogram column of <= 256 cells
  // per LDS tile) are built on that bound: refused, not truncated.
  if (d->bins < 2 || d->bins > kMaxWideBins) return fail(NIDREG_ERR_INVALID, "nidreg_create: bins must be in [2, " + std::to_string(kMaxWideBins) + "]");
  if (d->width < 1 || d->height < 1 || (!d->image && !opts.shard)) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image");
  const nidreg_handle* master = opts.shard ? opts.master : nullptr;
  if (opts.shard && !master) return fail(NIDREG_ERR_INVALID, "nidreg_create: shard without a master");
  // bins > 256: run on the occupied bins, compacted (WideBins above); a shard takes its master's compact layout
  WideBins wide_here;
  const WideBins* wide = opts.wide;
  nidreg_desc dd;
  if (!opts.shard && d->bins > NIDREG_MAX_BINS) {
    if (!wide) {
      if (d->image_dtype != NIDREG_IMAGE_F64 && d->image_dtype != NIDREG_IMAGE_U8) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image_dtype");
      if (!cloud && (d->num_points < 0 || (d->num_points > 0 && !d->intensities))) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
      const int rc = resolve_wide_bins(d, cloud, wide_here);
      if (rc) return rc;
      wide = &wide_here;
    }
    dd = *d;
    dd.bins = wide->compact_bins;
    d = &dd;
  } else if (opts.shard && d->bins > NIDREG_MAX_BINS) {
    return fail(NIDREG_ERR_INVALID, "nidreg_create: a shard is created with its master's compact bin count");
  }
  const int64_t n_in = master ? master->gcount[size_t(opts.group_hi)] - master->gcount[size_t(opts.group_lo)] : (cloud ? cloud->n : d->num_points);
  if (n_in < 0 || n_in > int64_t(INT_MAX)) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad num_points");
  if (!master && !cloud && n_in > 0 && (!d->points || !d->intensities)) return fail(NIDREG_ERR_INVALID, "nidreg_create: null points");
  if (cloud && cloud->device != d->device_id) return fail(NIDREG_ERR_INVALID, "nidreg_create_from_cloud: cloud lives on another device");
  if (d->mode != NIDREG_MODE_SPLINE && d->mode != NIDREG_MODE_NEAREST) return fail(NIDREG_ERR_INVALID, "nidreg_create: bad mode");
  // (NIDREG_PREC_FP32 -- float transform / projection, everything else as now -- existed until round 4: +8 % on the headline, for
  // |dNID| <= 2e-5; a mode that cheap to lose was not worth its kernel instantiations and was removed rather than kept half-built)
  if (d->precision != NIDREG_PREC_FP64)
    return fail(NIDREG_ERR_INVALID, d->precision == NIDREG_PREC_FP32 ? "nidreg_create: NIDREG_PREC_FP32 was removed (it bought 8 %); the core computes in double" : "nidreg_create: bad precision");

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(NIDREG_ERR_NO_DEVICE, "nidreg_create: no HIP device (the NID core has no CPU path)");
  if (d->device_id < 0 || d->device_id >= ndev) return fail(NIDREG_ERR_INVALID, "nidreg_create: device_id out of range");
  HIP_TRY(hipSetDevice(d->device_id));

  nidreg_handle* h = new nidreg_handle();
  h->device = d->device_id;
  h->model = d->model_id;
  h->mode = d->mode;
  h->precision = d->precision;
  h->bins = d->bins;
  h->nearest_exact = (d->flags & NIDREG_FLAG_NEAREST_EXACT) != 0;
  if (wide) {
    h->bins_user = wide->user_bins;
    h->inv_img = wide->inv_img;
    h->inv_pts = wide->inv_pts;
  } else if (master && master->bins_user) {
    h->bins_user = master->bins_user;
    h->inv_img = master->inv_img;
    h->inv_pts = master->inv_pts;
  }
  h->W = d->width;
  h->H = d->height;
  h->num_points = n_in;
  h->max_fov = d->max_fov;
  std::memcpy(h->intr, d->intrinsics, sizeof(h->intr));
  std::memcpy(h->dist, d->distortion, sizeof(h->dist));
  const int B = h->bins;
  int64_t N = h->num_points;  // becomes the number of records (after culling on the cloud path)

#define CREATE_TRY(expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      free_handle(h);                                                                        \
      return fail(NIDREG_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));        \
    }                                                                                        \
  } while (0)

  // ---- tiling: a workgroup owns GW histogram columns (= GW * B cells) in LDS, each cell replicated
  // 2^cshift times (lane-private copies, see k_spline_hist).  Default: ~256 cells x 16 copies = 32 KB.
  int GW = d->columns_per_group > 0 ? d->columns_per_group : std::max(1, 256 / B);
  GW = std::min(GW, B);
  int copies = d->lds_copies > 0 ? d->lds_copies : 16;
  int cshift = 0;
  while ((2 << cshift) <= copies && cshift < 4) cshift++;
  while (size_t(GW) * B * 8 > 128 * 1024 && GW > 1) GW /= 2;
  while ((size_t(GW) * B * 8 << cshift) > 64 * 1024 && cshift > 0) cshift--;
  // the headline shape (256 bins, one column per workgroup, default tuning) takes the WIDE histogram
  // kernel: 512 threads, 32 copies, one-instruction tap address (k_spline_hist); an explicit
  // lds_copies keeps the generic kernel (tests compare the two bit for bit)
  h->wide = (d->mode == NIDREG_MODE_SPLINE && B == 256 && GW == 1 && d->lds_copies == 0) ? 1 : 0;
  if (h->wide) cshift = kWideShift;
  h->GW = GW;
  h->cshift = cshift;
  h->NG = (B + GW - 1) / GW;
  h->NEB = (B + kEntropyCols - 1) / kEntropyCols;
  h->lds_hist = d->mode == NIDREG_MODE_NEAREST ? nearest_hist_lds_bytes(B, GW, cshift) : (size_t(GW) * B * 8 << cshift) + size_t(GW) * 8 + 16;
  // gradient pass: a single-column workgroup (GW = 1) keeps ONE copy of its G column (k_spline_grad<.., GW1>)
  h->lds_grad = spline_grad_lds_bytes(B, GW, cshift, false);  // G tile, reduction scratch, phi(q_r), flag (+ staged columns once the table is known to need them)
  h->lds_entropy = size_t(B) * 8 + size_t(GW) * 8 + size_t(kWaves) * 8;

  // ---- fixed point: sum over a bin <= N * 2^frac must stay below 2^63
  const int64_t scaleN = std::max<int64_t>(N, d->scale_points);
  int nbits = 1;
  while ((int64_t(1) << nbits) <= scaleN) nbits++;
  h->frac_bits = d->mode == NIDREG_MODE_NEAREST ? 0 : std::min(40, 62 - nbits);

  // ---- everything below is built ON THE DEVICE from one upload of the caller's arrays (the reference constructs
  // a cost object per pair per outer iteration, visual_camera_calibration.cpp:199-208, so construction time counts):
  // the bin image, then [ViewCulling::cull ->] histogram column + Morton key -> rocPRIM radix sort -> record gather
  // (nid_build.hip).  Temporaries live in the per-device scratch arena.
  const int W = h->W, H = h->H;
  h->pitch = ((W + 8) + 3) & ~3;  // padded width in pixels
  const int PH = H + 3;
  const int nstrips = (PH + 3) / 4 + 1;  // rows are stored in strips of four (nid_device.hpp load_patch)
  const size_t img_bytes = size_t(h->pitch) * 4 * nstrips + 64;
  const bool img_f64 = d->image_dtype == NIDREG_IMAGE_F64;
  if (d->image_dtype != NIDREG_IMAGE_F64 && d->image_dtype != NIDREG_IMAGE_U8) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: bad image_dtype");
  }
  const size_t src_row = size_t(W) * (img_f64 ? 8 : 1);
  if (d->image_row_stride < int64_t(src_row)) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: image_row_stride smaller than a row");
  }
  const int64_t pstride = d->point_stride > 0 ? d->point_stride : 32;
  if (!cloud && n_in > 0 && (pstride < 32 || pstride % 8 != 0)) {
    free_handle(h);
    return fail(NIDREG_ERR_INVALID, "nidreg_create: point_stride must be a multiple of 8 and >= 32 ((x y z 1) doubles)");
  }
  std::vector<int64_t> gcount;
  h->img_bytes = img_bytes;
  if (master) {
    // a shard: the master handle (same tiling, same bins) has built the padded bin image and the bucketed, Morton-ordered
    // records on the owner device; this shard takes the records of its column groups and a copy of the image
    if (master->GW != h->GW || master->NG != h->NG || master->bins != B || master->pitch != h->pitch || master->img_bytes != img_bytes) {
      free_handle(h);
      return fail(NIDREG_ERR_INVALID, "nidreg_create: shard / master layout mismatch");
    }
    const size_t rec_bytes = master->rec64 ? sizeof(Rec64) : sizeof(Rec32);
    const int64_t lo = master->gcount[size_t(opts.group_lo)], hi = master->gcount[size_t(opts.group_hi)];
    CREATE_TRY(hipMalloc(&h->d_img, img_bytes));
    CREATE_TRY(hipMemcpyPeer(h->d_img, h->device, master->d_img, master->device, img_bytes));
    CREATE_TRY(hipMalloc(&h->d_pts, std::max<size_t>(size_t(hi - lo), 1) * rec_bytes));
    if (hi > lo) CREATE_TRY(hipMemcpyPeer(h->d_pts, h->device, static_cast<const char*>(master->d_pts) + size_t(lo) * rec_bytes, master->device, size_t(hi - lo) * rec_bytes));
    h->rec64 = master->rec64;
    gcount.assign(master->gcount.size(), 0);
    for (size_t g = 0; g < master->gcount.size(); g++) gcount[g] = std::min(std::max(master->gcount[g], lo), hi) - lo;
    N = hi - lo;
    h->num_points = N;
    h->is_shard = true;
    h->col_lo = std::min(B, opts.group_lo * h->GW);
    h->col_hi = std::min(B, opts.group_hi * h->GW);
  } else
  {
    ScratchArena& arena = ScratchArena::of(h->device);
    std::lock_guard<ScratchArena> guard(arena);
    const size_t up_img = ((src_row * size_t(H)) + 255) & ~size_t(255);
    const size_t up_pts = cloud ? 0 : ((size_t(std::max<int64_t>(n_in, 1)) * 32 + 255) & ~size_t(255));
    const size_t up_int = cloud ? 0 : ((size_t(std::max<int64_t>(n_in, 1)) * 8 + 255) & ~size_t(255));
    const size_t up_lut = wide ? ((size_t(wide->user_bins) * 2 + 255) & ~size_t(255)) : 0;
    CREATE_TRY(arena.reserve(up_img + up_pts + up_int + 2 * up_lut + build_scratch_bytes(n_in, T_cull != nullptr, W, H) + 4096));
    const int Bsrc = wide ? wide->user_bins : B;  // the bin count the caller's values are binned with
    uint16_t *d_lut_img = nullptr, *d_lut_pts = nullptr;
    if (wide) {
      d_lut_img = static_cast<uint16_t*>(arena.carve(up_lut));
      d_lut_pts = static_cast<uint16_t*>(arena.carve(up_lut));
      CREATE_TRY(hipMemcpy(d_lut_img, wide->lut_img.data(), size_t(wide->user_bins) * 2, hipMemcpyHostToDevice));
      CREATE_TRY(hipMemcpy(d_lut_pts, wide->lut_pts.data(), size_t(wide->user_bins) * 2, hipMemcpyHostToDevice));
    }

    // bin image: bin_image = min(int(pix * bins), bins - 1) (nid_cost.hpp:78-79) for CV_64FC1 input;
    // max(0, min(bins-1, int(u8 / 255.0 * bins))) (cost_calculator_nid.cpp:43-46) for CV_8UC1 input;
    // padded by 1 (left/top) and >= 2 (right/bottom), edge replicated (= the clamp of knots_x / knots_y, :70-73)
    void* d_src = arena.carve(up_img);
    CREATE_TRY(hipMalloc(&h->d_img, img_bytes));
    CREATE_TRY(hipMemcpy2D(d_src, src_row, d->image, size_t(d->image_row_stride), src_row, size_t(H), hipMemcpyHostToDevice));
    CREATE_TRY(build_bin_image_device(d_src, img_f64 ? 1 : 0, (long long)src_row, W, H, Bsrc, d_lut_img, h->pitch, nstrips, h->d_img, nullptr));

    // points: bin_points = max(0, min(bins-1, int(intensity * bins))) (nid_cost.hpp:49, cost_calculator_nid.cpp:47)
    // is pose independent -> records are bucketed by column group, so a workgroup owns GW histogram columns;
    // inside a group they follow a Morton curve of the LiDAR-frame bearing (any order gives the same bits -- the
    // sums are integers --, a spatially coherent one makes a wave's gathers share cache lines for ANY pose).
    // Records are float32 when that is lossless (PLY data is float32 at source); otherwise double.
    CullArgs ca;
    if (T_cull) {
      ca.model = d->model_id;
      std::memcpy(ca.intr, d->intrinsics, sizeof(ca.intr));
      std::memcpy(ca.dist, d->distortion, sizeof(ca.dist));
      std::memcpy(ca.T, T_cull, sizeof(ca.T));
      ca.W = d->width;
      ca.H = d->height;
      ca.min_z = min_z;
      ca.depth = enable_depth ? 1 : 0;
    }
    const double* d_cloud_pts = cloud ? cloud->d_pts : nullptr;
    const double* d_cloud_int = cloud ? cloud->d_int : nullptr;
    if (!cloud) {
      double* up_p = static_cast<double*>(arena.carve(up_pts));
      double* up_i = static_cast<double*>(arena.carve(up_int));
      if (n_in > 0) {
        if (pstride == 32) {
          CREATE_TRY(hipMemcpy(up_p, d->points, size_t(n_in) * 32, hipMemcpyHostToDevice));
        } else {
          CREATE_TRY(hipMemcpy2D(up_p, 32, d->points, size_t(pstride), 32, size_t(n_in), hipMemcpyHostToDevice));
        }
        CREATE_TRY(hipMemcpy(up_i, d->intensities, size_t(n_in) * 8, hipMemcpyHostToDevice));
      }
      d_cloud_pts = up_p;
      d_cloud_int = up_i;
    }
    void* recs = nullptr;
    int rec64 = 0;
    CREATE_TRY(build_records_device(d_cloud_pts, d_cloud_int, n_in, T_cull ? &ca : nullptr, Bsrc, d_lut_pts, GW, h->NG, false, (d->flags & NIDREG_FLAG_INPUT_ORDER) != 0,
                                    arena, &recs, &rec64, gcount, nullptr));
    h->d_pts = recs;
    h->rec64 = rec64;
    N = gcount[size_t(h->NG)];
    h->num_points = N;
    CREATE_TRY(hipStreamSynchronize(nullptr));  // the bin-image kernel, before the arena is handed to the next construction
  }

  // ---- chunk tables (split_groups): by default 4 workgroups per CU for the 256-thread kernels, 2 per CU for the WIDE
  // histogram kernel (64 KB LDS each), which therefore has its own table
  {
    std::vector<uint32_t> gend(static_cast<size_t>(h->NG));
    for (int g = 0; g < h->NG; g++) gend[size_t(g)] = uint32_t(gcount[size_t(g) + 1]);
    CREATE_TRY(hipMalloc(&h->d_gend, gend.size() * sizeof(uint32_t)));
    CREATE_TRY(hipMemcpy(h->d_gend, gend.data(), gend.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    int num_cus = 256;
    if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || num_cus <= 0) num_cus = 256;
    // (Tried and dropped: emitting a group's parts part-major -- part j of every group in dispatch slot j of the CUs -- and
    // sizing them by slot weights.  The first-dispatched workgroup of a CU does run ~8 % faster than the second, systematically
    // (profiles/archive/r03d_workgroup_spread.txt), but the workgroups of one CU share its issue capacity: what ends a pass is the
    // slowest CU, not the slowest workgroup, and no weighting moved the kernel times (profiles/archive/r03e_slot_weights_no_gain.txt;
    // the part-major order itself cost 3 us in the gradient pass).)
    const int max_segs = max_segments(d->mode, GW);
    auto build_chunks = [&](int target, bool wide_hist, std::vector<Chunk>& chunks) { return split_groups(gcount.data(), h->NG, target, segment_overhead(wide_hist), max_segs, -1, chunks); };
    // workgroups per CU that are really co-resident for THIS kernel instantiation: 4 for the pinhole family, 3 for the
    // fisheye / equirectangular gradient kernels (154-161 VGPRs) -- 1024 chunks there meant 1.33 rounds
    int per_cu_grad = 4, per_cu_hist = h->wide ? 2 : 4;
    if (d->mode == NIDREG_MODE_SPLINE) {
      PassArgs oa;
      fill_pass_args(h, oa);
      const int og = occupancy_spline_grad<double>(oa);
      const int oh = occupancy_spline_hist<double>(oa);
      if (og > 0) per_cu_grad = std::min(og, 8);
      if (oh > 0) per_cu_hist = std::min(oh, 8);
    } else {  // NEAREST: the fast-tier kernels of the wide-angle models hold three (equirectangular) or four waves per SIMD
      PassArgs oa;
      fill_pass_args(h, oa);
      const int on = occupancy_nearest_hist<double>(oa);
      if (on > 0) per_cu_grad = per_cu_hist = std::min(on, 4);
    }
    h->gcount = gcount;
    h->num_cus = num_cus;
    h->per_cu_grad = h->wide ? per_cu_grad : std::min(per_cu_grad, per_cu_hist);
    h->per_cu_hist = per_cu_hist;
    std::vector<Chunk> chunks;
    auto own_target = [&](int per_cu) {
      if (d->target_blocks > 0) return int(d->target_blocks);
      const int64_t full = int64_t(per_cu) * num_cus;
      return int(snap_to_groups(round_chunks(per_cu, num_cus, N), gcount.data(), h->NG, full));
    };
    h->nslots = int(build_chunks(own_target(h->per_cu_grad), false, chunks));
    h->nchunks = int(chunks.size());
    for (const Chunk& c : chunks) h->longest_chunk = std::max<int64_t>(h->longest_chunk, c.count);
    h->seg = h->nslots > h->nchunks ? 1 : 0;
    if (h->seg && GW == 1) h->lds_grad = spline_grad_lds_bytes(B, GW, cshift, true);
    h->chunks_cap = std::max<size_t>(chunks.size(), 1);
    CREATE_TRY(hipMalloc(&h->d_chunks, h->chunks_cap * sizeof(Chunk)));
    if (!chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks, chunks.data(), chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    if (h->wide) {
      std::vector<Chunk> wide_chunks;
      const int64_t wide_slots = build_chunks(own_target(per_cu_hist), true, wide_chunks);
      h->nchunks_hist = int(wide_chunks.size());
      h->seg_hist = wide_slots > int64_t(wide_chunks.size()) ? 1 : 0;
      h->chunks_hist_cap = std::max<size_t>(wide_chunks.size(), 1);
      CREATE_TRY(hipMalloc(&h->d_chunks_hist, h->chunks_hist_cap * sizeof(Chunk)));
      if (!wide_chunks.empty()) CREATE_TRY(hipMemcpy(h->d_chunks_hist, wide_chunks.data(), wide_chunks.size() * sizeof(Chunk), hipMemcpyHostToDevice));
    }
  }

  // ---- NEAREST on an equirectangular camera: the pixel-boundary tables of the fast decision tier (nid_kernels.hpp NearestFast).
  // Boundary u = k sits at longitude theta_k = 2 pi (k / W - 1/2), boundary v = j at latitude pi (j / H - 1/2), W and H the
  // INTRINSICS (equirectangular.hpp:14-28 projects with them; the image size only enters the in-image test).
  if (d->mode == NIDREG_MODE_NEAREST && h->model == NIDREG_MODEL_EQUIRECTANGULAR && h->intr[0] >= 8.0 && h->intr[1] >= 8.0 && h->intr[0] <= 65536.0 && h->intr[1] <= 65536.0) {
    const double pi = 3.14159265358979323846;
    h->eq_kmax = int(std::ceil(h->intr[0]));
    h->eq_jmax = int(std::ceil(h->intr[1]));
    std::vector<double> tab(2 * size_t(h->eq_kmax + 1) + size_t(h->eq_jmax + 1));
    for (int k = 0; k <= h->eq_kmax; k++) {
      const double th = 2.0 * pi * (double(k) / h->intr[0] - 0.5);
      tab[2 * size_t(k)] = std::cos(th);
      tab[2 * size_t(k) + 1] = std::sin(th);
    }
    for (int j = 0; j <= h->eq_jmax; j++) {
      const double sj = std::sin(pi * (double(j) / h->intr[1] - 0.5));
      tab[2 * size_t(h->eq_kmax + 1) + size_t(j)] = sj * std::fabs(sj);
    }
    CREATE_TRY(hipMalloc(&h->d_eq_tab, tab.size() * sizeof(double)));
    CREATE_TRY(hipMemcpy(h->d_eq_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
  }

#ifdef NID_EXP_HANDOFF
  {  // EXPERIMENT: the (u, v) hand-off buffer of the process' one handle (leaked at destruction: an experiment build)
    void* uvb = nullptr;
    CREATE_TRY(hipMalloc(&uvb, size_t(std::max<int64_t>(N, 1)) * 16 + 64));
    CREATE_TRY(hipMemset(uvb, 0, size_t(std::max<int64_t>(N, 1)) * 16 + 64));
    CREATE_TRY(set_handoff_buffer(uvb));
  }
#endif
  // ---- per-evaluation scratch
  h->hist_words = nidreg_hist_words(B);
  if (d->ext_stream || (d->flags & NIDREG_FLAG_EXT_STREAM)) {
    h->stream = static_cast<hipStream_t>(d->ext_stream);
  } else {
    CREATE_TRY(pool_stream(h->device, &h->stream));
    h->own_stream = true;
  }
  if (d->ext_hist) {
    h->d_hist = static_cast<u64*>(d->ext_hist);
  } else {
    if (opts.shard) {
      // a shard's two buffers are replicas of the WHOLE pair's histogram: the owners of the other columns store into them from
      // their own devices (nid_kernels.hpp k_entropy_repl) -- fine-grained (coherent) device memory, mapped into every peer
      CREATE_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&h->d_hist_buf[0]), size_t(h->hist_words) * sizeof(u64), hipDeviceMallocFinegrained));
      CREATE_TRY(hipExtMallocWithFlags(reinterpret_cast<void**>(&h->d_hist_buf[1]), size_t(h->hist_words) * sizeof(u64), hipDeviceMallocFinegrained));
    } else {
      CREATE_TRY(hipMalloc(&h->d_hist_buf[0], size_t(h->hist_words) * sizeof(u64)));
      CREATE_TRY(hipMalloc(&h->d_hist_buf[1], size_t(h->hist_words) * sizeof(u64)));
    }
    CREATE_TRY(hipMemset(h->d_hist_buf[1], 0, size_t(h->hist_words) * sizeof(u64)));
    h->d_hist = h->d_hist_buf[0];
    h->hist_cur = 0;
    h->hist_zeroed[1] = true;  // [0] is zeroed below and read by nidreg_get_hist before the first evaluation
    h->own_hist = true;
  }
  if (d->ext_out) {
    h->d_out = static_cast<double*>(d->ext_out);
  } else {
    CREATE_TRY(hipMalloc(&h->d_out, NIDREG_OUT_DOUBLES * sizeof(double)));
    h->own_out = true;
  }
  CREATE_TRY(hipMemset(h->d_out, 0, NIDREG_OUT_DOUBLES * sizeof(double)));
  CREATE_TRY(hipMemset(h->d_hist, 0, size_t(h->hist_words) * sizeof(u64)));
  {
    // one allocation, carved (256-byte aligned) and zeroed: nidreg_get_hist before the first evaluation then
    // reads zeros, not uninitialised memory
    size_t off = 0;
    auto carve = [&](size_t bytes) {
      const size_t at = off;
      off = (off + bytes + 255) & ~size_t(255);
      return at;
    };
    const size_t o_part_hj = carve(size_t(h->NEB) * sizeof(long long));   // (split-phase ABI scratch: per column block)
    const size_t o_row_part = carve(size_t(h->NEB) * B * sizeof(u64));    // [NEB][B]
    const size_t o_phi_q = carve(size_t(B) * sizeof(double));
    const size_t o_hist_image = carve(size_t(B) * sizeof(double));
    const size_t o_hist_points = carve(size_t(B) * sizeof(double));
    const size_t o_scal = carve(sizeof(EntropyScalars));
    h->partials_cap = partial_slots(h);
    const size_t o_partials = carve(size_t(h->partials_cap) * 12 * sizeof(double));
    const size_t o_counters = carve(8 * sizeof(unsigned int));
    CREATE_TRY(hipMalloc(&h->d_scratch, off));
    CREATE_TRY(hipMemset(h->d_scratch, 0, off));
    char* base = static_cast<char*>(h->d_scratch);
    h->d_part_hj = reinterpret_cast<long long*>(base + o_part_hj);
    h->d_row_part = reinterpret_cast<u64*>(base + o_row_part);
    h->d_phi_q = reinterpret_cast<double*>(base + o_phi_q);
    h->d_hist_image = reinterpret_cast<double*>(base + o_hist_image);
    h->d_hist_points = reinterpret_cast<double*>(base + o_hist_points);
    h->d_scal = reinterpret_cast<EntropyScalars*>(base + o_scal);
    h->d_partials = reinterpret_cast<double*>(base + o_partials);
    h->d_counters = reinterpret_cast<unsigned int*>(base + o_counters);
  }
  {
    void* blk = nullptr;
    CREATE_TRY(pool_host_block(h->device, false, NIDREG_OUT_DOUBLES * sizeof(double), &blk));
    h->h_out = static_cast<double*>(blk);
  }
  std::memset(h->h_out, 0, NIDREG_OUT_DOUBLES * sizeof(double));
  if (!d->ext_out) {
    // results are written straight into host-mapped memory by the finalising workgroups: no D2H copy
    void* dp = nullptr;
    CREATE_TRY(hipHostGetDevicePointer(&dp, h->h_out, 0));
    h->d_out_host = static_cast<double*>(dp);
  }
  for (int i = 0; i < 6; i++) CREATE_TRY(hipEventCreate(&h->ev[i]));
  // The clears above (result block, histogram buffers, scratch incl. the ticket counters) are hipMemset calls on the null
  // stream, which return before they have run (2.9 us per call in the API trace, profiles/archive/r04m_hip_api_stats.csv), and the
  // handle's own stream is non-blocking: the first evaluation must not be able to overtake them.
  CREATE_TRY(hipStreamSynchronize(nullptr));
#undef CREATE_TRY
  *out = h;
  return NIDREG_OK;
}
}  // namespace nidreg_detail
