#include "nidreg_internal.hpp"

namespace nidreg_detail {
RcclApi* rccl_api() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {std::getenv("NIDREG_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* nm : names) {
      if (!nm || !*nm) continue;
      api.lib = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (!api.lib) {
      const char* e = dlerror();
      api.error = std::string("cannot open librccl.so (set NIDREG_RCCL_LIB): ") + (e ? e : "?");
      return;
    }
    auto sym = [&](const char* nm) {
      void* p = dlsym(api.lib, nm);
      if (!p && api.error.empty()) api.error = std::string("librccl.so has no symbol ") + nm;
      return p;
    };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
  });
  return &api;
}
int rccl_fail(const char* what, ncclResult_t r) {
  RcclApi* api = rccl_api();
  return fail(NIDREG_ERR_HIP, std::string(what) + ": " + (api->GetErrorString ? api->GetErrorString(r) : "RCCL error"));
}

void rccl_release(nidreg_handle* h) {
  if (h->rccl_comm && h->rccl_owned) {
    RcclApi* api = rccl_api();
    if (api->CommDestroy) (void)api->CommDestroy(static_cast<ncclComm_t>(h->rccl_comm));
  }
  h->rccl_comm = nullptr;
  h->rccl_owned = false;
}

// Every rank of the communicator must run the SAME fixed-point histogram: the all-reduce adds the ranks' int64 words as they are.
// One collective at attach time -- max over the ranks of {v, -v} for the handle's table parameters -- and a refusal on every rank
// alike when they differ (a rank created without desc.scale_points = the pair's total, or with other bins / mode, would
// otherwise produce a silently wrong cost, identical on all ranks).
int rccl_check_agreement(nidreg_handle* h, ncclComm_t comm, const char* who) {
  RcclApi* api = rccl_api();
  const int kN = 5;
  long long v[2 * kN] = {h->frac_bits, h->bins, (long long)h->hist_words, h->mode, h->bins_user};
  for (int k = 0; k < kN; k++) v[kN + k] = -v[k];
  long long* d = nullptr;
  HIP_TRY(hipMalloc(&d, sizeof(v)));
  hipError_t e = hipMemcpyAsync(d, v, sizeof(v), hipMemcpyHostToDevice, h->stream);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) r = api->AllReduce(d, d, size_t(2 * kN), ncclInt64, ncclMax, comm, h->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(v, d, sizeof(v), hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(h->stream);
  (void)hipFree(d);
  if (r != ncclSuccess) return rccl_fail(who, r);
  if (e != hipSuccess) return fail(NIDREG_ERR_HIP, std::string(who) + ": " + hipGetErrorString(e));
  static const char* names[kN] = {"frac_bits (desc.scale_points must be the pair's TOTAL point count on every rank)", "bins", "hist_words", "mode", "bins (caller's count)"};
  for (int k = 0; k < kN; k++)
    if (v[k] != -v[kN + k])
      return fail(NIDREG_ERR_INVALID, std::string(who) + ": the ranks of the communicator disagree on " + names[k] + ": max " + std::to_string(v[k]) + ", min " + std::to_string(-v[kN + k]) +
                                        "; the handle stays detached");
  return NIDREG_OK;
}

int rccl_attachable(const nidreg_handle* h, const char* who) {
  if (!h) return fail(NIDREG_ERR_INVALID, std::string(who) + ": null handle");
  if (h->set || h->is_shard) return fail(NIDREG_ERR_INVALID, std::string(who) + ": the handle is already sharded inside the library (desc.device_ids / NIDREG_DEVICES)");
  if (!h->own_hist || !h->own_out) return fail(NIDREG_ERR_INVALID, std::string(who) + ": the handle must own its histogram and result buffers (no ext_hist / ext_out)");
  if (h->async_outstanding != 0) return fail(NIDREG_ERR_INVALID, std::string(who) + ": collect the handle's outstanding tickets first");
  return NIDREG_OK;
}

// one evaluation of a handle with a communicator; mode SPLINE: pose = se3[7], NEAREST: row-major 4x4.  Collective: every rank
// of the communicator calls it with the same pose.
int rccl_eval(nidreg_handle* h, int mode, const double* pose, double* cost, double* grad7) {
  if (h->mode != mode) return fail(NIDREG_ERR_INVALID, mode == NIDREG_MODE_SPLINE ? "nidreg_eval: handle was created in NEAREST mode" : "nidreg_eval_iso: handle was created in SPLINE mode");
  RcclApi* api = rccl_api();
  ncclComm_t comm = static_cast<ncclComm_t>(h->rccl_comm);
  HIP_TRY(hipSetDevice(h->device));
  InflightGuard guard(h->device);
  bump_seq(h);
  const bool grad = mode == NIDREG_MODE_SPLINE && grad7 != nullptr;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[0], h->stream));
  if (h->num_points == 0) {  // a rank without points launches no histogram kernel: its (cleared) buffer still takes part in the sum
    HIP_TRY(begin_histogram(h));
    if (h->timing == 1) HIP_TRY(hipEventRecord(h->ev[1], h->stream));  // (the marker launch_hist_* would have recorded: nidreg_get_timing reads it)
    if (mode == NIDREG_MODE_SPLINE) {
      for (int k = 0; k < 4; k++) h->last_q[k] = pose[k];
      pose_from_se3(pose, h->last_R, h->last_t);
    }
  } else {
    const int rc = mode == NIDREG_MODE_SPLINE ? launch_hist_spline(h, pose, guard.alone) : launch_hist_nearest(h, pose);
    if (rc) return rc;
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[2], h->stream));
  RCCL_TRY(api->AllReduce(h->d_hist, h->d_hist, size_t(h->hist_words), ncclInt64, ncclSum, comm, h->stream));
  int rc = launch_entropy(h, 0.0, true);
  if (rc) return rc;
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[3], h->stream));
  if (grad) {
    rc = launch_grad(h, guard.alone, 0);
    if (rc) return rc;
    RCCL_TRY(api->AllReduce(h->d_out + 1, h->d_out + 1, 7, ncclFloat64, ncclSum, comm, h->stream));
  } else if (h->timing) {
    HIP_TRY(hipEventRecord(h->ev[4], h->stream));
  }
  if (h->timing) HIP_TRY(hipEventRecord(h->ev[5], h->stream));
  h->ev_grad = grad;
  HIP_TRY(hipMemcpyAsync(h->h_out, h->d_out, NIDREG_OUT_DOUBLES * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (cost) *cost = h->h_out[0];
  if (grad)
    for (int k = 0; k < 7; k++) grad7[k] = h->h_out[1 + k];
  return h->h_out[8] != 0.0 ? NIDREG_FALSE : NIDREG_OK;
}
}  // namespace nidreg_detail
