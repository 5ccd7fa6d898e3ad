"""Multi-GPU evaluation of the NID cost, one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Two ways the path shards:

* independent LiDAR-camera pairs -> pair k on rank k mod P, no data-path collective; the optimiser's
  ``sum_i NID_i`` (visual_camera_calibration.cpp:166-170) is 8 doubles per evaluation
  (``PairParallelNIDCost``);
* one pair, points sharded -> NID is nonlinear in the histogram, so the partial fixed-point
  histograms are all-reduced (int64 sum: exact, order independent) BEFORE the entropy tail, and the
  7-double gradient partials after the gradient pass (``ShardedNIDCost``).

The collective protocol is separated from the compute backend (``ShardedEvaluator``) so the
world_size-2 gloo tests can drive it on CPU tensors.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_slice(num_points, rank, world):
    """Contiguous, disjoint, exhaustive point slices."""
    lo = num_points * rank // world
    hi = num_points * (rank + 1) // world
    return lo, hi


class ShardedEvaluator:
    """Split-phase protocol over any backend exposing
    ``shard_hist(x)``, ``hist_tensor`` (int64), ``shard_entropy()``, ``shard_grad()``,
    ``grad_tensor`` (float64[7]) and ``shard_finish(want_grad) -> (ok, cost, grad)``."""

    def __init__(self, backend, group=None):
        self.backend = backend
        self.group = group

    def __call__(self, x, want_grad=True):
        b = self.backend
        b.shard_hist(x)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(b.hist_tensor, op=dist.ReduceOp.SUM, group=self.group)
        b.shard_entropy()
        if want_grad:
            b.shard_grad()
            if dist.is_initialized() and dist.get_world_size(self.group) > 1:
                dist.all_reduce(b.grad_tensor, op=dist.ReduceOp.SUM, group=self.group)
        return b.shard_finish(want_grad)


class _GpuShardBackend:
    """One rank's shard on its GPU: the handle accumulates into torch-owned device buffers on
    torch's current stream, so the RCCL all-reduce is ordered with the kernels without host syncs."""

    def __init__(self, proj, normalized_image, points, intensities, bins, total_points, device, precision, **tuning):
        from . import _lib, nid

        lib = _lib.load()
        dev = torch.device("cuda", device)
        self.words = int(lib.nidreg_hist_words(int(bins)))
        self._hist = torch.zeros(self.words, dtype=torch.int64, device=dev)
        self._out = torch.zeros(_lib.NIDREG_OUT_DOUBLES, dtype=torch.float64, device=dev)
        self.hist_tensor = self._hist
        self.grad_tensor = self._out[1:8]
        stream = torch.cuda.current_stream(dev).cuda_stream  # 0 = the default stream: still "external"
        flags = int(tuning.pop("flags", 0)) | _lib.FLAG_EXT_STREAM
        self.cost = nid.NIDCost(proj, normalized_image, points, intensities, bins, device=device, precision=precision, scale_points=int(total_points), flags=flags,
                                ext_stream=stream or None, ext_hist=self._hist.data_ptr(), ext_out=self._out.data_ptr(), **tuning)

    def shard_hist(self, x):
        self.cost.shard_hist(x)

    def shard_entropy(self):
        self.cost.shard_entropy()

    def shard_grad(self):
        self.cost.shard_grad()

    def shard_finish(self, want_grad):
        return self.cost.shard_finish(want_grad)


class ShardedNIDCost:
    """``NIDCost`` of ONE pair whose points are split over the ranks of ``group``.  Every rank
    passes ITS slice of the cloud (see ``shard_slice``) and the pair's full point count."""

    def __init__(self, proj, normalized_image, points, intensities, bins=16, device=0, precision="fp64", total_points=None, group=None, **tuning):
        n_local = int(np.asarray(points).shape[0])
        if total_points is None:
            t = torch.tensor([n_local], dtype=torch.int64, device=torch.device("cuda", device))
            if dist.is_initialized():
                dist.all_reduce(t, group=group)
            total_points = int(t.item())
        self.backend = _GpuShardBackend(proj, normalized_image, points, intensities, bins, total_points, device, precision, **tuning)
        self.inner = self.backend.cost
        self.eval = ShardedEvaluator(self.backend, group)

    def __call__(self, x, want_grad=True):
        return self.eval(x, want_grad)

    def close(self):
        self.inner.close()


class InLibShardedNIDCost:
    """The same split as ``ShardedNIDCost`` with the collectives INSIDE libnidreg.so (``nidreg_shard_comm_init``: the library
    opens librccl.so itself and runs histogram -> ncclAllReduce(int64) -> entropy -> gradient -> ncclAllReduce(f64 x 7) on the
    handle's stream) -- what a C++ caller of the drop-in gets.  torch.distributed is used once, to hand rank 0's
    ncclUniqueId to the other ranks."""

    def __init__(self, proj, normalized_image, points, intensities, bins=16, device=0, precision="fp64", total_points=None, group=None, **tuning):
        from . import nid

        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = torch.device("cuda", device)
        n_local = int(np.asarray(points).shape[0])
        if total_points is None:
            t = torch.tensor([n_local], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(t, group=group)
            total_points = int(t.item())
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid = torch.frombuffer(bytearray(nid.NIDCost.rccl_unique_id()), dtype=torch.uint8).to(dev)
        if world > 1:
            dist.broadcast(uid, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        self.inner = nid.NIDCost(proj, normalized_image, points, intensities, bins, device=device, precision=precision, scale_points=int(total_points), **tuning)
        self.inner.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))

    def __call__(self, x, want_grad=True):
        return self.inner(x, want_grad)

    def close(self):
        self.inner.close()


class PairParallelNIDCost:
    """``MultiNIDCost`` with the pairs spread over ranks: each rank evaluates its own pairs (one
    ``nidreg_eval_multi`` over its local handles), then ONE all-reduce of 9 doubles
    [ok_count, cost, grad7].  The trust gate is evaluated identically on every rank."""

    def __init__(self, local_multi, num_local_pairs, group=None, device=None):
        self.multi = local_multi
        self.n_local = num_local_pairs
        self.group = group
        self.device = device

    def __call__(self, x, want_grad=True):
        ok, c, g = self.multi(x, want_grad) if self.n_local else (True, 0.0, np.zeros(7))
        buf = torch.zeros(9, dtype=torch.float64, device=self.device)
        buf[0] = 0.0 if ok else 1.0
        buf[1] = c if ok else 0.0
        if want_grad and ok:
            buf[2:9] = torch.as_tensor(g, dtype=torch.float64)
        if dist.is_initialized() and dist.get_world_size(self.group) > 1:
            dist.all_reduce(buf, group=self.group)
        h = buf.cpu().numpy()
        return h[0] == 0.0, float(h[1]), (h[2:9].copy() if want_grad else None)
