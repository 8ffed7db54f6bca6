"""Multi-GPU side of the hot path.  One process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm) -- PyTorch is plumbing here: device tensors for the all-reduce buffer, nothing else.

* Frames and local-BA windows are independent units: they shard with NO collective
  (`shard_range` gives each rank its contiguous share; results are gathered by the host application).
* Global bundle adjustment has exactly one exchange step per LM trial: landmarks (with all their edges) are
  partitioned over ranks, poses are replicated, and the additive reduced camera system
  [S | b_schur | b_p | diag Hpp] is summed with ONE all-reduce; chi2 / scale / max-diagonal scalars ride in a
  second, tiny all-reduce.  `sharded_bundle_adjustment` is the LM driver (g2o control flow,
  reference Thirdparty/g2o/g2o/core/optimization_algorithm_levenberg.cpp:61-169) written against a small
  shard interface so that the same code runs on the HIP shard (capi.LbaShard) and, in the CPU gloo tests,
  on the oracle's shard.
"""
import math

import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block partition of n_items units (frames, windows, landmarks) over `world` ranks."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def partition_landmarks(w, rank, world):
    """Local problem of one rank: all poses, landmarks [lo, hi) and every edge of those landmarks.
    Returns (local problem dict, landmark index range, global indices of the local edges)."""
    lo, hi = shard_range(len(w["points"]), rank, world)
    sel = np.nonzero((w["edge_point"] >= lo) & (w["edge_point"] < hi))[0]
    loc = dict(w)
    loc["points"] = np.ascontiguousarray(w["points"][lo:hi])
    loc["edge_point"] = (w["edge_point"][sel] - lo).astype(np.int32)
    loc["edge_pose"] = np.ascontiguousarray(w["edge_pose"][sel])
    loc["edge_obs"] = np.ascontiguousarray(w["edge_obs"][sel])
    loc["edge_inv_sigma2"] = np.ascontiguousarray(w["edge_inv_sigma2"][sel])
    loc["edge_stereo"] = np.ascontiguousarray(w["edge_stereo"][sel])
    return loc, (lo, hi), sel


class _NoDist:
    """world_size 1: the collectives are identities."""
    world = 1

    def sum_(self, t, shard=None):
        return t

    def sum_scalars(self, vals):
        return list(vals)

    def max_scalars(self, vals):
        return list(vals)


class TorchDist:
    """torch.distributed collectives.  `device` is where scalar packs live (cuda for nccl/RCCL, cpu for gloo)."""

    def __init__(self, dist, device):
        import torch
        self.torch = torch
        self.dist = dist
        self.device = device
        self.world = dist.get_world_size()
        self.timing = True          # HIP events around the reduce-buffer all-reduces (device tensors only); off: no events at all
        self._events = []           # (start, end) event pairs not yet folded into _ms
        self._ms = 0.0

    def sum_(self, t, shard=None):
        """All-reduce of the reduce buffer.  RCCL enqueues it on torch's current stream while the shard's kernels run on the
        shard's own stream: with a shard that offers stream fences (capi.LbaShard) the two streams are ordered by events and the
        host never waits here -- the factorisation is enqueued behind the collective while it is still running; otherwise the
        host synchronises the collective's stream."""
        cuda = getattr(t, "is_cuda", False)
        fences = cuda and shard is not None and hasattr(shard, "fence_out")
        timed = cuda and self.timing
        if cuda:
            st = self.torch.cuda.current_stream(t.device)
            if fences:
                shard.fence_out(st.cuda_stream)          # the collective waits for the shard's partial sums
            if timed:
                e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
                e0.record(st)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        if cuda:
            if timed:
                e1.record(st)
                self._events.append((e0, e1))
                self._fold(False)
            if fences:
                shard.fence_in(st.cuda_stream)           # the shard's next kernels wait for the summed buffer
            else:
                st.synchronize()
        return t

    def _fold(self, wait):
        """completed event pairs go into the running total and are dropped (a long-lived comm must not grow)"""
        keep = []
        for a, b in self._events:
            if wait:
                b.synchronize()
            if b.query():
                self._ms += a.elapsed_time(b)
            else:
                keep.append((a, b))
        self._events = keep

    @property
    def allreduce_seconds(self):
        """device time spent in the reduce-buffer all-reduces so far (HIP events on the collective's stream); None on CPU backends
        or when nothing was timed.  Waits for the last collective."""
        if not self._events and self._ms == 0.0:
            return None
        self._fold(True)
        return 1e-3 * self._ms

    def sum_scalars(self, vals):
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def max_scalars(self, vals):
        t = self.torch.tensor(list(vals), dtype=self.torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return t.cpu().tolist()


def sharded_bundle_adjustment(shard, reduce_tensor, comm=None, max_iters=10, lambda_init=0.0, stop_flag=None):
    """LM loop over a landmark-sharded BA.  `shard` offers linearize() -> (chi2, maxdiag_p, maxdiag_l),
    reduce(lambda), finish(lambda) -> (solved, chi2_new, scale_poses, scale_landmarks), accept(bool).
    `reduce_tensor` is a torch tensor (or None when comm is None) aliasing the shard's reduce buffer.
    Every rank takes identical decisions because every decision input is all-reduced.  Returns stats dict."""
    comm = comm or _NoDist()
    lam, ni, n_bad = -1.0, 2.0, 0
    stats = dict(iterations=0, trials=0, stop_reason=0, chi2_trace=[])

    def terminate():
        local = 1.0 if (stop_flag is not None and stop_flag[0]) else 0.0
        return comm.max_scalars([local])[0] > 0      # all ranks must agree on stopping

    for it in range(max_iters):
        if terminate():
            stats["stop_reason"] = 3
            break
        hint = getattr(shard, "hint_lambda", None)
        if hint is not None and (it > 0 or lambda_init > 0):    # the lambda of the coming first trial is already known
            hint(lam if it > 0 else lambda_init)
        chi_l, mdp_l, mdl_l = shard.linearize()
        current_chi = comm.sum_scalars([chi_l])[0]
        ini_chi = current_chi
        if it == 0:
            stats["chi2_initial"] = current_chi
            if lambda_init > 0:
                lam = lambda_init
            else:
                # pose diagonals are partial sums -> they are summed inside the reduce buffer; use the landmark max
                # plus the summed pose diagonal (section 4 of the buffer) after a lambda-free reduce
                if reduce_tensor is not None:       # world size 1: linearize() already returned the true maxima
                    shard.reduce(0.0)
                    comm.sum_(reduce_tensor, getattr(shard, "s", None))
                mdp = shard.max_pose_diag()
                mdl = comm.max_scalars([mdl_l])[0]
                lam = 1e-5 * max(mdp, mdl)
            ni, n_bad = 2.0, 0
        rho, qmax, stopped = 0.0, 0, False
        while True:
            shard.reduce(lam)
            if reduce_tensor is not None:
                comm.sum_(reduce_tensor, getattr(shard, "s", None))
            solved, chi_new_l, sp, sl_l = shard.finish(lam)
            chi_new, sl, ok_all = comm.sum_scalars([chi_new_l, sl_l, float(solved)])
            temp_chi = chi_new if ok_all >= comm.world - 0.5 else float("inf")
            rho = (current_chi - temp_chi) / (sp + sl + 1e-3)
            if rho > 0 and math.isfinite(temp_chi):
                alpha = min(1.0 - (2 * rho - 1) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                ni = 2.0
                current_chi = temp_chi
                shard.accept(True)
            else:
                lam *= ni
                ni *= 2
                shard.accept(False)
            qmax += 1
            stats["trials"] += 1
            stopped = terminate()
            if not (rho < 0 and qmax < 10 and not stopped):
                break
        stats["iterations"] += 1
        stats["chi2_trace"].append(current_chi)
        stats["chi2_final"] = current_chi
        if qmax == 10 or rho == 0:
            stats["stop_reason"] = 1
            break
        n_bad = n_bad + 1 if (ini_chi - current_chi) * 1e3 < ini_chi else 0
        if n_bad >= 3:
            stats["stop_reason"] = 2
            break
    stats["lambda_"] = lam
    return stats


class LocalHipShard:
    """Adapter for world_size 1 without torch: the pose diagonal maximum comes straight from linearize()."""

    def __init__(self, lba_shard):
        self.s = lba_shard
        self._mdp = 0.0
        lba_shard.set_local(True)       # no collective between reduce() and finish()

    def hint_lambda(self, lam):
        self.s.hint_lambda(lam)

    def linearize(self):
        chi, mdp, mdl = self.s.linearize()
        self._mdp = mdp
        return chi, mdp, mdl

    def reduce(self, lam):
        self.s.reduce(lam)

    def finish(self, lam):
        return self.s.finish(lam)

    def accept(self, ok):
        self.s.accept(ok)

    def max_pose_diag(self):
        return self._mdp


class HipShard:
    """Adapter: capi.LbaShard (HIP kernels) + a torch.float64 CUDA tensor that owns the reduce buffer, so that
    torch.distributed (RCCL) all-reduces it in place over xGMI with no staging copy."""

    def __init__(self, lba_shard, torch, device):
        self.s = lba_shard
        self.n_red = lba_shard.reduce_len()
        self.tensor = torch.zeros(self.n_red, dtype=torch.float64, device=device)
        lba_shard.set_reduce_buffer(self.tensor.data_ptr())
        if hasattr(lba_shard, "set_async_reduce"):
            lba_shard.set_async_reduce(True)        # TorchDist.sum_ orders the two streams with events: no host wait per trial
        # n*n + 3n = n_red  ->  n
        self.n = int(round((-3 + math.sqrt(9 + 4 * self.n_red)) / 2))

    def hint_lambda(self, lam):
        self.s.hint_lambda(lam)

    def linearize(self):
        return self.s.linearize()

    def reduce(self, lam):
        self.s.reduce(lam)

    def finish(self, lam):
        return self.s.finish(lam)

    def accept(self, ok):
        self.s.accept(ok)

    def max_pose_diag(self):
        if self.n == 0:
            return 0.0
        return float(self.tensor[self.n * self.n + 2 * self.n:].abs().max().item())


def host_staged_allreduce(dist, torch):
    """An all-reduce callback for capi.LbaShard.optimize() / lba_shard_optimize on a backend without device collectives (gloo
    rehearsals with several ranks on ONE GPU): the buffer goes through host memory.  With RCCL the callback is a single
    ncclAllReduce on the given stream instead."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    calls = {"n": 0, "doubles": 0}

    def cb(dev_ptr, count, op, stream):
        calls["n"] += 1; calls["doubles"] += count
        if hip.hipStreamSynchronize(stream):
            return 1
        host = torch.empty(count, dtype=torch.float64)
        if hip.hipMemcpy(host.data_ptr(), dev_ptr, 8 * count, 2):
            return 1
        dist.all_reduce(host, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        return 1 if hip.hipMemcpy(dev_ptr, host.data_ptr(), 8 * count, 1) else 0
    cb.calls = calls
    return cb


def wrap_device_doubles(torch, dev_ptr, count, device):
    """a float64 tensor view of `count` doubles at a raw device pointer (no copy)"""
    class _Arr:         # __cuda_array_interface__ view
        __cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(dev_ptr), False), "version": 2}
    return torch.as_tensor(_Arr(), device=device)


def rccl_allreduce(dist, torch, device):
    """The same callback on RCCL (backend "nccl"): the buffer is wrapped as a tensor and all-reduced on torch's current stream,
    ordered against the shard's stream by events (the C++ form needs neither: ncclAllReduce takes the shard's stream itself)."""
    def cb(dev_ptr, count, op, stream):
        t = wrap_device_doubles(torch, dev_ptr, count, device)
        ext = torch.cuda.ExternalStream(int(stream) if stream else 0, device=device)
        cur = torch.cuda.current_stream(device)
        cur.wait_stream(ext)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM)
        ext.wait_stream(cur)
        return 0
    return cb
