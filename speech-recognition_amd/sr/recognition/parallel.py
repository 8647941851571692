# -*- coding: utf-8 -*-
"""Utterance sharding across the GPUs of one node and the single exchange step of
the training path: the all-reduce of the EM sufficient statistics.

One process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for tests).  Decode needs no collective: each rank decodes its
shard and the results are concatenated on the host.  Training all-reduces ONE packed
fp64 buffer per EM iteration: for every state the k x (1+2D)
block of `gh_em_accumulate` plus the frame count -- 253 KB for the 50-state / 8-mix /
39-dim model.  The statistics are centred on the CURRENT means, which every rank
holds identically, so summing the per-rank blocks is exact.

Nothing here touches likelihood arithmetic: `accumulate` is a callable supplied by the
caller (the HIP E-step in production, anything in tests).
"""
import numpy as np

__all__ = ["shard_utterances", "StatsAllReducer", "m_step", "distributed_em_iteration"]


def shard_utterances(lengths, world_size):
    """Greedy longest-first balancing of utterances over ranks by frame count.
    Returns a list of index arrays (one per rank), each sorted ascending."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for u in order:
        r = int(np.argmin(load))
        shards[r].append(int(u))
        load[r] += lengths[u]
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


class StatsAllReducer:
    """Sum a packed fp64 statistics buffer over all ranks (one collective per call).

    Two forms.  `device_buffer(n)` + `reduce_device()`: the buffer lives in HBM (a torch tensor whose
    `data_ptr()` is handed to `gh_bw_accumulate(stats_dev=...)`), RCCL reduces it where the kernels left
    it and only the reduced result crosses to the host -- the path of a multi-GPU run (backend "nccl").
    `__call__(ndarray)`: host buffer through the process group's backend (gloo in the CPU tests)."""

    def __init__(self, device=None, gpu_index=None):
        """device: torch device the buffer is reduced on.  Default: the GPU `gpu_index` (or torch's
        current one) when the process group runs on RCCL ("nccl"), the host for gloo."""
        import sys
        self.torch = self.dist = None
        self.enabled = False
        self.device = device
        self.calls, self.seconds = 0, 0.0          # collectives issued / host time spent in them (bench.py reports it)
        self._buf = None
        if "torch" not in sys.modules:   # no process group can exist: do not pay for importing torch (seconds to minutes cold)
            return
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.enabled = dist.is_available() and dist.is_initialized()
        if device is None and self.enabled and dist.get_backend() == "nccl":
            device = torch.device("cuda", torch.cuda.current_device() if gpu_index is None else int(gpu_index))
        self.device = device

    @property
    def world_size(self):
        return self.dist.get_world_size() if self.enabled else 1

    @property
    def on_gpu(self):
        """True when the collective runs on device memory (RCCL): use device_buffer / reduce_device."""
        return bool(self.enabled and self.device is not None and getattr(self.device, "type", str(self.device)) == "cuda")

    def device_buffer(self, n):
        """(tensor, device pointer) of an fp64 buffer of n entries on the reduce device (reused between calls)."""
        if self._buf is None or self._buf.numel() != n:
            self._buf = self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)
        return self._buf, int(self._buf.data_ptr())

    def reduce_device(self, tensor=None):
        """All-reduce (sum) the device buffer in place; returns the reduced buffer as a host ndarray."""
        import time
        t = self._buf if tensor is None else tensor
        t0 = time.perf_counter()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        out = t.cpu().numpy()                      # waits for the collective
        self.calls += 1
        self.seconds += time.perf_counter() - t0
        return out

    def __call__(self, stats):
        """stats: numpy fp64 array (any shape) -> summed over ranks, same shape."""
        if not self.enabled or self.world_size == 1 and not self.on_gpu:
            return stats
        import time
        t0 = time.perf_counter()
        t = self.torch.from_numpy(np.ascontiguousarray(stats, dtype=np.float64))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        out = t.cpu().numpy().reshape(stats.shape)
        self.calls += 1
        self.seconds += time.perf_counter() - t0
        return out


def m_step(stats, counts, means):
    """M-step of GMM.em from (all-reduced) centred statistics (hmm_state.py:134-148).

    stats [S, k, 1+2D] (S0, S1 = sum r (x - mean), S2 = sum r (x - mean)^2), counts [S]
    frames per state, means [S, k, D] the means the statistics were centred on.
    Returns (new means, new variances, new weights)."""
    D = means.shape[2]
    s0 = stats[:, :, 0]
    occ = np.where(s0 == 0, 10 ** (-5), s0)
    mu = (means * s0[:, :, None] + stats[:, :, 1:1 + D]) / occ[:, :, None]
    delta = mu - means
    sigma = (stats[:, :, 1 + D:] - delta * (2.0 * stats[:, :, 1:1 + D] - delta * s0[:, :, None])) / occ[:, :, None]
    with np.errstate(divide="ignore", invalid="ignore"):
        w = s0 / np.asarray(counts, dtype=np.float64)[:, None]
    return mu, sigma, w


def distributed_em_iteration(accumulate, means, vars_, weights, reducer=None):
    """One lock-step EM iteration over all states of a sharded training set.

    accumulate(means, vars, weights) -> (stats [S,k,1+2D], counts [S]) for THIS rank's frames;
    `reducer` sums the packed buffer over ranks.  Returns the updated (means, vars, weights)."""
    stats, counts = accumulate(means, vars_, weights)
    S = stats.shape[0]
    packed = np.concatenate([stats.reshape(S, -1), np.asarray(counts, dtype=np.float64).reshape(S, 1)], axis=1)
    if reducer is not None:
        packed = reducer(packed)
    stats = packed[:, :-1].reshape(stats.shape)
    counts = packed[:, -1]
    return m_step(stats, counts, means)
