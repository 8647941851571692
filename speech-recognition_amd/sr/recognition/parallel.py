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

__all__ = ["shard_utterances", "StatsAllReducer", "NativeReducer", "exchange_from_rank0", "m_step",
           "distributed_em_iteration"]


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


_MAGIC = b"GHCOMM1\0"


def _comm_ports():
    """Candidate TCP ports of the id hand-over: GMMHMM_COMM_PORT, else MASTER_PORT + 1 .. + 8 (MASTER_PORT itself
    belongs to the launcher's store)."""
    import os
    if os.environ.get("GMMHMM_COMM_PORT"):
        return [int(os.environ["GMMHMM_COMM_PORT"])]
    base = int(os.environ.get("MASTER_PORT", "29500"))
    return [base + 1 + i if base + 1 + i < 65536 else base - 1 - i for i in range(8)]


def exchange_from_rank0(rank, world, make_payload, addr=None, ports=None, timeout=180.0, token=b""):
    """Rank 0 calls make_payload() -> bytes and hands the result to every other rank over a plain TCP socket on
    MASTER_ADDR (the rendezvous of `gh_comm_create`: the 128-byte RCCL id must reach every rank before the first
    collective exists).  Every connection opens with a magic word, the world size, the caller's rank and `token`
    (e.g. the launcher's run id); a peer that answers anything else is not ours and the next candidate port is
    tried.  Returns the payload on every rank."""
    import os
    import socket
    import struct
    import time
    if world <= 1:
        return make_payload()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    ports = list(ports) if ports is not None else _comm_ports()
    hello = _MAGIC + struct.pack("<ii", world, 0) + struct.pack("<i", len(token)) + token
    deadline = time.monotonic() + timeout
    if rank == 0:
        payload = make_payload()
        srv, err = None, None
        for port in ports:
            try:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind((addr, port))
                srv.listen(world)
                break
            except OSError as e:
                err = e
                srv.close()
                srv = None
        if srv is None:
            raise RuntimeError("exchange_from_rank0: no free port among %s on %s (%s)" % (ports, addr, err))
        served = set()
        try:
            while len(served) < world - 1:
                srv.settimeout(max(0.1, deadline - time.monotonic()))
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    raise RuntimeError("exchange_from_rank0: %d of %d ranks reported within %.0f s"
                                       % (len(served), world - 1, timeout))
                with conn:
                    conn.settimeout(10.0)
                    try:
                        head = _recv_exact(conn, len(hello))
                    except (OSError, EOFError):
                        continue
                    peer = struct.unpack("<i", head[12:16])[0]
                    if head[:12] != hello[:12] or head[16:] != hello[16:] or not 0 < peer < world:
                        continue                          # not one of ours
                    conn.sendall(struct.pack("<q", len(payload)) + payload)
                    served.add(peer)
        finally:
            srv.close()
        return payload
    mine = _MAGIC + struct.pack("<ii", world, rank) + struct.pack("<i", len(token)) + token
    k = 0
    while True:
        port = ports[k % len(ports)]
        k += 1
        try:
            with socket.create_connection((addr, port), timeout=5.0) as c:
                c.settimeout(10.0)
                c.sendall(mine)
                n = struct.unpack("<q", _recv_exact(c, 8))[0]
                if 0 < n < (1 << 24):
                    return _recv_exact(c, n)
        except (OSError, EOFError):
            pass
        if time.monotonic() > deadline:
            raise RuntimeError("exchange_from_rank0: rank %d could not reach rank 0 on %s ports %s within %.0f s"
                               % (rank, addr, ports, timeout))
        time.sleep(0.05 if k < 40 else 0.5)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise EOFError("peer closed the connection")
        buf += chunk
    return bytes(buf)


class NativeReducer:
    """The exchange step through the library's own RCCL communicator (`gh_comm_create` / `gh_stats_allreduce`): no
    torch in the process.  One rank per GPU; rank / world size come from the launcher's environment (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR, MASTER_PORT) unless given.  Same surface as `StatsAllReducer` where the trainers use it:
    `enabled`, `world_size`, `rank`, `__call__(ndarray)` (small host buffers), plus `comm` for the device-resident
    paths (`_hip.EMSession.iteration(comm=...)`)."""

    on_gpu = False      # (no torch tensor to reduce: device-resident paths take `comm`, host arrays go through __call__)
    native = True

    def __init__(self, ctx=None, rank=None, world=None, timeout=180.0):
        import os
        from . import _hip
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.ctx = ctx if ctx is not None else _hip.default_context()
        token = os.environ.get("TORCHELASTIC_RUN_ID", "").encode()
        uid = exchange_from_rank0(self.rank, self.world, _hip.Comm.unique_id, timeout=timeout, token=token)
        self.comm = _hip.Comm(self.ctx, self.rank, self.world, uid)
        self.enabled = True
        self.calls, self.seconds = 0, 0.0

    @property
    def world_size(self):
        return self.world

    def __call__(self, stats):
        import time
        t0 = time.perf_counter()
        out = self.comm.allreduce_host(stats)
        self.calls += 1
        self.seconds += time.perf_counter() - t0
        return out

    def max(self, a):
        return self.comm.allreduce_host(a, op="max")

    def barrier(self):
        self.comm.barrier()

    def abort(self):
        """This rank leaves its collectives for good (gh_comm_abort): called on the way out after a failure, so that the
        peers' waits end in `_hip.CommError` (connection closed / deadline) instead of a hang."""
        self.comm.abort()

    def close(self):
        self.comm.close()


def exit_rank_on_failure(fn, reducer_of=lambda: None):
    """Run `fn()`; on ANY exception print it, abort the communicator `reducer_of()` returns (if any) and leave the process
    with status 1 -- from where it stands (os._exit: no interpreter teardown that could wait on a collective, and never a
    re-exec of a process that has touched the GPU).  A launcher that sees the status takes the other ranks down; ranks
    that are already inside a collective with this one get GH_ERR_COMM from the library (GMMHMM_COMM_TIMEOUT)."""
    import os
    import sys
    import traceback
    try:
        return fn()
    except SystemExit:
        raise
    except BaseException:
        traceback.print_exc()
        try:
            red = reducer_of()
            if red is not None and hasattr(red, "abort"):
                red.abort()
        except BaseException:
            traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)


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
