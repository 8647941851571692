#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Headline benchmark: frame-state log-likelihoods/s + utterances/s Viterbi decode
(BASELINE.json) of the GMM-HMM hot path on MI355X, through the C ABI.

    python bench.py --gpus N --steps K --warmup W

N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` (RANK /
LOCAL_RANK / WORLD_SIZE in the environment), or started directly -- then this process only spawns the N rank
processes (before anything touches HIP or torch.cuda) and relays their exit code.  One rank per GPU, each pinned to
the host cores next to its GPU.  The ranks talk through the LIBRARY'S OWN RCCL communicator (`--comm native`:
gh_comm_create / gh_stats_allreduce, collectives on the library's stream, no torch in the data path; `--comm torch`
is the torch.distributed form of round 2).  `--same-gpu` rehearses N native RCCL ranks on a one-GPU box (every rank
on GPU 0 with a host id of its own, socket transport); `--backend gloo --device 0` is the gloo rehearsal.

One "step" = one pass of the hot path over the batch: batched GMM log-likelihood
of every frame against every state (A3) + isolated-word Viterbi of every
utterance through the 10 stacked word models (A6) + arg-min word per utterance
(sr/core.py:82-87).  Workload = BASELINE.json configs[1]: 10-digit topology,
5 states/digit, 8-mix GMM, 39-dim features, 10k synthetic utterances per GPU
(weak scaling: every rank decodes its own 10k utterances; no data-path
collective -- decode shards by utterance).  Features are resident in HBM before
the timed region.  Arithmetic: fp64 (the reference's), which is what keeps the
Viterbi paths bit-identical; --dtype f32 runs the fp32 likelihood kernel.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra
objects: "roofline" for the dominant kernel (the likelihood kernel; HIP-event
timed live on the library's stream, gh_event_*), "cpu_baseline" (the numpy oracle,
reference-shaped scalar loops, 1 core, on a bounded sample), "em" (configs[2]: soft-EM
iterations on every rank's shard with ONE all-reduce of the device-resident statistics
per iteration -- em_utterances_per_s, allreduce_ms, rccl_ranks), "configs" (the other
BASELINE.json configs timed after the headline region: C1 x1000, C4, C5 K-layer and
loop grammar at the per-GPU size) and "pcie_inclusive" (the headline step with the
feature upload inside the step).  None of these touch `value` / `ms_per_step`.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "speech-recognition_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_HBM = 8.0e12      # B/s, MI355X_MICROARCH.md "HBM3E peak BW" (spec)
PEAK_F32 = 157.3e12    # flop/s, fp32 MFMA == fp32 vector (MI355X_MICROARCH.md)
PEAK_F64 = 78.6e12     # flop/s, fp64 vector == fp64 MFMA on MI355X (AMD spec; half the fp32 rate)


def synth_workload(seed, n_utts, W=10, n=5, M=8, D=39, tmin=50, tmax=150, utt_seed=None, words=None, T=None):
    """means ~ N(0,1), vars ~ U[.5,1.5], w ~ Dirichlet(1); left-to-right
    costs (self -log .9, next -log .1, last self 0); frames from a uniform segmentation.
    words / T given: those utterances (a rank's shard of a globally drawn list) instead of drawing n_utts."""
    rng = np.random.default_rng(seed)
    means = rng.normal(size=(W, n, M, D))
    vars_ = rng.uniform(0.5, 1.5, size=(W, n, M, D))
    w = rng.dirichlet(np.ones(M), size=(W, n))
    trans = np.full((n, n), np.inf)
    for i in range(n):
        trans[i, i] = -np.log(0.9) if i < n - 1 else 0.0
        if i < n - 1:
            trans[i + 1, i] = -np.log(0.1)
    if utt_seed is not None:
        rng = np.random.default_rng(utt_seed)
    if words is None:
        words = rng.integers(0, W, size=n_utts)
        T = rng.integers(tmin, tmax + 1, size=n_utts)
    else:
        words, T = np.asarray(words), np.asarray(T)
        n_utts = len(words)
    off = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    N = int(off[-1])
    utt = np.repeat(np.arange(n_utts), T)
    t = np.arange(N) - off[utt]
    st = np.minimum(t * n // T[utt], n - 1)
    comp = rng.integers(0, M, size=N)
    gidx = (words[utt] * n + st) * M + comp
    mflat, sflat = means.reshape(-1, D), np.sqrt(vars_).reshape(-1, D)
    X = np.empty((N, D))
    CH = 1 << 15  # chunked, in place: large temporaries are slow to fault in on some hosts
    tmp = np.empty((CH, D))
    for a in range(0, N, CH):
        b = min(N, a + CH)
        xs, ts = X[a:b], tmp[:b - a]
        rng.standard_normal(out=xs)
        np.take(sflat, gidx[a:b], axis=0, out=ts)
        xs *= ts
        np.take(mflat, gidx[a:b], axis=0, out=ts)
        xs += ts
    return dict(means=means, vars=vars_, w=w, trans=trans, words=words, off=off, X=X, W=W, n=n, M=M, D=D)


def stacked_graph(W, n, trans):
    to, frm = np.nonzero(~np.isinf(trans))
    return dict(row_state=np.arange(W * n),
                arc_to=np.concatenate([to + i * n for i in range(W)]),
                arc_from=np.concatenate([frm + i * n for i in range(W)]),
                arc_cost=np.tile(trans[to, frm], W),
                start_rows=[i * n for i in range(W)], end_rows=[i * n + n - 1 for i in range(W)])


def profiled_traffic(dtype, n_frames):
    """HBM bytes per launch of the likelihood kernel from the committed PMC passes (the newest
    profiles/*_pmc_traffic_<dtype>.json: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs of this
    same command, folded by tools/pmc_traffic.py); None when no profile matches this dtype / workload size."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % dtype)))
    esz = 8 if dtype == "f64" else 4
    for path in reversed(paths):      # newest first; profiles of other workloads (C5, forced alignment) do not match the size
        try:
            d = json.load(open(path))["kernels"]
            k = next(v for name, v in d.items() if name.startswith("loglik"))
            alg = k["algorithmic_bytes_per_launch"]
            if alg and abs(alg - n_frames * esz * (39 + 50)) <= 1e-3 * alg:
                return k["hbm_bytes_per_launch"]
        except Exception:
            continue
    return None


def profiled_fused_traffic(n_frames, D, esz):
    """HBM bytes per launch of the fused single-Gaussian sweep from the committed PMC passes (the newest
    profiles/*_pmc_fused_traffic.json: FETCH_SIZE calibrated on tools/hbm_stream.hip in the kernel's own access width
    -- 8 B per lane reads are tallied at half their bytes, like the 16 B ones -- plus WRITE_SIZE; tools/pmc_calibrated.py);
    None when no profile matches this workload (compulsory reads = n_frames x D x esz)."""
    import glob
    for path in reversed(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_fused_traffic.json")))):
        try:
            ks = json.load(open(path))["kernels"]
            for k in ks.values():
                want = float(n_frames) * D * esz
                if abs(k["compulsory_read_bytes"] - want) <= 0.02 * want and k.get("corrected_read_bytes"):
                    return k["corrected_read_bytes"] + k.get("write_bytes", 0.0)
        except Exception:
            continue
    return None


def cpu_baseline(wl, n_utts=20):
    """The numpy oracle run the way the reference runs (per-frame GMM.evaluate through dense
    inverse covariances, per-cell Python DP), one core, on the first `n_utts` utterances."""
    from oracle import ref_numpy as O
    W, n = wl["W"], wl["n"]
    nes = np.zeros(n, dtype=bool)
    t0 = time.perf_counter()
    frames = 0
    correct = 0
    for u in range(n_utts):
        x = wl["X"][wl["off"][u]:wl["off"][u + 1]]
        ev = []
        for i in range(W):
            states = [(wl["means"][i, s], wl["vars"][i, s], wl["w"][i, s]) for s in range(n)]
            costs, _ = O.decode_states(O.emission_matrix(x, states, dense=True), nes, wl["trans"])
            ev.append(costs[-1, -1])
        correct += int(np.argmin(ev) == wl["words"][u])
        frames += len(x)
    dt = time.perf_counter() - t0
    out = dict(value=frames * W * n / dt, unit="frame-state loglik/s", cores=1, kind="port",
               sample="first %d utterances (%d frames) of the same workload, oracle/ref_numpy.py "
                      "(emission_matrix dense=True + decode_states), %.1f s" % (n_utts, frames, dt),
               utterances_per_s=n_utts / dt, accuracy=correct / n_utts)
    # beside it: the same arithmetic VECTORISED with numpy (log-domain batch likelihoods, oracle.gmm_neg_loglik_batch;
    # the DP stays per cell), i.e. what a CPU user who is not bound to the reference's per-frame objects would run
    nv = min(len(wl["words"]), 3 * n_utts)
    Xs = wl["X"][:wl["off"][nv]]
    S = W * n
    t1 = time.perf_counter()
    O.gmm_neg_loglik_batch(Xs, wl["means"].reshape(S, *wl["means"].shape[2:]), wl["vars"].reshape(S, *wl["vars"].shape[2:]),
                           wl["w"].reshape(S, -1))
    dv = time.perf_counter() - t1
    out["vectorized_numpy_loglik"] = dict(value=len(Xs) * S / dv, unit="frame-state loglik/s", frames=int(len(Xs)),
                                          seconds=dv, threads="numpy default")
    return out


_CPU_WL = None


def _cpu_all_cores_worker(rng_):
    """utterances [lo, hi) of the forked parent's workload on the vectorised oracle path: (frames, correct)"""
    from oracle import ref_numpy as O
    wl = _CPU_WL
    W, n = wl["W"], wl["n"]
    S = W * n
    nes = np.zeros(n, dtype=bool)
    means, vars_, w = (wl["means"].reshape(S, *wl["means"].shape[2:]), wl["vars"].reshape(S, *wl["vars"].shape[2:]),
                       wl["w"].reshape(S, -1))
    frames = correct = 0
    for u in range(*rng_):
        x = wl["X"][wl["off"][u]:wl["off"][u + 1]]
        nll = O.gmm_neg_loglik_batch(x, means, vars_, w)                     # [T, S], log domain, numpy-vectorised
        ev = [O.decode_states(nll[:, i * n:(i + 1) * n].T, nes, wl["trans"])[0][-1, -1] for i in range(W)]
        correct += int(np.argmin(ev) == wl["words"][u])
        frames += len(x)
    return frames, correct


def cpu_all_cores(wl, seconds=8.0):
    """SURVEY.md 8(d)'s all-cores CPU leg: the same workload on EVERY host core of the box -- multiprocessing over
    utterances, each worker the oracle's vectorised path (batch log-likelihoods in numpy, the DP per cell) -- on a sample
    sized for about `seconds`.  Runs BEFORE anything touches the GPU (forked workers, no exec) and before the rank pins
    itself to its cores.  A reported baseline, not a target."""
    import multiprocessing as mp
    global _CPU_WL
    _CPU_WL = wl
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    t0 = time.perf_counter()
    _cpu_all_cores_worker((0, 4))                                         # one core's rate, to size the sample
    per_utt = (time.perf_counter() - t0) / 4
    n_utts = int(max(cores * 4, min(len(wl["words"]), seconds * cores / max(per_utt, 1e-4))))
    chunks = [(i, min(n_utts, i + 8)) for i in range(0, n_utts, 8)]
    ctx = mp.get_context("fork")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_all_cores_worker, chunks)
    dt = time.perf_counter() - t0
    frames = sum(r[0] for r in res)
    S = wl["W"] * wl["n"]
    return dict(value=frames * S / dt, unit="frame-state loglik/s", cores=cores, kind="port",
                sample="first %d utterances (%d frames), oracle/ref_numpy.py vectorised likelihoods + per-cell DP, "
                       "multiprocessing over utterances, %.1f s" % (n_utts, frames, dt),
                utterances_per_s=n_utts / dt, accuracy=sum(r[1] for r in res) / n_utts)


class native_stdout_to_stderr:
    """RCCL prints a version banner on file descriptor 1 when its first communicator comes up; the contract of this
    script is ONE JSON line on stdout, so native writes to fd 1 are sent to stderr while a process group starts."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


# ------------------------------------------------------------------------------------------ N-rank launch
def spawn_ranks(n, argv, same_gpu=False):
    """`bench.py --gpus N` started without a launcher: start N rank processes with the torchrun environment and
    relay the worst exit code.  This parent never imports torch and never calls HIP (a process that has touched the
    GPU must not fork / exec rank processes on this pool)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        if same_gpu:
            # N native RCCL ranks on ONE GPU: RCCL refuses two ranks of one host on one device, so every rank announces
            # a host id of its own and the ranks talk over the socket transport on the loopback interface
            env.update(NCCL_HOSTID="gmmhmm-bench-%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                       NCCL_SHM_DISABLE="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    live = list(procs)
    while live:            # a rank that dies leaves the others waiting in a collective: end them with it
        for p in list(live):
            r = p.poll()
            if r is not None:
                live.remove(p)
                rc = max(rc, abs(r))
        if rc and live:
            time.sleep(2.0)
            for p in live:
                p.kill()
        time.sleep(0.05)
    return rc


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_local_cpus(index):
    """Host cores next to GPU `index` WITHOUT touching the GPU: the KFD topology lists the GPUs in the order HIP
    enumerates them (minus *_VISIBLE_DEVICES filtering) with their PCI address; sysfs knows that device's local
    cpulist.  Returns a set of CPU numbers or None."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        gpus = []
        for node in sorted(os.listdir(base), key=int):
            props = dict(l.split(None, 1) for l in open(os.path.join(base, node, "properties")).read().splitlines() if " " in l)
            if int(props.get("simd_count", "0")) > 0 and int(props.get("gfx_target_version", "0")) > 0:
                gpus.append(props)
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        if vis and all(v.strip().isdigit() for v in vis.split(",")):
            gpus = [gpus[int(v)] for v in vis.split(",") if int(v) < len(gpus)]
        g = gpus[index]
        loc, dom = int(g["location_id"]), int(g.get("domain", "0"))
        addr = "%04x:%02x:%02x.%x" % (dom, (loc >> 8) & 0xff, (loc >> 3) & 0x1f, loc & 7)
        cpus = _parse_cpulist(open("/sys/bus/pci/devices/%s/local_cpulist" % addr).read())
        return cpus or None
    except Exception:
        return None


def pin_rank(local_rank, local_world, dev):
    """Pin this rank process to the host cores of its GPU's NUMA node (shared evenly with the other ranks whose GPUs
    sit on the same node); without topology information: an even slice of the allowed cores.  Called before anything
    touches the GPU.  Returns a description for the JSON line."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return None
    if local_world <= 1 or len(allowed) < 2 * local_world:
        return {"cpus": len(allowed), "how": "unpinned"}
    near = gpu_local_cpus(dev)
    how = "numa"
    mine = None
    if near:
        near = sorted(near & set(allowed))
        sharers = [r for r in range(local_world) if (gpu_local_cpus(r) or set()) & set(near)] or [local_rank]
        if local_rank in sharers and len(near) >= 2 * len(sharers):
            k = sharers.index(local_rank)
            per = len(near) // len(sharers)
            mine = near[k * per:(k + 1) * per]
    if not mine:
        per = len(allowed) // local_world
        mine = allowed[local_rank * per:(local_rank + 1) * per]
        how = "even slice"
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {"cpus": len(allowed), "how": "unpinned"}
    return {"cpus": len(mine), "first": mine[0], "last": mine[-1], "how": how}


class Group:
    """What the bench needs from the ranks of a run: barrier, max / sum of a few host numbers, a device-wide sync.
    kind "native": the library's own RCCL communicator (parallel.NativeReducer; torch is never imported);
    kind "torch": torch.distributed (nccl or gloo), the form of round 2; kind "single": one rank."""

    def __init__(self, args, rank, world, dev):
        from sr.recognition import _hip
        self.rank, self.world, self.dev = rank, world, dev
        self.torch = self.dist = self.native = None
        self.kind = "single"
        self.red_dev = "cpu"
        if world > 1 and args.comm == "torch":
            import torch
            import torch.distributed as dist
            self.torch, self.dist, self.kind = torch, dist, "torch"
            with native_stdout_to_stderr():
                if args.backend == "nccl":
                    torch.cuda.set_device(dev)
                    dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
                    self.red_dev = "cuda"
                else:
                    dist.init_process_group("gloo")
                dist.barrier()      # communicators come up here (and RCCL's banner is printed), not inside the timed region
        self.ctx = _hip.default_context(dev)
        if world > 1 and args.comm == "native":
            from sr.recognition.parallel import NativeReducer
            if self.ctx.lib.gh_comm_version() <= 0:
                raise RuntimeError("librccl cannot be opened by libgmmhmm.so (use --comm torch): %s"
                                   % self.ctx.lib.gh_last_error().decode())
            with native_stdout_to_stderr():
                self.native = NativeReducer(self.ctx, rank, world)
                Group.active_native = self.native      # (what a failing rank aborts on its way out)
                self.native.barrier()
            self.kind = "native"

    def sync_device(self):
        self.ctx.device_sync()
        if self.torch is not None and self.red_dev == "cuda":
            self.torch.cuda.synchronize()

    def barrier(self):
        if self.kind == "native":
            self.native.barrier()
        elif self.kind == "torch":
            self.dist.barrier()

    def max(self, x):
        if self.kind == "native":
            return float(self.native.max(np.array([float(x)]))[0])
        if self.kind == "torch":
            t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self.red_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return float(t.item())
        return float(x)

    def maxv(self, a):
        a = np.asarray(a, dtype=np.float64)
        if self.kind == "native":
            return self.native.max(a)
        if self.kind == "torch":
            t = self.torch.from_numpy(a.copy()).to(self.red_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return t.cpu().numpy()
        return a

    def sum(self, a):
        a = np.asarray(a, dtype=np.float64)
        if self.kind == "native":
            return self.native(a)
        if self.kind == "torch":
            t = self.torch.from_numpy(a.copy()).to(self.red_dev)
            self.dist.all_reduce(t)
            return t.cpu().numpy()
        return a

    def close(self):
        if self.kind == "native":
            self.native.barrier()
            self.native.close()
        elif self.kind == "torch":
            self.dist.barrier()
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ EM leg
def em_leg(args, group):
    """configs[2]: soft-EM iterations on this rank's shard.  Default (weak scaling): `--em-utts` utterances per GPU
    (100k / 8 GPUs = 12 500); `--em-total T` (strong scaling): T utterances drawn once, sharded over the ranks by
    parallel.shard_utterances (greedy longest-first by frames).  The iteration is the device-resident session
    (gh_em_iteration): own-state likelihoods -> forward-backward -> statistics -> ONE ncclAllReduce of the packed fp64
    buffer on the same stream -> M-step -> model re-pack; the host only reads (log P, converged) back."""
    from sr.recognition import _hip
    from sr.recognition.train import BaumWelchTrainer
    from sr.recognition.parallel import NativeReducer, StatsAllReducer, shard_utterances
    rank, world, dev = group.rank, group.world, group.dev
    info = {}
    own_red = None
    if group.kind == "native":
        red = group.native
    elif group.kind == "torch":
        red = StatsAllReducer(gpu_index=dev)
    else:
        # single-GPU run: a one-rank RCCL communicator, so that the collective of the training path runs on every bench
        try:
            with native_stdout_to_stderr():
                red = own_red = NativeReducer(group.ctx, 0, 1)
        except Exception as e:  # reported, not fatal: the EM leg then runs without a communicator
            info["rccl_error"] = repr(e)[:200]
            red = None
    if args.em_total:
        rng = np.random.default_rng(1003)
        g_words = rng.integers(0, 10, size=args.em_total)
        g_T = rng.integers(50, 151, size=args.em_total)
        mine = shard_utterances(g_T, world)[rank]
        wl = synth_workload(1003, len(mine), utt_seed=1003 + 7919 * rank, words=g_words[mine], T=g_T[mine])
        U = len(mine)
    else:
        U = args.em_utts
        wl = synth_workload(1003, U, utt_seed=None if rank == 0 else 1003 + 7919 * rank)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
    labels = [[int(w)] for w in wl["words"]]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)   # perturbed start, same on every rank
    tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels, device=dev, reducer=red)
    del data

    def fence():
        tr.ctx.sync()
        group.sync_device()
        group.barrier()
    hist = [tr.iteration()]                       # warm-up iteration (allocations, RCCL channel set-up)
    if red is not None:
        red.calls, red.seconds = 0, 0.0
    # the same iteration with occ_floor = 0 (every non-zero posterior enters the statistics), timed beside the default
    dt_exact = None
    if tr.session is not None and group.world == 1:
        tr0 = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)],
                               labels, device=dev, reducer=red, occ_floor=0.0)
        tr0.iteration()
        tr0.ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.em_iters):
            tr0.iteration()
        dt_exact = (time.perf_counter() - t0) / args.em_iters
        tr0.close()
    # (a) one 32-byte D2H of (log P, converged) per iteration -- what a training loop with a stop rule does
    fence()
    t0 = time.perf_counter()
    for _ in range(args.em_iters):
        hist.append(tr.iteration())
    fence()
    dt = group.max(time.perf_counter() - t0)
    # (b) the same iterations enqueued back to back, history read once at the end (no stop rule to evaluate in between)
    dt_q = None
    if tr.session is not None:
        fence()
        t0 = time.perf_counter()
        for _ in range(args.em_iters):
            tr.iteration(sync=False)
        tr.drain()
        fence()
        dt_q = group.max(time.perf_counter() - t0)
        hist = list(tr.history)
    frames = float(tr.batch.N)
    # (c) the phases of an iteration from HIP events on the kernels' stream (gh_em_profile): rooflines of the E-step kernels
    phase_roofline = None
    if tr.session is not None:
        try:
            tr.session.profile(True)
            ph = np.zeros(4)
            for _ in range(10):
                tr.iteration(sync=False)
                ph += tr.session.phase_ms()
            tr.drain()
            tr.session.profile(False)
            ph /= 10.0
            n_, M_, D_ = tr.n, tr.M, tr.D
            fl_ll = 2.0 * 2 * D_ * n_ * M_ * frames                       # the utterance's own word: n states x M components
            fl_bw = (2.0 * 2 * D_ + 2.0 * (2 * D_ + 1)) * n_ * M_ * frames  # densities again + the statistics GEMM
            by_fb = (8.0 * n_ + 8.0 * tr.session_lanes()) * frames if hasattr(tr, "session_lanes") else (8.0 * n_ + 64.0) * frames
            phase_roofline = [
                {"kernel": "loglik_mfma_kernel (own-state subset)", "ms": ph[0], "bound": "mfma", "achieved": fl_ll / (ph[0] * 1e-3) / 1e12,
                 "peak": PEAK_F64 / 1e12, "unit": "TFLOP/s", "frac": fl_ll / (ph[0] * 1e-3) / PEAK_F64},
                {"kernel": "fb_chain2_kernel + fb_chain2_cells_kernel", "ms": ph[1], "bound": "hbm (latency bound at this size)",
                 "achieved": by_fb / (ph[1] * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": by_fb / (ph[1] * 1e-3) / PEAK_HBM,
                 "bytes_per_frame": by_fb / frames,
                 "note": "algorithmic bytes: the n own-state likelihoods read + one 64-byte line of compact gamma written per frame; "
                         "12 500 utterances are 1.5 waves per SIMD walking ~150 dependent columns each: the recursion waits, HBM idles"},
                {"kernel": "bw_fused_kernel + bw_fused_reduce_kernel", "ms": ph[2], "bound": "mfma", "achieved": fl_bw / (ph[2] * 1e-3) / 1e12,
                 "peak": PEAK_F64 / 1e12, "unit": "TFLOP/s", "frac": fl_bw / (ph[2] * 1e-3) / PEAK_F64,
                 "note": "nominal flops (every frame x every own state); blocks without occupancy above occ_floor are skipped"},
                {"kernel": "em_tail + all-reduce + em_mstep + gmm re-pack + em_finish", "ms": ph[3]}]
        except Exception as e:     # (a library without gh_em_profile, two streams: the line goes out without the entry)
            phase_roofline = {"error": repr(e)[:200]}
    tot = group.sum([float(U), frames])
    all_utts, all_frames = float(tot[0]), float(tot[1])
    # the collective by itself: the packed buffer's size through the same communicator, stream-synchronised
    ar_ms, n_ranks = None, 0
    comm = tr._comm()
    if comm is not None:
        n_ranks = comm.count                                   # ncclCommCount
        buf = np.zeros(tr._packed_len())
        red(buf)
        group.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            red(buf)
        ar_ms = group.max((time.perf_counter() - t0) / 10) * 1e3
    elif group.kind == "torch":
        ar_ms = (red.seconds / red.calls * 1e3) if red.calls else None
        n_ranks = group.dist.get_world_size() if group.dist.get_backend() == "nccl" else 0
    per_it = dt / args.em_iters
    info.update({
        "workload": "configs[2]: soft EM (fwd-bwd E-step + all-reduced statistics + M-step), 10x5 states, 8-mix, 39-dim, "
                    + ("%d utterances sharded over %d GPUs (strong scaling)" % (args.em_total, world) if args.em_total
                       else "%d utterances per GPU (weak scaling)" % U),
        "scaling": "strong" if args.em_total else "weak",
        "iterations": args.em_iters, "ms_per_iteration": per_it * 1e3,
        "ms_per_iteration_enqueued": None if dt_q is None else dt_q / args.em_iters * 1e3,
        "em_utterances_per_s": all_utts / per_it, "em_frames_per_s": all_frames / per_it,
        "occ_floor": tr.occ_floor, "ms_per_iteration_occ_floor_0": None if dt_exact is None else dt_exact * 1e3,
        "roofline": phase_roofline,
        "device_resident_iteration": tr.session is not None,
        "host_syncs_per_iteration": 1 if tr.session is not None else 3,
        "allreduce_ms": ar_ms, "allreduce_bytes": tr._packed_len() * 8,
        "allreduce_note": "host-staged round trip of the same buffer through the same communicator; inside the iteration "
                          "the collective sits on the kernels' stream (gh_stats_allreduce)" if comm is not None else None,
        "allreduce_on_device_buffer": bool(comm is not None or getattr(red, "on_gpu", False)),
        "comm": group.kind if group.kind != "single" else ("native" if comm is not None else None),
        "backend": "nccl" if comm is not None else (group.dist.get_backend() if group.kind == "torch" else None),   # ("nccl" IS RCCL on ROCm)
        "rccl_ranks": n_ranks, "rccl_library": tr.ctx.lib.gh_comm_library().decode() if comm is not None else None,
        "loglik_per_frame": [h / all_frames for h in hist],
        "loglik_monotone": bool(all(b >= a - 1e-9 * abs(a) for a, b in zip(hist, hist[1:]))),
    })
    tr.close()
    if own_red is not None:
        own_red.close()
    return info


# ----------------------------------------------------------------------------- the other BASELINE configs
def _timeit(fn, reps=3, ramp=0.3):
    """Mean wall time of fn() (each call synchronises) after `ramp` seconds of the same work (clock governor)."""
    t_r = time.perf_counter()
    out = fn()
    while time.perf_counter() - t_r < ramp:
        out = fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    return (time.perf_counter() - t0) / reps, out


def _isolated_config(ctx, group, name, seed, U, W, n, M, D, npdt, peak_flops, mismatch=False):
    from sr.recognition import _hip
    wl = synth_workload(seed, U, W=W, n=n, M=M, D=D, utt_seed=None if group.rank == 0 else seed + 7919 * group.rank)
    S = W * n
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=npdt)
    lat = _hip.Lattices(ctx, [stacked_graph(W, n, wl["trans"])])
    group.barrier()
    t_ll, _ = _timeit(lambda: (b.loglik(gmm, fetch=False), ctx.sync()))
    # (the recognised word = the cheapest word end: `best_end`, found on the device; the [U, W] end costs stay there)
    t_vit, r = _timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False))
    words = r["best_end"]
    esz = np.dtype(npdt).itemsize
    N = b.N
    flops, bytes_ll, bytes_vit = 2.0 * 2 * D * S * M * N, float(esz * (D + S) * N), float(esz * S * N)
    hbm_bound = flops / bytes_ll < peak_flops / PEAK_HBM
    tm = group.maxv([t_ll, t_vit])
    tot = group.sum([float(U), float(N), float(np.mean(words == wl["words"]))])
    g_ll, g_vit = float(tm[0]), float(tm[1])
    out = {"workload": name, "n_gpus": group.world, "utterances": int(tot[0]), "frames": int(tot[1]), "states": S, "mixtures": M, "dim": D,
           "ms": (g_ll + g_vit) * 1e3, "loglik_ms": g_ll * 1e3, "viterbi_ms": g_vit * 1e3,
           "value": tot[1] * S / (g_ll + g_vit), "unit": "frame-state loglik/s", "utterances_per_s": tot[0] / (g_ll + g_vit),
           "decode_accuracy": tot[2] / group.world,
           "roofline": ({"kernel": "loglik", "bound": "hbm", "achieved": bytes_ll / t_ll / 1e9, "peak": PEAK_HBM / 1e9,
                         "unit": "GB/s", "frac": bytes_ll / t_ll / PEAK_HBM} if hbm_bound else
                        {"kernel": "loglik", "bound": "mfma", "achieved": flops / t_ll / 1e12, "peak": peak_flops / 1e12,
                         "unit": "TFLOP/s", "frac": flops / t_ll / peak_flops}),
           "viterbi_roofline": {"kernel": "viterbi_chain", "bound": "hbm", "achieved": bytes_vit / t_vit / 1e9,
                                "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": bytes_vit / t_vit / PEAK_HBM},
           "timing": "wall time of the synchronous C-ABI calls (includes their host side), mean of 3 after a 0.3 s ramp; "
                     "rooflines are rank 0's, times the slowest rank's"}
    if M == 1:
        # configs[0]'s own path: HMM.evaluate of a single-Gaussian model scores a cell while it fills the cost matrix
        # (hmm.py:133-134: dtw + mahalanobis) -- gh_viterbi_fused, no [N, S] matrix.  `ms` / `value` are the fused sweep's
        # (the reference's semantics for these models: log-domain distance); the two-kernel form stays beside it.
        t_f, rf = _timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False, fused_gmm=gmm, log_domain=True))
        assert ctx.last_fused, "gh_viterbi_fused fell back to two kernels"
        t_fg, rg = _timeit(lambda: lat.viterbi(b, want_path=False, want_end_cost=False, fused_gmm=gmm))
        same = bool(np.array_equal(rf["best_end"], words) and np.array_equal(rg["best_end"], words))
        tf = group.maxv([t_f, t_fg])
        g_f, g_fg = float(tf[0]), float(tf[1])
        bytes_f = float((esz * D + 4) * N)                      # features in once + 4 B of result per frame
        flops_f = 4.0 * D * S * N                               # two fma per (state, dimension)
        out.update({
            "two_kernel_ms": out["ms"], "two_kernel_value": out["value"], "ms": g_f * 1e3, "fused_ms": g_f * 1e3,
            "fused_gmm_evaluate_ms": g_fg * 1e3, "value": tot[1] * S / g_f, "utterances_per_s": tot[0] / g_f,
            "fused_words_equal_two_kernel_words": same,
            "fused_roofline": {"kernel": "viterbi_fused", "bound": "valu", "achieved": flops_f / t_f / 1e12, "peak": peak_flops / 1e12,
                               "unit": "TFLOP/s", "frac": flops_f / t_f / peak_flops,
                               "hbm_achieved": bytes_f / t_f / 1e9, "hbm_frac": bytes_f / t_f / PEAK_HBM,
                               "bytes_per_frame": esz * D + 4, "flop_per_frame": 4 * D * S,
                               "traffic": profiled_fused_traffic(N, D, esz),
                               "note": "the sweep is bound by the vector pipe (2 D fma + 7 recurrence instructions per cell "
                                       "column, 50 of 64 lanes), not by HBM; wall time of the synchronous call"}})
    lat.close(); b.close(); gmm.close()
    return out


def fp32_mismatch_c2(ctx, wl):
    """configs[1] at bench size: how often the isolated-word decode from FP32 likelihoods (fp64 DP) differs from the fp64
    decode -- the recognised word, and the Viterbi state path through the whole stacked graph (best end)."""
    from sr.recognition import _hip
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    lat = _hip.Lattices(ctx, [stacked_graph(W, n, wl["trans"])])
    res = {}
    for dt in (np.float64, np.float32):
        b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"], dtype=dt)
        b.loglik(gmm, fetch=False)
        r = lat.viterbi(b, want_path=True)
        res[dt] = (np.argmin(r["end_cost_flat"].reshape(b.U, W), axis=1), r["paths"])
        b.close()
    lat.close(); gmm.close()
    (w64, p64), (w32, p32) = res[np.float64], res[np.float32]
    return {"utterances": int(len(w64)), "fp32_word_mismatch_rate": float(np.mean(w64 != w32)),
            "fp32_path_mismatch_rate": float(np.mean([not np.array_equal(x, y) for x, y in zip(p64, p32)]))}


def _lattice_roofline(key, bytes_dp, t_dec, N, K, W, n):
    """The bound that binds.  K-layer lattice (viterbi_layers_kernel): 114 vector instructions per column and wave for all
    K x W x n cell slots at ~4 cycles each on one of 1024 SIMDs -- the sweep is VALU-issue bound, its HBM
    fraction says how far the likelihood stream is from mattering.  Loop grammar (viterbi_loop_kernel): four utterances
    per wave, the sweep runs at the rate HBM delivers the likelihood matrix."""
    hbm = {"hbm_achieved": bytes_dp / t_dec / 1e9, "hbm_frac": bytes_dp / t_dec / PEAK_HBM}
    if key == "C5_K7_lattice":
        issue_s = 114.0 * 4.0 * N / (1024 * 2.4e9)        # instructions x cycles x columns / (SIMDs x clock)
        return dict({"kernel": "viterbi_layers_kernel + lattice_backtrace_kernel", "bound": "valu", "achieved": issue_s / t_dec,
                     "peak": 1.0, "unit": "fraction of the VALU issue rate", "frac": issue_s / t_dec,
                     "instructions_per_column": 114,
                     "note": "rank 0: 114 VALU instructions x 4 cycles per column and wave over 1024 SIMDs at 2.4 GHz against the "
                             "wall time of gh_viterbi_labels (kernels + label copy-back + host slicing)"}, **hbm)
    return dict({"kernel": "viterbi_loop_kernel + lattice_backtrace_kernel", "bound": "hbm", "achieved": bytes_dp / t_dec / 1e9,
                 "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": bytes_dp / t_dec / PEAK_HBM,
                 "note": "rank 0: algorithmic bytes (esz*S + 4) per frame over the wall time of gh_viterbi_labels "
                         "(kernels + label copy-back + host slicing)"}, **hbm)


def lse_f32exp_leg(dev, wl, c5_distinct=2000, K=7):
    """The fp64 likelihood kernel with the exponentials of its log-sum-exp in FP32 (gh_ctx_set_compat bit 1; OFF by
    default) next to the all-fp64 epilogue: kernel time from HIP events on the launch stream and the fraction of the fp64
    matrix peak both ways, the largest |delta nll| over configs[1]'s 50 M likelihoods, and how many decodes change: the
    recognised word and the whole state path of every configs[1] utterance, the state paths of `c5_distinct` distinct
    configs[4] utterances through the K-layer lattice and the loop grammar (the tiled copies decode like the originals)."""
    from sr.recognition import _hip
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    ctx = _hip.Context(dev)          # a context of its own: the switch is a property of the context
    gmm = _hip.PackedGMM(ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=wl["off"])
    lat = _hip.Lattices(ctx, [stacked_graph(W, n, wl["trans"])])

    def kernel_ms(reps=20, ramp=0.4):
        t_r = time.perf_counter()
        while time.perf_counter() - t_r < ramp:
            b.loglik(gmm, fetch=False); ctx.sync()      # (launches are asynchronous: without the sync the ramp queues seconds of work)
        ctx.sync()
        e0, e1 = ctx.new_event(), ctx.new_event()
        ctx.record(e0)
        for _ in range(reps):
            b.loglik(gmm, fetch=False)
        ctx.record(e1)
        ctx.sync()
        return ctx.elapsed_ms(e0, e1) / reps

    res = {}
    for name, fe in (("fp64", False), ("f32exp", True)):
        ctx.set_compat(underflow=True, lse_f32=fe)
        ms = kernel_ms()
        nll = b.loglik(gmm)
        r = lat.viterbi(b, want_path=True)
        res[name] = (ms, nll.copy(), r["best_end"].copy(), r["paths"])
    flops = 2.0 * 2 * D * S * M * b.N
    d = np.abs(res["fp64"][1] - res["f32exp"][1])
    out = {"kernel": "loglik_mfma_kernel<double, 20, 8>", "kernel_ms_fp64_epilogue": res["fp64"][0], "kernel_ms_f32exp": res["f32exp"][0],
           "frac_fp64_epilogue": flops / (res["fp64"][0] * 1e-3) / PEAK_F64, "frac_f32exp": flops / (res["f32exp"][0] * 1e-3) / PEAK_F64,
           "max_abs_delta_nll": float(d.max()), "max_rel_delta_nll": float((d / np.abs(res["fp64"][1])).max()),
           "C2_utterances": int(b.U), "C2_word_mismatch_rate": float(np.mean(res["fp64"][2] != res["f32exp"][2])),
           "C2_path_mismatch_rate": float(np.mean([not np.array_equal(x, y) for x, y in zip(res["fp64"][3], res["f32exp"][3])])),
           "default": "off (gh_ctx_set_compat bit 1 / GMMHMM_LSE=f32exp turns it on)"}
    del res
    b.close(); lat.close(); gmm.close()
    # configs[4]: distinct K-word utterances of the configs[4] model
    rng = np.random.default_rng(1005)
    wl5 = synth_workload(1005, 1, W=W, n=n, M=M, D=D)
    means, vars_, trans = wl5["means"], wl5["vars"], wl5["trans"]
    words = rng.integers(0, W, size=(c5_distinct, K))
    Tw = rng.integers(30, 61, size=(c5_distinct, K))
    seg_len = Tw.reshape(-1)
    seg_off = np.concatenate([[0], np.cumsum(seg_len)])
    seg = np.repeat(np.arange(len(seg_len)), seg_len)
    t = np.arange(int(seg_off[-1])) - seg_off[seg]
    st = np.minimum(t * n // seg_len[seg], n - 1)
    idx = (words.reshape(-1)[seg] * n + st) * M + rng.integers(0, M, size=len(seg))
    X = means.reshape(-1, D)[idx] + np.sqrt(vars_).reshape(-1, D)[idx] * rng.standard_normal((len(seg), D))
    off = np.concatenate([[0], np.cumsum(Tw.sum(axis=1))]).astype(np.int64)
    gmm5 = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl5["w"].reshape(S, M))
    b5 = _hip.Batch(ctx, feats=X, offsets=off)
    for key, graph in (("C5_K7_lattice", packed_lattice([trans] * W, n, [list(range(W))] * K)[0]),
                       ("C5_loop_grammar", packed_loop_lattice([trans] * W, n)[0])):
        lat5 = _hip.Lattices(ctx, [graph])
        got = {}
        for name, fe in (("fp64", False), ("f32exp", True)):
            ctx.set_compat(underflow=True, lse_f32=fe)
            b5.loglik(gmm5, fetch=False)
            got[name] = lat5.viterbi(b5, want_path=True)["paths"]
        out[key + "_path_mismatch_rate"] = float(np.mean([not np.array_equal(x, y) for x, y in zip(got["fp64"], got["f32exp"])]))
        lat5.close()
    out["C5_distinct_utterances"] = int(c5_distinct)
    b5.close(); gmm5.close(); ctx.close()
    return out


def _continuous_config(ctx, group, U_total, U_base, npdt, K=7, W=10, n=5, M=8, D=39):
    """configs[4] at its per-GPU size (1 M utterances / 8 GPUs): `U_base` distinct synthetic K-word utterances per
    rank, tiled on the device to `U_total` (synthesising 39 M distinct frames costs ~12 GB of host memory and ~1 min
    of numpy per rank -- x8 ranks on one host; the decode does not care that utterances repeat); K-layer lattice (the
    reference's grammar, main.py:35) and the word-loop grammar, decoded to label sequences on the device
    (gh_viterbi_labels).  Every rank runs its own share (main.py:60 shards by utterance: no data-path collective);
    reported: all ranks' utterances / the slowest rank's time.  Also: how often a decode from FP32 likelihoods
    differs from the fp64 decode (paths and label sequences, on the distinct utterances)."""
    from sr.recognition import _hip
    from sr.recognition.continuous_speech import packed_lattice, packed_loop_lattice
    rng = np.random.default_rng(1005 + 7919 * group.rank)
    wl = synth_workload(1005, 1, W=W, n=n, M=M, D=D)
    means, vars_, trans = wl["means"], wl["vars"], wl["trans"]
    S = W * n
    words = rng.integers(0, W, size=(U_base, K))
    Tw = rng.integers(30, 61, size=(U_base, K))
    seg_len = Tw.reshape(-1)
    seg_off = np.concatenate([[0], np.cumsum(seg_len)])
    Nb = int(seg_off[-1])
    seg = np.repeat(np.arange(len(seg_len)), seg_len)
    t = np.arange(Nb) - seg_off[seg]
    st = np.minimum(t * n // seg_len[seg], n - 1)
    idx = (words.reshape(-1)[seg] * n + st) * M + rng.integers(0, M, size=Nb)
    X = means.reshape(-1, D)[idx] + np.sqrt(vars_).reshape(-1, D)[idx] * rng.standard_normal((Nb, D))
    T = Tw.sum(axis=1)
    reps = max(1, int(round(U_total / U_base)))
    U = U_base * reps
    off_b = np.concatenate([[0], np.cumsum(T)]).astype(np.int64)
    base = _hip.Batch(ctx, feats=np.ascontiguousarray(X, dtype=npdt), offsets=off_b, dtype=npdt)
    b = base.tile(reps)
    # every copy becomes an utterance of its own: independent noise of a quarter of a standard deviation per feature,
    # generated on the device (gh_batch_jitter) -- synthesising 39 M distinct frames on the host costs ~12 GB and ~1 min
    # per rank; copy 0 stays what the host made, so the distinct-utterance checks below still see `base`
    jitter = 0.25
    if reps > 1 and hasattr(b, "jitter"):
        b.jitter(seed=1005 + 7919 * group.rank, scale=jitter)
    else:
        jitter = 0.0
    N = b.N
    gmm = _hip.PackedGMM(ctx, means.reshape(S, M, D), vars_.reshape(S, M, D), wl["w"].reshape(S, M))
    group.barrier()
    t_ll, _ = _timeit(lambda: (b.loglik(gmm, fetch=False), ctx.sync()), reps=3, ramp=0.2)
    esz = np.dtype(npdt).itemsize
    out = {}
    truth = [list(w) for w in words]
    other = np.float32 if npdt == np.float64 else np.float64     # the decode from the OTHER precision's likelihoods
    base_o = _hip.Batch(ctx, feats=np.ascontiguousarray(X, dtype=other), offsets=off_b, dtype=other)
    base.loglik(gmm, fetch=False)
    base_o.loglik(gmm, fetch=False)
    for key, (graph, _nes), max_labels in (
            ("C5_K7_lattice", packed_lattice([trans] * W, n, [list(range(W))] * K), K + 1),
            ("C5_loop_grammar", packed_loop_lattice([trans] * W, n), None)):
        lat = _hip.Lattices(ctx, [graph])
        R = len(graph["row_state"])
        row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
        ml = max_labels if max_labels is not None else b.lengths // (n - 1) + 2
        group.barrier()
        t_dec, r = _timeit(lambda: lat.viterbi_labels(b, row_word, max_labels=ml, as_lists=False, want_end_cost=False), reps=5, ramp=0.25)   # (the first calls grow the scratch arenas)
        lf, lo, ln = r["labels_flat"], r["label_off"], r["n_labels"]
        acc = float(np.mean([[int(v) for v in lf[lo[u]:lo[u] + ln[u]]] == truth[u % U_base] for u in range(0, U, max(1, U // 4000))]))
        # fp32-likelihood decode against the fp64 one, on the distinct utterances: state paths and label sequences
        pa = lat.viterbi(base, want_path=True)["paths"]
        pb = lat.viterbi(base_o, want_path=True)["paths"]
        path_mis = float(np.mean([not np.array_equal(x, y) for x, y in zip(pa, pb)]))
        mlb = max_labels if max_labels is not None else base.lengths // (n - 1) + 2
        la = lat.viterbi_labels(base, row_word, max_labels=mlb)["labels"]
        lb = lat.viterbi_labels(base_o, row_word, max_labels=mlb)["labels"]
        lab_mis = float(np.mean([not np.array_equal(x, y) for x, y in zip(la, lb)]))
        del pa, pb
        bytes_dp = float((esz * S + 4) * N)     # un-fused Viterbi over materialised likelihoods
        # all ranks: utterances / slowest rank (each leg's times are max'ed over the ranks)
        tm = group.maxv([t_ll, t_dec])
        tot = group.sum([float(U), float(N), acc, path_mis, lab_mis])
        g_ll, g_dec = float(tm[0]), float(tm[1])
        out[key] = {"workload": "configs[4] per-GPU share: %d utterances (%d synthesised on the host, tiled x%d on the device%s), "
                                "K=%d words, %d lattice rows"
                                % (U, U_base, reps, ", every copy with its own N(0, %.2f^2) noise per feature" % jitter if jitter else "", K, R),
                    "distinct_utterances": int(U if jitter else U_base),
                    "n_gpus": group.world, "utterances": int(tot[0]), "frames": int(tot[1]), "lattice_rows": R,
                    "ms": (g_ll + g_dec) * 1e3, "loglik_ms": g_ll * 1e3, "viterbi_labels_ms": g_dec * 1e3,
                    "value": tot[0] / (g_ll + g_dec), "unit": "utterances/s", "dp_cells_per_s": tot[1] * R / g_dec,
                    "sequence_accuracy_sampled": tot[2] / group.world,
                    "fp32_path_mismatch_rate": tot[3] / group.world, "fp32_label_mismatch_rate": tot[4] / group.world,
                    "fp32_mismatch_sample": "%d distinct utterances per rank, fp32 vs fp64 likelihoods, fp64 DP" % U_base,
                    "roofline": _lattice_roofline(key, bytes_dp, t_dec, N, K, W, n)}
        lat.close()
    b.close(); base_o.close()
    # ---- the same decode PIPELINED: the batch in 4 pieces over two contexts (HIP stream + host thread each), so that a
    # piece's dynamic program (HBM / VALU bound) runs under the next piece's likelihoods (matrix-core bound); the
    # likelihood kernels themselves are chained by events (each fills the GPU: overlapping two gains nothing)
    try:
        pipe = _pipelined_decode(ctx.device, base, reps, gmm_arrays=(means.reshape(S, M, D), vars_.reshape(S, M, D), wl["w"].reshape(S, M)),
                                 graphs={"C5_K7_lattice": (packed_lattice([trans] * W, n, [list(range(W))] * K)[0], K + 1),
                                         "C5_loop_grammar": (packed_loop_lattice([trans] * W, n)[0], None)}, n=n, npdt=npdt, group=group)
        for key, r in pipe.items():
            out[key].update(r)
    except Exception as e:
        if group.world > 1:
            raise
        out["C5_pipelined_error"] = repr(e)[:300]
    base.close(); gmm.close()
    return out


def _pipelined_decode(dev, base, reps, gmm_arrays, graphs, n, npdt, group, pieces=4, lanes=2):
    import threading
    from sr.recognition import _hip
    per = max(1, reps // pieces)
    ctxs = [_hip.Context(dev) for _ in range(lanes)]
    host = base.features()
    off = base.offsets
    X = np.ascontiguousarray(np.concatenate(host)) if len(host) else np.zeros((0, base.D), dtype=npdt)
    state = []
    for l, c in enumerate(ctxs):
        small = _hip.Batch(c, feats=X, offsets=off, dtype=npdt)
        state.append(dict(ctx=c, gmm=_hip.PackedGMM(c, *gmm_arrays), batches=[small.tile(per) for _ in range(pieces // lanes)], small=small))
    U = sum(bb.U for st in state for bb in st["batches"])
    out = {}
    for key, (graph, max_labels) in graphs.items():
        row_word = np.where(graph["row_state"] >= 0, graph["row_state"] // n, -1).astype(np.int32)
        for st in state:
            st["lat"] = _hip.Lattices(st["ctx"], [graph])
        chain_lock, chain = threading.Lock(), {"last": None}

        def lane(st):
            for bb in st["batches"]:
                with chain_lock:
                    if chain["last"] is not None:
                        st["ctx"].wait_event(chain["last"])
                    bb.loglik(st["gmm"], fetch=False)
                    ev = st["ctx"].new_event()
                    st["ctx"].record(ev)
                    chain["last"] = ev
                ml = max_labels if max_labels is not None else bb.lengths // (n - 1) + 2
                st["lat"].viterbi_labels(bb, row_word, max_labels=ml, as_lists=False, want_end_cost=False)

        def step():
            chain["last"] = None
            th = [threading.Thread(target=lane, args=(st,)) for st in state]
            for t in th:
                t.start()
            for t in th:
                t.join()
        group.barrier()
        t_pipe, _ = _timeit(step, reps=4, ramp=0.3)
        g_t = group.max(t_pipe)
        tot = group.sum([float(U)])
        out[key] = {"pipelined_ms": g_t * 1e3, "pipelined_value": float(tot[0]) / g_t, "pipelined_utterances": int(tot[0]),
                    "pipelined_how": "%d pieces over %d contexts: a piece's DP under the next piece's likelihoods" % (pieces, lanes)}
        for st in state:
            st["lat"].close()
    for st in state:
        for bb in st["batches"]:
            bb.close()
        st["small"].close(); st["gmm"].close(); st["ctx"].close()
    return out


def _training_config(ctx, U, K=7):
    """configs[2]'s model on K-word transcripts (the reference's own training data are digit strings,
    continuous_speech.py:56-179): (a) the alignment + regrouping step of continuous_train -- own-state likelihoods,
    forced-alignment Viterbi through one graph per distinct transcript, frames regrouped per state
    (gh_lattices_create_transcripts + gh_loglik_sets + gh_align_runs); (b) one soft-EM iteration on the same
    transcripts: the device-resident session over word strings (gh_em_create_transcripts -- likelihoods of the
    transcripts' words, sequence-form forward-backward with lane = cell, statistics, M-step and re-pack on one stream;
    round 3 ran it call by call with a host M-step and graphs rebuilt from the new costs: 5.6 ms)."""
    from sr.recognition import _hip
    from sr.recognition.train import BaumWelchTrainer
    wl = synth_workload(1003, U * K)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    off = wl["off"][::K]
    labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
    data = [wl["X"][off[u]:off[u + 1]] for u in range(U)]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
    gmm = _hip.PackedGMM(ctx, means0.reshape(W * n, M, D), wl["vars"].reshape(W * n, M, D), wl["w"].reshape(W * n, M))
    b = _hip.Batch(ctx, feats=wl["X"], offsets=off)
    keys, utt_graph = {}, np.empty(U, dtype=np.int32)
    for u, l in enumerate(labels):
        utt_graph[u] = keys.setdefault(tuple(l), len(keys))
    from sr.recognition.continuous_speech import transcript_state_sets
    sets = transcript_state_sets(labels, n, W)

    def align():
        lat = _hip.Lattices.from_transcripts(ctx, [wl["trans"]] * W, n, list(keys))
        b.loglik(gmm, fetch=False, state_sets=sets)
        r = lat.align_runs(b, utt_lattice=utt_graph)       # the runs of every best path (state, first row, rows): ~N / 20 of them
        lat.close()
        return r
    t_align, r = _timeit(align)
    assigned = float(r["length"].sum()) / float(b.N)
    b.close(); gmm.close()
    tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels, device=ctx.device)
    hist = [tr.iteration()]
    t_em, _ = _timeit(lambda: hist.append(tr.iteration()), reps=3)
    resident = tr.session is not None and getattr(tr.session, "word_strings", False)
    phases = None
    if resident:      # HIP events between the phases of an iteration, on the stream its kernels run on (gh_em_profile)
        tr.session.profile(True)
        ph = []
        for _ in range(3):
            hist.append(tr.iteration())
            ph.append(tr.session.phase_ms())
        tr.session.profile(False)
        ph = np.median(np.array(ph), axis=0)
        phases = {"loglik_states_of_the_transcripts_words_ms": float(ph[0]), "fb_seq_cell_plus_ranges_ms": float(ph[1]),
                  "bw_fused_statistics_ms": float(ph[2]), "tail_allreduce_mstep_repack_ms": float(ph[3])}
    tr.close()
    N = int(off[-1])
    return {"workload": "configs[2] model, %d utterances of %d words (%d frames, %d distinct transcripts)" % (U, K, N, len(keys)),
            "align_and_regroup": {"ms_per_call": t_align * 1e3, "utterances_per_s": U / t_align, "frames_per_s": N / t_align,
                                  "frames_assigned_to_a_state": assigned, "runs": int(len(r["state"]))},
            "soft_em_iteration": {"ms_per_iteration": t_em * 1e3, "utterances_per_s": U / t_em, "frames_per_s": N / t_em,
                                  "device_resident_iteration": bool(resident), "phases": phases,
                                  "loglik_monotone": bool(all(y >= x - 1e-7 * abs(x) for x, y in zip(hist, hist[1:])))}}


def _c4_em_config(ctx, group, U):
    """Soft-EM iterations at the configs[3] shape (64 words x 16 states x 32 mixtures).  Round 3: the device-resident
    session covers it -- own-state likelihoods (32 Gaussian tiles per 32-frame block instead of 2048), chain
    forward-backward with 16 lanes per utterance, the matrix-core statistics kernel with one wave per 16 components of a
    state (normalised by the likelihood kernel's per-state sums), the 20.7 MB statistics buffer all-reduced on the
    stream by the library's communicator, M-step and model re-pack on the device.  (Round 2 and early round 3: generic
    kernels, call by call, M-step in numpy: 36.9 ms per iteration for 1 000 utterances.)"""
    from sr.recognition.train import BaumWelchTrainer
    wl = synth_workload(1004, U, W=64, n=16, M=32, D=39, utt_seed=None if group.rank == 0 else 1004 + 7919 * group.rank)
    W = wl["W"]
    data = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U)]
    labels = [[int(w)] for w in wl["words"]]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
    red = group.native if group.kind == "native" else None
    if group.kind == "torch":
        from sr.recognition.parallel import StatsAllReducer
        red = StatsAllReducer(gpu_index=group.dev)
    kw = {"occ_floor": float(os.environ["EM_OCC_FLOOR"])} if "EM_OCC_FLOOR" in os.environ else {}    # (experiments: tools/time_c4_em.py)
    tr = BaumWelchTrainer(means0, wl["vars"], wl["w"], [wl["trans"]] * W, data, labels, device=ctx.device, reducer=red, **kw)
    del data
    hist = [tr.iteration()]
    iters = 4
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        hist.append(tr.iteration())
    dt = group.max((time.perf_counter() - t0) / iters)
    tot = group.sum([float(U), float(tr.batch.N)])
    resident = tr.session is not None
    out = {"workload": "configs[3] shape, soft EM: 64 x 16 states x 32 mixtures, 39-dim, %d utterances per GPU (%s)"
                       % (U, "device-resident iteration" if resident else "generic kernels, call by call"),
           "n_gpus": group.world, "utterances": int(tot[0]), "frames": int(tot[1]), "ms_per_iteration": dt * 1e3,
           "em_frames_per_s": tot[1] / dt, "em_utterances_per_s": tot[0] / dt, "device_resident_iteration": resident,
           "allreduce_bytes": tr._packed_len() * 8,
           "loglik_per_frame": [h / tot[1] for h in hist],
           "loglik_monotone": bool(all(y >= x - 1e-9 * abs(x) for x, y in zip(hist, hist[1:])))}
    tr.close()
    return out


def _ctrain_config(ctx, U, K=7, iters=8):
    """The reference's ACTUAL training algorithm on configs[2]'s model: continuous_train (continuous_speech.py:56-179) --
    per outer iteration forced alignment of every utterance, frames regrouped per state, every state refit by
    binary-split k-means + EM, transition costs re-estimated, models pickled -- on `U` synthetic K-word utterances.
    Alignment, regrouping gather, partition variances, lock-step k-means / EM with their stop rules all run on the
    device (gh_align_segments, gh_batch_gather, gh_fit_*); reported: wall time per outer iteration in steady state."""
    import contextlib
    import io
    import tempfile
    from sr.recognition import continuous_speech as cs
    from sr.recognition.model_io import models_from_arrays
    wl = synth_workload(1003, U * K)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    iso = [wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in range(U * K)]
    data = [np.concatenate(iso[i * K:(i + 1) * K]) for i in range(U)]
    labels = [[int(w) for w in wl["words"][i * K:(i + 1) * K]] for i in range(U)]
    means0 = wl["means"] + 0.3 * np.random.default_rng(0).normal(size=wl["means"].shape)
    models = models_from_arrays(means0, wl["vars"], wl["w"], [wl["trans"]] * W, mu=means0[:, :, 0], sigma=wl["vars"][:, :, 0])

    class Stamps(io.StringIO):
        def __init__(self):
            super().__init__()
            self.t = []

        def write(self, text):
            if text.startswith("Continuous training iteration:"):
                self.t.append(time.perf_counter())
            return len(text)
    np.random.seed(0)
    stamps = Stamps()
    import warnings
    with contextlib.redirect_stdout(stamps), warnings.catch_warnings(), tempfile.TemporaryDirectory() as out:
        warnings.simplefilter("ignore")
        cs.continuous_train(data, models, labels, out, n_gaussians=M, n_segments=n, max_iteration=iters)
    per = np.diff(stamps.t + [time.perf_counter()])
    N = int(sum(len(x) for x in data))
    # steady state: the second half WITHOUT the last iteration (its stamp runs until the call has returned: the wait for the
    # pickle writer, the handles, the temporary directory).  The workload of every iteration is the same -- two runs give the
    # same bits and the same launch counts -- so what the median leaves out are host hiccups
    steady = float(np.median(per[len(per) // 2:-1])) if len(per) >= 4 else float(np.median(per[len(per) // 2:]))
    return {"workload": "configs[2] model, continuous_train on %d utterances of %d words (%d frames), %d mixtures" % (U, K, N, M),
            "outer_iterations": int(len(per)), "ms_per_outer_iteration": [round(1e3 * float(x), 1) for x in per],
            "ms_per_outer_iteration_steady": steady * 1e3, "frames_per_s": N / steady, "utterances_per_s": U / steady,
            "note": "steady = median of the second half of the iterations without the last one, which carries the call's teardown (the first ones grow the scratch arenas); round 2: 750 ms"}


def _train_words_config(ctx, W=10, templates=200, n=5, ng=4):
    """Isolated-word training of all word models in one pass (`batch.train_words`: sr/core.py:47-60 trains word after
    word): segmental k-means of all words in lock-step, the W x n mixtures refit in one device-resident session, all
    templates re-aligned in one launch."""
    import contextlib
    import io
    import warnings
    from sr.recognition.batch import train_words
    wl = synth_workload(1006, W * templates, W=W, n=n, M=ng)
    order = np.argsort(wl["words"], kind="stable")
    words = [[wl["X"][wl["off"][u]:wl["off"][u + 1]] for u in order[wl["words"][order] == w]] for w in range(W)]
    N = int(wl["off"][-1])

    def run():
        np.random.seed(0)
        with contextlib.redirect_stdout(io.StringIO()), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return train_words(words, n, n_gaussians=ng)
    run()
    t0 = time.perf_counter()
    models = run()
    dt = time.perf_counter() - t0
    return {"workload": "isolated-word training, %d words x ~%d templates (%d frames, 39-dim), %d states, %d mixtures" % (W, templates, N, n, ng),
            "ms": dt * 1e3, "templates_per_s": W * templates / dt, "frames_per_s": N / dt, "models": len(models)}


def extra_configs(args, group, npdt, peak_flops, wl):
    """The other BASELINE configs, after the timed region.  With several ranks every rank runs its share of the legs
    that BASELINE defines on 8 GPUs (configs[4]: C5 K-layer lattice and loop grammar at 125 000 utterances per rank;
    configs[3]: C4) and rank 0 reports all ranks' units over the slowest rank's time; the single-GPU-only legs (C1 x1000,
    the word-string training step, the fp32-vs-fp64 decode comparison of configs[1]) run when there is one rank."""
    from sr.recognition import _hip
    ctx = _hip.Context(group.dev)
    out = {}
    one = group.world == 1
    c4_utts = args.c4_utts if args.c4_utts else (50000 if one else 10000)
    c4_name = ("configs[3]: 64 HMMs x 16 states x 32 mixtures, 39-dim, %d utterances%s"
               % (c4_utts, "" if c4_utts >= 50000 else " per GPU (reduced from 50 000)"))
    legs = []
    if one:
        legs.append(("C1x1000", lambda: _isolated_config(ctx, group, "configs[0] x1000: 10x5 states, 1 Gaussian, 13-dim, 100 000 utterances",
                                                         1001, 100000, 10, 5, 1, 13, npdt, peak_flops)))
    legs.append(("C4", lambda: _isolated_config(ctx, group, c4_name, 1004, c4_utts, 64, 16, 32, 39, npdt, peak_flops)))
    legs.append(("C4_em", lambda: _c4_em_config(ctx, group, args.c4_em_utts if args.c4_em_utts else c4_utts)))
    if one:
        legs.append(("C3_word_strings", lambda: _training_config(ctx, args.c3_utts)))
        legs.append(("C3_continuous_train", lambda: _ctrain_config(ctx, args.c3_utts)))
        legs.append(("C2_fp32_decode", lambda: fp32_mismatch_c2(ctx, wl)))
        if npdt == np.float64:
            legs.append(("C2_lse_f32exp", lambda: lse_f32exp_leg(ctx.device, wl)))
        legs.append(("C2_train_words", lambda: _train_words_config(ctx)))
    legs.append(("C5", lambda: _continuous_config(ctx, group, args.c5_utts, min(args.c5_utts, 5000), npdt)))
    for key, fn in legs:
        t0 = time.perf_counter()
        try:
            r = fn()
            if key == "C5":
                out.update(r)
            else:
                out[key] = r
        except Exception as e:
            if not one:
                raise            # a rank that drops out of a leg's collectives would hang the others: fail loudly
            out[key] = {"error": repr(e)[:300]}    # a single-rank extra must never take the headline line down with it
        out.setdefault("_seconds", {})[key] = round(time.perf_counter() - t0, 1)
    ctx.close()
    return out


def pcie_inclusive(dev, wl, npdt, steps=6):
    """The headline step with the feature upload INSIDE the step (gh_batch_create from pageable host memory): what a caller
    pays who hands over host buffers every time.  Measured twice: one context (upload, likelihoods, Viterbi one after the
    other) and two contexts on two host threads (one batch's upload travels while the other batch computes -- the link
    stays the bound: 312 MB per step).  Never part of `value`."""
    import threading
    from sr.recognition import _hip
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    X = np.ascontiguousarray(wl["X"], dtype=npdt)

    class Lane:
        def __init__(self):
            self.ctx = _hip.Context(dev)
            self.gmm = _hip.PackedGMM(self.ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D), wl["w"].reshape(S, M))
            self.lat = _hip.Lattices(self.ctx, [stacked_graph(W, n, wl["trans"])])

        def step(self):
            b = _hip.Batch(self.ctx, feats=X, offsets=wl["off"], dtype=npdt)
            b.loglik(self.gmm, fetch=False)
            self.lat.viterbi(b, want_path=False)
            b.close()

        def close(self):
            self.lat.close(); self.gmm.close(); self.ctx.close()

    def timed(n_lanes):
        lanes = [Lane() for _ in range(n_lanes)]
        for l in lanes:
            l.step()

        def work(l):
            for _ in range(steps):
                l.step()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(l,)) for l in lanes]
        for t in th:
            t.start()
        for t in th:
            t.join()
        dt = (time.perf_counter() - t0) / (steps * n_lanes)
        for l in lanes:
            l.close()
        return dt
    dt1, dt2 = timed(1), timed(2)
    # fp32 on the wire, fp64 in HBM and in every kernel (gh_batch_create_wire: half the bytes over the link, widened on the
    # device, host buffer page-locked for the copy); the features are fp32-rounded before the call -- a caller's choice
    wire = {}
    if npdt == np.float64:
        X32 = np.ascontiguousarray(wl["X"], dtype=np.float32)
        step64 = Lane.step

        def step_wire(self):
            b = _hip.Batch(self.ctx, feats=X32, offsets=wl["off"], dtype=np.float64, wire=np.float32, pin="keep")
            b.loglik(self.gmm, fetch=False)
            self.lat.viterbi(b, want_path=False)
            b.close()
        Lane.step = step_wire
        try:
            w1, w2 = timed(1), timed(2)
            wire = {"ms_per_step_f32_wire": w1 * 1e3, "ms_per_step_f32_wire_two_lanes": w2 * 1e3, "h2d_bytes_per_step_f32_wire": int(X32.nbytes),
                    "h2d_GBps_f32_wire_two_lanes": X32.nbytes / w2 / 1e9}
        finally:
            Lane.step = step64
            _hip.host_unpin(X32)
    return {**wire, "ms_per_step": dt1 * 1e3, "value": X.shape[0] * S / dt1, "unit": "frame-state loglik/s",
            "ms_per_step_two_lanes": dt2 * 1e3, "value_two_lanes": X.shape[0] * S / dt2,
            "h2d_bytes_per_step": int(X.nbytes), "h2d_GBps_two_lanes": X.nbytes / dt2 / 1e9,
            "note": "upload + likelihoods + Viterbi per step from pageable host memory; one context / two contexts on two host threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--inflight", type=int, default=2,
                    help="batches in flight per GPU (one HIP stream + one host thread each): the host side of a step "
                         "overlaps the other batch's kernels")
    ap.add_argument("--ramp-seconds", type=float, default=1.0,
                    help="untimed pre-warmup: keep the GPU busy with the same step this long so that its clock governor "
                         "reaches the sustained frequency (a cold MI355X runs ~1.9 GHz for the first few hundred ms, "
                         "2.4 GHz afterwards: tools/wave_timeline.py); not part of --warmup / --steps")
    ap.add_argument("--dtype", choices=["f64", "f32"], default="f64")
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-all-cores", action="store_true", help="skip the all-cores CPU leg (multiprocessing over utterances, ~10 s)")
    ap.add_argument("--cpu-utts", type=int, default=80)
    # rehearsal of the N > 1 path on a box with fewer GPUs: --backend gloo --device 0 lets every rank share GPU 0
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--comm", default=None, choices=["native", "torch"],
                    help="process group of an N-rank run: native = the library's own RCCL communicator (gh_comm_*; default "
                         "with --backend nccl), torch = torch.distributed (default with --backend gloo)")
    ap.add_argument("--same-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank on GPU 0, native RCCL over the socket transport (one "
                         "NCCL_HOSTID per rank); only when bench.py starts its own ranks")
    ap.add_argument("--no-pin", action="store_true", help="do not pin the rank to the cores of its GPU's NUMA node")
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (default: LOCAL_RANK)")
    ap.add_argument("--no-em", action="store_true", help="skip the EM leg (configs[2])")
    ap.add_argument("--em-utts", type=int, default=12500, help="utterances per GPU of the EM leg (100k / 8 GPUs)")
    ap.add_argument("--em-iters", type=int, default=5)
    ap.add_argument("--em-total", type=int, default=0,
                    help="strong scaling of the EM leg: this many utterances in total, sharded over the ranks by frames "
                         "(configs[2]: 100000); default 0 = --em-utts per GPU (weak)")
    ap.add_argument("--c4-em-utts", type=int, default=0, help="utterances per GPU of the C4-shape EM leg (default: those of the C4 leg)")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip C1 x1000 / C4 / C5 / PCIe legs (single-GPU runs only)")
    ap.add_argument("--c4-utts", type=int, default=0, help="utterances of the C4 leg (default: 50 000 = configs[3] on one "
                                                         "GPU, 10 000 per GPU in an N-rank run)")
    ap.add_argument("--c5-utts", type=int, default=125000, help="utterances of the C5 legs (1 M / 8 GPUs)")
    ap.add_argument("--c3-utts", type=int, default=2000, help="7-word utterances of the training-step leg (C3_word_strings)")
    args = ap.parse_args()

    if args.comm is None:
        args.comm = "native" if args.backend == "nccl" else "torch"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], same_gpu=args.same_gpu))      # nothing above this line touches HIP or torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d (launch one rank per GPU: torch.distributed.run "
                 "--nproc-per-node %d, or start bench.py --gpus %d without a launcher)" % (args.gpus, world, args.gpus, args.gpus))
    dev = (0 if args.same_gpu else local_rank) if args.device is None else args.device
    if args.same_gpu and world > 1:
        # (also under a launcher: RCCL reads these when the communicator comes up, not when the process starts)
        os.environ.update(NCCL_HOSTID="gmmhmm-bench-%d" % rank, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                          NCCL_SHM_DISABLE="1")
        # the refit's TAIL launches keep workgroups waiting for the rest of their grid: that needs the card to this process
        # alone (the occupancy check cannot see the other ranks' kernels), so the rehearsal on a shared card does without
        os.environ.setdefault("GMMHMM_REFIT_TAIL", "0")
    all_cores = None
    if world == 1 and not args.no_cpu_baseline and not args.no_all_cores:
        try:      # (before the GPU is touched and before this process pins itself: see cpu_all_cores)
            all_cores = cpu_all_cores(synth_workload(1002, args.utts))
        except Exception as e:
            all_cores = {"error": repr(e)[:200]}
    # host cores: before anything touches the GPU (and before numpy / the ranks' host threads start working)
    pinned = None if args.no_pin else pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))), dev)
    from sr.recognition import _hip
    group = Group(args, rank, world, dev)
    npdt = np.float64 if args.dtype == "f64" else np.float32
    wl = synth_workload(1002, args.utts, utt_seed=None if rank == 0 else 1002 + 7919 * rank)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n

    class Lane:
        """One in-flight batch: its own context (HIP stream, scratch, pinned buffers) and resident copies of the
        model, the batch and the decoding graph.  `--inflight` lanes are driven by one host thread each, so the
        host side of a step (result copy-back, arg-min, Python) overlaps the kernels of the other lane."""

        def __init__(self):
            self.ctx = _hip.Context(dev)
            self.gmm = _hip.PackedGMM(self.ctx, wl["means"].reshape(S, M, D), wl["vars"].reshape(S, M, D),
                                      wl["w"].reshape(S, M))
            self.batch = _hip.Batch(self.ctx, feats=wl["X"], offsets=wl["off"], dtype=npdt)
            self.lat = _hip.Lattices(self.ctx, [stacked_graph(W, n, wl["trans"])])
            self.ll_ms = []
            self.decoded = None
            self.untimed_done = self.ctx.new_event()

        def step(self, timed):
            """One pass of the hot path over the batch: likelihoods, Viterbi over every word model, arg-min.
            The likelihood kernels of different lanes are chained by events (each fills the whole GPU, so
            overlapping two of them gains nothing and would blur the per-launch timing)."""
            with chain_lock:
                if chain["last"] is not None:
                    self.ctx.wait_event(chain["last"])
                if timed:
                    a, b = self.ctx.new_event(), self.ctx.new_event()
                    self.ctx.record(a)
                else:
                    b = self.untimed_done
                self.batch.loglik(self.gmm, fetch=False)
                self.ctx.record(b)
                chain["last"] = b
                if timed:
                    self.ll_ms.append((a, b))
            r = self.lat.viterbi(self.batch, want_path=False)
            ec = r["end_cost_flat"].reshape(self.batch.U, W)
            self.decoded = np.argmin(ec, axis=1)

    import threading
    chain_lock, chain = threading.Lock(), {"last": None}
    lanes = [Lane() for _ in range(max(1, args.inflight))]
    N_frames, U = lanes[0].batch.N, lanes[0].batch.U

    def run_steps(count, timed, seconds=None):
        """`count` steps in total (or as many as fit `seconds`), shared by the lanes' host threads."""
        if len(lanes) == 1:
            t0, k = time.perf_counter(), 0
            while (k < count) if seconds is None else (time.perf_counter() - t0 < seconds):
                lanes[0].step(timed)
                k += 1
            return
        lock, state = threading.Lock(), {"next": 0, "err": None}
        t_end = None if seconds is None else time.perf_counter() + seconds

        def worker(lane):
            try:
                while True:
                    with lock:
                        k = state["next"]
                        if (k >= count) if t_end is None else (time.perf_counter() >= t_end):
                            return
                        state["next"] = k + 1
                    lane.step(timed)
            except BaseException as e:  # surface a worker failure in the main thread
                state["err"] = e
        th = [threading.Thread(target=worker, args=(l,)) for l in lanes]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if state["err"] is not None:
            raise state["err"]

    def fence():
        """Barrier + device-wide synchronisation on both sides of the timed region: every lane's stream, then
        hipDeviceSynchronize through the library's runtime (all contexts of this GPU; plus torch.cuda.synchronize when the
        group is torch's), then the barrier over the ranks (RCCL all-reduce + stream sync, or dist.barrier), then again."""
        for l in lanes:
            l.ctx.sync()
        group.sync_device()
        group.barrier()
        for l in lanes:
            l.ctx.sync()
        group.sync_device()

    run_steps(0, False, seconds=args.ramp_seconds)   # clock ramp, untimed (see --ramp-seconds)
    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    dt = time.perf_counter() - t0
    decoded = next(l.decoded for l in lanes if l.decoded is not None)
    ll_times = [lanes[0].ctx.elapsed_ms(a, b) for l in lanes for a, b in l.ll_ms]   # HIP events, the library's own runtime
    dt = group.max(dt)                                   # the slowest rank's time
    tot = group.sum([float(N_frames), float(U)])         # all ranks' units
    all_frames, all_utts = float(tot[0]), float(tot[1])
    accuracy = float(np.mean(decoded == wl["words"]))

    if rank == 0:
        ll_avg_s = float(np.mean(ll_times)) * 1e-3
        esz = 8 if args.dtype == "f64" else 4
        flops_per_frame = 2.0 * (2 * D) * S * M          # GEMM-form contraction
        bytes_per_frame = esz * D + esz * S              # features in once + likelihoods out once
        peak = PEAK_F64 if args.dtype == "f64" else PEAK_F32
        ach = flops_per_frame * N_frames / ll_avg_s
        out = {
            "metric": "frame-state loglik/s + utterances/s Viterbi decode",
            "value": all_frames * S * args.steps / dt,
            "unit": "frame-state loglik/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_ramp_s": args.ramp_seconds,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "configs[1]: 10-digit HMM, 5 states/digit, 8-mix GMM, 39-dim, "
                                   "%d utterances/GPU (%d frames on rank 0), isolated-word decode" % (U, N_frames),
                       "states": S, "mixtures": M, "dim": D, "parallelism": "utterance-sharded x%d" % world, "batches_in_flight": len(lanes),
                       "process_group": group.kind, "host_cores": pinned},
            "utterances_per_s": all_utts * args.steps / dt,
            "frames_per_s": all_frames * args.steps / dt,
            "decode_accuracy": accuracy,
            "roofline": {
                "kernel": "loglik_kernel (batched GMM.evaluate)", "bound": "mfma",
                "achieved": ach / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s", "frac": ach / peak,
                "traffic": profiled_traffic(args.dtype, N_frames),
                "kernel_ms": ll_avg_s * 1e3,
                "hbm_achieved_GBps": bytes_per_frame * N_frames / ll_avg_s / 1e9,
                "hbm_frac": bytes_per_frame * N_frames / ll_avg_s / PEAK_HBM,
                "note": "algorithmic flops = 2*(2D)*S*M per frame; peak = dense %s rate" % args.dtype,
                # which log-sum-exp epilogue `value` and `frac` were measured with: "fp64" is what a caller gets by default
                # (likelihoods to 1e-10 of the oracle); "f32exp" (GMMHMM_LSE=f32exp / gh_ctx_set_compat bit 1: the
                # exponentials in fp32, |delta nll| <= 2.6e-7 = 3e-9 relative, no decode changes) is the opt-in fast mode,
                # whose fraction is copied into `frac_f32exp` from the C2_lse_f32exp leg below when that leg runs
                "lse": ("f32exp" if os.environ.get("GMMHMM_LSE") == "f32exp" else "fp64") if args.dtype == "f64" else "f32",
            },
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is a rank-0, single-GPU-run measurement
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_utts)
            if all_cores is not None:
                out["cpu_baseline"]["all_cores"] = all_cores
    # ---- everything below runs after (and outside) the timed region; it never changes value / ms_per_step
    for l in lanes:
        l.lat.close(); l.batch.close(); l.gmm.close(); l.ctx.close()
    lanes.clear()
    em = None
    if not args.no_em:
        try:
            em = em_leg(args, group)
        except Exception as e:
            if world > 1:
                raise             # a rank that drops out of a collective would hang the others: fail loudly
            em = {"error": repr(e)[:300]}
    configs = None
    if not args.no_extra_configs:
        configs = extra_configs(args, group, npdt, PEAK_F64 if args.dtype == "f64" else PEAK_F32, wl)
    if rank == 0:
        if em is not None:
            out["em"] = em
        if configs is not None:
            out["configs"] = configs
            fe = configs.get("C2_lse_f32exp")
            if isinstance(fe, dict) and "frac_f32exp" in fe:
                out["roofline"]["frac_f32exp"] = fe["frac_f32exp"]          # (kernel alone, HIP events; the default mode's
                out["roofline"]["frac_fp64_alone"] = fe["frac_fp64_epilogue"]   #  fraction measured the same way beside it)
        if world == 1 and not args.no_extra_configs:
            try:
                out["pcie_inclusive"] = pcie_inclusive(dev, wl, npdt)
            except Exception as e:
                out["pcie_inclusive"] = {"error": repr(e)[:300]}
        print(json.dumps(out), flush=True)
    group.close()


if __name__ == "__main__":
    # a rank that fails aborts its communicator and exits 1 on the spot (its peers get an error from the library's
    # deadline / RCCL's asynchronous error instead of hanging in a collective; the spawner relays the status)
    # (sr.recognition.parallel.exit_rank_on_failure, spelled out: nothing of the package is imported before the spawner runs)
    try:
        main()
    except SystemExit:
        raise
    except BaseException:
        import traceback
        traceback.print_exc()
        try:
            if getattr(Group, "active_native", None) is not None:
                Group.active_native.abort()
        except BaseException:
            traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
