#!/usr/bin/env python3
# -*- coding: utf-8 -*-
"""
Headline benchmark: frame-state log-likelihoods/s + utterances/s Viterbi decode
(BASELINE.json) of the GMM-HMM hot path on MI355X, through the C ABI.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

One "step" = one pass of the hot path over the batch: batched GMM log-likelihood
of every frame against every state (A3) + isolated-word Viterbi of every
utterance through the 10 stacked word models (A6) + arg-min word per utterance
(sr/core.py:82-87).  Workload = BASELINE.json configs[1]: 10-digit topology,
5 states/digit, 8-mix GMM, 39-dim features, 10k synthetic utterances per GPU
(weak scaling: every rank decodes its own 10k utterances; no data-path
collective -- decode shards by utterance).  Features are resident in HBM before
the timed region.  Arithmetic: fp64 (the reference's), which is what keeps the
Viterbi paths bit-identical; --dtype f32 runs the fp32 likelihood kernel.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra
objects: "roofline" for the dominant kernel (the likelihood kernel; HIP-event
timed live on the library's stream) and "cpu_baseline" (the numpy oracle,
reference-shaped scalar loops, 1 core, on a bounded sample).
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


def synth_workload(seed, n_utts, W=10, n=5, M=8, D=39, tmin=50, tmax=150, utt_seed=None):
    """means ~ N(0,1), vars ~ U[.5,1.5], w ~ Dirichlet(1); left-to-right
    costs (self -log .9, next -log .1, last self 0); frames from a uniform segmentation."""
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
    words = rng.integers(0, W, size=n_utts)
    T = rng.integers(tmin, tmax + 1, size=n_utts)
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


class HipEvents:
    """hipEvent timing on an explicit stream (torch.cuda.Event would only see torch's stream)."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        self.hip.hipEventSynchronize.argtypes = [C.c_void_p]
        self.hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]

    def new(self):
        e = C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(e)) == 0
        return e

    def record(self, e, stream):
        assert self.hip.hipEventRecord(e, C.c_void_p(stream)) == 0

    def wait(self, stream, e):
        """Work submitted to `stream` after this call starts only when `e` has completed."""
        assert self.hip.hipStreamWaitEvent(C.c_void_p(stream), e, 0) == 0

    def elapsed_ms(self, a, b):
        assert self.hip.hipEventSynchronize(b) == 0
        ms = C.c_float()
        assert self.hip.hipEventElapsedTime(C.byref(ms), a, b) == 0
        return float(ms.value)


def profiled_traffic(dtype, n_frames):
    """HBM bytes per launch of the likelihood kernel from the committed PMC passes (the newest
    profiles/*_pmc_traffic_<dtype>.json: FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc runs of this
    same command, folded by tools/pmc_traffic.py); None when no profile matches this dtype / workload size."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_%s.json" % dtype)))
    try:
        d = json.load(open(paths[-1]))["kernels"]
        k = next(v for name, v in d.items() if name.startswith("loglik"))
        esz = 8 if dtype == "f64" else 4
        if abs(k["algorithmic_bytes_per_launch"] - n_frames * esz * (39 + 50)) > 1e-3 * k["algorithmic_bytes_per_launch"]:
            return None
        return k["hbm_bytes_per_launch"]
    except Exception:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
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
    ap.add_argument("--cpu-utts", type=int, default=80)
    # rehearsal of the N > 1 path on a box with fewer GPUs: --backend gloo --device 0 lets every rank share GPU 0
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--device", type=int, default=None, help="GPU index for every rank (default: LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    dev = local_rank if args.device is None else args.device
    if world > 1:
        import torch
        import torch.distributed as dist
        if args.backend == "nccl":
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group("gloo")
    try:
        import torch
        have_torch_cuda = torch.cuda.is_available()
    except Exception:
        torch, have_torch_cuda = None, False
    red_dev = "cuda" if (world > 1 and args.backend == "nccl") else "cpu"

    from sr.recognition import _hip
    npdt = np.float64 if args.dtype == "f64" else np.float32
    wl = synth_workload(1002, args.utts, utt_seed=None if rank == 0 else 1002 + 7919 * rank)
    W, n, M, D = wl["W"], wl["n"], wl["M"], wl["D"]
    S = W * n
    ev = HipEvents()

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
            self.untimed_done = ev.new()

        def step(self, timed):
            """One pass of the hot path over the batch: likelihoods, Viterbi over every word model, arg-min.
            The likelihood kernels of different lanes are chained by events (each fills the whole GPU, so
            overlapping two of them gains nothing and would blur the per-launch timing)."""
            stream = self.ctx.stream
            with chain_lock:
                if chain["last"] is not None:
                    ev.wait(stream, chain["last"])
                if timed:
                    a, b = ev.new(), ev.new()
                    ev.record(a, stream)
                else:
                    b = self.untimed_done
                self.batch.loglik(self.gmm, fetch=False)
                ev.record(b, stream)
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
        for l in lanes:
            l.ctx.sync()
        if have_torch_cuda:
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        for l in lanes:
            l.ctx.sync()
        if have_torch_cuda:
            torch.cuda.synchronize()

    run_steps(0, False, seconds=args.ramp_seconds)   # clock ramp, untimed (see --ramp-seconds)
    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    dt = time.perf_counter() - t0
    decoded = next(l.decoded for l in lanes if l.decoded is not None)
    ll_ms = [p for l in lanes for p in l.ll_ms]
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tot = torch.tensor([float(N_frames), float(U)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tot)
        dt = float(tmax.item())
        all_frames, all_utts = float(tot[0].item()), float(tot[1].item())
    else:
        all_frames, all_utts = float(N_frames), float(U)
    accuracy = float(np.mean(decoded == wl["words"]))

    if rank == 0:
        ll_avg_s = float(np.mean([ev.elapsed_ms(a, b) for a, b in ll_ms])) * 1e-3
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
                       "states": S, "mixtures": M, "dim": D, "parallelism": "utterance-sharded x%d" % world, "batches_in_flight": len(lanes)},
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
            },
        }
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is a rank-0, single-GPU-run measurement
            out["cpu_baseline"] = cpu_baseline(wl, args.cpu_utts)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
