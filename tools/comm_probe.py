# -*- coding: utf-8 -*-
"""Probe of the native RCCL path (gh_comm_*): one rank, or N ranks as N processes.

    python tools/comm_probe.py                   one rank on GPU 0
    python tools/comm_probe.py --world 2 --same-gpu
        two ranks that share GPU 0: RCCL refuses two ranks on one device of one HOST, so every rank announces a host
        id of its own (NCCL_HOSTID) and the ranks talk through the socket transport on the loopback interface -- the
        bootstrap, the id hand-over, the collective and its stream ordering are the real ones, only the wire is not xGMI.
"""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "speech-recognition_amd"))


def rank_main(args):
    import numpy as np
    from sr.recognition import _hip
    from sr.recognition.parallel import NativeReducer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    ctx = _hip.default_context(0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0")))
    t0 = time.perf_counter()
    red = NativeReducer(ctx, rank, world, timeout=60)
    t1 = time.perf_counter()
    lib = ctx.lib
    out = red(np.arange(5, dtype=np.float64) + rank)
    expect = world * np.arange(5) + world * (world - 1) / 2
    assert np.array_equal(out, expect), (out, expect)
    mx = red.max(np.array([float(rank)]))
    assert mx[0] == world - 1
    red.barrier()
    print("rank %d/%d: comm up in %.2f s, ncclCommCount=%d, rccl %d from %s, allreduce OK" % (
        rank, world, t1 - t0, red.comm.count, lib.gh_comm_version(), lib.gh_comm_library().decode()), flush=True)
    red.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=1)
    ap.add_argument("--same-gpu", action="store_true")
    ap.add_argument("--port", type=int, default=29731)
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--torch-first", action="store_true", help="import torch (its HIP runtime and RCCL) before the library")
    args = ap.parse_args()
    if args.child or args.world == 1:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.torch_first:
            import torch  # noqa: F401
            torch.cuda.is_available()
        rank_main(args)
        return 0
    procs = []
    for r in range(args.world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(args.world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(args.port))
        if args.same_gpu:
            env.update(NCCL_HOSTID="gmmhmm-probe-%d" % r, NCCL_SOCKET_IFNAME="lo", NCCL_IB_DISABLE="1", NCCL_P2P_DISABLE="1",
                       NCCL_SHM_DISABLE="1")
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--world", str(args.world)] + (["--same-gpu"] if args.same_gpu else [])
        procs.append(subprocess.Popen(cmd, env=env))
    rc = 0
    deadline = time.time() + 150
    for p in procs:
        try:
            rc |= p.wait(timeout=max(1, deadline - time.time()))
        except subprocess.TimeoutExpired:
            p.kill()
            rc |= 1
            print("rank process killed after timeout", flush=True)
    return rc


if __name__ == "__main__":
    sys.exit(main())
