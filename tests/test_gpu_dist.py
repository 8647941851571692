# -*- coding: utf-8 -*-
"""Two ranks sharing the one GPU of the test box: the sharded EM iteration (HIP E-step per
rank + ONE all-reduce of the packed statistics, gloo here / RCCL in production) gives the same
model as the single-process iteration, and sharded decode needs no collective."""
import os
import socket

import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    g = load_golden("G11_continuous_train")
    W, U = int(g["n_words"]), int(g["n_utts"])
    data = [g["x%d" % i] for i in range(U)]
    labels = [list(g["labels%d" % i]) for i in range(U)]
    means = np.array([g["init%d_means" % wi] for wi in range(W)])
    vars_ = np.array([g["init%d_vars" % wi] for wi in range(W)])
    w = np.array([g["init%d_w" % wi] for wi in range(W)])
    trans = [g["init%d_transitions" % wi] for wi in range(W)]
    return means, vars_, w, trans, data, labels


def _worker(rank, world, port, out_dir):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), os.path.join(os.path.dirname(here), "speech-recognition_amd"), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["GMMHMM_DEVICE"] = "0"  # both ranks on the single GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from sr.recognition.train import BaumWelchTrainer
        from sr.recognition.parallel import shard_utterances, StatsAllReducer
        means, vars_, w, trans, data, labels = _problem()
        mine = shard_utterances([len(x) for x in data], world)[rank]
        tr = BaumWelchTrainer(means, vars_, w, trans, [data[i] for i in mine], [labels[i] for i in mine],
                              device=0, reducer=StatsAllReducer(), var_floor=1e-3)
        hist = tr.fit(2)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), means=tr.means, vars=tr.vars, w=tr.weights,
                 hist=np.array(hist), mine=mine)
        tr.close()
    finally:
        dist.destroy_process_group()


def test_sharded_em_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from sr.recognition.train import BaumWelchTrainer
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    means, vars_, w, trans, data, labels = _problem()
    tr = BaumWelchTrainer(means, vars_, w, trans, data, labels, var_floor=1e-3)
    hist = tr.fit(2)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(np.concatenate([r0["mine"], r1["mine"]]).tolist()) == list(range(len(data)))
    for r in (r0, r1):
        np.testing.assert_allclose(r["hist"], hist, rtol=1e-10)
        np.testing.assert_allclose(r["means"], tr.means, rtol=1e-8, atol=1e-10)
        np.testing.assert_allclose(r["vars"], tr.vars, rtol=1e-7)
        np.testing.assert_allclose(r["w"], tr.weights, rtol=1e-8, atol=1e-12)
    tr.close()
