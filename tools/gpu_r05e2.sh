#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 5 --same-gpu --utts 1000 --em-utts 1500 --c5-utts 6000 --c4-utts 300 --c4-em-utts 200 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_5ranks_same_gpu_torchrun.json 2> $O/bench_5ranks_same_gpu_torchrun.err; echo "bench 5 ranks torchrun exit $?"; tail -c 300 $O/bench_5ranks_same_gpu_torchrun.json; tail -3 $O/bench_5ranks_same_gpu_torchrun.err
