#!/bin/bash
# more than two ranks on the one GPU of a gpurun box (at most six processes may hold the card): the five-rank tests, then
# bench.py with six ranks, once self-spawned and once under the driver's launcher
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_em_session.py -m gpu -x -q -k "five or one_of_five" > $O/pytest_ranks.log 2>&1; echo "pytest ranks exit $?"; tail -5 $O/pytest_ranks.log
timeout -k 10 900 python3 bench.py --gpus 6 --same-gpu --utts 1000 --em-utts 1500 --c5-utts 6000 --c4-utts 300 --c4-em-utts 200 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_6ranks_same_gpu.json 2> $O/bench_6ranks_same_gpu.err; echo "bench 6 ranks exit $?"; tail -c 600 $O/bench_6ranks_same_gpu.json; tail -3 $O/bench_6ranks_same_gpu.err
timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 6 --master-addr 127.0.0.1 --master-port 29650 bench.py --gpus 6 --same-gpu --utts 1000 --em-utts 1500 --c5-utts 6000 --c4-utts 300 --c4-em-utts 200 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_6ranks_same_gpu_torchrun.json 2> $O/bench_6ranks_same_gpu_torchrun.err; echo "bench 6 ranks torchrun exit $?"; tail -c 300 $O/bench_6ranks_same_gpu_torchrun.json; tail -3 $O/bench_6ranks_same_gpu_torchrun.err
