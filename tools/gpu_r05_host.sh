#!/bin/bash
# host-side trims of round 5: tests of everything that opens refit sessions, then continuous_train / train_words timings
mkdir -p gpurun_out/r05c
timeout -k 10 900 python -m pytest tests/test_gpu_refit.py tests/test_gpu_lockstep.py tests/test_gpu_train_words.py tests/test_gpu_api.py tests/test_gpu_e2e.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r05c/host_tests.log 2>&1 || { tail -30 gpurun_out/r05c/host_tests.log; exit 1; }
tail -2 gpurun_out/r05c/host_tests.log
timeout -k 10 300 python3 tools/time_ctrain.py 2000 7 8 > gpurun_out/r05c/ctrain.log 2>&1 || exit 1
tail -3 gpurun_out/r05c/ctrain.log
GMMHMM_FIT_KEEP_MB=0 timeout -k 10 300 python3 tools/time_ctrain.py 2000 7 8 > gpurun_out/r05c/ctrain_nokeep.log 2>&1 || exit 1
tail -2 gpurun_out/r05c/ctrain_nokeep.log
REPS=5 timeout -k 10 300 python3 tools/prof_train_words.py > gpurun_out/r05c/prof_train_words.txt 2>&1 || exit 1
grep "train_words \[ms\]\|^call \|without" gpurun_out/r05c/prof_train_words.txt
