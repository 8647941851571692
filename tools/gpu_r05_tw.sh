#!/bin/bash
# train_words: tests, then stage times with 4 / 8 / 2 gather threads
mkdir -p gpurun_out/r05c
timeout -k 10 600 python -m pytest tests/test_gpu_train_words.py tests/test_gpu_api.py -x -q -m gpu > gpurun_out/r05c/tw_tests.log 2>&1 || { tail -20 gpurun_out/r05c/tw_tests.log; exit 1; }
tail -2 gpurun_out/r05c/tw_tests.log
for t in 4 8 2; do
  GMMHMM_HOST_THREADS=$t REPS=5 timeout -k 10 300 python3 tools/prof_train_words.py > gpurun_out/r05c/prof_train_words_t$t.txt 2>&1 || exit 1
  echo "threads $t"; grep "train_words \[ms\]\|^call \|without" gpurun_out/r05c/prof_train_words_t$t.txt
done
