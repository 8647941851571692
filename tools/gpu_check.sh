#!/bin/bash
# scratch check: sequence-form / alignment tests, then the kernel times of continuous_train
mkdir -p gpurun_out/r05e
timeout -k 10 900 python -m pytest tests/test_gpu_seq.py tests/test_gpu_e2e.py tests/test_gpu_api.py tests/test_gpu_train_words.py tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/r05e/seq_tests.log 2>&1 || { tail -30 gpurun_out/r05e/seq_tests.log; exit 1; }
tail -2 gpurun_out/r05e/seq_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/prof_ctrain -o ctrain -- python3 $GRAFT_REPO_ROOT/tools/time_ctrain.py 2000 7 6 > $GRAFT_REPO_ROOT/gpurun_out/r05e/ctrain_prof.log 2>&1 || exit 1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r05e/prof_ctrain -name "*kernel_stats.csv" | head -1)
cp "$f" gpurun_out/r05e/ctrain_kernel_stats.csv
rm -rf gpurun_out/r05e/prof_ctrain
grep -i "backtrace\|cut_segments" gpurun_out/r05e/ctrain_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}'
grep "steady\|per outer" gpurun_out/r05e/ctrain_prof.log
