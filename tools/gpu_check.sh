#!/bin/bash
# scratch check: EM tests, then fb_chain2's utterances per wave swept on the C2 EM iteration (kernel times under rocprofv3)
mkdir -p gpurun_out/r05e
timeout -k 10 900 python -m pytest tests/test_gpu_em_session.py tests/test_gpu_fused.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r05e/em_tests.log 2>&1 || { tail -20 gpurun_out/r05e/em_tests.log; exit 1; }
tail -2 gpurun_out/r05e/em_tests.log
cd /tmp && export TMPDIR=/tmp
for upw in 4 3 2 1; do
  export GMMHMM_FBCHAIN_UPW=$upw
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05e/prof_em_$upw -o em -- python3 $GRAFT_REPO_ROOT/tools/time_em.py 12500 > $GRAFT_REPO_ROOT/gpurun_out/r05e/em_upw$upw.log 2>&1 || exit 1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/r05e/prof_em_$upw -name "*kernel_stats.csv" | head -1)
  echo "upw $upw: $(grep -i 'fb_chain2_kernel' $f | awk -F'",' '{print $2}' | cut -d, -f1-3)  |  $(grep -i 'ms_per_iteration\|ms per iteration' $GRAFT_REPO_ROOT/gpurun_out/r05e/em_upw$upw.log | tail -1 | cut -c1-200)"
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/r05e/prof_em_$upw
done
