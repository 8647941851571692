#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
for v in "thread" "nothread" "thread_fastswitch"; do
  if [ $v = nothread ]; then export GMMHMM_CTRAIN_PICKLE_THREAD=0; else unset GMMHMM_CTRAIN_PICKLE_THREAD; fi
  if [ $v = thread_fastswitch ]; then export PYSWITCH=0.0003; else unset PYSWITCH; fi
  for rep in 1 2; do
  CTRAIN_PROFILE=0 timeout -k 10 600 python3 tools/time_ctrain.py 2000 7 10 > $O/ctrain_$v.log 2>&1; echo "$v: $(tail -2 $O/ctrain_$v.log | head -1) | $(tail -1 $O/ctrain_$v.log)"
  done
done
