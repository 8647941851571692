#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
for bpw in 1 2 4; do
  for dt in f32 f64; do
    GMMHMM_LOGLIK_BPW=$bpw python3 bench.py --dtype $dt --no-cpu-baseline --no-em --no-extra-configs > $O/bench_bpw${bpw}_$dt.json 2>/dev/null
    python3 - <<PY
import json
d=json.load(open("$O/bench_bpw${bpw}_$dt.json"))
print("bpw $bpw $dt: ms_per_step %.3f kernel_ms %.3f frac %.3f acc %.3f" % (d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["frac"], d["decode_accuracy"]))
PY
  done
done
