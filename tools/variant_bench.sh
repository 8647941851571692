#!/bin/bash
# Build libgmmhmm variants with extra -D flags on the likelihood kernel and print their bench numbers.
# usage (on the build container): tools/variant_bench.sh build NAME "-DFLAG=.. -DFLAG2=.."
#        (on the GPU box):        tools/variant_bench.sh run NAME [NAME...]
set -e
cd "$(dirname "$0")/.."
P=speech-recognition_amd
if [ "$1" = build ]; then
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-value -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form=1 $3 -c $P/csrc/gh_loglik_mfma.hip -o tools/bin/mf_$2.o
  objs=$(ls $P/build/*.o | grep -v gh_loglik_mfma.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o tools/bin/libgmmhmm_$2.so $objs tools/bin/mf_$2.o
  rm -f tools/bin/mf_$2.o
else
  shift
  for v in "$@"; do
    for dt in f64 f32; do
      GMMHMM_LIB=$PWD/tools/bin/libgmmhmm_$v.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-em --no-extra-configs --inflight 1 --dtype $dt 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', d['dtype'], 'kernel_ms %.4f frac %.3f step_ms %.3f' % (d['roofline']['kernel_ms'], d['roofline']['frac'], d['ms_per_step']))"
    done
  done
fi
