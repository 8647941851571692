#!/bin/bash
# Diagnostic builds of the fused single-Gaussian decode (timing only): what the LDS broadcasts cost.
# usage (on the GPU box): bash tools/fused_variants.sh [utterances] [D]
set -e
U=${1:-100000}; D=${2:-13}
for v in "" "-DGH_FUSED_NOLDS"; do
  touch speech-recognition_amd/csrc/gh_viterbi_fused.hip
  GMMHMM_EXTRA_FLAGS="$v" python speech-recognition_amd/build.py > /dev/null
  echo "== variant '${v}'"
  python tools/time_fused.py $U $D 2>&1 | tail -2
done
touch speech-recognition_amd/csrc/gh_viterbi_fused.hip
python speech-recognition_amd/build.py > /dev/null
