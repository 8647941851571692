#!/bin/bash
# Diagnostic builds of the sequence-form forward-backward (timing only): read-ahead depth of the recursion.
# usage (on the GPU box): bash tools/fbseq_variants.sh "2 4 8"
set -e
for pf in ${1:-2 4 8}; do
  touch speech-recognition_amd/csrc/gh_seq.hip
  GMMHMM_EXTRA_FLAGS="-DGH_FBSEQ_PF=$pf" python speech-recognition_amd/build.py > /dev/null
  echo "== PF=$pf"
  python tools/time_em_strings.py 2>&1 | tail -1
done
touch speech-recognition_amd/csrc/gh_seq.hip
python speech-recognition_amd/build.py > /dev/null
