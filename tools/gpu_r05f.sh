#!/bin/bash
# FETCH_SIZE calibrated for 8-byte-per-lane reads, then the fused single-Gaussian sweep (configs[0] x 1000) under the same counter
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05
mkdir -p $O
rm -rf $O/pmc_calib $O/pmc_fused_fetch $O/pmc_fused_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_calib -o c -- tools/bin/hbm_stream calib > $O/calib.log 2>&1; echo "calib exit $?"; tail -1 $O/calib.log
DT=f64 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fused_fetch -o p -- python3 tools/time_fused.py 100000 13 > /dev/null 2> $O/pmc_fused_fetch.err; echo "fused pmc fetch exit $?"
DT=f64 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_fused_write -o p -- python3 tools/time_fused.py 100000 13 > /dev/null 2> $O/pmc_fused_write.err; echo "fused pmc write exit $?"
c=$(find $O/pmc_calib -name "*counter_collection.csv" | head -1); f=$(find $O/pmc_fused_fetch -name "*counter_collection.csv" | head -1); w=$(find $O/pmc_fused_write -name "*counter_collection.csv" | head -1)
cp $c $O/pmc_calib.csv
python3 tools/pmc_calibrated.py $c $f viterbi_fused 1040000000 k_read8_tiles $w > $O/pmc_fused_traffic.json; echo "fold exit $?"
head -c 2500 $O/pmc_fused_traffic.json
rm -rf $O/pmc_calib $O/pmc_fused_fetch $O/pmc_fused_write
