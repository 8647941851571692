#!/bin/bash
# Round-4 evidence in ONE gpurun call: bench (plain + under rocprofv3), PMC traffic passes of the headline kernel, kernel
# stats + SQ counters of the fused single-Gaussian sweep (configs[0] x 1000), the fp32-exponential epilogue, kernel stats of
# the EM iteration / continuous_train / C5.  Outputs under gpurun_out/$TAG/ (TAG defaults to r04).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r04}
mkdir -p $O
python3 bench.py > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench f64 exit $?"
python3 bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs > $O/headline.json 2> $O/headline.err; echo "headline prof exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_fetch.err; echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_write.err; echo "pmc write exit $?"
DT=f64 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fused -o f -- python3 tools/time_fused.py 100000 13 > $O/fused.log 2>&1; echo "fused prof exit $?"
DT=f64 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fused_fetch -o p -- python3 tools/time_fused.py 100000 13 > /dev/null 2> $O/pmc_fused_fetch.err; echo "fused pmc fetch exit $?"
DT=f64 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_fused_write -o p -- python3 tools/time_fused.py 100000 13 > /dev/null 2> $O/pmc_fused_write.err; echo "fused pmc write exit $?"
DT=f64 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_fused_sq -o p -- python3 tools/time_fused.py 100000 13 > /dev/null 2> $O/pmc_fused_sq.err; echo "fused pmc sq exit $?"
python3 tools/time_fused.py 100000 13 > $O/fused_plain.log 2>&1; echo "fused plain exit $?"
python3 tools/time_fused.py 20000 39 > $O/fused_d39.log 2>&1; echo "fused d39 exit $?"
python3 tools/time_lse.py > $O/lse_f32exp.json 2> $O/lse_f32exp.err; echo "lse exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_em -o em -- python3 tools/time_em.py 12500 > $O/em.log 2>&1; echo "em prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_emstr -o es -- python3 tools/time_em_strings.py > $O/em_strings.log 2>&1; echo "em strings prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctrain -o ct -- python3 tools/time_ctrain.py 2000 7 6 > $O/ctrain_prof.log 2>&1; echo "ctrain prof exit $?"
CTRAIN_PROFILE=0 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain.log 2>&1; echo "ctrain exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 tools/time_c5.py 125000 > $O/c5.json 2> $O/c5.err; echo "c5 prof exit $?"
python3 bench.py --gpus 2 --same-gpu --steps 20 --warmup 5 --no-cpu-baseline --c5-utts 20000 --c4-utts 2000 > $O/bench_2ranks_same_gpu.json 2> $O/bench_2ranks_same_gpu.err; echo "2-rank bench exit $?"
# fold: kernel stats csv files next to the logs, raw profiler directories removed
for d in headline fused em emstr ctrain c5; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
for d in fetch write fused_fetch fused_write fused_sq; do f=$(find $O/pmc_$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $O/pmc_${d}.csv; done
python3 tools/pmc_traffic.py $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_traffic_f64.json 999364 f64 > /dev/null 2>&1
python3 tools/pmc_kernel.py $O/pmc_fused_sq.csv viterbi_fused > $O/pmc_fused_sq.txt 2>&1
python3 tools/pmc_kernel.py $O/pmc_fused_fetch.csv viterbi_fused > $O/pmc_fused_fetch.txt 2>&1
python3 tools/pmc_kernel.py $O/pmc_fused_write.csv viterbi_fused > $O/pmc_fused_write.txt 2>&1
rm -rf $O/prof_* $O/pmc_fetch $O/pmc_write $O/pmc_fused_fetch $O/pmc_fused_write $O/pmc_fused_sq $O/pmc_fetch.csv $O/pmc_write.csv $O/pmc_fused_sq.csv $O/pmc_fused_fetch.csv $O/pmc_fused_write.csv
ls $O
