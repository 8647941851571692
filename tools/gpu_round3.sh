#!/bin/bash
# Round-3 evidence in ONE gpurun call: bench (plain + under rocprofv3), PMC traffic passes of the headline kernel, kernel
# stats of the EM iteration / continuous_train / C5, SQ counters of the EM kernels.  Outputs under gpurun_out/r03/.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03
mkdir -p $O
python3 bench.py > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench f64 exit $?"
python3 bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs > $O/headline.json 2> $O/headline.err; echo "headline prof exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_fetch.err; echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_write.err; echo "pmc write exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_em -o em -- python3 tools/time_em.py 12500 > $O/em.log 2>&1; echo "em prof exit $?"
EM_OCC_FLOOR=1e-30 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_em_floor -o em -- python3 tools/time_em.py 12500 > $O/em_floor.log 2>&1; echo "em (floor) prof exit $?"
EM_OCC_FLOOR=1e-30 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_em -o p -- python3 tools/time_em.py 12500 > /dev/null 2> $O/pmc_em.err; echo "pmc em exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctrain -o ct -- python3 tools/time_ctrain.py 2000 7 6 > $O/ctrain_prof.log 2>&1; echo "ctrain prof exit $?"
CTRAIN_PROFILE=0 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain.log 2>&1; echo "ctrain exit $?"
python3 tools/time_ctrain.py 2000 7 4 > $O/ctrain_host_profile.log 2>&1; echo "ctrain host profile exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 tools/time_c5.py 125000 > $O/c5.json 2> $O/c5.err; echo "c5 prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4em -o c4em -- python3 tools/time_c4_em.py 50000 > $O/c4em.json 2> $O/c4em.err; echo "c4 em prof exit $?"
python3 tools/comm_probe.py --world 2 --same-gpu > $O/comm_probe_2ranks.log 2>&1; echo "comm probe exit $?"
python3 bench.py --gpus 2 --same-gpu --steps 20 --warmup 5 --no-cpu-baseline --c5-utts 20000 --c4-utts 2000 > $O/bench_2ranks_same_gpu.json 2> $O/bench_2ranks_same_gpu.err; echo "2-rank bench exit $?"
ls $O
