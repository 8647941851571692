set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02s
mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest exit $?" >> $O/tests.log; tail -3 $O/tests.log
python3 bench.py > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench f64 exit $?"
python3 bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs > $O/headline.json 2> $O/headline.err; echo "headline prof exit $?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_fetch.err; echo "pmc fetch exit $?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs --steps 30 > /dev/null 2> $O/pmc_write.err; echo "pmc write exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o c5 -- python3 tools/time_c5.py 125000 > $O/c5.json 2> $O/c5.err; echo "c5 prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_em -o em -- python3 tools/time_em.py 12500 > $O/em.log 2>&1; echo "em prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_em7 -o em7 -- python3 tools/time_em.py 87500 7 > $O/em7.log 2>&1; echo "em7 prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_seq -o seq -- python3 tools/time_seq.py 12500 7 > $O/seq.log 2>&1; echo "seq prof exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctrain -o ct -- python3 tools/time_ctrain.py 2000 7 2 > $O/ctrain.log 2>&1; echo "ctrain prof exit $?"
ls $O
