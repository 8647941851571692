#!/bin/bash
# Round-5 evidence in ONE gpurun call: bench (plain + under rocprofv3), kernel stats of the streaming refit kernels, of
# continuous_train (+ steady-state time, host profile), of train_words (+ its stage times) and the decode over 12- / 16-state
# word models.  Outputs under gpurun_out/$TAG/ (TAG defaults to r05a).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/${TAG:-r05a}
mkdir -p $O
python3 bench.py > $O/bench_f64.json 2> $O/bench_f64.err; echo "bench f64 exit $?"
python3 bench.py --dtype f32 --no-cpu-baseline > $O/bench_f32.json 2> $O/bench_f32.err; echo "bench f32 exit $?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_headline -o headline -- python3 bench.py --no-cpu-baseline --no-em --no-extra-configs > $O/headline.json 2> $O/headline.err; echo "headline prof exit $?"
for k in 4 8; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_refit_k$k -o r -- python3 tools/time_refit.py $k > $O/refit_k$k.log 2>&1; echo "refit k$k prof exit $?"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ctrain -o ct -- python3 tools/time_ctrain.py 2000 7 6 > $O/ctrain_prof.log 2>&1; echo "ctrain prof exit $?"
CTRAIN_PROFILE=0 python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain.log 2>&1; echo "ctrain exit $?"
python3 tools/time_ctrain.py 2000 7 8 > $O/ctrain_host_profile.log 2>&1; echo "ctrain host profile exit $?"
REPS=4 python3 tools/time_train_words.py > $O/train_words.log 2>&1; echo "train_words exit $?"
REPS=5 python3 tools/prof_train_words.py > $O/train_words_stages.txt 2>&1; echo "train_words stages exit $?"
python3 tools/time_layers_wide.py 5000 16 10 7 > $O/layers_wide.txt 2>&1 && python3 tools/time_layers_wide.py 5000 12 10 7 >> $O/layers_wide.txt 2>&1; echo "wide word models exit $?"
# the randomised sweeps (each a few seconds): streaming vs tile refit kernels, word-template vs lean Viterbi kernels, train_words vs
# the word-after-word loop, continuous_train's device path vs the compat_cov host path
python3 tools/stress_refit.py 400 2 > $O/stress_refit.txt 2>&1; echo "stress_refit exit $?"
python3 tools/stress_decode.py 400 2 > $O/stress_decode.txt 2>&1; echo "stress_decode exit $?"
python3 tools/stress_train_words.py 300 2 > $O/stress_train_words.txt 2>&1; echo "stress_train_words exit $?"
python3 tools/stress_ctrain.py 150 2 > $O/stress_ctrain.txt 2>&1; echo "stress_ctrain exit $?"
for d in headline refit_k4 refit_k8 ctrain; do f=$(find $O/prof_$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/${d}_kernel_stats.csv; done
rm -rf $O/prof_*
ls $O
