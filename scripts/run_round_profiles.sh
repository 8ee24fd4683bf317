#!/bin/bash
# round-end measurement set on the GPU box: bench line, rocprofv3 statistics of the bench command and of the roofline measurement,
# configs[3] / configs[4] lines.  usage: run_round_profiles.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r2_b}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
python bench.py --steps 5 --warmup 2 > $OUT/bench_line.json 2> $OUT/bench_stderr.log; echo "bench rc=$?"; cut -c1-400 $OUT/bench_line.json
python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg3.json 2>> $OUT/bench_stderr.log; echo "cfg3 rc=$?"; cut -c1-200 $OUT/bench_cfg3.json
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4.json 2>> $OUT/bench_stderr.log; echo "cfg4 rc=$?"; cut -c1-200 $OUT/bench_cfg4.json
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT/prof_bench $OUT/prof_roofline
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/prof_bench/bench_stdout.log 2>&1; echo "prof bench rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -o bench -- python $R/bench.py --roofline-only > $OUT/prof_roofline/bench_stdout.log 2>&1; echo "prof roofline rc=$?"
rm -f $OUT/prof_bench/*kernel_trace.csv $OUT/prof_roofline/*kernel_trace.csv   # keep the statistics; the traces are large
ls $OUT $OUT/prof_bench
