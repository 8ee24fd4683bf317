#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (on the GPU box). usage: run_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${1:-r1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/bench_stdout.log 2>&1
echo "rc=$?"; grep -o '{"metric.*' $OUT/bench_stdout.log | cut -c1-300; ls $OUT
