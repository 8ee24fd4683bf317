#!/bin/bash
# rocprofv3 kernel-trace stats of `bench.py --roofline-only` (one launch on the chip at a time). usage: run_roofline_profile.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_${1:-r2}_roofline
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o bench -- python $R/bench.py --roofline-only > $OUT/bench_stdout.log 2>&1
echo "rc=$?"; grep -o '{"metric.*\|{"roofline.*\|{.*avg_launch.*' $OUT/bench_stdout.log | cut -c1-600; ls $OUT
