#!/bin/bash
# One entry point for the measurements kept under profiles/ (run on the GPU box: `gpurun -- scripts/profile.sh <what> [args]`;
# everything lands in gpurun_out/<tag>/ and is copied into profiles/ by hand).  Replaces the round 1-3 collection of run_*.sh drivers.
#
#   round <tag>            bench lines (configs[1], [3], [4]) + rocprofv3 --kernel-trace --stats of the bench command and of
#                          `bench.py --roofline-only` (one launch on the chip at a time: what roofline.avg_launch_ms must agree with)
#   pmc <tag> [prec B T]   PMC passes (counters only, separate --pmc runs, --kernel-trace) of one score evaluation; then
#                          `python scripts/pmc_to_json.py gpurun_out/<tag> conv_v4 profiles/<name>.json` (FETCH_SIZE x2 on gfx950)
#   power [opts] [iters]   socket power + shader clock once a second while the evaluation runs back to back
#   train [fp32|bf16]      training-step timing + per-kernel statistics; `train-pmc <tag> [prec]` its PMC passes
#   ab <libA.so> <libB.so> same-box A/B of library builds (scripts/ab_libs.py: harness timings, bit-identity, end to end)
#   trace <lib> <wg> <case> cycle stamps of one workgroup of conv_v4 (trace build: make EXTRA=-DUSE_HIP_TRACE_BUILD)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WHAT=${1:-round}; shift
PMC_SETS=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM"
  "FETCH_SIZE GRBM_GUI_ACTIVE"
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum")
pmc_passes() {   # <outdir> <command...>
  local OUT=$1; shift; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
  local i=0
  for set in "${PMC_SETS[@]}"; do
    i=$((i+1))
    timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- "$@" > $OUT/pass$i.log 2>&1
    echo "pass $i rc=$?"
  done
}
case $WHAT in
round)
  OUT=$R/gpurun_out/${1:-round}; mkdir -p $OUT; cd $R
  python bench.py --steps 5 --warmup 2 > $OUT/bench_line.json 2> $OUT/bench_stderr.log; echo "bench rc=$?"; cut -c1-400 $OUT/bench_line.json
  python bench.py --config 3 --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_cfg3.json 2>> $OUT/bench_stderr.log; echo "cfg3 rc=$?"; cut -c1-200 $OUT/bench_cfg3.json
  python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/bench_cfg4.json 2>> $OUT/bench_stderr.log; echo "cfg4 rc=$?"; cut -c1-200 $OUT/bench_cfg4.json
  cd /tmp && export TMPDIR=/tmp; mkdir -p $OUT/prof_bench $OUT/prof_roofline
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > $OUT/prof_bench/bench_stdout.log 2>&1; echo "prof bench rc=$?"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_roofline -o bench -- python $R/bench.py --roofline-only > $OUT/prof_roofline/bench_stdout.log 2>&1; echo "prof roofline rc=$?"
  rm -f $OUT/prof_bench/*kernel_trace.csv $OUT/prof_roofline/*kernel_trace.csv   # keep the statistics; the traces are large
  for d in prof_bench prof_roofline; do python $R/scripts/rocprof_summary.py $OUT/$d > $OUT/${d}_kernel_stats.txt 2>/dev/null; done
  ls $OUT ;;
pmc)
  pmc_passes $R/gpurun_out/${1:-pmc} python $R/scripts/gpu_time_forward.py ${2:-bf16} ${3:-8} ${4:-640} 1 ;;
power)
  (USE_OPTS="${1:-}" python $R/scripts/gpu_time_forward.py bf16 8 640 ${2:-400} > $R/gpurun_out/power_run.log 2>&1) &
  PID=$!
  while kill -0 $PID 2>/dev/null; do
    P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
    C=$(rocm-smi --showclocks 2>/dev/null | grep sclk | grep -oE "\([0-9]+Mhz\)")
    echo "power $P W sclk $C"; sleep 1
  done
  tail -1 $R/gpurun_out/power_run.log | cut -c1-70 ;;
train)
  cd $R; mkdir -p gpurun_out; export TRAIN_PRECISION=${1:-fp32}
  python scripts/train_step_bench.py 4 512 5 > gpurun_out/train_bench.log 2>&1
  export TMPDIR=/tmp; rm -rf gpurun_out/train_prof
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof -o train -- python scripts/train_step_bench.py 4 512 2 > gpurun_out/train_prof.log 2>&1
  python scripts/rocprof_summary.py gpurun_out/train_prof > gpurun_out/train_kernel_stats.txt; tail -1 gpurun_out/train_bench.log; head -30 gpurun_out/train_kernel_stats.txt ;;
train-pmc)
  export TRAIN_PRECISION=${2:-bf16}; pmc_passes $R/gpurun_out/${1:-pmc_train} python $R/scripts/train_step_bench.py 2 512 1 ;;
ab)
  cd $R; python scripts/ab_libs.py --e2e "$@" ;;
trace)
  cd $R; USE_HIP_LIB=$R/$1 USE_HIP_TRACE=${2:-1400} python scripts/gpu_conv_trace.py 4 "${3:-L0 conv0 128->128}" ;;
*) echo "unknown: $WHAT"; exit 2 ;;
esac
