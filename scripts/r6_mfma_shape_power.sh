#!/bin/bash
# round 6: energy of the two bf16 MFMA shapes at a convolution's matrix duty.  Back to back, v_mfma_f32_32x32x16_bf16 (mode 0) holds the chip at 1.75 GHz / 1.30 kW and
# v_mfma_f32_16x16x32_bf16 (mode 2) at 2.06 GHz / 1.33 kW (neither PPT-limited: another limiter), so their joules per FLOP are taken at different voltages.  With idle
# time between the iterations (nap units of 256 cycles per 512 cycles of MFMA issue per wave) both run unthrottled at the same clock: the power difference is the instruction's.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
run() { python scripts/smi_probe.py --period 0.5 -- scripts/microbench/mfma_dtype_power $1 5 2 $2 > /tmp/o.txt 2>&1; python3 -c "
import json
L=open('/tmp/o.txt').read().strip().splitlines(); j=json.loads(L[-1]); print(L[-2], '|', j.get('socket_W'), 'W', j.get('current_gfxclk'), 'MHz ppt', (j.get('limit_reasons') or {}).get('ppt_pwr'))"; }
for nap in 0 1 2 4 8; do for m in 0 2 1; do run $m $nap; done; done
