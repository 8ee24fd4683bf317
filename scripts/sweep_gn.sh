#!/bin/bash
for r in 1 2; do for o in gn_inline=20480 gn_inline=81920 gn_inline=400000 gn_inline=0; do
  echo "$o: $(USE_OPTS=$o python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-45)"
done; done
