#!/bin/bash
# scheduling knobs re-sweep: one score evaluation (configs[1] shape)
for o in "subbatch=2,stagger_level=2" "subbatch=2,stagger_level=3" "subbatch=2,stagger_level=1" "subbatch=3,stagger_level=2" "subbatch=3,stagger_level=3" "subbatch=4,stagger_level=1" "subbatch=2,stagger_level=2,conv_sk_max_px=1280" "subbatch=2,stagger_level=2,gn_inline=81920" "subbatch=2,stagger_level=2"; do
  echo "$o: $(USE_OPTS=$o python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-45)"
done
