#!/bin/bash
# cost of the K6 HIP-event brackets inside bench.py's timed region: ms per iteration at different sampling periods
for k in 8 1 8 1 64; do
  line=$(timeout 100 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-gae-sweep --k6-sample $k 2>/dev/null | grep metric)
  echo "k6-sample $k: $(echo "$line" | grep -o '"ms_per_step": [0-9.]*') $(echo "$line" | grep -o '"avg_launch_us": [0-9.]*' | head -1) $(echo "$line" | grep -o '"launches_timed": [0-9]*' | head -1)"
done
