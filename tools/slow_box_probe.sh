#!/bin/bash
# Some boxes of the pool run K6 ~15 % slower than the rest (60 us instead of 51 inside the loop).  This prints the quick K6 number
# and, on such a box, the phase profile, the issue microbenchmark and the clocks, to see WHAT is slower there.
cd "$(dirname "$0")/.."
K6=$(K6_REPS=12 python tools/k6_ab.py | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['median_second_half_us'])")
echo "K6 back to back: $K6 us"
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk"
if python -c "import sys; sys.exit(0 if float('$K6') > 51.0 else 1)"; then
  echo "== slow box =="
  python tools/ppo_phase_profile.py 2>/dev/null | grep -A16 "actor: total"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mib tools/mfma_issue_bench.hip 2>/dev/null && timeout 120 /tmp/mib | grep -E "^(0|11|40|41|48|33|43) "
  rocm-smi --showpower --showtemp 2>/dev/null | grep -E "Power|Temp" | head -6
fi
