#!/bin/bash
# slab-store policy A/B (boxes of the pool differ: on some the minibatch kernel's stores are the slow part).  Variants built by
#   tools/build_variants.sh main "st1:-DERL_SLAB_ST=1:main:ppo_step_s3_pre.hip" "st2:-DERL_SLAB_ST=2:main:ppo_step_s3_pre.hip" ...
cd $GRAFT_REPO_ROOT
L=$GRAFT_REPO_ROOT/elegantrl_amd/lib
for v in "" _st0 _st1 _st3 _st4 _ld1 _ld2 ""; do
  lib=$L/liberl_hip$v.so; [ -f $lib ] || continue
  ERL_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-gae-sweep --repeats 2 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('slab store variant [$v]', d['value'], d['ms_per_step'], d['extra']['repeated_regions_ms_per_step'], 'k6 span', d['roofline']['avg_launch_us'], 'event', d['roofline']['event_bracket_us'], 'rest', d['breakdown']['per_minibatch_rest_us'])"
done
