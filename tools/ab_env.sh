#!/bin/bash
# A/B of one library switch inside bench.py on the GPU box:  tools/ab_env.sh HIPT_NO_EMBED_LN [reps]
# prints regions/s and the per-kernel averages for SWITCH=1 / SWITCH unset, interleaved, `reps` times (default 2).
SW=${1:?switch name}; REPS=${2:-2}
for rep in $(seq $REPS); do
  for v in 1 ""; do
    env ${v:+$SW=1} timeout -k 10 200 python bench.py --steps 10 --warmup 3 --slides 0 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('$SW=${v:-0}', round(d['value'],2), {n:round(x['avg_us']*x['launches_per_step']/1000,2) for n,x in k.items() if x['avg_us']*x['launches_per_step']>300}, d['selfcheck'])" || exit 1
  done
done
