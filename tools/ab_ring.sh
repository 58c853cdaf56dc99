#!/bin/bash
# A/B of ring-kernel variants on one box: headline bench (100 steps), rollout / update ms.  usage: tools/ab_ring.sh VAR v0 v1 [reps]
VAR=$1; V0=$2; V1=$3; REPS=${4:-2}
fmt='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["rollout_ms"], d["update_ms"])'
for i in $(seq $REPS); do
  for v in $V0 $V1; do
    env $VAR=$v python bench.py --no-configs --no-secondary --no-cpu-baseline --frame-store --steps 100 2>/dev/null | python -c "$fmt" "$VAR=$v lazy "
    env $VAR=$v python bench.py --no-configs --no-secondary --no-cpu-baseline --steps 100 2>/dev/null | python -c "$fmt" "$VAR=$v plain"
  done
done
