#!/bin/bash
# the ring kernel's phase stamps at several env thread counts: tools/ring_sweep.sh
for n in 8 12 16 24; do
  python tools/ring_timing.py $n bits 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print(d['env_threads'], 'threads', d['rollout_ms_timed'], 'ms', {k[:12]: v for k,v in d.items() if isinstance(v,float) and k[0] in 'wbfTpsc'})"
done
