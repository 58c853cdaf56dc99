#!/bin/bash
# sweep of the ring kernel's poll gap (64-cycle sleeps between polls of the rec granule): tools/ring_sweep.sh [threads]
for g in 1 4 16 64; do
  A2C_RING_POLL=$g python tools/ring_timing.py ${1:-16} bits 2>/dev/null | python -c "
import json,sys,os
d=json.load(sys.stdin)
print('gap', os.environ.get('G'), d['rollout_ms_timed'], {k[:12]: v for k,v in d.items() if isinstance(v,float) and k[0] in 'wbfTp'})" 
done
