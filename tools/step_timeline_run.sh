#!/bin/bash
# kernel trace of one relay-rollout bench run + the per-step timeline:  tools/step_timeline_run.sh <bench args...>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline --no-kernel-timers > /tmp/tr.json 2>/tmp/tr.err
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/step_timeline.py "$f" 200
tail -1 /tmp/tr.json | cut -c1-160
