#!/bin/bash
# kernel stats of the GRU+BPTT and ConvModel configs (tools/dbg/sites.py: eager update, native env threads, no child process)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
for wl in gru_bptt conv; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$wl -- python3 $R/tools/dbg/sites.py $wl > $O/trace_$wl.log 2> $O/trace_$wl.err
  tail -3 $O/trace_$wl.log
done
