cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_gru
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --workload gru_bptt --steps 5 --warmup 2 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline > $O/trace_gru.json 2> $O/trace_gru.err
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
cp $f $R/gpurun_out/gru_kernel_stats.csv
rm -rf $O/t
