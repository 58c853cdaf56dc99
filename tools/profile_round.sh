#!/bin/bash
# Profiling passes of a round, run as the LAST GPU action of the round (through gpurun):   tools/profile_round.sh r5
# Raw outputs under /tmp (scratch); what travels back is the folded summaries in gpurun_out/prof_<round>_folded/, to be
# copied to profiles/ and committed together with <round>_manifest.json (HEAD is filled in by tools/profile_commit.py
# in the container: the GPU box has no .git).  The program after `--` is always python3 bench.py itself (native env
# threads: no child process under the profiler); counters (--pmc) run in passes of their own, never together with a trace.
set -x
RND=${1:-r6}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=/tmp/prof_$RND
rm -rf $O; mkdir -p $O
COMMON="--sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
# 1. kernel trace + stats: headline (A3CModel 256x128, packed transport, ring kernel) and the other BASELINE configs
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_a3c -- python3 $R/bench.py --steps 20 --warmup 3 $COMMON > $O/trace_a3c.json 2> $O/trace_a3c.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_conv32 -- python3 $R/bench.py --workload conv --steps 10 --warmup 3 $COMMON > $O/trace_conv32.json 2> $O/trace_conv32.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_gru -- python3 $R/bench.py --workload gru_bptt --steps 5 --warmup 2 $COMMON > $O/trace_gru.json 2> $O/trace_gru.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_shard -- python3 $R/bench.py --workload conv --n-envs 256 --grey --steps 3 --warmup 2 $COMMON > $O/trace_shard.json 2> $O/trace_shard.err
# 2. HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the headline's kernels and of the GRU+BPTT config's
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_a3c_$c -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_a3c_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_gru_$c -- python3 $R/bench.py --workload gru_bptt --steps 2 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_gru_$c.log 2>&1
done
# 3. SQ counters (two passes of 8) on the same two runs
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_a3c_sq1 -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_a3c_sq1.log 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d $O/pmc_a3c_sq2 -- python3 $R/bench.py --steps 4 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_a3c_sq2.log 2>&1
rocprofv3 --pmc $SQ1 --output-format csv -d $O/pmc_gru_sq1 -- python3 $R/bench.py --workload gru_bptt --steps 2 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_gru_sq1.log 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d $O/pmc_gru_sq2 -- python3 $R/bench.py --workload gru_bptt --steps 2 --warmup 2 $COMMON --no-kernel-timers > $O/pmc_gru_sq2.log 2>&1
python3 $R/tools/profile_round_fold.py $RND $O $O/folded
ls $O/folded
# only the folded summaries travel back (gpurun_out is capped at 64 MiB)
mkdir -p $R/gpurun_out/prof_${RND}_folded && cp $O/folded/* $R/gpurun_out/prof_${RND}_folded/ && cp $O/*.err $R/gpurun_out/prof_${RND}_folded/ 2>/dev/null
rm -rf $O
