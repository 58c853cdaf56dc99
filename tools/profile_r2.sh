#!/bin/bash
# Round-2 profiling passes (run on the GPU box through gpurun; outputs under gpurun_out/prof_r2/).
# The bench runs with native env threads only: no child process is spawned under the profiler.
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r2
mkdir -p $O
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline"
# 1. kernel trace + stats of the headline bench (host-pinned ingest)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $BENCH > $O/trace.json 2> $O/trace.err
# 2. the same with the device tape (per-launch step kernel, round-1 style)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_tape -- $BENCH --ingest device-tape > $O/trace_tape.json 2> $O/trace_tape.err
# 3. PMC passes on the isolated step kernel (tools/run_kernel.py step 256): HBM traffic, then SQ counters
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_step_$c -- python3 $R/tools/run_kernel.py step 256 3 > $O/pmc_step_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $O/pmc_step_sq1 -- python3 $R/tools/run_kernel.py step 256 3 > $O/pmc_step_sq1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS --output-format csv -d $O/pmc_step_sq2 -- python3 $R/tools/run_kernel.py step 256 3 > $O/pmc_step_sq2.log 2>&1
# 4. HBM traffic of the persistent rollout kernel and of the update kernels in the headline bench
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_bench_$c -- $BENCH --steps 4 > $O/pmc_bench_$c.log 2>&1
done
ls -R $O | head -60
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $O/sq_counters.txt
