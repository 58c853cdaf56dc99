#!/bin/bash
# PMC passes on isolated generic conv kernels (GRUModel layer 2: 16->24 3x3 s2 on 84x84), B samples
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_conv_pmc
mkdir -p $O
SPEC=${1:-16,84,84,24,3,2,1}
B=${2:-8192}
for what in fwd bwd_data wgrad; do
  python3 $R/tools/run_kernel.py spec:$SPEC:$what $B 3 2>&1 | grep -v amdgpu | tail -1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/${what}_$c -- python3 $R/tools/run_kernel.py spec:$SPEC:$what $B 2 > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/${what}_sq -- python3 $R/tools/run_kernel.py spec:$SPEC:$what $B 2 > /dev/null 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_WAVES --output-format csv -d $O/${what}_sq2 -- python3 $R/tools/run_kernel.py spec:$SPEC:$what $B 2 > /dev/null 2>&1
done
cd $R && python3 - <<PY
import csv, glob, collections
O = "gpurun_out/prof_conv_pmc"
for what in ("fwd", "bwd_data", "wgrad"):
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for d in glob.glob(f"{O}/{what}_*"):
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"][:60]
                tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])].add(r["Dispatch_Id"])
    for k, cs in tot.items():
        if "prep" in k or "elementwise" in k or "distribution" in k or "reduce" in k: continue
        print(what, k, {c: round(v / len(cnt[(k, c)]), 1) for c, v in cs.items()})
PY
