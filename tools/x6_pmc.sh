# SQ counters + HBM traffic of gemm_x6_kernel on the update shapes (tools/gemm_x6_bench.py); counters in passes of their own
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/x6pmc; rm -rf $O; mkdir -p $O
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS"
rocprofv3 --pmc $SQ1 --output-format csv -d $O/sq1 -- python3 $R/tools/gemm_x6_bench.py ${1:-2048} > $O/sq1.log 2>&1
rocprofv3 --pmc $SQ2 --output-format csv -d $O/sq2 -- python3 $R/tools/gemm_x6_bench.py ${1:-2048} > $O/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/tools/gemm_x6_bench.py ${1:-2048} > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/gemm_x6_bench.py ${1:-2048} > $O/trace.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ("sq1", "sq2", "fetch"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % tag, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "x6" not in k: continue
            acc[(k[:40], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            print(tag, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n", len(next(iter(d.values()))))
for f in glob.glob("$O/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "x6" in r["Name"] or "splitk" in r["Name"]: print(r["Name"][:50], r["Calls"], r["AverageNs"])
PY
