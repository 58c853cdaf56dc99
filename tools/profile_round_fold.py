#!/usr/bin/env python3
"""Fold the raw rocprofv3 outputs of tools/profile_round.sh into the small files that go to profiles/:
    python tools/profile_round_fold.py <round, e.g. r5> <raw dir> <out dir>
  <round>_manifest.json       what the counters were measured on: sha256 of the kernel sources (bench.csrc_sha256: bench.py only
                              quotes roofline.traffic from a traffic file whose manifest matches the running tree), sha256 of the
                              three product libraries, the bench commands (git HEAD is added in the container: the box has no .git)
  <round>_kernel_stats_<cfg>.csv   the --stats per-kernel table of each traced bench run (top 40 rows)
  <round>_bench_profiled_<cfg>.json   the JSON line bench.py printed under the profiler
  <round>_traffic.json           HBM bytes per launch per kernel: (2 x FETCH_SIZE + WRITE_SIZE) KB, FETCH doubled as
                              MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads
  <round>_pmc_sq.json           SQ counters per kernel (two passes), with the derived shares the guide names:
                              parked = WAIT_ANY / WAVE_CYCLES, issue_stall = WAIT_INST_ANY / WAVE_CYCLES,
                              mfma_busy = VALU_MFMA_BUSY_CYCLES / (32 x BUSY_CYCLES: SIMD cycles), lds_conflict =
                              LDS_BANK_CONFLICT / LDS_IDX_ACTIVE, valu_per_mfma = INSTS_VALU / MFMA instructions"""
import csv
import glob
import json
import os
import re
import sys

rnd, raw, out = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(out, exist_ok=True)


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:90]


for cfg in ("a3c", "conv32", "gru", "shard"):
    fs = glob.glob(os.path.join(raw, f"trace_{cfg}", "**", "*kernel_stats.csv"), recursive=True)
    if fs:
        rows = list(csv.DictReader(open(max(fs, key=os.path.getmtime))))[:40]
        with open(os.path.join(out, f"{rnd}_kernel_stats_{cfg}.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    j = os.path.join(raw, f"trace_{cfg}.json")
    if os.path.exists(j):
        lines = [l for l in open(j) if l.startswith("{")]
        if lines:
            open(os.path.join(out, f"{rnd}_bench_profiled_{cfg}.json"), "w").write(lines[-1])


def counters(d):
    """{kernel: {counter: (sum over dispatches, n dispatches)}}"""
    res = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            k = (short(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])
            per[k] = per.get(k, 0.0) + float(r["Counter_Value"])
        for (kn, cn, _), v in per.items():
            a = res.setdefault(kn, {}).setdefault(cn, [0.0, 0])
            a[0] += v
            a[1] += 1
    return res


traffic = {}
for cfg in ("a3c", "gru"):
    fe, wr = counters(os.path.join(raw, f"pmc_{cfg}_FETCH_SIZE")), counters(os.path.join(raw, f"pmc_{cfg}_WRITE_SIZE"))
    for kn in sorted(set(fe) & set(wr)):
        f_, nf = fe[kn]["FETCH_SIZE"]
        w_, nw = wr[kn]["WRITE_SIZE"]
        hbm = (2.0 * f_ / nf + w_ / nw) * 1024.0
        if hbm < 4e6:
            continue
        traffic[f"{cfg}:{kn}"] = dict(fetch_KB_per_launch=round(f_ / nf, 1), write_KB_per_launch=round(w_ / nw, 1),
                                     hbm_bytes_per_launch=round(hbm), launches_averaged=[nf, nw])
for k in list(traffic):                 # the key bench.py looks the headline kernel's traffic up by
    if "a3c_ring_kernel" in k:      # bench.py's default headline keeps the single-frame store (no fp32 state rows)
        traffic["a3c_ring_lazy"] = dict(traffic[k], alias_of=k)
json.dump(traffic, open(os.path.join(out, f"{rnd}_traffic.json"), "w"), indent=1)

sq = {}
for cfg in ("a3c", "gru"):
    a, b = counters(os.path.join(raw, f"pmc_{cfg}_sq1")), counters(os.path.join(raw, f"pmc_{cfg}_sq2"))
    for kn in sorted(set(a) | set(b)):
        c = {}
        for src in (a.get(kn, {}), b.get(kn, {})):
            for cn, (v, n) in src.items():
                c[cn] = v / max(n, 1)
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        if wc < 1e5:
            continue
        mf = c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0)
        d = dict(counters={k: round(v) for k, v in c.items()})
        d["parked_share"] = round(c.get("SQ_WAIT_ANY", 0.0) / wc, 3)
        d["issue_stall_share"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
        d["issuing_share"] = round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c.get("SQ_BUSY_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") is not None:
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 CUs = 32 SIMDs each), SQ_VALU_MFMA_BUSY_CYCLES over the SIMDs
            # (= 8 cycles per 512-flop MFMA op: check against SQ_INSTS_VALU_MFMA_MOPS_F32): busy share of the SIMD cycles
            d["mfma_busy_share_of_simd_cycles"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (32.0 * c["SQ_BUSY_CYCLES"]), 3)
            if c.get("SQ_INSTS_VALU") and c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"):
                d["valu_per_mfma_op"] = round(c["SQ_INSTS_VALU"] / c["SQ_INSTS_VALU_MFMA_MOPS_F32"], 2)
        sq[f"{cfg}:{kn}"] = d
json.dump(sq, open(os.path.join(out, f"{rnd}_pmc_sq.json"), "w"), indent=1)
print("kernels with traffic:", len(traffic), "with SQ counters:", len(sq))

# ---- manifest
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402
libs = {}
for n in ("liba2c_mi355x.so", "liba2c_hostpool.so", "liba2c_torch_ops.so"):
    pth = os.path.join(root, "pytorch-a2c_amd", "a2c_amd", n)
    if os.path.exists(pth):
        libs[n] = hashlib.sha256(open(pth, "rb").read()).hexdigest()
json.dump(dict(round=rnd, csrc_sha256=bench.csrc_sha256(), libraries_sha256=libs, git_head=None,
               commands=["python3 bench.py --steps 20 --warmup 3 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline (trace, headline)",
                         "... --workload conv | --workload gru_bptt | --workload conv --n-envs 256 (traces)",
                         "--pmc FETCH_SIZE / WRITE_SIZE / two SQ passes on the headline and on gru_bptt, --no-kernel-timers"],
               note="git_head is filled in by tools/profile_commit.py in the container (the GPU box has no .git); bench.py compares "
                    "csrc_sha256 with the running tree before it quotes roofline.traffic from the traffic file of this round"),
          open(os.path.join(out, f"{rnd}_manifest.json"), "w"), indent=1)
