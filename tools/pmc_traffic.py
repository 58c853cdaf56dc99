#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes into profiles/<round>_traffic.json.

    python tools/pmc_traffic.py KEY KERNEL_SUBSTRING ALG_BYTES_PER_LAUNCH FETCH_DIR WRITE_DIR [out.json]

FETCH_DIR / WRITE_DIR: output directories of two SEPARATE passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`)
over the same command.  Units and the gfx950 correction follow MI355X_MICROARCH.md (HBM section):
both counters are in KB; FETCH_SIZE counts the 128-B requests of wide coalesced reads as 64 B, so it
is doubled.  Values are averaged over the launches of the matching kernel."""
import csv
import glob
import json
import os
import sys


def avg_counter(d, name, sub):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = {}
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and sub in r["Kernel_Name"]:
                per[r["Dispatch_Id"]] = per.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        tot += sum(per.values())
        n += len(per)
    if n == 0:
        raise SystemExit(f"no {name} rows for kernel '{sub}' under {d}")
    return tot / n, n


def main():
    key, sub, alg, fdir, wdir = sys.argv[1], sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5]
    out = sys.argv[6] if len(sys.argv) > 6 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "profiles", "r1_traffic.json")
    fetch, nf = avg_counter(fdir, "FETCH_SIZE", sub)
    write, nw = avg_counter(wdir, "WRITE_SIZE", sub)
    hbm = (2.0 * fetch + write) * 1024.0
    data = json.load(open(out)) if os.path.exists(out) else {}
    data[key] = dict(fetch_KB_per_launch=fetch, write_KB_per_launch=write, hbm_bytes_per_launch=hbm,
                     algorithmic_bytes_per_launch=alg, traffic_over_algorithmic=round(hbm / alg, 3), kernel=sub,
                     launches_averaged=[nf, nw])
    json.dump(data, open(out, "w"), indent=1)
    print(key, json.dumps(data[key]))


if __name__ == "__main__":
    main()
