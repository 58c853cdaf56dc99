#!/usr/bin/env python3
"""Container side of tools/profile_round.sh: copy gpurun_out/prof_<round>_folded/* to profiles/, stamp the manifest with git
HEAD, and refuse when the tree's kernel sources differ from what was profiled.   python tools/profile_commit.py r5"""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

rnd = sys.argv[1] if len(sys.argv) > 1 else "r5"
src = os.path.join(ROOT, "gpurun_out", f"prof_{rnd}_folded")
man = json.load(open(os.path.join(src, f"{rnd}_manifest.json")))
if man["csrc_sha256"] != bench.csrc_sha256():
    raise SystemExit(f"kernel sources changed since the profile was taken ({man['csrc_sha256'][:12]} vs {bench.csrc_sha256()[:12]}): re-run tools/profile_round.sh")
man["git_head"] = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
for n in sorted(os.listdir(src)):
    if n.endswith((".csv", ".json")):
        shutil.copy(os.path.join(src, n), os.path.join(ROOT, "profiles", n))
json.dump(man, open(os.path.join(ROOT, "profiles", f"{rnd}_manifest.json"), "w"), indent=1)
print("profiles/ updated from", src, "HEAD", man["git_head"][:12])
