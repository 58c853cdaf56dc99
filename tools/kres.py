#!/usr/bin/env python3
"""Register / scratch use of the kernels of one .hip file: python tools/kres.py conv.hip [name-filter]"""
import os, re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pytorch-a2c_amd", "csrc")
out = "/tmp/_kres.s"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-ffp-contract=off", "--offload-arch=gfx950", "-std=c++17",
                "--cuda-device-only", "-S", os.path.join(csrc, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
txt = open(out).read()
for b in txt.split("  - .agpr_count:")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    if flt not in name:
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).split("(")[0].replace("void ", "")
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print(f"{dem:50s} vgpr {g('vgpr_count'):>4s} agpr {b.splitlines()[0].strip():>4s} spill {g('vgpr_spill_count'):>3s} "
          f"scratch {g('private_segment_fixed_size'):>5s} sgpr {g('sgpr_count'):>4s}")
