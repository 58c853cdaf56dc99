import os, sys, glob, subprocess
for p in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(p, open(p).read().strip())
print(subprocess.run("lscpu | grep -i numa", shell=True, capture_output=True, text=True).stdout)
for f in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
    t = open(f).read()
    if "simd_count 0" in t.splitlines()[1:3] or "cpu_cores_count 0" not in t:
        continue
    print(f)
