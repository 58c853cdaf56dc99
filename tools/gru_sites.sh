#!/bin/bash
# per-site update timers of the GRU+BPTT config:  tools/gru_sites.sh [ENV=val ...]
env "$@" timeout 500 python bench.py --workload gru_bptt --steps 6 --warmup 2 --sustain-steps 0 --no-configs --no-secondary --no-cpu-baseline > /dev/null 2>/tmp/gru_err.log
python - <<PY
import json
d=json.load(open("gpurun_out/bench_full_gru_bptt_n1.json"))
ls=d.get("update_launch_sites_ms") or {}
print("$*", d["value"], d["rollout_ms"], d["update_ms"], " ".join(f"{k}={v['avg_ms']:.3f}" for k,v in ls.items() if k.startswith("conv") and "data" in k))
PY
