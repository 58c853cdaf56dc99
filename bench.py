#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the A2C rollout+update hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload a3c|conv|gru|gru_bptt|fc] [--ingest ...]

One "step" = one pass of the hot path over one batch: a rollout of n_envs x n_tsteps env-steps followed
by one Updater.update_model (GAE/returns scan, forward, loss, backward, clip + optimiser).

Where the frames come from (``--ingest``, default ``host-pinned``): the north star keeps env stepping on
the host.  The synthetic envs (84x84 binary uint8 frames, rewards, dones: SURVEY.md 8d) therefore run in
HOST env workers; every env step's frame crosses PCIe from pinned host memory inside the timed region:
  host-pinned  worker threads/processes behind one pinned, device-mapped region; A3CModel: the persistent
               one-launch rollout kernel reads the uint8 frames zero-copy (a2c_a3c_rollout); other models:
               hipMemcpyAsync of the frames block per step ("memcpy" forces that for A3CModel too);
  device-tape  round-1 style: the same tape pre-generated in HBM, the rollout replayed as a hipGraph --
               reported as the secondary key ``value_device_tape`` (kernel-side evidence, no ingest).
``--env-workers native`` (default) steps the tape envs in C threads of liba2c_hostpool (no Python per env
step); ``process`` steps Python TapeEnv objects in worker processes (a2c_amd.hostpool_worker), reported as
``host_pinned_process_workers``.

Multi-GPU: weak scaling, every rank plays its own n_envs envs (own pool); one RCCL all-reduce of the flat
gradient arena (+ 2 tiny ones for the whole-batch statistics) per update.  ``python bench.py --gpus N``
without a launcher spawns ``torch.distributed.run`` itself (before this process touches the GPU).

Prints ONE JSON line (rank 0, <= 4 KB: compact_line), the full report goes to gpurun_out/bench_full_<tag>.json and stderr:
  "roofline":     the kernel the epoch spends most time in.  For the persistent ring rollout kernel `frac` is PER PIPE:
                  achieved / the rate at which this instruction mix would run with every MFMA at its own pipe's peak
                  (conv1 is issued as three exact bf16 products, the rest as fp32 MFMAs: ring_roofline) = matrix time at
                  each instruction's own peak / launch duration; `frac_fp32_equiv` = algorithmic fp32 flops over the fp32
                  peak (rounds 1-5's figure); `traffic` = HBM bytes per launch from profiles/<round>_traffic.json when its
                  manifest matches the running tree,
  "cpu_baseline": the CPU oracle (restatement of the reference's algorithm) timed on this box's usable host
                  cores on a bounded sample of the same workload (N=1 only); "cpu_baseline_values": the same for the other
                  configs of the line (side file: cpu_baselines),
  value_<config>: the other BASELINE.json configs (conv 32x64, gru+BPTT 256x128, a3c at 32 / 2048 envs, and the per-GPU shard
                  of configs[4]: ConvModel 256x128 on Breakout-like grey frames over the uint8 transport, --grey),
  value_one_env_thread / host_us_per_env_step / predicted_8rank_weak: the headline with ONE env thread (what a rank of an
                  8-rank run on a 16-CPU quota has) and 8 x that -- a prediction, not a scaling measurement.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "pytorch-a2c_amd"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {   # model, n_envs, n_tsteps, use_bptt, n_actions
    "a3c": ("A3CModel", 256, 128, False, 3),        # headline: Pong-v0, a3c, 256 x 128
    "conv": ("ConvModel", 32, 64, False, 3),        # configs[1]; --n-envs 256 with T=128 is the per-GPU shard of configs[4]
    "gru": ("GRUModel", 256, 128, False, 3),
    "gru_bptt": ("GRUModel", 256, 128, True, 3),    # configs[3]
    "fc": ("FCModel", 256, 128, False, 3),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # fp32 matrix == fp32 vector peak
BF16_PEAK_TFLOPS = 2516.6    # dense bf16 MFMA (16 x the fp32 matrix rate, MI355X_MICROARCH.md)
PCIE_PEAK_GBS = 63.0         # host link: PCIe Gen5 x16 (spec)
SS = (4, 84, 84)
FRAME_BYTES = {"u8": 84 * 84, "bits": 84 * 84 // 8}      # bytes per frame over the host link, by transport


def hyps_for(model, n_envs, T, use_bptt, optim):
    return dict(gamma=.99, lambda_=.98, n_tsteps=T, n_rollouts=n_envs, n_envs=n_envs, n_frame_stack=4, action_shift=0,
                render=False, env_type="Pong-synthetic", use_bptt=use_bptt, use_nstep_rets=False, norm_advs=True,
                entr_coef=.005, pi_coef=1.0, val_coef=.5, max_norm=.5, lr=1e-4, optim_type=optim, is_discrete=True,
                h_size=256, model=model, env_timeout_s=30.0)


class SyntheticDevicePool:
    """Device-resident synthetic env pool: i.i.d. binary 84x84 frames, rewards -1/0/+1 with P=(.02,.96,.02),
    real done with P=1/800; the same T-step tape is replayed every epoch.  Implements the Runner's device-pool
    protocol (start / device_step).  Only behind ``--ingest device-tape`` / the ``value_device_tape`` key."""

    def __init__(self, n_envs, T, device, seed, frame_shape=(1, 84, 84)):
        self.B, self.T, self.frame_shape = n_envs, T, frame_shape
        hw = int(np.prod(frame_shape))
        g = torch.Generator(device=device).manual_seed(1234 + seed)
        self.frames = (torch.rand((T + 1, n_envs, hw), device=device, generator=g) < 0.25).float()
        r = torch.rand((T, n_envs), device=device, generator=g)
        self.rew = (r < 0.02).float() - (r > 0.98).float()
        self.done = (torch.rand((T, n_envs), device=device, generator=g) < 1.0 / 800).float()
        self.uniforms = torch.rand((T, n_envs), device=device, generator=g)

    def __len__(self):
        return self.B

    def start(self, runner):
        from a2c_amd import ops
        ones = torch.ones(self.B, device=self.frames.device)
        ops.frame_stack_push(self.frames[self.T], ones, runner.bookmark.data_ptr(), runner.S,
                             runner.bookmark.data_ptr(), runner.S, self.B, runner.C, runner.HW)

    def device_step(self, t, env0, B):
        sl = slice(env0, env0 + B)
        return self.frames[t, sl], self.rew[t, sl], self.done[t, sl], self.done[t, sl]


def make_host_pool(n_envs, T, kind, n_workers, seed, transport="bits", grey=False):
    """host env workers stepping synthetic.TapeEnv tapes (binary uint8 frames, like pong_prep's) behind the pinned
    region; transport "bits": the workers pack every frame to one bit per pixel on its way into the pinned slot.
    ``grey``: Breakout-like tapes (grey levels 0..255, what breakout_prep hands on: preprocessing.py:19-23) -- uint8
    transport only (the packed one refuses them), real dones only (no Pong done-on-reward override, runner.py:213-214)"""
    from a2c_amd.hostpool import ProcessEnvPool, ThreadEnvPool
    from a2c_amd.synthetic import TapeEnv
    L = min(T + 1, 33)          # tape length per env: content does not affect cost, keeps host memory small
    kws = [dict(env_id=seed * 100000 + j, length=L, grey=grey) for j in range(n_envs)]
    bits = transport == "bits"
    if grey and bits:
        raise ValueError("grey-level frames need the uint8 transport")
    if kind == "native":
        return ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=n_workers, pong=not grey, frame_bits=bits)
    return ProcessEnvPool(TapeEnv, n_envs, env_kwargs=kws, n_workers=n_workers, pong=not grey, frame_shape=(1, 84, 84),
                          frame_dtype=np.uint8, frame_bits=bits)



# ---------------------------------------------------------------- the ONE line the driver parses
LINE_MAX = 4096      # the driver keeps an 8 KB tail of stdout: the line must fit it with room to spare (tests/test_bench_line.py)
_CFG_SHORT = (("conv_32x64", "conv_32x64"), ("gru_bptt_256x128", "gru_bptt_256x128"), ("a3c_32", "a3c_32"),
              ("a3c_2048", "a3c_2048"), ("conv_2048x128_per_gpu_shard_256x128", "conv_shard_256x128"))


def _finite(x):
    """strict JSON has no NaN / Infinity: such a figure is reported as null"""
    if isinstance(x, float) and (x != x or x in (float("inf"), float("-inf"))):
        return None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    return x


def compact_line(full, side_file=None):
    """The printed line: the contract's keys, `config` as short tokens, `roofline`, `cpu_baseline` and the scalar copies of
    the secondary figures.  Everything else of `full` (per-site tables of the other configs, secondary legs, notes) lives in
    the side file and on stderr -- round 4's 22.7 KB line did not fit the driver's stdout tail and was recorded unparsed."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg[k] for k in ("workload", "n_envs", "n_tsteps", "optimizer", "transport", "ingest", "states_layout",
                                          "update", "info_readback", "env_workers", "usable_host_cpus", "parallelism")
                      if k in cfg}
    for k in ("rollout_ms", "update_ms"):
        line[k] = full.get(k)
    rf = full.get("roofline")
    if rf:
        line["roofline"] = {k: rf[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale",
                                               "traffic_source", "alg_flops_per_launch", "alg_bytes_per_launch", "avg_launch_us",
                                               "launches_per_rollout", "hbm_GBs", "pipes", "frac_fp32_equiv", "frac_of_pipe_time") if k in rf}
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "cpu_model", "kind", "rollout_steps_per_s",
                                                   "update_samples_per_s", "update_batch", "extrapolated", "sample", "error")
                                if k in cb}
    sus = full.get("sustained")
    if sus:
        line["sustained"] = {k: sus[k] for k in ("steps", "ms_per_step", "value") if k in sus}
    # scalar copies of the nested figures the verdicts quote
    sat = (full.get("scan_roofline") or {}).get("saturating_2^19x128")
    if sat:
        line["scan_roofline_frac_saturating"] = sat["frac"]
        line["scan_roofline_GBs_saturating"] = sat["achieved_GBs"]
    cfgs = full.get("configs") or {}
    ms = {}
    for key, short in _CFG_SHORT:
        v = cfgs.get(key) or {}
        if v.get("value") is not None:
            line["value_" + short] = v["value"]
            ms[short] = [v.get("rollout_ms"), v.get("update_ms")]
    if ms:
        line["configs_rollout_update_ms"] = ms
    cbs = full.get("cpu_baselines") or {}
    cbv = {short: (cbs.get(key) or {}).get("value") for key, short in _CFG_SHORT if (cbs.get(key) or {}).get("value")}
    if cbv:
        line["cpu_baseline_values"] = cbv          # env-steps/s of the oracle port on this box's cores, per config
    for key in ("value_device_tape", "value_states_rows_written", "host_pinned_u8_transport", "host_pinned_process_workers",
                "value_one_env_thread"):
        v = full.get(key)
        if isinstance(v, dict) and v.get("value") is not None:
            line[key if key.startswith("value_") else "value_" + key] = v["value"]
    for k in ("host_us_per_env_step", "rccl_ranks", "dist_backend", "allreduce_ms_per_update", "allreduce_bytes"):
        if full.get(k) is not None:
            line[k] = full[k]
    p8 = full.get("predicted_8rank_weak")
    if isinstance(p8, dict) and p8.get("value") is not None:
        line["predicted_8rank_weak"] = p8["value"]       # 8 x the one-env-thread leg: a prediction, not a measurement
    if rf and rf.get("frac") is not None:
        line["roofline_frac"] = rf["frac"]
    if (cb or {}).get("value") and full.get("value"):
        line["speedup_vs_cpu_baseline"] = round(full["value"] / cb["value"], 1)
    if side_file:
        line["full_report"] = side_file
    line = _finite(line)
    s = json.dumps(line, allow_nan=False)
    if len(s) > LINE_MAX:       # never hand the driver a line it cannot keep: drop the prose first, then the extras
        for victim in (("cpu_baseline", "sample"), ("roofline", "traffic_source"), ("config", "info_readback"),
                       ("config", "ingest"), ("configs_rollout_update_ms", None), ("sustained", None)):
            if victim[1] is None:
                line.pop(victim[0], None)
            elif isinstance(line.get(victim[0]), dict):
                line[victim[0]].pop(victim[1], None)
            s = json.dumps(line, allow_nan=False)
            if len(s) <= LINE_MAX:
                break
    return s


def write_side_file(full, tag):
    """the full report (every table the line no longer carries) next to the line: gpurun_out/bench_full_<tag>.json"""
    d = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, f"bench_full_{tag}.json")
        with open(path, "w") as f:
            json.dump(_finite(full), f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        print(f"[bench] could not write the full report ({e})", file=sys.stderr)
        return None

# ---------------------------------------------------------------- CPU baseline (oracle, host cores)
def _cpu_rollout_worker(args):
    model, T, A, seconds = args
    import torch as th
    th.set_num_threads(1)
    from oracle import a2c_oracle as O
    net = O.OracleNet(model, SS, A, 256)
    hyps = hyps_for(model, 1, T, False, "RMSprop")
    n_slots = 4
    N = T * n_slots
    D = dict(states=th.zeros(N, *SS), deltas=th.zeros(N), rewards=th.zeros(N), dones=th.zeros(N),
             actions=th.zeros(N).long())
    if net.is_recurrent:
        D["h_states"] = th.zeros(N, 256)
    frames = (np.random.default_rng(0).random((64, 1, 84, 84)) < 0.25).astype(np.float64)

    class TapeEnv:          # env cost excluded on both sides: frames come from a pre-generated tape
        t = 0

        def reset(self):
            return frames[0]

        def step(self, a):
            self.t += 1
            return frames[self.t % 64], (1.0 if self.t % 50 == 0 else 0.0), self.t % 800 == 0, {}
    r = O.SlotRunner(TapeEnv(), D, hyps)
    r.start(net)
    r.rollout(net, 0)                       # warm-up slot
    t0 = time.perf_counter()
    done_slots = 0
    while time.perf_counter() - t0 < seconds:
        r.rollout(net, 1 + done_slots % (n_slots - 1))
        done_slots += 1
    return done_slots * T, time.perf_counter() - t0


def cpu_baseline(model, n_envs, T, use_bptt, A, optim, roll_s=5.0, budget=12.0):
    """Oracle timed like the reference runs on the host cores this container may use: rollout =
    min(n_envs, cores) processes x batch-1 forwards, 1 torch thread each (no scale-up); update = one process,
    all cores, on as large a batch as ~12 s allow (full N when it fits).  env-steps/sec of one epoch of the
    workload = N / (N / rollout_rate + N / update_rate)."""
    import multiprocessing as mp
    from a2c_amd.hostpool import usable_cpus
    from oracle import a2c_oracle as O
    cores = usable_cpus()
    workers = max(1, min(cores, n_envs))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_rollout_worker, [(model, T, A, roll_s)] * workers)
    roll_rate = sum(r[0] / r[1] for r in res)          # aggregate env-steps/s of `workers` processes, measured
    torch.set_num_threads(cores)
    net = O.OracleNet(model, SS, A, 256)
    N_full = n_envs * T

    def time_update(R):
        hyps = hyps_for(model, R, T, use_bptt, optim)
        N = R * T
        g = torch.Generator().manual_seed(0)
        D = dict(states=(torch.rand(N, *SS, generator=g) < 0.25).float(), deltas=torch.randn(N, generator=g),
                 rewards=torch.randn(N, generator=g).round(), dones=(torch.rand(N, generator=g) < 0.01).float(),
                 actions=torch.randint(0, A, (N,), generator=g))
        D["dones"][T - 1::T] = 1
        if net.is_recurrent:
            D["h_states"] = torch.randn(N, 256, generator=g)
        upd = O.OracleUpdater(net, hyps)
        t1 = time.perf_counter()
        upd.update_model(D)
        return N, time.perf_counter() - t1

    n_small, dt_small = time_update(4 if model != "ConvModel" else 1)
    n_small, dt_small = time_update(4 if model != "ConvModel" else 1)          # second call: warm
    R_big = int(max(1, min(n_envs, budget / max(dt_small / n_small, 1e-9) / T)))
    N_upd, dt_upd = time_update(R_big)
    upd_rate = N_upd / dt_upd
    value = 1.0 / (1.0 / roll_rate + 1.0 / upd_rate)
    cpu_model = ""
    try:
        cpu_model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:      # noqa: BLE001
        pass
    return dict(value=round(value, 1), unit="env-steps/s", cores=cores, kind="port",
                rollout_steps_per_s=round(roll_rate, 1), update_samples_per_s=round(upd_rate, 1),
                rollout_processes=workers, update_batch=N_upd, extrapolated=bool(N_upd < N_full), cpu_model=cpu_model,
                sample=f"oracle port: rollout {workers} procs x batch-1 fwd {roll_s:g} s each (measured aggregate); update_model "
                       f"N={N_upd}/{N_full}, {cores} torch threads; value=1/(1/r+1/u); wall {time.perf_counter() - t0:.0f}s")


# ---------------------------------------------------------------- roofline helpers
def conv_alg_bytes(d, B, in_bytes_per_sample=None):
    """input + output bytes of one pass over B samples (fp32 both sides unless the input side is given)"""
    return B * ((4.0 * d.Cin * d.H * d.W if in_bytes_per_sample is None else float(in_bytes_per_sample)) + 4.0 * d.Cout * d.OH * d.OW)


def conv_flops(d, B):
    return 2.0 * B * d.Cout * d.OH * d.OW * d.Cin * d.ks * d.ks


def step_alg_flops(A):
    """one rollout step of A3CModel per env: conv 8x8/s4 4->16 on 84x84, conv 4x4/s2 16->32 on 20x20, (A+1) x 2592 heads"""
    return 2 * 16 * 400 * 256 + 2 * 32 * 81 * 256 + 2 * (A + 1) * 2592


def step_alg_bytes(u8_frame=False, stash=True, ring=False, lazy=False):
    """per env-step: 3 planes of the previous state + the new frame in (fp32 or uint8), the new state out
    (+ the conv1 / conv2 activations stashed for the update: 16x20x20 + 32x9x9 floats).  ring: the persistent ring
    kernel keeps the state in LDS -- HBM sees the state row and the stash going out, nothing coming in (the frame
    crosses PCIe)"""
    if ring and lazy:       # single-frame store: one uint8 frame out instead of the 4-plane fp32 row
        return 84 * 84 + (4 * (6400 + 2592) if stash else 0)
    if ring:
        return 4 * 84 * 84 * 4 + (4 * (6400 + 2592) if stash else 0)
    return 3 * 84 * 84 * 4 + (84 * 84 if u8_frame else 84 * 84 * 4) + 4 * 84 * 84 * 4 + (4 * (6400 + 2592) if stash else 0)


def scan_roofline(device):
    """GAE+returns scan at the config size and at a bandwidth-saturating size (20 B/element)."""
    from a2c_amd import ops
    out = {}
    for label, n_seg, T in (("config_256x128", 256, 128), ("saturating_2^19x128", 1 << 19, 128)):
        N = n_seg * T
        x, r = torch.randn(N, device=device), torch.randn(N, device=device)
        d = (torch.rand(N, device=device) < 0.01).float()
        d[T - 1::T] = 1
        a, b = torch.empty_like(x), torch.empty_like(x)
        for _ in range(3):
            ops.gae_returns(x, r, d, .9702, .99, n_seg, T, a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.gae_returns(x, r, d, .9702, .99, n_seg, T, a, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = 20.0 * N / (ms * 1e-3) / 1e9
        out[label] = dict(elements=N, avg_ms=round(ms, 4), achieved_GBs=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
        del x, r, d, a, b
    return out



def csrc_sha256():
    """identity of the kernel sources of the running tree (the GPU box has no .git): sha256 over name + bytes of every file
    under pytorch-a2c_amd/csrc (build products excluded) and of include/*.h"""
    import hashlib
    h = hashlib.sha256()
    files = []
    for d in (os.path.join(ROOT, "pytorch-a2c_amd", "csrc"), os.path.join(ROOT, "include")):
        for n in sorted(os.listdir(d)):
            if n.endswith((".hip", ".h", ".c", ".cpp")) or n == "Makefile":
                files.append(os.path.join(d, n))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def lookup_traffic(key, profiles=None):
    """(hbm_bytes_per_launch | None, source | None, stale) for `key` from the newest profiles/r*_traffic.json whose manifest
    (profiles/r*_manifest.json, written by tools/profile_round.sh as the last GPU action of a round) matches this tree"""
    profiles = profiles or os.path.join(ROOT, "profiles")
    if key is None or not os.path.isdir(profiles):
        return None, None, False
    mine = csrc_sha256()
    stale = False
    rounds = sorted((n for n in os.listdir(profiles) if n.startswith("r") and n.endswith("_traffic.json")),
                    key=lambda n: -int("".join(c for c in n.split("_")[0] if c.isdigit()) or 0))
    for tname in rounds:
        try:
            tj = json.load(open(os.path.join(profiles, tname)))
        except (OSError, ValueError):
            continue
        if key not in tj:
            continue
        mfile = os.path.join(profiles, tname.replace("_traffic.json", "_manifest.json"))
        try:
            man = json.load(open(mfile))
        except (OSError, ValueError):
            man = {}
        if man.get("csrc_sha256") == mine:
            return round(tj[key]["hbm_bytes_per_launch"]), \
                f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, KB; manifest head {str(man.get('git_head'))[:12]})", False
        stale = True
    return None, None, stale


# ---------------------------------------------------------------- one workload on this rank
class Bench:
    """net + rollout buffers + env pool + Runner + Updater of one workload; ``step()`` = rollout + update."""

    def __init__(self, workload, n_envs, optim, ingest, env_workers, n_workers, shard, dev, update_graph=True,
                 transport="bits", frame_store=False, grey=False):
        import a2c_amd
        from a2c_amd.runner import Runner
        from a2c_amd.updater import Updater
        self.model, n0, self.T, self.use_bptt, self.A = WORKLOADS[workload]
        self.n_envs = n_envs or n0
        if workload == "conv" and self.n_envs >= 256:
            self.T = 128                               # configs[4]: n_tsteps=128
        self.hyps = hyps_for(self.model, self.n_envs, self.T, self.use_bptt, optim)
        self.grey = grey
        if grey:            # configs[4] is BreakoutNoFrameskip-v4: 4 actions, grey-level frames, no Pong done override
            self.A = 4
            self.hyps["env_type"] = "Breakout-synthetic"
        if frame_store:     # single-frame uint8 store, first conv layer stacked on load, fp32 states on demand
            self.hyps.update(frame_store=True, lazy_states=True)
        self.frame_store = frame_store
        self.shard, self.dev, self.ingest, self.optim_name = shard, dev, ingest, optim
        torch.manual_seed(20260101)                # the reference's default init, identical on every rank
        self.net = net = getattr(a2c_amd.models, self.model)(list(SS), self.A, h_size=256)
        self.N = N = self.n_envs * self.T
        self.D = D = dict(states=torch.zeros(N, *SS, device=dev), deltas=torch.zeros(N, device=dev),
                          rewards=torch.zeros(N, device=dev), dones=torch.zeros(N, device=dev),
                          actions=torch.zeros(N, dtype=torch.int64, device=dev))
        if net.is_recurrent:
            D["h_states"] = torch.zeros(N, 256, device=dev)
        self.slots = list(range(self.n_envs))
        self.graph = self.ugraph = None
        self.graph_stash = False
        self.env_workers, self.transport = env_workers, transport
        if ingest == "device-tape":
            self.pool = pool = SyntheticDevicePool(self.n_envs, self.T, dev, seed=shard.rank)
            self.runner = Runner(D, self.hyps, None, None, None, env_pool=pool,
                                 uniform_fn=lambda t, B, e0: pool.uniforms[t, e0:e0 + B])
        else:
            self.pool = pool = make_host_pool(self.n_envs, self.T, env_workers, n_workers, seed=shard.rank, transport=transport,
                                              grey=grey)
            self.runner = Runner(D, self.hyps, None, None, None, env_pool=pool,
                                 ingest="memcpy" if ingest == "memcpy" else None)
        self.updater = Updater(net, self.hyps, shard=shard)
        self.want_update_graph = update_graph
        self.info = None
        self._pending = None

    # the rollout: a hipGraph replay (device tape) or the live host-pinned ingest
    def rollout(self):
        if self.graph is not None:
            self.graph.replay()
            self.net._dirty = False        # the captured rollout re-derived the inference weights (conv fragments, Wc)
            if self.graph_stash:           # ... and stashed the conv activations of every state for the update
                self.net.stash_commit(self.D["states"], self.N)
        else:
            self.runner.rollout(self.net, self.slots, self.hyps)

    def update(self):
        if self.ugraph is not None:
            return self.ugraph.replay()
        return self.updater.update_model(self.D)

    def update_async(self):
        """enqueue the update; its five scalars are collected one step later (Updater.collect): the host then enqueues the
        NEXT rollout -- which needs nothing from the host -- while this update still runs, instead of leaving the GPU idle
        behind a blocking read-back + ~150 us of launch preparation per epoch"""
        if self.ugraph is not None:
            return self.ugraph.replay_async()
        return self.updater.update_model_async(self.D)

    def step(self):
        self.rollout()
        self.info = self.update()
        return self.info

    def step_pipelined(self):
        self.rollout()
        tok = self.update_async()
        if self._pending is not None:
            self.info = self.updater.collect(self._pending)       # the PREVIOUS step's losses (its copy finished long ago)
        self._pending = tok

    def drain(self):
        if self._pending is not None:
            self.info = self.updater.collect(self._pending)
            self._pending = None

    def capture(self):
        """after one eager step: the device-tape rollout and (single GPU, RMSprop) the update as hipGraphs"""
        if self.ingest == "device-tape":
            try:
                from a2c_amd import ops as a2c_ops
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                self.net.mark_dirty()
                with a2c_ops.graph_capture(g):
                    self.runner.rollout(self.net, self.slots, self.hyps)
                self.graph = g
                self.graph_stash = self.runner._stash_bufs is not None and self.runner._stash_used
            except Exception as e:      # noqa: BLE001
                print(f"[bench] rollout hipGraph capture failed ({type(e).__name__}: {e}); eager rollout", file=sys.stderr)
                self.graph = None
                torch.cuda.synchronize()
        # the update as hipGraph(s): ~40 dependent launches whose stream gaps shrink to graph-edge gaps.  A sharded
        # update is cut at its two collectives (advantage moments, gradient arena): graph, all-reduce on the stream,
        # graph, all-reduce, graph -- no collective inside a capture (Updater.capture_update).  Adam's step count is a
        # kernel argument, so its update stays eager.
        if self.want_update_graph and self.optim_name == "RMSprop" and os.environ.get("A2C_NO_UPDATE_GRAPH") != "1":
            try:
                # an update always follows a rollout: capture it in that state (the rollout's activation stash is
                # valid, so the captured forward starts behind the conv layers), then replay it once so that the
                # rollout that was just played is consumed like any other
                self.rollout()
                self.ugraph = self.updater.capture_update(self.D)
                self.info = self.update()
            except Exception as e:      # noqa: BLE001
                print(f"[bench] update hipGraph capture failed ({type(e).__name__}: {e}); eager update", file=sys.stderr)
                self.ugraph = None
                torch.cuda.synchronize()

    def timed(self, steps, split=True):
        """time `steps` steps (barrier + sync on both sides); returns (elapsed_s, rollout_ms, update_ms) with the
        two halves from HIP events on the launch stream"""
        ev = []
        if split:
            # the HIP events of the timed region exist BEFORE it starts (torch creates a hipEvent at its first record():
            # three creations per step on the host path between the rollout launch and the update replay cost 3 % of the
            # headline); inside the region a step only re-records them
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
            for e in ev:
                for x in e:
                    x.record()
        self.shard.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipelined = os.environ.get("A2C_BENCH_SYNC_INFO") != "1"      # read the losses back one step late (see update_async)
        for k in range(steps):
            if split:
                e = ev[k]
                e[0].record()
                self.rollout()
                e[1].record()
                if pipelined:
                    tok = self.update_async()
                    if self._pending is not None:
                        self.info = self.updater.collect(self._pending)
                    self._pending = tok
                else:
                    self.info = self.update()
                e[2].record()
            elif pipelined:
                self.step_pipelined()
            else:
                self.step()
        self.drain()
        torch.cuda.synchronize()
        self.shard.barrier()
        elapsed = time.perf_counter() - t0
        if self.ingest != "device-tape":
            self.runner.check()
        r_ms = sum(e[0].elapsed_time(e[1]) for e in ev) / max(len(ev), 1)
        u_ms = sum(e[1].elapsed_time(e[2]) for e in ev) / max(len(ev), 1)
        if ev and os.environ.get("A2C_BENCH_STEP_TIMES") == "1":        # per-step halves of the timed region (stderr)
            print("[bench] per-step rollout ms: " + " ".join(f"{e[0].elapsed_time(e[1]):.2f}" for e in ev[:40]), file=sys.stderr)
            print("[bench] per-step update  ms: " + " ".join(f"{e[1].elapsed_time(e[2]):.2f}" for e in ev[:40]), file=sys.stderr)
            print("[bench] start-to-start   ms: " + " ".join(f"{a[0].elapsed_time(b[0]):.2f}" for a, b in zip(ev[:39], ev[1:40])), file=sys.stderr)
        return elapsed, r_ms, u_ms

    def site_timers(self, n=2):
        """HIP events per launch site of the update, `n` EAGER updates (the timed ones may be graph replays)"""
        from a2c_amd import ops
        timers = ops.KernelTimers()
        for _ in range(n):
            self.rollout()
            ops.TIMERS = timers
            self.updater.update_model(self.D)
            ops.TIMERS = None
        torch.cuda.synchronize()
        return timers.summary()

    def rollout_site_timers(self):
        """HIP events per launch site of ONE eager rollout (per-layer models: the timed rollouts replay a whole-slot
        hipGraph, which has no per-site events); tuners are warm, so these are the kernels the graph replays"""
        from a2c_amd import ops
        if self.ingest == "device-tape" or self.runner._zero_copy_ok(self.net):
            return None
        timers = ops.KernelTimers()
        hy = dict(self.hyps, rollout_graphs=False)
        ops.TIMERS = timers
        try:
            self.runner.rollout(self.net, self.slots, hy)
            self.runner.finish()
        finally:
            ops.TIMERS = None
        self.update()                      # consume the rollout like any other
        return timers.summary()

    def describe_ingest(self):
        if self.ingest == "device-tape":
            return "device-tape fp32 (frames pre-generated in HBM, rollout replayed as a hipGraph)"
        if self.runner._zero_copy_ok(self.net):
            mode = "zero-copy persistent rollout kernel"
        elif self.ingest == "memcpy" or os.environ.get("A2C_NO_RELAY") == "1":
            mode = "hipMemcpyAsync per step behind a host hand-off"
        else:
            mode = "per-step hipGraph segments; actions published and frames fetched by the device, a2c_pool_publish_actions / a2c_pool_ingest"
        tr = ("packed 1 bit/pixel (binary preprocessor, 882 B/frame over the link, expanded to uint8 {0,1} on the device)"
              if self.transport == "bits" else "uint8 (7056 B/frame over the link)")
        return (f"host-pinned {tr} ({mode}; {self.pool.n_workers} {'native env threads' if self.env_workers == 'native' else 'env worker processes'}"
                f" behind one pinned device-mapped region)")

    def close(self):
        try:
            self.runner.close()
        except Exception:      # noqa: BLE001
            pass


def linear_on_bf16_x6(batch, n, k, what):
    """does `linear.<what> NxK` over `batch` rows run as gemm_x6_kernel?  (gemm.hip: x9_eligible -- every dimension >= 1024 and
    2e10 multiply-adds -- or ConvModel's resize_emb forward at rollout batch 128..256 from the prebuilt weight image)"""
    if os.environ.get("A2C_GEMM_X9", "") in ("0", "1"):
        return False
    if min(batch, n, k) >= 1024 and float(batch) * n * k >= 2e10:
        return True
    return (what == "fwd" and 128 <= batch <= 256 and n * k >= (1 << 24) and os.environ.get("A2C_NO_X6_FWD") != "1")


def a3c_x6_layers(model, batch):
    """layers whose two backward passes run as six bf16 piece products (conv.hip: bwd_x6_kernel, wgrad_x6_kernel): A3CModel's conv2
    at streaming batch (>= 8 samples per CU), unless switched off"""
    if model == "A3CModel" and batch >= 2048 and os.environ.get("A2C_BWD_X6") != "0" and os.environ.get("A2C_WGRAD_X6") != "0":
        return ("conv2",)
    return ()


def site_roofline(name, site, conv_layers, batch, u8_store=False, bf16_pipe=False, x6_layers=()):
    """roofline entry of one launch site: algorithmic bytes and flops / its average HIP-event duration, each priced on what
    the site really reads and really issues.  ``u8_store``: the first layer reads the single-frame uint8 store (one 7,056-byte
    frame per sample: every frame is a plane of four consecutive states) instead of the 4 x 84 x 84 fp32 row.  ``bf16_pipe``:
    conv1's weight gradient of A3CModel runs on the bf16 matrix pipe as three exact bf16 pieces per fp32 value
    (`wgrad_stream_bf16_kernel`): its matrix time is 3 x flops at the bf16 peak, not flops at the fp32 peak -- priced the
    fp32 way a fast launch "exceeds" the peak (round 5's side report printed 1.20), which is not a roofline."""
    lname, _, what = name.partition(".")
    out = dict(site=name, avg_ms=round(site["avg_ms"], 4), launches=site["launches"])
    if "+" in name:                  # a fused pass (GRUModel: conv2.bwd_data+conv1.bwd_weight): two layers' flops, one layer's bytes
        l2, l1 = conv_layers.get("conv2"), conv_layers.get("conv1")
        if l2 is not None and l1 is not None:
            sec = site["avg_ms"] * 1e-3
            by = batch * (4.0 * l2.d.Cout * l2.d.OH * l2.d.OW + l1.d.H * l1.d.W + l2.d.Cin * l2.d.H * l2.d.W / 8.0)
            out.update(batch=batch, alg_bytes=by, hbm_GBs=round(by / sec / 1e9, 1), frac_of_hbm_peak=round(by / sec / 1e9 / HBM_PEAK_GBS, 4),
                       alg_flops=conv_flops(l2.d, batch) + conv_flops(l1.d, batch),
                       note="layer 2's input gradient stays in LDS: bytes = its dOut + the uint8 frames + the sign words; the "
                            "first layer's weight gradient runs on the bf16 pipe (3 x its flops), layer 2's backward on fp32 MFMAs")
            t_pk = conv_flops(l2.d, batch) / (F32_PEAK_TFLOPS * 1e12) + 3.0 * conv_flops(l1.d, batch) / (BF16_PEAK_TFLOPS * 1e12)
            out["frac_of_pipe_peak"] = round(t_pk / sec, 4)
        return out
    if lname == "linear":            # "linear.<pass> NxK": a dense fp32 GEMM over `batch` rows
        try:
            n_, k_ = (int(v) for v in what.split(" ")[1].split("x"))
            tf = 2.0 * batch * n_ * k_ / (site["avg_ms"] * 1e-3) / 1e12
            if linear_on_bf16_x6(batch, n_, k_, what.split(" ")[0]):
                # gemm_x6_kernel: six exact bf16 piece products per element pair -- 6 x the flops on the bf16 pipe
                out.update(batch=batch, pipe="bf16 MFMA x6 (exact 3-way split of both operands, fp32 accumulate)",
                           tflops_fp32_equiv=round(tf, 2), frac_of_pipe_peak=round(6.0 * tf / BF16_PEAK_TFLOPS, 4))
            else:
                out.update(batch=batch, pipe="fp32 MFMA", tflops=round(tf, 2), frac_of_f32_mfma_peak=round(tf / F32_PEAK_TFLOPS, 4),
                           frac_of_pipe_peak=round(tf / F32_PEAK_TFLOPS, 4))
            if batch <= 64:          # skinny: the weights are the traffic
                gbs = 4.0 * n_ * k_ / (site["avg_ms"] * 1e-3) / 1e9
                out.update(weight_GBs=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4))
        except (ValueError, IndexError):
            pass
    if lname in conv_layers:
        d = conv_layers[lname].d
        sec = site["avg_ms"] * 1e-3
        fl = conv_flops(d, batch)
        from_store = u8_store and lname == "conv1" and what.split(" ")[0] in ("fwd", "bwd_weight")
        by = conv_alg_bytes(d, batch, in_bytes_per_sample=d.H * d.W if from_store else None)
        tf, gbs = fl / sec / 1e12, by / sec / 1e9
        out.update(batch=batch, tflops=round(tf, 2), hbm_GBs=round(gbs, 1), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 4),
                   alg_bytes=by, alg_flops=fl)
        if from_store:
            out["input"] = "uint8 single-frame store: 7,056 B per sample"
        if bf16_pipe and lname == "conv1" and what.split(" ")[0] == "bwd_weight":
            t_pk = 3.0 * fl / (BF16_PEAK_TFLOPS * 1e12)
            out.update(pipe="bf16 MFMA x3 (exact split, fp32 accumulate)", tflops_fp32_equiv=round(tf, 2),
                       frac_of_pipe_peak=round(t_pk / sec, 4))
        elif lname in x6_layers and what.split(" ")[0] in ("bwd_data", "bwd_weight"):
            # bwd_x6_kernel / wgrad_x6_kernel (A3CModel's conv2 at update batch): both operands split, six piece products
            t_pk = 6.0 * fl / (BF16_PEAK_TFLOPS * 1e12)
            out.update(pipe="bf16 MFMA x6 (exact 3-way split of both operands, fp32 accumulate)", tflops_fp32_equiv=round(tf, 2),
                       frac_of_pipe_peak=round(t_pk / sec, 4))
            del out["tflops"]
        else:
            out.update(pipe="fp32 MFMA", frac_of_f32_mfma_peak=round(tf / F32_PEAK_TFLOPS, 4), frac_of_pipe_peak=round(tf / F32_PEAK_TFLOPS, 4))
    return out


def ring_roofline(A, n_envs, T, launches, us, bf16_conv1):
    """`roofline` of the persistent A3CModel rollout launch, per pipe.  One launch = (T + 1) steps x n_envs / launches envs of
    4.64 MFLOP each (step_alg_flops).  With conv1 on the bf16 pipe (three exact bf16 pieces per fp32 weight, uint8 pixels
    exact in bf16) 71 % of those algorithmic fp32 flops are ISSUED as 3 x as many flops on a pipe 16 x as fast, so the peak
    the launch is held against is the rate at which THIS instruction mix would run with every MFMA at its own pipe's peak:
        peak = alg_flops / (3 * conv1_flops / bf16_peak + (alg_flops - conv1_flops) / fp32_peak)
    frac = achieved / peak = matrix time at each instruction's own peak / launch duration  (<= 1 by construction).
    `frac_fp32_equiv` = achieved / the fp32 MFMA peak: rounds 1-5's definition, kept for comparison only -- it is not a
    roofline fraction once part of the work runs on the bf16 pipe."""
    fl = step_alg_flops(A) * n_envs * (T + 1) / launches
    tf = fl / (us * 1e-6) / 1e12
    c1 = 2 * 16 * 400 * 256
    if bf16_conv1:
        t_step = 3.0 * c1 / (BF16_PEAK_TFLOPS * 1e12) + (step_alg_flops(A) - c1) / (F32_PEAK_TFLOPS * 1e12)
        peak = step_alg_flops(A) / t_step / 1e12
        pipes = "conv1: bf16 MFMA x3 (exact split, fp32 accumulate) @ %.1f TF; conv2/heads: fp32 MFMA @ %.1f TF" % (BF16_PEAK_TFLOPS, F32_PEAK_TFLOPS)
    else:
        peak, pipes = F32_PEAK_TFLOPS, "fp32 MFMA"
    return dict(bound="mfma", achieved=round(tf, 2), peak=round(peak, 1), unit="TFLOP/s", frac=round(tf / peak, 4),
                frac_fp32_equiv=round(tf / F32_PEAK_TFLOPS, 4), pipes=pipes, avg_launch_us=round(us, 2),
                launches_per_rollout=launches, alg_flops_per_launch=fl)


def run_config(workload, n_envs, optim, ingest, env_workers, n_workers, shard, dev, steps, warmup, transport="bits",
               frame_store=False, grey=False):
    """one extra BASELINE config: ms per step, env-steps/s, its dominant update and rollout launch sites"""
    b = Bench(workload, n_envs, optim, ingest, env_workers, n_workers, shard, dev, transport=transport, frame_store=frame_store,
              grey=grey)
    try:
        b.step()
        b.capture()
        for _ in range(max(warmup - 1, 0)):
            b.step()
        elapsed, r_ms, u_ms = b.timed(steps)
        out = dict(workload=f"{b.model} n_envs={b.n_envs} n_tsteps={b.T}{' +BPTT' if b.use_bptt else ''}"
                            f"{' Breakout-like grey frames, 4 actions' if grey else ''}", steps=steps, transport=transport,
                   ms_per_step=round(1e3 * elapsed / steps, 3), value=round(b.N * steps / elapsed, 1), unit="env-steps/s",
                   rollout_ms=round(r_ms, 3), update_ms=round(u_ms, 3), ingest=b.describe_ingest(),
                   update="hipGraph" if b.ugraph is not None else "eager")
        if frame_store:
            out["states_layout"] = ("single-frame uint8 store (f4): ingest writes frames[slot][t+3], conv1 forward / weight "
                                    "gradient stack on load, fp32 `states` rows expanded on demand (Runner.materialize_states)")
            out["frame_store_live"] = bool(getattr(b.runner, "_fstore", None) is not None and getattr(b.runner, "_fstore_ok", True))
        layers = getattr(b.net, "_cl", None) or ([b.net._c1, b.net._c2] if hasattr(b.net, "_c1") else [])
        conv = {l.name: l for l in layers}
        summ = b.site_timers(1)
        # what the first layer reads / which pipe its weight gradient runs on (site_roofline prices each site on that)
        st_live = bool(frame_store and getattr(b.runner, "_fstore", None) is not None and getattr(b.runner, "_fstore_ok", True))
        bf_w = st_live and b.model == "A3CModel" and os.environ.get("A2C_WGRAD_F32") != "1"
        x6l = a3c_x6_layers(b.model, b.N)
        if summ:
            dom = max(summ, key=lambda k: summ[k]["total_ms"])
            out["dominant_update_site"] = site_roofline(dom, summ[dom], conv, b.N, st_live, bf_w, x6l)
            out["update_conv_sites"] = {k: site_roofline(k, v, conv, b.N, st_live, bf_w, x6l) for k, v in summ.items()
                                        if k.split(".")[0] in conv}
        rs = b.rollout_site_timers()
        if rs:
            # the rollout's forwards run at batch n_envs, T+1 times per slot (north star: HBM GB/s on the conv forward)
            dom = max(rs, key=lambda k: rs[k]["total_ms"])
            out["dominant_rollout_site"] = site_roofline(dom, rs[dom], conv, b.n_envs, st_live)
            out["rollout_conv_fwd_sites"] = {k: site_roofline(k, v, conv, b.n_envs, st_live) for k, v in rs.items()
                                             if k.endswith(".fwd") and k.split(".")[0] in conv}
            out["rollout_sites_total_ms"] = round(sum(v["total_ms"] for v in rs.values()), 3)
        if b.ingest != "device-tape" and r_ms > 0:
            out["h2d_GBs"] = round(b.N * FRAME_BYTES[transport] / (r_ms * 1e-3) / 1e9, 2)
        return out
    finally:
        b.close()


def spawn_ranks(args, argv):
    """``python bench.py --gpus N`` without a launcher: start N ranks with torch.distributed.run as a CHILD
    process (this process has not touched the GPU and never execs) and relay rank 0's JSON line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for l in proc.stdout.splitlines():
        if l.startswith("{") and '"metric"' in l:
            line = l
        else:
            print(l, file=sys.stderr)
    if line:
        print(line)
    sys.exit(proc.returncode if proc.returncode else (0 if line else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="a3c", choices=sorted(WORKLOADS))
    ap.add_argument("--n-envs", type=int, default=None, help="envs per GPU (default: the workload's)")
    ap.add_argument("--global-envs", type=int, default=None, help="strong scaling: this many envs over all GPUs")
    ap.add_argument("--optim", default="RMSprop", choices=["RMSprop", "Adam"])
    ap.add_argument("--ingest", default="host-pinned", choices=["host-pinned", "memcpy", "device-tape"])
    ap.add_argument("--transport", default="bits", choices=["bits", "u8"],
                    help="what crosses the host link per frame: bits = 1 bit/pixel (the synthetic frames are binary like "
                         "pong_prep's, preprocessing.py:15-16), u8 = one byte per pixel (any uint8 preprocessor)")
    ap.add_argument("--grey", action="store_true",
                    help="Breakout-like tapes: grey levels 0..255, 4 actions, real dones only; implies --transport u8 "
                         "(configs[4]; the per-GPU shard leg of the default run sets it itself)")
    ap.add_argument("--env-workers", default="native", choices=["native", "process"])
    ap.add_argument("--n-workers", type=int, default=None, help="env worker threads/processes per rank")
    ap.add_argument("--sustain-steps", type=int, default=200,
                    help="if --steps is smaller, an additional region of this many steps is timed and reported")
    ap.add_argument("--no-update-graph", action="store_true", help="do not capture the update into a hipGraph")
    ap.add_argument("--frame-store", action="store_true",
                    help="force the single-frame uint8 rollout store + lazy fp32 states (SURVEY.md 8 row f4); it is the default "
                         "with the host-pinned ingest (a3c: ring kernel, <= 256 envs; conv-stack nets: relay path)")
    ap.add_argument("--no-frame-store", action="store_true",
                    help="write the fp32 `states` rows in the rollout (the reference's layout, round 3's) instead of the single-frame store")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--no-secondary", action="store_true", help="skip value_device_tape / process-worker runs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    args = ap.parse_args()
    if args.grey:
        args.transport = "u8"

    # the ONE JSON line goes to the real stdout; whatever else prints on the way (the model classes mirror the
    # reference's "Flat Features Size" print) goes to stderr
    json_out, sys.stdout = sys.stdout, sys.stderr
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.stdout = json_out
        spawn_ranks(args, sys.argv[1:])
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    from a2c_amd import ops
    from a2c_amd.hostpool import usable_cpus
    from a2c_amd.parallel import Shard

    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("A2C_BENCH_ONE_DEVICE") == "1":      # test hook: N ranks share cuda:0 (use with A2C_DIST_BACKEND=gloo)
        local = 0
    if usable_cpus() < 4 * max(world, 1):
        # fewer CPUs than (main thread + env thread + RCCL proxy + slack) per rank: do not let waiting host threads spin
        from a2c_amd import _lib
        _lib.load().a2c_set_blocking_sync(1)
    torch.cuda.set_device(local)
    shard = Shard.from_env()
    dev = torch.device("cuda", local)
    n_envs = args.n_envs
    scaling = "weak"
    if args.global_envs:
        lo, hi = shard.slot_range(args.global_envs)
        n_envs, scaling = hi - lo, "strong"
    # env workers per rank: the usable CPUs are shared by all ranks of the node
    n_workers = args.n_workers
    if n_workers is None:
        # this rank's share of the usable CPUs (the cgroup quota is shared by all ranks of the node), one CPU per rank set
        # aside for its main thread (kernel launches, RCCL proxy): NEVER more spinning env threads than CPUs -- a spinning
        # thread that loses its core stalls the device-side hand-shake of all its envs for a scheduler period (round 3 had a
        # floor of 4 here: 32 spinning threads on a 16-CPU quota at 8 ranks)
        per_rank = max(1, (usable_cpus() - max(shard.world, 1)) // max(shard.world, 1))
        n_workers = max(1, min(14 if args.env_workers == "native" else 48, per_rank))
    print(f"[bench] rank {shard.rank}/{shard.world}: env_workers={n_workers} ({args.env_workers}), usable_cpus={usable_cpus()}",
          file=sys.stderr)

    # SURVEY.md 8 row f4 is the headline's storage layout: the ring kernel keeps ONE uint8 frame per env step, the update's
    # first-layer weight gradient stacks the frames on load, the reference-layout fp32 `states` rows are expanded on demand
    # (Runner.materialize_states; not needed by rollout -> update -> rollout).  --no-frame-store writes the rows.
    fs_main = args.frame_store or (args.ingest == "host-pinned" and not args.no_frame_store)
    b = Bench(args.workload, n_envs, args.optim, args.ingest, args.env_workers, n_workers, shard, dev,
              update_graph=not args.no_update_graph, transport=args.transport, frame_store=fs_main, grey=args.grey)
    model, T, A, N = b.model, b.T, b.A, b.N
    b.step()
    b.capture()
    for _ in range(max(args.warmup - 1, 0)):
        b.step()
    elapsed, rollout_ms, update_ms = b.timed(args.steps, split=not args.no_kernel_timers)
    sustained = None
    if args.steps < args.sustain_steps:
        e2, r2, u2 = b.timed(args.sustain_steps, split=not args.no_kernel_timers)
        sustained = dict(steps=args.sustain_steps, seconds=round(e2, 3), ms_per_step=round(1e3 * e2 / args.sustain_steps, 3),
                         value=None, rollout_ms=round(r2, 3), update_ms=round(u2, 3))
    allreduce_ms = None
    if shard.active:
        t = torch.tensor([elapsed, sustained["seconds"] if sustained else 0.0], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t[0].item())
        if sustained:
            sustained["seconds"] = round(float(t[1].item()), 3)
        # cost of the one gradient all-reduce per update, timed on its own
        g = b.net._arena.train_grads()
        for _ in range(3):
            shard.allreduce_(g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            shard.allreduce_(g)
        e1.record()
        torch.cuda.synchronize()
        allreduce_ms = e0.elapsed_time(e1) / 10
    summ = None
    if not args.no_kernel_timers:
        summ = b.site_timers(2)
    ingest_desc = b.describe_ingest()
    zero_copy = args.ingest != "device-tape" and b.runner._zero_copy_ok(b.net)
    pool_workers = getattr(b.pool, "n_workers", None)

    if shard.rank != 0:
        b.close()
        return
    total_steps = N * shard.world * args.steps
    if sustained:
        sustained["value"] = round(N * shard.world * args.sustain_steps / sustained["seconds"], 1)
    parallelism = f"dp{shard.world} (rollout shards, 1 RCCL grad all-reduce/update)"
    out = dict(metric="env-steps/sec (rollout+update)", value=round(total_steps / elapsed, 1), unit="env-steps/s",
               n_gpus=shard.world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3),
               higher_is_better=True, scaling=scaling, vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"{model} n_envs={b.n_envs} n_tsteps={T} 84x84x4 synthetic frames"
                                    f"{' +BPTT' if b.use_bptt else ''}, {args.optim}, per GPU",
                           n_envs=b.n_envs, n_tsteps=T, optimizer=args.optim, transport=args.transport,
                           ingest=("device-tape" if args.ingest == "device-tape" else
                                   f"host-pinned/{'zero-copy-persistent' if zero_copy else ('memcpy' if args.ingest == 'memcpy' else 'device-relay')}"
                                   f"/{args.env_workers}"),
                           states_layout="u8-frame-store+lazy-fp32-states" if fs_main else "fp32-states-rows",
                           update=("hipGraph" if len(b.ugraph.graphs) == 1 else f"{len(b.ugraph.graphs)}-hipGraphs/"
                                   f"{len(b.ugraph.colls)}-collectives") if b.ugraph is not None else "eager",
                           info_readback="one-step-late" if os.environ.get("A2C_BENCH_SYNC_INFO") != "1" else "blocking",
                           env_workers=pool_workers, usable_host_cpus=usable_cpus(), parallelism=f"dp{shard.world}"),
               config_notes=dict(
                   ingest=ingest_desc,
                   states_layout=("single-frame uint8 store (SURVEY 8 f4): 7 KB frame per env step written by the rollout, "
                                  "conv1 weight gradient stacked on load, fp32 `states` rows on demand "
                                  "(Runner.materialize_states); `value_states_rows_written` = the same run writing the rows")
                   if fs_main else "fp32 `states` rows (N, 4, 84, 84) written by the rollout (the reference's layout)",
                   info_readback=("one step late: every step enqueues rollout + update, the update's five scalars travel to "
                                  "pinned host memory behind it and are collected while the NEXT rollout runs "
                                  "(Updater.update_model_async / collect); the timed region ends with a full drain + sync"
                                  if os.environ.get("A2C_BENCH_SYNC_INFO") != "1" else "blocking read-back after every update"),
                   frames="84x84 binary, i.i.d. Bernoulli(0.25) per pixel from default_rng(1234 + env_id) "
                          "(SURVEY 8d says uniform{0,1}: cost-neutral, the kernels are data-independent)",
                   parallelism=parallelism),
               rollout_ms=round(rollout_ms, 3), update_ms=round(update_ms, 3),
               last_info={k: round(float(v), 6) for k, v in (b.info or {}).items()})
    if sustained:
        out["sustained"] = sustained
    if args.ingest != "device-tape" and rollout_ms > 0:
        fb = FRAME_BYTES[args.transport]
        gbs = b.n_envs * fb * T / (rollout_ms * 1e-3) / 1e9
        out["h2d"] = dict(transport=args.transport, bytes_per_update=b.n_envs * fb * T, bytes_per_step=b.n_envs * fb,
                          achieved_GBs=round(gbs, 2), peak_GBs=PCIE_PEAK_GBS, frac=round(gbs / PCIE_PEAK_GBS, 4),
                          note="frames, pinned host memory -> device, inside the timed region; rate over the rollout half")
    if shard.active:
        out["rccl_ranks"] = torch.distributed.get_world_size() if torch.distributed.get_backend() == "nccl" else 0
        out["dist_backend"] = torch.distributed.get_backend()
        out["allreduce_ms_per_update"] = round(allreduce_ms, 4)
        out["allreduce_bytes"] = int(b.net._arena.n_train * 4)

    # ---- rooflines
    if summ is not None:
        kern = {k: dict(avg_ms=round(v["avg_ms"], 4), launches=v["launches"]) for k, v in
                sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        out["update_launch_sites_ms"] = kern
        out["update_launch_sites_note"] = "HIP events per launch site, 2 eager updates after the timed region"
        layers = getattr(b.net, "_cl", None) or ([b.net._c1, b.net._c2] if hasattr(b.net, "_c1") else [])
        conv_layers = {l.name: l for l in layers}
        dom = next(iter(kern), None)
        dom_ms = summ[dom]["total_ms"] / 2 if dom is not None else 0.0
        fused_step = getattr(b.net, "_step_supported", lambda: False)()
        if fused_step and rollout_ms > dom_ms:
            cus_ = torch.cuda.get_device_properties(dev).multi_processor_count
            ring = zero_copy and os.environ.get("A2C_NO_RING") != "1" and \
                (b.n_envs <= cus_ or os.environ.get("A2C_RING_BLOCKS") != "0")
            # (more envs than CUs: ceil(n_envs / CUs) ring launches one after the other, each a block of interleaved envs)
            launches = ((b.n_envs + cus_ - 1) // cus_ if ring else 1) if zero_copy else T + 1
            us = rollout_ms * 1e3 / launches
            fl = step_alg_flops(A) * b.n_envs * (T + 1) / launches
            lazy_ring = bool(ring and fs_main and getattr(b.runner, "_states_stale", False))
            by = (step_alg_bytes(u8_frame=args.ingest != "device-tape", ring=ring, lazy=lazy_ring) -
                  (0 if ring else (84 * 84 - FRAME_BYTES[args.transport] if args.ingest != "device-tape" else 0))) \
                * b.n_envs * (T + 1) / launches
            tf = fl / (us * 1e-6) / 1e12
            name = ("%s (a2c_a3c_rollout: 1 launch = %d steps x %d envs, host-paced)"
                    % ("a3c_ring_kernel" if ring else "a3c_step_kernel<persistent>", T + 1, b.n_envs // launches)) if zero_copy else \
                   f"a3c_step_kernel (B={b.n_envs}, {T + 1} launches/rollout)"
            bf_ring = bool(ring and os.environ.get("A2C_RING_F32") != "1")
            out["roofline"] = dict(kernel=name, traffic=None, **ring_roofline(A, b.n_envs, T, launches, us, bf_ring),
                                   hbm_GBs=round(by / (us * 1e-6) / 1e9, 1),
                                   hbm_frac=round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), alg_bytes_per_launch=by)
            if zero_copy:
                link = dict(out["h2d"], note="PCIe frame bytes / launch duration vs the 63 GB/s link")
                if args.transport == "u8" and link["frac"] > out["roofline"]["frac"]:
                    # the launch is paced by the host link (uint8 transport: 7 KB per env step): that fraction is the primary
                    # figure.  (Packed transport: 0.9 KB per env step, ~0.2 of the link -- the launch is not link-bound, and
                    # its per-pipe matrix fraction happens to be of the same size: no switching on a comparison of the two.)
                    out["roofline"].update(bound="host_link", achieved=link["achieved_GBs"], peak=PCIE_PEAK_GBS, unit="GB/s",
                                           frac=link["frac"], mfma_TFLOPs=round(tf, 2), mfma_frac=out["roofline"]["frac"])
                out["roofline"]["host_link"] = link
        elif dom is not None:
            ms = summ[dom]["avg_ms"]
            lname, _, what = dom.partition(".")
            if lname in conv_layers and what == "fwd":
                d = conv_layers[lname].d
                ach = conv_alg_bytes(d, N) / (ms * 1e-3) / 1e9
                out["roofline"] = dict(kernel=f"{dom} (B={N})", bound="hbm", achieved=round(ach, 1),
                                       peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 4), traffic=None,
                                       tflops=round(conv_flops(d, N) / (ms * 1e-3) / 1e12, 2))
            elif lname in conv_layers:
                d = conv_layers[lname].d
                tf = conv_flops(d, N) / (ms * 1e-3) / 1e12
                out["roofline"] = dict(kernel=f"{dom} (B={N})", bound="mfma", achieved=round(tf, 2), peak=F32_PEAK_TFLOPS,
                                       unit="TFLOP/s", frac=round(tf / F32_PEAK_TFLOPS, 4), traffic=None)
            else:
                out["roofline"] = dict(kernel=dom, bound="mfma", achieved=None, peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                                       frac=None, traffic=None)
        # HBM bytes per launch from the PMC passes committed under profiles/ (collected separately: rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE passes of this same command, tools/profile_round.sh).  Only counters taken from THIS tree are
        # quoted: the traffic file's manifest records the sha256 of the kernel sources it was measured on; a tree whose
        # sources differ prints traffic: null, traffic_stale: true instead of another kernel's counters.
        if "roofline" in out and out["roofline"].get("traffic") is None and args.workload == "a3c" and N == 32768:
            key = {"conv1.fwd": "conv1_fwd", "conv1.bwd_weight": "conv1_wgrad", "conv2.bwd_data": "conv2_bwd_data"}.get(dom)
            k2 = ((("a3c_ring_lazy" if lazy_ring else "a3c_ring") if out["roofline"]["kernel"].startswith("a3c_ring") else "a3c_rollout") if zero_copy else "a3c_step") \
                if out["roofline"]["kernel"].startswith("a3c_") else key
            tr, src, stale = lookup_traffic(k2)
            out["roofline"]["traffic"] = tr
            if src:
                out["roofline"]["traffic_source"] = src
            if stale:
                out["roofline"]["traffic_stale"] = True
        # always report conv1 forward (north star: HBM GB/s on the conv forward)
        if "conv1" in conv_layers and "conv1.fwd" in summ:
            d = conv_layers["conv1"].d
            ms = summ["conv1.fwd"]["avg_ms"]
            ach = conv_alg_bytes(d, N) / (ms * 1e-3) / 1e9
            out["conv1_fwd_roofline"] = dict(bound="hbm", avg_ms=round(ms, 4), achieved_GBs=round(ach, 1),
                                             frac=round(ach / HBM_PEAK_GBS, 4),
                                             tflops=round(conv_flops(d, N) / (ms * 1e-3) / 1e12, 2))
    b.close()
    del b
    torch.cuda.empty_cache()

    if shard.world == 1:
        out["scan_roofline"] = scan_roofline(dev)
        if not args.no_secondary and args.workload == "a3c":
            # kernel-side evidence without the ingest: the round-1 path (frames already in HBM, rollout as a hipGraph of
            # T+1 a3c_step_kernel launches) -- NOT the north-star number
            try:
                d = Bench(args.workload, n_envs, args.optim, "device-tape", "native", 1, shard, dev)
                d.step(); d.capture(); d.step()
                d.timed(10)                 # (the first process on a fresh box spends milliseconds per step on the host in
                e, r_ms, u_ms = d.timed(20)   #  this phase for its first steps: page-ins, not the path under test)
                us = r_ms * 1e3 / (T + 1)
                tf = step_alg_flops(A) * d.n_envs / (us * 1e-6) / 1e12
                out["value_device_tape"] = dict(value=round(d.N * 20 / e, 1), unit="env-steps/s", steps=20,
                                                ms_per_step=round(1e3 * e / 20, 3), rollout_ms=round(r_ms, 3),
                                                update_ms=round(u_ms, 3), gpu_ms_per_step=round(r_ms + u_ms, 3),
                                                note="frames pre-generated in HBM (no host ingest), rollout hipGraph; "
                                                     "ms_per_step is wall clock, gpu_ms_per_step the two halves from HIP events: "
                                                     "the FIRST process on a fresh box spends 2-3 ms on the host per step in "
                                                     "this phase (first submit after each blocking read-back; 7.4 ms wall in "
                                                     "every later process on the same box)")
                out["step_kernel_roofline"] = dict(kernel=f"a3c_step_kernel (B={d.n_envs}, {T + 1} launches per hipGraph replay)",
                                                   bound="mfma", achieved=round(tf, 2), peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                                                   frac=round(tf / F32_PEAK_TFLOPS, 4), avg_launch_us=round(us, 2),
                                                   hbm_GBs=round(step_alg_bytes() * d.n_envs / (us * 1e-6) / 1e9, 1))
                d.close()
                del d
            except Exception as e:      # noqa: BLE001
                out["value_device_tape"] = dict(value=None, error=f"{type(e).__name__}: {e}")
            if fs_main and args.workload == "a3c":
                # the same headline with the fp32 `states` rows written by the rollout (round 3's layout)
                try:
                    d = Bench(args.workload, n_envs, args.optim, args.ingest, args.env_workers, n_workers, shard, dev,
                              transport=args.transport, frame_store=False)
                    d.step(); d.capture(); d.step()
                    e, r_ms, u_ms = d.timed(60)
                    out["value_states_rows_written"] = dict(value=round(d.N * 60 / e, 1), unit="env-steps/s", steps=60,
                                                            ms_per_step=round(1e3 * e / 60, 3), rollout_ms=round(r_ms, 3),
                                                            update_ms=round(u_ms, 3),
                                                            note="fp32 states rows (113 KB per env step) written by the ring kernel, "
                                                                 "conv1 weight gradient from the fp32 rows")
                    d.close()
                    del d
                except Exception as e:      # noqa: BLE001
                    out["value_states_rows_written"] = dict(value=None, error=f"{type(e).__name__}: {e}")
            if args.ingest == "host-pinned" and args.transport == "bits":
                # the same headline with the generic uint8 transport (any uint8 preprocessor): host-link bound
                try:
                    d = Bench(args.workload, n_envs, args.optim, "host-pinned", args.env_workers, n_workers, shard, dev,
                              transport="u8")
                    d.step(); d.capture(); d.step()
                    e, r_ms, u_ms = d.timed(40)
                    gbs = d.N * FRAME_BYTES["u8"] / (r_ms * 1e-3) / 1e9
                    out["host_pinned_u8_transport"] = dict(value=round(d.N * 40 / e, 1), unit="env-steps/s", steps=40,
                                                           ms_per_step=round(1e3 * e / 40, 3), rollout_ms=round(r_ms, 3),
                                                           update_ms=round(u_ms, 3),
                                                           h2d=dict(transport="u8", achieved_GBs=round(gbs, 2), peak_GBs=PCIE_PEAK_GBS,
                                                                    frac=round(gbs / PCIE_PEAK_GBS, 4)),
                                                           note="one byte per pixel over the link (round 2's headline transport)")
                    d.close()
                    del d
                except Exception as e:      # noqa: BLE001
                    out["host_pinned_u8_transport"] = dict(value=None, error=f"{type(e).__name__}: {e}")
            if args.ingest == "host-pinned" and args.env_workers == "native":
                # the host floor an 8-rank run on a 16-CPU quota meets (one env thread per rank): the same headline
                # with ONE native env thread serving all envs; host_us_per_env_step = that rollout / (envs x steps) is the
                # upper bound of the thread's service time per env step (the kernel's own step is inside it)
                try:
                    d = Bench(args.workload, n_envs, args.optim, "host-pinned", "native", 1, shard, dev,
                              transport=args.transport, frame_store=fs_main)
                    d.step(); d.capture(); d.step()
                    e, r_ms, u_ms = d.timed(40)
                    out["value_one_env_thread"] = dict(value=round(d.N * 40 / e, 1), unit="env-steps/s", steps=40,
                                                       ms_per_step=round(1e3 * e / 40, 3), rollout_ms=round(r_ms, 3),
                                                       update_ms=round(u_ms, 3), env_threads=1,
                                                       us_per_rollout_step=round(r_ms * 1e3 / (T + 1), 2))
                    out["host_us_per_env_step"] = round(r_ms * 1e3 / ((T + 1) * d.n_envs), 4)
                    # NOT a scaling measurement (the driver measures that): what 8 ranks add up to if each is paced like this
                    # one-thread leg, i.e. on a box whose CPU quota leaves ONE env thread per rank (16 CPUs / 8 ranks)
                    out["predicted_8rank_weak"] = dict(value=round(8 * d.N * 40 / e, 1), unit="env-steps/s",
                                                       vs_1rank=round(8 * d.N * 40 / e / out["value"], 2),
                                                       basis="8 x value_one_env_thread; prediction, host-bound, unmeasured")
                    d.close()
                    del d
                except Exception as e:      # noqa: BLE001
                    out["value_one_env_thread"] = dict(value=None, error=f"{type(e).__name__}: {e}")
            if args.ingest == "host-pinned" and args.env_workers == "native":
                try:
                    nw = max(1, min(48, usable_cpus() - 2))
                    d = Bench(args.workload, n_envs, args.optim, "host-pinned", "process", nw, shard, dev, transport=args.transport)
                    d.step(); d.capture(); d.step()
                    e, r_ms, u_ms = d.timed(40)
                    out["host_pinned_process_workers"] = dict(value=round(d.N * 40 / e, 1), unit="env-steps/s", steps=40,
                                                              ms_per_step=round(1e3 * e / 40, 3), rollout_ms=round(r_ms, 3),
                                                              update_ms=round(u_ms, 3), env_worker_processes=d.pool.n_workers,
                                                              note="Python TapeEnv objects stepped in worker processes "
                                                                   "(a2c_amd.hostpool_worker), same pinned-region protocol")
                    d.close()
                    del d
                except Exception as e:      # noqa: BLE001
                    out["host_pinned_process_workers"] = dict(value=None, error=f"{type(e).__name__}: {e}")
        if not args.no_configs and args.workload == "a3c" and not args.n_envs and not args.global_envs:
            cfgs = {}
            # BASELINE.json configs[1], [3], the north star's n_envs in {32, 2048} for the headline model, and the
            # per-GPU shard of configs[4] (ConvModel, 2048 envs over 8 GPUs = 256 envs x 128 steps per GPU)
            for key, wl, ne, st_, wu in (("conv_32x64", "conv", None, 20, 3), ("gru_bptt_256x128", "gru_bptt", None, 10, 2),
                                         ("a3c_32", "a3c", 32, 40, 3), ("a3c_2048", "a3c", 2048, 10, 2),
                                         ("conv_2048x128_per_gpu_shard_256x128", "conv", 256, 10, 2)):
                try:
                    torch.cuda.empty_cache()
                    # (the single-frame store, row f4, is the layout of every config that keeps it: the ring kernel's for
                    # A3CModel up to 256 envs, the relay path's for the conv-stack nets; --no-frame-store = fp32 rows)
                    fs_cfg = (not args.no_frame_store) and (wl != "a3c" or fs_main)
                    # configs[4] is Breakout: grey levels 0..255 over the uint8 transport (the packed one is Pong's)
                    brk = key.startswith("conv_2048x128") and args.ingest != "device-tape"
                    cfgs[key] = run_config(wl, ne, args.optim, args.ingest, args.env_workers, n_workers, shard, dev, st_, wu,
                                           transport="u8" if brk else args.transport, frame_store=fs_cfg, grey=brk)
                except Exception as e:      # noqa: BLE001
                    cfgs[key] = dict(error=f"{type(e).__name__}: {e}")
            # row f4 measured: the same conv-stack configs writing the reference's fp32 `states` rows in the rollout
            for key, wl, ne, st_, wu in (("conv_32x64", "conv", None, 20, 3), ("gru_bptt_256x128", "gru_bptt", None, 10, 2)):
                if args.no_frame_store:
                    break
                try:
                    torch.cuda.empty_cache()
                    r = run_config(wl, ne, args.optim, args.ingest, args.env_workers, n_workers, shard, dev, st_, wu,
                                   transport=args.transport, frame_store=False)
                    cfgs[key + "_states_rows"] = {k: r[k] for k in ("workload", "value", "ms_per_step", "rollout_ms", "update_ms",
                                                                    "states_layout") if k in r}
                    c1 = (r.get("update_conv_sites") or {}).get("conv1.bwd_weight")
                    if c1:
                        cfgs[key + "_states_rows"]["conv1.bwd_weight"] = c1
                except Exception as e:      # noqa: BLE001
                    cfgs[key + "_states_rows"] = dict(error=f"{type(e).__name__}: {e}")
            out["configs"] = cfgs
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, WORKLOADS[args.workload][1] if not n_envs else n_envs, T,
                                                   WORKLOADS[args.workload][3], A, args.optim)
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"] = dict(value=None, error=f"{type(e).__name__}: {e}")
            # north_star: every throughput figure "next to the reference's CPU updater timed on the same box".  The other
            # configs of this line get their own bounded sample (3 s of rollout processes + ~6 s of update_model each: the
            # per-sample cost of the oracle does not depend on the batch beyond that, `extrapolated` says when N was cut);
            # the two other A3CModel sizes reuse the headline's measured rates (same model, same min(n_envs, cores) workers).
            if out.get("configs") and out["cpu_baseline"].get("value"):
                cbs = {}
                for key, wl, ne in (("conv_32x64", "conv", 32), ("gru_bptt_256x128", "gru_bptt", 256)):
                    try:
                        m_, _, T_, bp_, A_ = WORKLOADS[wl]
                        cbs[key] = cpu_baseline(m_, ne, T_, bp_, A_, args.optim, roll_s=3.0, budget=6.0)
                    except Exception as e:      # noqa: BLE001
                        cbs[key] = dict(value=None, error=f"{type(e).__name__}: {e}")
                hb = out["cpu_baseline"]
                for key, ne in (("a3c_32", 32), ("a3c_2048", 2048)):
                    cbs[key] = dict(value=hb["value"], unit="env-steps/s", cores=hb["cores"], kind="port", extrapolated=True,
                                    sample=f"the headline's measured A3CModel rates (rollout {hb['rollout_steps_per_s']}/s, update "
                                           f"{hb['update_samples_per_s']}/s) at n_envs={ne}: same model, same worker count")
                c32 = cbs.get("conv_32x64") or {}
                if c32.get("value"):
                    cbs["conv_2048x128_per_gpu_shard_256x128"] = dict(
                        value=c32["value"], unit="env-steps/s", cores=c32["cores"], kind="port", extrapolated=True,
                        sample="ConvModel's measured per-sample rates (conv_32x64 entry) at N = 32,768 per GPU")
                out["cpu_baselines"] = cbs
    tag = f"{args.workload}_n{shard.world}" + (f"_e{args.n_envs}" if args.n_envs else "")
    side = write_side_file(out, tag)
    print("[bench] full report: " + json.dumps(_finite(out)), file=sys.stderr)
    print(compact_line(out, side), file=json_out)
    json_out.flush()


if __name__ == "__main__":
    main()
