#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the A2C rollout+update hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload a3c|conv|gru|gru_bptt|fc]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one batch: a rollout of n_envs x n_tsteps env-steps
(batched forward, sampling, TD deltas, frame stacking -- all HIP kernels) followed by one
Updater.update_model (GAE/returns scan, forward, loss, backward, clip + optimiser).  Inputs are
synthetic and already resident in HBM when the timed region starts (84x84 binary frames,
rewards, dones and sampling uniforms pre-generated on the device: SURVEY.md section 8d).
Default workload = the headline config of BASELINE.json: A3CModel, n_envs=256, n_tsteps=128.
Multi-GPU: weak scaling, every rank plays its own n_envs envs; one RCCL all-reduce of the flat
gradient arena (+ 2 tiny ones for the whole-batch statistics) per update.

Prints ONE JSON line (rank 0): metric/value/... plus
  "roofline":     the dominant kernel's algorithmic bytes (or flops) / its HIP-event duration,
  "cpu_baseline": the CPU oracle (restatement of the reference's algorithm) timed on this box's
                  host cores on a bounded sample of the same workload (N=1 only).
"""
import argparse
import json
import os
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
    "conv": ("ConvModel", 32, 64, False, 3),        # configs[1]
    "gru": ("GRUModel", 256, 128, False, 3),
    "gru_bptt": ("GRUModel", 256, 128, True, 3),    # configs[3]
    "fc": ("FCModel", 256, 128, False, 3),
}
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_PEAK_TFLOPS = 157.3      # fp32 matrix == fp32 vector peak


def hyps_for(model, n_envs, T, use_bptt, optim):
    return dict(gamma=.99, lambda_=.98, n_tsteps=T, n_rollouts=n_envs, n_envs=n_envs, n_frame_stack=4, action_shift=0,
                render=False, env_type="Pong-synthetic", use_bptt=use_bptt, use_nstep_rets=False, norm_advs=True,
                entr_coef=.005, pi_coef=1.0, val_coef=.5, max_norm=.5, lr=1e-4, optim_type=optim, is_discrete=True,
                h_size=256, model=model)


class SyntheticDevicePool:
    """Device-resident synthetic env pool: i.i.d. binary 84x84 frames, rewards
    -1/0/+1 with P=(.02,.96,.02), real done with P=1/800; the same T-step tape is replayed every
    epoch.  Implements the Runner's device-pool protocol (start / device_step)."""

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


# ---------------------------------------------------------------- CPU baseline (oracle, host cores)
def _cpu_rollout_worker(args):
    model, T, n_slots, A = args
    import torch as th
    th.set_num_threads(1)
    from oracle import a2c_oracle as O
    ss = (4, 84, 84)
    net = O.OracleNet(model, ss, A, 256)
    hyps = hyps_for(model, 1, T, False, "RMSprop")
    N = T * n_slots
    D = dict(states=th.zeros(N, *ss), deltas=th.zeros(N), rewards=th.zeros(N), dones=th.zeros(N),
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
    for i in range(1, n_slots):
        r.rollout(net, i)
    return (n_slots - 1) * T, time.perf_counter() - t0


def cpu_baseline(model, n_envs, T, use_bptt, A, optim):
    """Oracle timed like the reference runs: rollout = one process per core, batch-1
    forwards, 1 torch thread each; update = one process, all cores.  Bounded sample, scaled to the
    workload's env-steps/sec: 1 / (1/rollout_rate + 1/update_rate)."""
    import multiprocessing as mp
    from oracle import a2c_oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = min(cores, n_envs, 32)
    slots = 3
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        res = pool.map(_cpu_rollout_worker, [(model, T, slots, A)] * workers)
    steps = sum(r[0] for r in res)
    roll_rate = steps / max(r[1] for r in res)          # aggregate env-steps/s of `workers` processes
    roll_rate *= cores / workers
    # update on a bounded batch
    R = 8 if model != "ConvModel" else 2
    torch.set_num_threads(cores)
    ss = (4, 84, 84)
    net = O.OracleNet(model, ss, A, 256)
    hyps = hyps_for(model, R, T, use_bptt, optim)
    N = R * T
    g = torch.Generator().manual_seed(0)
    D = dict(states=(torch.rand(N, *ss, generator=g) < 0.25).float(), deltas=torch.randn(N, generator=g),
             rewards=torch.randn(N, generator=g).round(), dones=(torch.rand(N, generator=g) < 0.01).float(),
             actions=torch.randint(0, A, (N,), generator=g))
    D["dones"][T - 1::T] = 1
    if net.is_recurrent:
        D["h_states"] = torch.randn(N, 256, generator=g)
    upd = O.OracleUpdater(net, hyps)
    # torch's CPU kernels do not scale to hundreds of threads on such small batches: time a few
    # thread counts and keep the BEST one for the baseline (favours the CPU)
    upd_rate, upd_threads = 0.0, cores
    for nt in sorted({min(cores, c) for c in (8, 16, 32, 64, cores)}):
        torch.set_num_threads(nt)
        upd.update_model(D)
        t1 = time.perf_counter()
        upd.update_model(D)
        rate = N / (time.perf_counter() - t1)
        if rate > upd_rate:
            upd_rate, upd_threads = rate, nt
        if time.perf_counter() - t0 > 45:
            break
    value = 1.0 / (1.0 / roll_rate + 1.0 / upd_rate)
    return dict(value=round(value, 1), unit="env-steps/s", cores=cores, kind="port",
                sample=f"oracle (CPU restatement of the reference): rollout {workers} procs x {slots - 1} slots x {T} "
                       f"batch-1 steps, scaled to {cores} cores = {roll_rate:.0f} steps/s; update_model on N={N}, best of "
                       f"several thread counts ({upd_threads} threads) = {upd_rate:.0f} samples/s; combined as "
                       f"1/(1/r+1/u); wall {time.perf_counter() - t0:.0f}s")


# ---------------------------------------------------------------- roofline helpers
def conv_alg_bytes(d, B):
    return 4.0 * B * (d.Cin * d.H * d.W + d.Cout * d.OH * d.OW)


def conv_flops(d, B):
    return 2.0 * B * d.Cout * d.OH * d.OW * d.Cin * d.ks * d.ks


def step_alg_flops(A):
    """a2c_a3c_step per env: conv 8x8/s4 4->16 on 84x84, conv 4x4/s2 16->32 on 20x20, (A+1) x 2592 heads"""
    return 2 * 16 * 400 * 256 + 2 * 32 * 81 * 256 + 2 * (A + 1) * 2592


def step_alg_bytes():
    """a2c_a3c_step per env: 3 planes of the previous state + the new frame in, the new state out"""
    return 2 * 4 * 84 * 84 * 4


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="a3c", choices=sorted(WORKLOADS))
    ap.add_argument("--n-envs", type=int, default=None)
    ap.add_argument("--optim", default="RMSprop", choices=["RMSprop", "Adam"])
    ap.add_argument("--no-graph", action="store_true", help="do not capture the rollout into a hipGraph")
    ap.add_argument("--no-update-graph", action="store_true", help="do not capture the update into a hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timers", action="store_true")
    args = ap.parse_args()

    import a2c_amd
    from a2c_amd import ops
    from a2c_amd.parallel import Shard
    from a2c_amd.runner import Runner
    from a2c_amd.updater import Updater

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("A2C_BENCH_ONE_DEVICE") == "1":      # test hook: N ranks share cuda:0 (use with A2C_DIST_BACKEND=gloo)
        local = 0
    torch.cuda.set_device(local)
    shard = Shard.from_env()
    dev = torch.device("cuda", local)

    model, n_envs, T, use_bptt, A = WORKLOADS[args.workload]
    if args.n_envs:
        n_envs = args.n_envs
    hyps = hyps_for(model, n_envs, T, use_bptt, args.optim)
    ss = (4, 84, 84)
    torch.manual_seed(20260101)                # the reference's default init, identical on every rank
    net = getattr(a2c_amd.models, model)(list(ss), A, h_size=256)
    N = n_envs * T
    D = dict(states=torch.zeros(N, *ss, device=dev), deltas=torch.zeros(N, device=dev),
             rewards=torch.zeros(N, device=dev), dones=torch.zeros(N, device=dev),
             actions=torch.zeros(N, dtype=torch.int64, device=dev))
    if net.is_recurrent:
        D["h_states"] = torch.zeros(N, 256, device=dev)
    pool = SyntheticDevicePool(n_envs, T, dev, seed=shard.rank)
    runner = Runner(D, hyps, None, None, None, env_pool=pool,
                    uniform_fn=lambda t, B, e0: pool.uniforms[t, e0:e0 + B])
    updater = Updater(net, hyps, shard=shard)
    slots = list(range(n_envs))

    graph = None

    def rollout():
        if graph is not None:
            graph.replay()
            net._dirty = False        # the captured rollout re-derived the inference weights (conv fragments, Wc)
        else:
            net.mark_dirty()
            runner.rollout(net, slots, hyps)

    def step():
        rollout()
        return updater.update_model(D)

    info = None
    for i in range(max(args.warmup, 1)):
        info = step()
        if i == 0 and not args.no_graph:
            # capture the whole n_tsteps rollout (T x ~8 launches) into ONE hipGraph: no host
            # round trip exists inside it because the env tape is device resident
            try:
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                net.mark_dirty()
                # thread_local: the RCCL watchdog thread of a multi-GPU run may query events while
                # this thread captures; that must not invalidate the capture
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    runner.rollout(net, slots, hyps)
                graph = g
            except Exception as e:      # noqa: BLE001
                if shard.rank == 0:
                    print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); eager rollout", file=sys.stderr)
                graph = None
                torch.cuda.synchronize()
    # the update as a second hipGraph (single GPU, RMSprop: no collective and no per-step scalar
    # argument inside): ~40 dependent launches whose 5-10 us stream gaps shrink to graph-edge gaps
    ugraph, udev = None, None
    if graph is not None and not shard.active and args.optim == "RMSprop" and not args.no_update_graph:
        try:
            torch.cuda.synchronize()
            ug = torch.cuda.CUDAGraph()
            net._dirty = False
            with torch.cuda.graph(ug, capture_error_mode="thread_local"):
                udev, u_nglobal = updater._enqueue_update(D)
            ugraph = ug
            net._dirty = True          # the capture itself did not run: the optimiser step has not happened
        except Exception as e:      # noqa: BLE001
            print(f"[bench] update hipGraph capture failed ({type(e).__name__}: {e}); eager update", file=sys.stderr)
            ugraph = None
            torch.cuda.synchronize()

    def update():
        if ugraph is not None:
            ugraph.replay()
            updater.optim._steps += 1
            return updater._finish_update(udev, u_nglobal)
        return updater.update_model(D)

    timers = None
    if not args.no_kernel_timers and ugraph is None:
        timers = ops.KernelTimers()

    shard.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    roll_ev = []
    for _ in range(args.steps):
        ops.TIMERS = None
        if not args.no_kernel_timers:  # HIP events on the launch stream around the rollout (one hipGraph replay)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            rollout()
            ev[1].record()
            roll_ev.append(ev)
        else:
            rollout()
        ops.TIMERS = timers
        info = update()
    torch.cuda.synchronize()
    shard.barrier()
    elapsed = time.perf_counter() - t0
    ops.TIMERS = None
    sites_note = "HIP events per launch site inside the timed region"
    if ugraph is not None and not args.no_kernel_timers:
        # per-site timings cannot be taken inside a graph replay: two extra EAGER updates after the
        # timed region (same kernels, same data) provide them; the rollout is still timed live above
        timers = ops.KernelTimers()
        for _ in range(2):
            rollout()
            ops.TIMERS = timers
            updater.update_model(D)
            ops.TIMERS = None
        torch.cuda.synchronize()
        sites_note = "HIP events per launch site, 2 eager updates after the timed region (the timed updates are hipGraph replays)"
    if shard.active:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if shard.rank != 0:
        return
    total_steps = N * shard.world * args.steps
    out = dict(metric="env-steps/sec (rollout+update)", value=round(total_steps / elapsed, 1), unit="env-steps/s",
               n_gpus=shard.world, steps=args.steps, warmup=args.warmup, ms_per_step=round(1e3 * elapsed / args.steps, 3),
               higher_is_better=True, scaling="weak", vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=f"{model} n_envs={n_envs} n_tsteps={T} 84x84x4 synthetic frames"
                                    f"{' +BPTT' if use_bptt else ''}, {args.optim}, per GPU",
                           n_envs_per_gpu=n_envs, n_tsteps=T, optimizer=args.optim,
                           rollout="hipGraph" if graph is not None else "eager",
                           parallelism=f"dp{shard.world} (rollout shards, 1 RCCL grad all-reduce/update)"),
               last_info={k: round(float(v), 6) for k, v in (info or {}).items()})

    # ---- roofline of the dominant kernel of the update (HIP events on the launch stream)
    if timers is not None:
        summ = timers.summary()
        kern = {k: dict(avg_ms=round(v["avg_ms"], 4), launches=v["launches"]) for k, v in
                sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"])}
        out["update_launch_sites_ms"] = kern
        out["update_launch_sites_note"] = sites_note
        if ugraph is not None:
            out["config"]["update"] = "hipGraph"
        layers = getattr(net, "_cl", None)
        if layers is None and hasattr(net, "_c1"):
            layers = [net._c1, net._c2]
        conv_layers = {l.name: l for l in (layers or [])}
        dom = next(iter(kern), None)
        rollout_ms = sum(a.elapsed_time(b) for a, b in roll_ev) / max(len(roll_ev), 1)
        out["rollout_ms"] = round(rollout_ms, 3)
        fused_step = graph is not None and getattr(net, "_step_supported", lambda: False)()
        n_upd = 2 if ugraph is not None else args.steps          # updates the launch-site timers saw
        dom_ms = summ[dom]["total_ms"] / n_upd if dom is not None else 0.0
        if fused_step and rollout_ms > dom_ms:
            # the one-launch rollout step is the kernel the epoch spends most time in: T+1 launches
            # per hipGraph replay; average launch duration = replay time / (T+1) (includes the
            # inter-node gaps, so `achieved` is a lower bound; profiles/ holds the rocprofv3 average)
            us = rollout_ms * 1e3 / (T + 1)
            fl = step_alg_flops(A) * n_envs
            by = step_alg_bytes() * n_envs
            tf = fl / (us * 1e-6) / 1e12
            out["roofline"] = dict(kernel=f"a3c_step_kernel (B={n_envs}, {T + 1} launches/rollout)", bound="mfma",
                                   achieved=round(tf, 2), peak=F32_PEAK_TFLOPS, unit="TFLOP/s",
                                   frac=round(tf / F32_PEAK_TFLOPS, 4), traffic=None, avg_launch_us=round(us, 2),
                                   hbm_GBs=round(by / (us * 1e-6) / 1e9, 1), hbm_frac=round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                   alg_flops_per_launch=fl, alg_bytes_per_launch=by)
            dom = "a3c_step"
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
        # HBM bytes per launch from the PMC passes committed under profiles/ (collected separately:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on tools/run_kernel.py at this exact size)
        tfile = os.path.join(ROOT, "profiles", "r1_traffic.json")
        key = {"conv1.fwd": "conv1_fwd", "conv1.bwd_weight": "conv1_wgrad", "conv2.bwd_data": "conv2_bwd_data",
               "a3c_step": "a3c_step"}.get(dom)
        if "roofline" in out and key and args.workload == "a3c" and N == 32768 and os.path.exists(tfile) \
                and key in json.load(open(tfile)):
            out["roofline"]["traffic"] = round(json.load(open(tfile))[key]["hbm_bytes_per_launch"])
            out["roofline"]["traffic_source"] = "profiles/r1_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, KB)"
        # always report conv1 forward (north star: HBM GB/s on the conv forward) and the scan
        if "conv1" in conv_layers and "conv1.fwd" in summ:
            d = conv_layers["conv1"].d
            ms = summ["conv1.fwd"]["avg_ms"]
            ach = conv_alg_bytes(d, N) / (ms * 1e-3) / 1e9
            out["conv1_fwd_roofline"] = dict(bound="hbm", avg_ms=round(ms, 4), achieved_GBs=round(ach, 1),
                                             frac=round(ach / HBM_PEAK_GBS, 4),
                                             tflops=round(conv_flops(d, N) / (ms * 1e-3) / 1e12, 2))
    if shard.world == 1:
        out["scan_roofline"] = scan_roofline(dev)
        if not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(model, n_envs, T, use_bptt, A, args.optim)
            except Exception as e:      # noqa: BLE001
                out["cpu_baseline"] = dict(value=None, error=f"{type(e).__name__}: {e}")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
