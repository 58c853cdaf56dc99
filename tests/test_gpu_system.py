"""-m gpu: system-level behaviour of the drop-in -- sharded update == single-process update,
Runner.run queue protocol, StatsRunner, norm_returns, public autograd path vs Updater path."""
import os
import queue
import socket
import threading

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from oracle import a2c_oracle as O  # noqa: E402
from cases import base_hyps, hashf, synth_shared  # noqa: E402
from test_gpu_kernels import close  # noqa: E402
from test_gpu_models import make_net, _fake_pool, _datas  # noqa: E402

DEV = "cuda"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, kind, opt, q, ss=(4, 84, 84), backend="gloo", force=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", A2C_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if force:
        os.environ["A2C_FORCE_COLLECTIVES"] = "1"
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "pytorch-a2c_amd"), os.path.join(root, "tests", "golden"),
                    os.path.join(root, "tests")]
    import torch.distributed as dist
    from a2c_amd.parallel import Shard
    from a2c_amd.updater import Updater
    from test_gpu_models import make_net
    from cases import base_hyps, synth_shared
    torch.cuda.set_device(0)
    A, h, R, T = 3, 256 if ss[-1] == 84 else 32, 4, 6
    sh = Shard.from_env()
    assert sh.active and dist.get_backend() == backend
    net = make_net(kind, ss, A, h)
    lo, hi = sh.slot_range(R)
    hyps = base_hyps(n_tsteps=T, n_rollouts=hi - lo, optim_type=opt, h_size=h)
    upd = Updater(net, hyps, shard=sh)
    infos = []
    for u in range(2):
        D = synth_shared(kind, ss, A, h, R, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
        Dl = {k: v[lo * T:hi * T].cuda() for k, v in D.items()}          # this rank's contiguous slots
        infos.append(upd.update_model(Dl))
    q.put((rank, infos, [p.detach().cpu().numpy() for p in net.parameters()]))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,opt,ss", [("A3CModel", "RMSprop", (4, 84, 84)), ("GRUModel", "Adam", (4, 84, 84)),
                                         ("ConvModel", "RMSprop", (4, 84, 84)), ("ConvModel", "Adam", (4, 20, 20))])
def test_sharded_update_equals_single_process(kind, opt, ss):
    """world_size 2 (two processes, gloo all-reduce of the CUDA gradient arena) must reproduce
    the single-process update on the whole batch: losses, GradNorm and every parameter.
    ConvModel = BASELINE config 5's model (230 MB gradient message at 84x84)."""
    from a2c_amd.updater import Updater
    A, h, R, T = 3, 256 if ss[-1] == 84 else 32, 4, 6
    net = make_net(kind, ss, A, h)
    upd = Updater(net, base_hyps(n_tsteps=T, n_rollouts=R, optim_type=opt, h_size=h))
    ref_infos = []
    for u in range(2):
        D = synth_shared(kind, ss, A, h, R, T, seed=700 + 10 * u, recurrent=net.is_recurrent)
        ref_infos.append(upd.update_model({k: v.cuda() for k, v in D.items()}))
    ref_params = [p.detach().cpu().numpy() for p in net.parameters()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, kind, opt, q, ss)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, infos, params in res:
        for u in range(2):
            for k in ref_infos[u]:
                assert infos[u][k] == pytest.approx(ref_infos[u][k], rel=2e-5, abs=1e-7), (rank, u, k)
        for a, b in zip(params, ref_params):
            # (two RMSprop steps of lr 1e-4: a weight whose gradient is at fp32-noise level moves by up to lr * 10 in either
            # direction; the shards sum S = db^T a2 per rank before the all-reduce, the single process over the whole batch)
            np.testing.assert_allclose(a, b, rtol=0, atol=4e-6)
    for a, b in zip(res[0][2], res[1][2]):          # ranks stay bit-identical without any broadcast
        assert np.array_equal(a, b)


def test_rccl_branch_executes_at_world_1():
    """backend "nccl" (= RCCL on ROCm) with the collectives forced on at world_size 1: the all-reduce of the
    gradient arena, of the advantage moments and of the loss sums really go through RCCL, and the update equals
    the plain single-process one."""
    from a2c_amd.updater import Updater
    kind, opt, ss, A, h, R, T = "A3CModel", "RMSprop", (4, 84, 84), 3, 256, 4, 6
    net = make_net(kind, ss, A, h)
    upd = Updater(net, base_hyps(n_tsteps=T, n_rollouts=R, optim_type=opt, h_size=h))
    ref_infos = []
    for u in range(2):
        D = synth_shared(kind, ss, A, h, R, T, seed=700 + 10 * u, recurrent=False)
        ref_infos.append(upd.update_model({k: v.cuda() for k, v in D.items()}))
    ref_params = [p.detach().cpu().numpy() for p in net.parameters()]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_shard_worker, args=(0, 1, _free_port(), kind, opt, q, ss, "nccl", True))
    p.start()
    rank, infos, params = q.get(timeout=600)
    p.join(60)
    assert p.exitcode == 0
    for u in range(2):
        for k in ref_infos[u]:
            assert infos[u][k] == pytest.approx(ref_infos[u][k], rel=2e-5, abs=1e-7), (u, k)
    for a, b in zip(params, ref_params):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)


def test_runner_run_queue_protocol():
    """Runner.run keeps the reference's gate/stop protocol (runner.py:169-172)"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 3, 4, 3, (4, 84, 84)
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, n_envs=B)
    net = make_net("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False)
    gate, stop, rew = queue.Queue(), queue.Queue(), queue.Queue(1)
    rew.put(-1)
    ekws = [dict(env_id=j, rew_period=2, done_period=5) for j in range(B)]
    r = Runner(D, hyps, gate, stop, rew, env_pool=_fake_pool(ekws))
    th = threading.Thread(target=r.run, args=(net,), daemon=True)
    for i in range(B):
        gate.put(i)
    th.start()
    got = sorted(stop.get(timeout=120) for _ in range(B))
    assert got == list(range(B))
    torch.cuda.synchronize()
    assert float(D["dones"].view(B, T)[:, -1].min()) == 1.0          # every slot ends done (runner.py:244)
    assert float(D["states"].abs().sum()) > 0
    for i in range(B):                                                 # a second epoch re-opens the gate
        gate.put(i)
    assert sorted(stop.get(timeout=120) for _ in range(B)) == list(range(B))


def test_runner_run_plays_a_partial_batch_after_the_gate_timeout():
    """a caller that re-opens the gate with FEWER tokens than hyps['n_rollouts'] (or two Runners on one gate_q) must not
    hang Runner.run forever: after hyps['gate_timeout_s'] without a further token what arrived is played and answered"""
    from a2c_amd.runner import Runner
    B, T, A, ss = 3, 4, 3, (4, 84, 84)
    hyps = base_hyps(env_type="FakePong-v0", n_tsteps=T, n_rollouts=B, n_envs=B, gate_timeout_s=0.5)
    net = make_net("A3CModel", ss, A, 256)
    D = _datas(B * T, ss, False)
    gate, stop, rew = queue.Queue(), queue.Queue(), queue.Queue(1)
    rew.put(-1)
    ekws = [dict(env_id=j, rew_period=2, done_period=5) for j in range(B)]
    r = Runner(D, hyps, gate, stop, rew, env_pool=_fake_pool(ekws))
    th = threading.Thread(target=r.run, args=(net,), daemon=True)
    gate.put(0)
    gate.put(1)                      # two of three tokens
    th.start()
    assert sorted(stop.get(timeout=120) for _ in range(2)) == [0, 1]
    torch.cuda.synchronize()
    d = D["dones"].view(B, T)
    assert float(d[:2, -1].min()) == 1.0 and float(D["states"].view(B, -1)[2].abs().sum()) == 0.0      # slot 2 untouched
    gate.put(None)                   # shutdown token
    th.join(30)
    assert not th.is_alive() and r.error is None


def test_stats_runner_and_get_action():
    from a2c_amd.runner import StatsRunner

    class Env:          # SequentialEnvironment surface over the fake env
        def __init__(self):
            self.e = O.FakeEnv(env_id=3, rew_period=2, done_period=6)

        def reset(self):
            return self.e.reset()

        def step(self, a):
            return self.e.step(a)

        def get_action(self, logits):
            from a2c_amd.utils import sample_action
            p = torch.softmax(logits, -1)
            return int(sample_action(p, torch.full((1,), 0.5)).item())
    hyps = base_hyps(env_type="FakeBreakout", n_test_eps=3)
    for kind in ("A3CModel", "GRUModel"):
        net = make_net(kind, (4, 84, 84), 3, 256)
        val = StatsRunner(hyps, env=Env()).rollout(net)
        assert np.isfinite(val)


def test_stats_runner_serial_matches_reference_golden(golden):
    """one env: the reference's loop on the device net reproduces the reference's own evaluation (g8)"""
    from cases import STATS_CASES
    from a2c_amd.runner import StatsRunner
    from a2c_amd.utils import sample_action
    g = golden["g8_stats"]
    for (name, kind, env_type, n_eps, ekw, A) in STATS_CASES:
        hyps = base_hyps(env_type=env_type, n_test_eps=n_eps, action_shift=1 if "Pong" in env_type else 0)
        it = iter(g[f"{name}_uniforms"])

        class Env:          # SequentialEnvironment surface over the fake env, uniforms from the fixture
            def __init__(self):
                self.e = O.FakeEnv(**ekw)
                self.e.reset()

            def reset(self):
                return self.e.reset()

            def step(self, a):
                return self.e.step(a)

            def get_action(self, logits):
                return int(sample_action(torch.softmax(logits, -1), torch.tensor([float(next(it))])).item())
        net = make_net(kind, (4, 84, 84), A, 256)
        assert StatsRunner(hyps, env=Env()).rollout(net) == pytest.approx(float(g[f"{name}_avg_rew"]), abs=1e-12)


@pytest.mark.parametrize("kind", ["A3CModel", "GRUModel", "FCModel"])
def test_stats_runner_batched_matches_oracle(kind):
    """row f3: n_test_eps envs in lock-step on the device, one episode each == the oracle's restatement of the
    reference loop (runner.py:274-314) run on each env for one episode with the same uniforms"""
    from a2c_amd.runner import StatsRunner
    E, A, ss = 6, 3, (4, 84, 84)
    pong = kind == "A3CModel"
    hyps = base_hyps(env_type="FakePong-v0" if pong else "FakeBreakout", n_test_eps=E, action_shift=1 if pong else 0)
    ekws = [dict(env_id=20 + j, rew_period=3 + j % 3, done_period=4 + j) for j in range(E)]
    # uniforms < 0.999: at u ~ 1 the reference falls through to -1 where the rollout samplers return the last action
    us = torch.from_numpy(hashf(64 * E, 4711, 0, 0.999).reshape(64, E))
    usd = us.to(DEV)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)

    def mk(k):
        e = O.FakeEnv(**k)
        e.reset()
        return e
    got = StatsRunner(hyps, envs=[mk(k) for k in ekws], uniform_fn=lambda t, n: usd[t, :n].contiguous()).rollout(net)
    want = 0.0
    for j in range(E):
        it = iter([float(us[t, j]) for t in range(64)])
        want += O.stats_rollout(onet, mk(ekws[j]), hyps, 1, lambda it=it: next(it))
    assert got == pytest.approx(want / E, abs=1e-12)


def test_norm_returns_matches_oracle():
    from a2c_amd.updater import Updater
    kind, ss, A, h, R, T = "A3CModel", (4, 84, 84), 3, 256, 3, 5
    hyps = base_hyps(n_tsteps=T, n_rollouts=R, norm_returns=True)
    net, onet = make_net(kind, ss, A, h), O.OracleNet(kind, ss, A, h)
    upd, oupd = Updater(net, hyps), O.OracleUpdater(onet, hyps)
    for u in range(2):
        D = synth_shared(kind, ss, A, h, R, T, seed=800 + u, recurrent=False)
        info = upd.update_model({k: v.cuda() for k, v in D.items()})
        oinfo = oupd.update_model(D)
        for k in oinfo:
            assert info[k] == pytest.approx(oinfo[k], rel=5e-5, abs=2e-6), (u, k)
    assert upd.ret_mean == pytest.approx(float(oupd.ret_mean), rel=1e-5)
    assert upd.ret_std == pytest.approx(float(oupd.ret_std), rel=1e-5)


def test_updater_bptt_api_and_new_lr_quirk():
    from a2c_amd.updater import Updater
    kind, ss, A, h, R, T = "GRUModel", (4, 84, 84), 3, 256, 2, 4
    hyps = base_hyps(n_tsteps=T, n_rollouts=R, use_bptt=True)
    net, onet = make_net(kind, ss, A, h), O.OracleNet(kind, ss, A, h)
    D = synth_shared(kind, ss, A, h, R, T, seed=810, recurrent=True)
    upd = Updater(net, hyps)
    v, l = upd.bptt(D["states"].cuda(), D["h_states"].cuda(), D["dones"].cuda())
    with torch.no_grad():
        ov, ol = O.bptt(onet, D["states"], D["h_states"], D["dones"], hyps)
    close("bptt vals", v, ov, 1e-5, 1e-5)
    close("bptt logits", l, ol, 1e-5, 1e-5)
    upd.update_model({k: t.cuda() for k, t in D.items()})
    upd.new_lr(5e-5)                       # like the reference, loading the old state restores the old lr
    assert upd.optim.param_groups[0]["lr"] == 1e-4


def test_train_driver_two_epochs(tmp_path):
    """row f1: the reference's epoch loop (gate/stop barrier, save cadence, files) on the engine"""
    from a2c_amd.training import train
    hyps = dict(exp_name="t", main_path=str(tmp_path), model="A3CModel", env_type="FakeBreakout", n_envs=3, n_rollouts=3,
                n_tsteps=5, max_tsteps=1e9, action_size=3, seed=1, n_test_eps=1)
    best = train(None, hyps, verbose=False, env_fn=lambda j: O.FakeEnv(env_id=j, rew_period=3, done_period=7),
                 max_epochs=2)
    folder = os.path.join(str(tmp_path), "t", "t_0")
    assert sorted(os.listdir(folder)) == ["best_net.p", "log.txt", "net.p", "optim.p"]
    sd = torch.load(os.path.join(folder, "net.p"))
    assert set(sd.keys()) == set(O.formula_state_dict("A3CModel", (4, 84, 84), 3, 256).keys())     # reference key set
    osd = torch.load(os.path.join(folder, "optim.p"))
    assert len(osd["param_groups"][0]["params"]) == 12 and 6 not in osd["state"]                    # emb_bnorm: no state
    log = open(os.path.join(folder, "log.txt")).read()
    assert "Step:15" in log and "Step:30" in log and "BestRew" in log
    assert np.isfinite(best)
    # a checkpoint written by the driver loads into a fresh net and into the reference-shaped oracle
    net = make_net("A3CModel", (4, 84, 84), 3, 256)
    net.load_state_dict(sd)
    onet = O.OracleNet("A3CModel", (4, 84, 84), 3, 256, state_dict={k: v.cpu() for k, v in sd.items()})
    x = torch.from_numpy(O.formula_frames(2, (4, 84, 84), seed=9, binary=True))
    with torch.no_grad():
        a, b = net(x), onet(x)
    close("val", a[0], b[0], 1e-5, 1e-5)
    close("pi", a[1], b[1], 1e-5, 1e-5)


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as the driver calls it (no launcher): the parent spawns torch.distributed.run
    before touching the GPU and relays rank 0's JSON line.  The two ranks share cuda:0 here (gloo), each with
    its own pinned pool region and env threads."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, A2C_BENCH_ONE_DEVICE="1", A2C_DIST_BACKEND="gloo")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--n-envs", "16", "--n-workers", "1", "--sustain-steps", "0", "--no-cpu-baseline", "--no-configs",
                          "--no-secondary"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["dist_backend"] == "gloo" and d["allreduce_bytes"] > 0 and d["allreduce_ms_per_update"] > 0
    assert d["config"]["ingest"].startswith("host-pinned/") and d["config"]["transport"] == "bits"
    assert d["config"]["update"] == "3-hipGraphs/2-collectives"            # the sharded update is not eager
    assert len(lines[0]) <= 4096


def test_bench_strong_scaling_conv_world2():
    """--global-envs over two ranks (contiguous slot shards of ONE global batch) with ConvModel: the 230 MB gradient
    arena as one message per update, loss means over the global batch; gloo, both ranks on cuda:0"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, A2C_BENCH_ONE_DEVICE="1", A2C_DIST_BACKEND="gloo")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--workload", "conv", "--global-envs", "16", "--n-workers", "1", "--sustain-steps", "0",
                          "--no-cpu-baseline", "--no-configs", "--no-secondary"], capture_output=True, text=True, env=env,
                         timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0
    assert d["config"]["n_envs"] == 8 and d["allreduce_bytes"] > 200e6
    full = json.load(open(os.path.join(root, d["full_report"])))           # (everything the line no longer carries)
    assert all(np.isfinite(v) for v in full["last_info"].values())


def test_bench_world2_a3c_ring_with_two_env_threads_per_rank():
    """Multi-GPU readiness without the node: two ranks (gloo, both on cuda:0), the HEADLINE path per rank -- ring kernel,
    packed transport, single-frame store -- with only 2 native env threads each (what a rank gets when 8 ranks share a
    16-CPU quota): no hand-shake time-out, finite losses, the sharded update as hipGraphs around its two collectives."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, A2C_BENCH_ONE_DEVICE="1", A2C_DIST_BACKEND="gloo")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
                          "--n-envs", "64", "--n-workers", "2", "--sustain-steps", "0", "--no-cpu-baseline", "--no-configs",
                          "--no-secondary"], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["env_workers"] == 2 and d["config"]["n_envs"] == 64
    assert d["config"]["ingest"] == "host-pinned/zero-copy-persistent/native"
    assert d["config"]["states_layout"] == "u8-frame-store+lazy-fp32-states"
    assert d["config"]["update"] == "3-hipGraphs/2-collectives"
    full = json.load(open(os.path.join(root, d["full_report"])))
    assert all(np.isfinite(v) for v in full["last_info"].values()), full["last_info"]
    assert d["roofline"]["kernel"].startswith("a3c_ring_kernel")


# ---------------------------------------------------------------------------------------------- the update through torch.ops
@pytest.mark.parametrize("name", ["a3c_rms", "a3c_adam", "conv_small_rms", "gru_bptt_rms", "fc_cartpole_rms", "grufc_bptt_rms"])
def test_eager_update_through_torch_custom_ops_is_bit_identical(name, monkeypatch):
    """north_star / SURVEY 8b: "called from Python through PyTorch-ROCm custom ops".  A2C_TORCH_OPS=1 routes every kernel launch
    of the eager Updater.update_model (scans, moments, loss, every conv / GEMM / GRU / LayerNorm pass, clip + optimiser, the
    re-derivation of the inference weights) through torch.ops.a2c_mi355x.abi_<entry point> -- one in-place op per launcher of
    include/a2c_mi355x.h, generated from the header, taking (owning tensor, byte offset) for every pointer -- instead of the
    ctypes binding.  Same C function, same arguments: the two updates of each g6 case leave bit-identical infos, gradient
    arenas and parameters; no launch of the update falls back to ctypes and no address goes unresolved."""
    from a2c_amd import ops
    from a2c_amd.updater import Updater
    from cases import UPDATE_CASES, synth_shared
    case = next(c for c in UPDATE_CASES if c[0] == name)
    _, kind, ss, A, h, R_, T, opt, norm_advs, nstep, use_bptt, n_upd = case
    res = {}
    for mode in ("ctypes", "torch_ops"):
        if mode == "torch_ops":
            monkeypatch.setenv("A2C_TORCH_OPS", "1")
            ab = ops.torch_abi()
            ab.stats.update(torch_ops=0, ctypes=0, by_name={}, unresolved={})
        net = make_net(kind, ss, A, h)
        hyps = base_hyps(n_tsteps=T, n_rollouts=R_, optim_type=opt, norm_advs=norm_advs, use_nstep_rets=nstep, use_bptt=use_bptt,
                         h_size=h)
        upd = Updater(net, hyps)
        out = []
        for u in range(n_upd):
            D = {k: v.to(DEV) for k, v in synth_shared(kind, ss, A, h, R_, T, seed=700 + 10 * u, recurrent=net.is_recurrent).items()}
            info = upd.update_model(D)
            torch.cuda.synchronize()
            out.append((dict(info), net._arena.grads.cpu().clone(), net._arena.params.cpu().clone()))
        res[mode] = out
    monkeypatch.delenv("A2C_TORCH_OPS")
    st = ops.torch_abi().stats
    assert st["unresolved"] == {} and st["ctypes"] == 0, st
    assert st["torch_ops"] >= 10 * n_upd and "a2c_gae_returns_fused" in st["by_name"] and "a2c_loss_fwd_bwd" in st["by_name"], st
    assert any(k.startswith("a2c_clip_") for k in st["by_name"]) and any(k.startswith("a2c_gemm") for k in st["by_name"]), st
    for (ia, ga, pa), (ib, gb, pb) in zip(res["ctypes"], res["torch_ops"]):
        assert ia == ib, (ia, ib)
        assert torch.equal(ga, gb) and torch.equal(pa, pb)
    print(f"[torch ops] {name}: {st['torch_ops']} launches through torch.ops.a2c_mi355x.abi_*: "
          + ", ".join(f"{k[4:]} x{v}" for k, v in sorted(st["by_name"].items())))


def test_torch_abi_ops_validate_their_arguments():
    """the generated ops refuse CPU tensors, offsets outside the owning tensor and wrong argument counts, and surface the
    launcher's own error codes"""
    from a2c_amd import ops
    o = ops.load_torch_ops()
    x = torch.arange(8, dtype=torch.float32, device=DEV)
    y = torch.zeros(8, device=DEV)
    o.abi_add([x, x, y], [0, 0, 0], [8], [])
    torch.cuda.synchronize()
    assert torch.equal(y, 2 * x)
    o.abi_add([x, x, y], [16, 0, 16], [4], [])                  # rows of larger buffers: byte offsets
    torch.cuda.synchronize()
    assert torch.equal(y[4:], x[4:] + x[:4])
    with pytest.raises(RuntimeError):
        o.abi_add([x, x, y], [0, 0, 64], [8], [])               # offset outside the tensor
    with pytest.raises(RuntimeError):
        o.abi_add([x, x], [0, 0], [8], [])                      # wrong number of buffers
    with pytest.raises((RuntimeError, NotImplementedError)):
        o.abi_add([x.cpu(), x.cpu(), y.cpu()], [0, 0, 0], [8], [])
    with pytest.raises(RuntimeError, match="invalid argument"):
        o.abi_discount_scan([x, x, y, torch.empty(0, device=DEV)], [0, 0, 0, 0], [-1, 4], [0.9])


@pytest.mark.parametrize("kind,ingest", [("GRUModel", "relay"), ("ConvModel", "relay"), ("A3CModel", "memcpy"), ("FCModel", "relay")])
def test_rollout_through_torch_custom_ops_matches_oracle(kind, ingest, monkeypatch):
    """A2C_TORCH_OPS=1 on the ROLLOUT side: every per-step launch whose pointers live in torch tensors (conv / GEMM / GRU cell /
    heads / bookkeeping kernels) goes through torch.ops.a2c_mi355x.abi_*; the launches that are handed addresses of the pinned
    pool region (a2c_pool_ingest*, a2c_pool_publish_actions, a2c_store_u32_system: host memory mapped into the device, not a
    tensor) and the argument-block launchers fall back to ctypes, call by call.  Two rounds against the oracle, as
    tests/test_gpu_ingest.py::test_process_pool_rollouts_match_oracle does for the ctypes binding."""
    from a2c_amd import ops
    from a2c_amd.runner import Runner
    from cases import U8FakeEnv
    from test_gpu_ingest import _compare_round, _oracle_rollouts, _pool
    monkeypatch.setenv("A2C_TORCH_OPS", "1")
    ab = ops.torch_abi()
    ab.stats.update(torch_ops=0, ctypes=0, by_name={}, unresolved={})
    B, T, A, ss = 5, 6, 3, (4, 84, 84)
    ekws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + 2 * j) for j in range(B)]
    hyps = base_hyps(env_type="FakePong-v0" if kind == "A3CModel" else "FakeBreakout", n_tsteps=T, n_rollouts=B,
                     action_shift=0, n_envs=B, env_timeout_s=20.0)
    net = make_net(kind, ss, A, 256)
    onet = O.OracleNet(kind, ss, A, 256)
    D = _datas(B * T, ss, net.is_recurrent, actions_on_host=False)
    us = torch.from_numpy(hashf(2 * T * B, 901, 0, 1).reshape(2, T, B))
    usd = us.to(DEV)
    rnd = [0]
    pool = _pool(U8FakeEnv, ekws, 2, pong="Pong" in hyps["env_type"])
    r = Runner(D, hyps, None, None, None, env_pool=pool, ingest=ingest,
               uniform_fn=lambda t, Bn, env0: usd[rnd[0], t, env0:env0 + Bn].contiguous())
    try:
        refs = _oracle_rollouts(kind, onet, hyps, ekws, us, 2, B, T, ss)
        for rnd[0] in range(2):
            r.rollout(net, list(range(B)), hyps)
            r.finish()
            _compare_round(D, refs[rnd[0]], net.is_recurrent)
    finally:
        r.close()
    st = ab.stats
    # (A3CModel's per-step work is ONE argument-block launch, a2c_a3c_step: ctypes by design; its weight preparation and the
    # first frame stack still go through the ops)
    assert st["torch_ops"] > (20 if kind != "A3CModel" else 2), st
    # what may fall back: launches that take addresses inside the pinned pool region (not torch memory)
    assert set(st["unresolved"]) <= {"a2c_pool_ingest", "a2c_pool_ingest_bits", "a2c_pool_ingest_post", "a2c_pool_publish_actions",
                                     "a2c_store_u32_system", "a2c_heads_fused_publish", "a2c_memcpy_async", "a2c_unpack_bits"}, st["unresolved"]
    print(f"[torch ops] rollout {kind}/{ingest}: {st['torch_ops']} launches through abi_* ops, {st['ctypes']} through ctypes "
          f"({dict(st['unresolved'])})")


def test_integration_md_torch_ops_examples_run():
    """the two abi_* calls INTEGRATION.md shows a reference-side caller: the in-place GAE + returns scan and conv1 over row t
    of every slot of a rollout-major states buffer (byte offsets into the owning tensors, not views) -- against the ctypes path"""
    from a2c_amd import ops
    o = ops.load_torch_ops()
    R, T, gamma, lam = 3, 5, 0.99, 0.98
    N = R * T
    deltas, rewards = torch.from_numpy(hashf(N, 1, -1, 1)).to(DEV), torch.from_numpy(hashf(N, 2, -1, 1)).to(DEV)
    dones = (torch.from_numpy(hashf(N, 3)) < 0.2).float().to(DEV)
    dones[T - 1::T] = 1
    advs, rets, a2, r2 = (torch.empty(N, device=DEV) for _ in range(4))
    null = torch.empty(0, device=DEV)
    o.abi_gae_returns_fused([deltas, rewards, dones, advs, rets, null], [0] * 6, [R, T], [gamma * lam, gamma])
    ops.gae_returns(deltas, rewards, dones, gamma * lam, gamma, R, T, a2, r2)
    torch.cuda.synchronize()
    assert torch.equal(advs, a2) and torch.equal(rets, r2)
    # conv1 of A3CModel over state t of every slot
    states = torch.from_numpy(hashf(N * 28224, 4).reshape(N, 4, 84, 84)).to(DEV)
    w, bias = torch.from_numpy(hashf(16 * 4 * 64, 5, -.1, .1).reshape(16, 4, 8, 8)).to(DEV), torch.from_numpy(hashf(16, 6)).to(DEV)
    d = ops.conv_desc(4, 84, 84, 16, 8, 4, 0)
    wprep = torch.empty(ops.conv_prep_floats(d, 0), device=DEV)
    ops.conv_prep(d, 0, w, wprep)
    t = 2
    out, want = torch.empty(R, 16, 20, 20, device=DEV), torch.empty(R, 16, 20, 20, device=DEV)
    o.abi_conv2d_fwd([states, wprep, bias, out], [4 * t * 28224, 0, 0, 0], [4, 84, 84, 16, 8, 4, 0, 20, 20, T * 28224, 1, 6400, R], [])
    ops.conv_fwd(d, states[t].data_ptr(), T * 28224, wprep, bias, True, want, R)
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    ref = torch.relu(torch.nn.functional.conv2d(states[t::T].cpu().double(), w.cpu().double(), bias.cpu().double(), stride=4))
    close("conv1 rows", out, ref, 1e-5, 1e-5)
