"""CPU: host-side logic of the product that needs no kernel -- model containers (state-dict
keys / parameter order / shapes vs the reference's, via the golden-pinned oracle tables), the
sharding helpers, and the world_size-2 gloo path of the multi-GPU update."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import a2c_oracle as O
from cases import MODEL_CASES


@pytest.mark.parametrize("case", MODEL_CASES, ids=[f"{c[0]}-{c[1]}" for c in MODEL_CASES])
def test_model_containers_match_reference_layout(case):
    import a2c_amd
    kind, ss, A, h, _ = case
    net = getattr(a2c_amd.models, kind)(list(ss), A, h_size=h, bnorm=False)
    sd = O.formula_state_dict(kind, ss, A, h)
    mine = net.state_dict()
    assert set(mine) == set(sd)
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    shapes, _ = O.param_shapes(kind, ss, A, h)
    prim = [k for k in shapes if "running" not in k and "num_batches" not in k]
    assert [n for n, _ in net.named_parameters()] == prim
    assert set(net._arena_order()) == set(prim)
    assert net.is_recurrent == (kind in ("GRUModel", "GRUFCModel"))
    net.load_state_dict(sd)
    net.req_grads(False)
    assert not any(p.requires_grad for p in net.parameters())


def test_same_seed_same_init_as_torch_modules():
    """containers are real torch modules created in the reference's order -> same RNG stream"""
    import a2c_amd
    torch.manual_seed(0)
    a = a2c_amd.models.A3CModel([4, 84, 84], 3, h_size=256)
    torch.manual_seed(0)
    c1 = torch.nn.Conv2d(4, 16, 8, stride=4)
    assert torch.equal(a.state_dict()["conv1.0.weight"], c1.weight)


def test_shard_slot_ranges():
    from a2c_amd.parallel import Shard
    for world in (1, 2, 3, 8):
        got = []
        for r in range(world):
            lo, hi = Shard(r, world).slot_range(2048 if world != 3 else 10)
            got += list(range(lo, hi))
        assert got == list(range(2048 if world != 3 else 10))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from a2c_amd.parallel import Shard, moments_to_mean_std
    sh = Shard.from_env()
    assert sh.active and sh.world == world and sh.rank == rank
    # the sharded update's two exchanges: advantage moments (2 doubles) and the flat gradient arena
    full = torch.arange(24, dtype=torch.float32) * 0.37 - 3
    lo, hi = sh.slot_range(24)
    mine = full[lo:hi].double()
    sums = torch.stack([mine.sum(), (mine * mine).sum()])
    sh.allreduce_(sums)
    n_global = sh.global_count(hi - lo)
    mean, std = moments_to_mean_std(sums, n_global)
    grads = torch.full((1000,), float(rank + 1))
    sh.allreduce_(grads)
    sh.barrier()
    q.put((rank, n_global, mean, std, float(grads[0]), float(grads.sum())))
    dist.destroy_process_group()


def test_gloo_world2_sharded_statistics_and_gradient_allreduce():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    full = torch.arange(24, dtype=torch.float32) * 0.37 - 3
    for rank, n_global, mean, std, g0, gs in res:
        assert n_global == 24
        assert mean == pytest.approx(float(full.double().mean()), rel=1e-12)
        assert std == pytest.approx(float(full.double().std()), rel=1e-12)     # unbiased, like Tensor.std
        assert g0 == 3.0 and gs == 3000.0                                       # sum over ranks: 1 + 2


def test_split_count_rule_for_few_tile_products(monkeypatch):
    """ops.pick_splitk: products of a handful of 128 x 128 tiles over a short K (GRUModel's resize_emb at rollout batch) get
    at most 256 / tiles slabs -- the reduce reads every slab back; big products keep the 512-workgroup target; the
    environment override wins; products that fill the chip are not split."""
    from a2c_amd import ops
    monkeypatch.delenv("A2C_SPLITK_TARGET", raising=False)
    monkeypatch.delenv("A2C_SPLITK_MIN_K", raising=False)
    assert ops.pick_splitk(256, 256, 2304) == 64                 # 4 tiles: 256 / 4
    assert ops.pick_splitk(32, 2000, 28224) == 32                # 16 tiles, long K: 512 / 16 (the 226 MB weight stream)
    assert ops.pick_splitk(256, 256, 8192) == 128                # K > 4096: the general rule
    assert ops.pick_splitk(32768, 2000, 28224) == 1
    monkeypatch.setenv("A2C_SPLITK_TARGET", "512")
    assert ops.pick_splitk(256, 256, 2304) == 72                 # min(512 / 4, 2304 / 32)
