"""CPU: the host env pool (worker processes + pinned-region protocol, include/a2c_hostpool.h) without a GPU.
The test process plays the device's part of the protocol (post actions, wait for the rec granules, read
the frames) and checks every frame byte, reward and done flag against the same envs stepped in-process
exactly like the reference's loop body does (runner.py:207-227)."""
import os
import sys
import time

import numpy as np
import pytest

from cases import F32FakeEnv, FailingEnv, U8FakeEnv
from a2c_amd.hostpool import IDLE, ROLLOUT, ProcessEnvPool, pool_lib
from a2c_amd.synthetic import TapeEnv


def _ref_step(env, a, shift):
    obs, rew, done, _ = env.step(a + shift)
    if done:
        obs = env.reset()
    return np.asarray(obs), rew, done


@pytest.mark.parametrize("cls,dtype", [(U8FakeEnv, np.uint8), (F32FakeEnv, np.float32), (TapeEnv, np.uint8)])
def test_pool_matches_in_process_envs(cls, dtype):
    B, W, K, shift = 7, 3, 40, 1
    if cls is TapeEnv:
        kws = [dict(env_id=j, length=17, p_done=0.1) for j in range(B)]
    else:
        kws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + j) for j in range(B)]
    pool = ProcessEnvPool(cls, B, env_kwargs=kws, n_workers=W, action_shift=shift, pong=True, register=False)
    try:
        pool.start()
        assert pool.frame_dtype == dtype and pool.frame_shape == (1, 84, 84)
        assert pool.header.n_workers == W and pool.header.frame_stride % 16 == 0
        pool.set_phase(ROLLOUT)
        refs = [cls(**kw) for kw in kws]
        fr = pool.frames_view()
        rew, done = np.zeros(B, np.float32), np.zeros(B, np.float32)
        pool.wait_frames(0)
        pool.unpack(rew, done)
        assert (done == 1).all() and (rew == 0).all()           # frame 0: env.reset(), frame stack restarts
        for j in range(B):
            assert np.array_equal(fr[j], np.asarray(refs[j].reset()).astype(dtype))
        ema, ep = -1.0, [0.0] * B
        n_eps = 0
        for k in range(K):
            acts = ((np.arange(B) * 7 + k) % 3 - (k % 5 == 0)).astype(np.int64)     # includes the -1 of sample_action
            pool.post_actions(acts, seq=k)
            pool.wait_frames(k + 1)
            pool.unpack(rew, done)
            for j in range(B):
                o, r, d = _ref_step(refs[j], int(acts[j]), shift)
                assert np.array_equal(fr[j], o.astype(dtype)), (k, j)
                assert rew[j] == np.float32(r) and done[j] == float(d), (k, j)
                ep[j] += r
                if d or r != 0:       # Pong: an episode ends whenever a point is scored (runner.py:212-217)
                    n_eps += 1
                    ep[j] = 0.0
        assert pool.header.episodes == n_eps and n_eps > 0
        # the EMA is order dependent across workers (as across the reference's processes): check its range only
        assert -1.0 <= pool.rew_ema() <= 1.0
        pool.set_phase(IDLE)
    finally:
        pool.close()
    assert not os.path.exists("/dev/shm/" + pool.name)


def test_pool_worker_exception_reaches_the_gpu_process():
    B = 4
    kws = [dict(env_id=j, fail_at=2 if j == 2 else 10 ** 9) for j in range(B)]
    pool = ProcessEnvPool(FailingEnv, B, env_kwargs=kws, n_workers=2, register=False)
    try:
        pool.start()
        pool.set_phase(ROLLOUT)
        acts = np.zeros(B, np.int64)
        with pytest.raises(RuntimeError, match="env worker"):
            for k in range(5):
                pool.post_actions(acts, seq=k)
                pool.wait_frames(k + 1, timeout=20.0)
    finally:
        pool.close()


def test_pool_workers_do_not_import_torch_or_hip():
    """the worker program and the pool library must stay free of the GPU runtime"""
    import subprocess
    code = ("import sys; sys.argv=['x']; import a2c_amd.hostpool, a2c_amd.hostpool_worker, a2c_amd.synthetic; "
            "a2c_amd.hostpool.pool_lib(); assert 'torch' not in sys.modules, 'torch imported'; "
            "assert not any('amdhip' in l for l in open('/proc/self/maps')), 'HIP runtime mapped'; print('clean')")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(sys.path))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert out.stdout.strip() == "clean", out.stderr


def test_pool_idle_phase_does_not_spin():
    pool = ProcessEnvPool(TapeEnv, 2, env_kwargs=[dict(env_id=j, length=9) for j in range(2)], n_workers=2, register=False)
    try:
        pool.start()
        pool.set_phase(IDLE)
        time.sleep(0.2)
        import psutil
        for p in pool.procs:
            psutil.Process(p.pid).cpu_percent()
        time.sleep(0.5)
        busy = [psutil.Process(p.pid).cpu_percent() for p in pool.procs]
        assert max(busy) < 50.0, busy
    finally:
        pool.close()


def test_native_thread_pool_is_byte_identical_to_python_tape_envs():
    """ThreadEnvPool (pthreads + native tape env, no Python in the loop) == stepping synthetic.TapeEnv objects"""
    from a2c_amd.hostpool import ThreadEnvPool
    B, K = 9, 120
    kws = [dict(env_id=j, length=13, p_done=0.07) for j in range(B)]
    pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, register=False, pong=True)
    try:
        pool.start()
        pool.set_phase(ROLLOUT)
        assert pool.frame_dtype == np.uint8 and pool.header.n_workers == 2
        refs = [TapeEnv(**k) for k in kws]
        fr = pool.frames_view()
        pool.wait_frames(0)
        for j in range(B):
            assert np.array_equal(fr[j], refs[j].reset())
        rew, done = np.zeros(B, np.float32), np.zeros(B, np.float32)
        n_eps = 0
        for k in range(K):
            pool.post_actions(np.full(B, k % 3, np.int64), seq=k)
            pool.wait_frames(k + 1)
            pool.unpack(rew, done)
            for j in range(B):
                o, r, d = _ref_step(refs[j], k % 3, 0)
                assert np.array_equal(fr[j], o) and rew[j] == np.float32(r) and done[j] == float(d), (k, j)
                n_eps += bool(d or r != 0)
        assert pool.header.episodes == n_eps
    finally:
        pool.close()


@pytest.mark.parametrize("kind", ["process", "thread"])
def test_packed_bits_transport_carries_the_same_frames(kind):
    """A2C_FRAME_BITS (binary preprocessors, preprocessing.py:15-16): one bit per pixel in the pinned slots; unpacked
    they are byte for byte the uint8 frames of the same envs, rewards / dones unchanged"""
    from a2c_amd.hostpool import FRAME_BITS, ThreadEnvPool
    B, K = 5, 30
    if kind == "process":
        kws = [dict(env_id=j, rew_period=3 + j % 3, done_period=5 + j) for j in range(B)]
        pool = ProcessEnvPool(U8FakeEnv, B, env_kwargs=kws, n_workers=2, pong=True, register=False, frame_bits=True)
        refs = [U8FakeEnv(**kw) for kw in kws]
    else:
        kws = [dict(env_id=j, length=13, p_done=0.07) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, register=False, pong=True, frame_bits=True)
        refs = [TapeEnv(**k) for k in kws]
    try:
        pool.start()
        pool.set_phase(ROLLOUT)
        h = pool.header
        assert h.frame_dtype == FRAME_BITS and h.frame_bytes == 882 and h.frame_stride == 896 and h.frame_elems == 7056
        assert pool.transport == "bits" and pool.frame_dtype == np.uint8
        pool.wait_frames(0)
        for j in range(B):
            assert np.array_equal(pool.frames_view()[j], np.asarray(refs[j].reset()).astype(np.uint8))
        rew, done = np.zeros(B, np.float32), np.zeros(B, np.float32)
        for k in range(K):
            pool.post_actions(np.full(B, k % 3, np.int64), seq=k)
            pool.wait_frames(k + 1)
            pool.unpack(rew, done)
            fr = pool.frames_view()
            for j in range(B):
                o, r, d = _ref_step(refs[j], k % 3, 0)
                assert np.array_equal(fr[j], np.asarray(o).astype(np.uint8)), (k, j)
                assert rew[j] == np.float32(r) and done[j] == float(d), (k, j)
    finally:
        pool.close()


class _GreyEnv(U8FakeEnv):
    """uint8 frames that are NOT binary"""

    def _frame(self):
        return (super()._frame() * 7).astype(np.uint8)


def test_packed_bits_transport_refuses_non_binary_frames():
    pool = ProcessEnvPool(_GreyEnv, 2, env_kwargs=[dict(env_id=j) for j in range(2)], n_workers=1, register=False,
                          frame_bits=True, frame_shape=(1, 84, 84), frame_dtype=np.uint8)
    try:
        with pytest.raises(RuntimeError, match="env worker"):
            pool.start(timeout=30.0)
            pool.wait_frames(0, timeout=10.0)
    finally:
        pool.close()


@pytest.mark.parametrize("kind", ["process", "thread"])
def test_step_counter_wraps_at_2_31_on_the_rec_side(kind):
    """rec carries (seq << 1 | done) in 32 bits = seq modulo 2^31, cmd carries seq modulo 2^32: an env that has taken
    more than 2^31 steps must keep answering (the host side compares modulo 2^31, like the kernels do)"""
    from a2c_amd.hostpool import ThreadEnvPool
    B, s0 = 3, (1 << 31) - 3
    if kind == "process":
        kws = [dict(env_id=j, rew_period=3, done_period=5 + j) for j in range(B)]
        pool = ProcessEnvPool(U8FakeEnv, B, env_kwargs=kws, n_workers=2, register=False, seq_start=s0)
        refs = [U8FakeEnv(**kw) for kw in kws]
    else:
        kws = [dict(env_id=j, length=13, p_done=0.07) for j in range(B)]
        pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, register=False, seq_start=s0)
        refs = [TapeEnv(**k) for k in kws]
    try:
        pool.start()
        pool.set_phase(ROLLOUT)
        assert pool.seq == s0
        pool.wait_frames(s0, timeout=10.0)
        for j in range(B):
            refs[j].reset()
        rew, done = np.zeros(B, np.float32), np.zeros(B, np.float32)
        for k in range(8):                                    # crosses 2^31 - 1 -> 2^31
            pool.post_actions(np.zeros(B, np.int64), seq=s0 + k)
            pool.wait_frames(s0 + k + 1, timeout=10.0)
            pool.unpack(rew, done)
            fr = pool.frames_view()
            for j in range(B):
                o, r, d = _ref_step(refs[j], 0, 0)
                assert np.array_equal(fr[j], np.asarray(o).astype(np.uint8)) and rew[j] == np.float32(r) and done[j] == float(d)
    finally:
        pool.close()


def test_self_validating_tagged_mirror_of_the_packed_frames(monkeypatch):
    """A2C_TAGGED=1: packed pools keep a second copy of every frame as 16-byte chunks that each carry the step number's low
    16 bits (a2c_hostpool.h): 14 bytes of packed pixels + tag, the last chunk = {reward, done, seq, tag}.  Checked here from
    the host side for both worker kinds: chunk contents == the packed frame, tags == seq & 0xffff, record == rec granule."""
    from a2c_amd.hostpool import ThreadEnvPool
    monkeypatch.setenv("A2C_TAGGED", "1")
    B = 3
    for kind in ("thread", "process"):
        if kind == "process":
            kws = [dict(env_id=j, rew_period=3, done_period=5 + j) for j in range(B)]
            pool = ProcessEnvPool(U8FakeEnv, B, env_kwargs=kws, n_workers=2, register=False, frame_bits=True, seq_start=65533)
        else:
            kws = [dict(env_id=j, length=13, p_done=0.2) for j in range(B)]
            pool = ThreadEnvPool.from_tape_envs([TapeEnv(**k) for k in kws], n_threads=2, register=False, frame_bits=True,
                                                seq_start=65533)
        try:
            pool.start()
            pool.set_phase(ROLLOUT)
            h = pool.header
            assert h.off_tagged and h.tagged_chunks == 64 and h.tagged_stride == 1024          # 84 x 84 = 63 x 112 pixels + the record
            mm = np.frombuffer(pool.region.mm, dtype=np.uint8)
            rew, done = np.zeros(B, np.float32), np.zeros(B, np.float32)
            for k in range(6):                                    # crosses seq & 0xffff == 0
                seq = 65533 + k
                pool.wait_frames(seq, timeout=10.0)
                pool.unpack(rew, done)
                packed = pool.region.frames[:, :pool.frame_bytes]
                for j in range(B):
                    t = mm[h.off_tagged + j * h.tagged_stride:h.off_tagged + (j + 1) * h.tagged_stride].reshape(64, 16)
                    tags = t[:, 14:16].copy().view(np.uint16).reshape(-1)
                    assert (tags == (seq & 0xffff)).all(), (kind, k, j, tags[:4])
                    data = t[:63, :14].reshape(-1)[:pool.frame_bytes]
                    assert np.array_equal(data, packed[j]), (kind, k, j)
                    meta = t[63]
                    assert meta[0:4].copy().view(np.float32)[0] == rew[j] and int(meta[4:8].copy().view(np.uint32)[0]) == int(done[j])
                    assert int(meta[8:12].copy().view(np.uint32)[0]) == seq
                pool.post_actions(np.zeros(B, np.int64), seq=seq)
        finally:
            pool.close()
