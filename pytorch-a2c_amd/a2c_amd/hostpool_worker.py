"""Env worker process of the host pool (``python -m a2c_amd.hostpool_worker``, spec pickled on stdin).

Plays, for a block of envs, the host half of one iteration of the reference's rollout loop
(runner.py:207-227): take the sampled action, ``env.step(action + action_shift)``, keep the episode
reward and its EMA (runner.py:211-217), ``env.reset()`` on a real done (utils.py:36-38), and hand the
prepped frame + reward + done to the device through the pinned region.  numpy + ctypes only.
"""
import ctypes
import os
import pickle
import sys
import traceback

import numpy as np


def main():
    spec = pickle.loads(sys.stdin.buffer.read())
    from a2c_amd import hostpool as hp
    lib = hp.pool_lib()
    reg = hp.Region(spec["shm"])
    if lib.a2c_pool_check(reg.base):
        raise RuntimeError("pool region is not formatted")
    reg.bind()
    base, env0, n, wid = reg.base, int(spec["env0"]), int(spec["n"]), int(spec["worker"])
    try:
        run(spec, lib, reg, base, env0, n)
    except BaseException:      # noqa: BLE001 -- the GPU process must learn about it instead of waiting forever
        traceback.print_exc()
        lib.a2c_pool_worker_failed(base, wid)
        raise


def run(spec, lib, reg, base, env0, n):
    from a2c_amd import hostpool as hp
    shift, pong = int(spec["action_shift"]), bool(spec["pong"])
    fdt = np.dtype(spec["frame_dtype"])
    fshape = tuple(spec["frame_shape"])
    bits = reg.header.frame_dtype == hp.FRAME_BITS      # packed transport: the frame is packed 8 pixels to a byte
    fbytes = int(np.prod(fshape)) * fdt.itemsize
    assert (fbytes + 7) // 8 == reg.header.frame_bytes if bits else fbytes == reg.header.frame_bytes
    parent = int(spec["parent"])
    envs = [spec["factory"](**kw) for kw in spec["env_kwargs"]]
    if spec.get("probe_reset"):
        for env in envs:
            env.reset()

    def as_frame(obs):
        f = np.ascontiguousarray(obs, dtype=fdt)
        if f.nbytes != fbytes:
            raise ValueError(f"env returned a frame of {f.nbytes} bytes, the pool was sized for {fbytes}")
        return f

    take, episode = lib.a2c_pool_take, lib.a2c_pool_episode
    if bits:
        def publish(b, j, ptr, s, rew, done):
            if lib.a2c_pool_publish_bits(b, j, ptr, s, rew, done):
                raise ValueError("frame_bits transport: the env returned a pixel that is neither 0 nor 1")
    else:
        publish = lib.a2c_pool_publish
    seq0 = int(reg.header.seq_start)
    for i, env in enumerate(envs):          # frame 0 = env.reset(); done = 1: the frame stack starts from zeros
        f = as_frame(env.reset())
        publish(base, env0 + i, f.ctypes.data, seq0, 0.0, 1)
    lib.a2c_pool_worker_ready(base)
    next_seq = np.full(n, seq0, dtype=np.uint32)  # the env step each env waits for (cmd granules: modulo 2^32)
    nsp = next_seq.ctypes.data
    ep_rew = [0.0] * n
    act = ctypes.c_int32(0)
    actp = ctypes.addressof(act)
    spin_ns = 500_000_000
    while True:
        i = take(base, env0, n, nsp, spin_ns, actp)
        if i == -2:
            break
        if i < 0:
            if os.getppid() != parent:      # the GPU process is gone
                break
            continue
        obs, rew, done, _ = envs[i].step(act.value + shift)
        ep_rew[i] += rew
        reset = bool(done)
        if pong and rew != 0:
            done = True
        if done:                             # runner.py:215-217
            episode(base, ep_rew[i])
            ep_rew[i] = 0.0
        if reset:
            obs = envs[i].reset()
        f = as_frame(obs)
        s = (int(next_seq[i]) + 1) & 0xffffffff
        next_seq[i] = s
        publish(base, env0 + i, f.__array_interface__["data"][0], s, rew, 1 if reset else 0)


if __name__ == "__main__":
    main()
