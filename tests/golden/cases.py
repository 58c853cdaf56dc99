"""Case tables and closed-form input generators shared by make_golden.py (which
runs the REFERENCE on them, build container only) and by the tests (which run
the oracle / the HIP path on the same inputs and compare with the recorded
reference outputs).  Nothing here comes from the reference."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import a2c_oracle as O  # noqa: E402

null_prep = lambda pic: pic[None]   # preprocessing.py:8-9 restated


def hashf(n, seed, lo=0.0, hi=1.0):
    return (lo + (hi - lo) * O.formula_frames(1, (n,), seed=seed)[0]).astype(np.float32)



MODEL_CASES = [  # kind, state_shape, n_actions, h_size, batch
    ("A3CModel", (4, 84, 84), 3, 256, 3),
    ("ConvModel", (4, 84, 84), 4, 256, 2),
    ("ConvModel", (4, 20, 20), 3, 32, 3),
    ("GRUModel", (4, 84, 84), 3, 256, 3),
    ("FCModel", (4, 4), 2, 200, 5),
    ("FCModel", (4, 84, 84), 3, 256, 2),
    ("GRUFCModel", (4, 4), 2, 64, 5),
]



def base_hyps(**kw):
    h = dict(gamma=.99, lambda_=.98, n_tsteps=8, n_rollouts=2, n_frame_stack=4, action_shift=0,
             render=False, env_type="FakeBreakout", use_bptt=False, use_nstep_rets=False,
             norm_advs=True, entr_coef=.005, pi_coef=1.0, val_coef=.5, max_norm=.5, lr=1e-4,
             optim_type="RMSprop", is_discrete=True, h_size=256, preprocessor=null_prep, seed=0)
    h.update(kw)
    return h


ROLLOUT_CASES = [
    # name, kind, env_type, T, n_slots, env kwargs, A
    ("a3c_pong", "A3CModel", "FakePong-v0", 12, 2, dict(env_id=1, rew_period=5, done_period=17), 3),
    ("gru_brk", "GRUModel", "FakeBreakout", 8, 3, dict(env_id=0, rew_period=3, done_period=8), 4),
    ("a3c_brk", "A3CModel", "FakeBreakout", 6, 2, dict(env_id=2, rew_period=4, done_period=9), 4),
]



def synth_shared(kind, ss, A, h, R_, T, seed, recurrent):
    """Closed-form shared_data (regenerated identically by the tests)."""
    N = R_ * T
    D = dict(states=torch.from_numpy(O.formula_frames(N, ss, seed=seed, binary=(ss[-1] == 84))),
             rewards=torch.from_numpy(np.round(hashf(N, seed + 1, -1.4, 1.4)).astype(np.float32)),
             deltas=torch.from_numpy(hashf(N, seed + 2, -1, 1)),
             actions=torch.from_numpy((hashf(N, seed + 3) * A).astype(np.int64).clip(0, A - 1)))
    d = (hashf(N, seed + 4) < 0.15).astype(np.float32)
    d[T - 1::T] = 1.0
    D["dones"] = torch.from_numpy(d)
    if recurrent:
        D["h_states"] = torch.from_numpy(hashf(N * h, seed + 5, -1, 1).reshape(N, h))
    return D


UPDATE_CASES = [
    # name, kind, state_shape, A, h, n_rollouts, T, optim, norm_advs, nstep, bptt, n_updates
    ("a3c_rms", "A3CModel", (4, 84, 84), 3, 256, 4, 8, "RMSprop", True, False, False, 2),
    ("a3c_adam", "A3CModel", (4, 84, 84), 3, 256, 4, 8, "Adam", True, False, False, 2),
    ("a3c_nonorm_nstep", "A3CModel", (4, 84, 84), 3, 256, 3, 5, "RMSprop", False, True, False, 1),
    ("conv_small_rms", "ConvModel", (4, 20, 20), 3, 32, 3, 4, "RMSprop", True, False, False, 2),
    ("conv_full_adam", "ConvModel", (4, 84, 84), 4, 256, 2, 3, "Adam", True, False, False, 1),
    ("gru_rms", "GRUModel", (4, 84, 84), 3, 256, 3, 6, "RMSprop", True, False, False, 2),
    ("gru_bptt_rms", "GRUModel", (4, 84, 84), 3, 256, 3, 6, "RMSprop", True, False, True, 2),
    ("gru_bptt_adam_nstep", "GRUModel", (4, 84, 84), 3, 256, 2, 5, "Adam", True, True, True, 1),
    ("fc_cartpole_rms", "FCModel", (4, 4), 2, 200, 4, 32, "RMSprop", True, False, False, 2),
    ("fc_full_adam", "FCModel", (4, 84, 84), 3, 256, 2, 4, "Adam", True, False, False, 1),
    ("grufc_bptt_rms", "GRUFCModel", (4, 4), 2, 64, 3, 7, "RMSprop", True, False, True, 2),
]
# evaluation rollouts (StatsRunner.rollout, runner.py:274-314)
STATS_CASES = [
    # name, kind, env_type, n_test_eps, env kwargs, A
    ("a3c_pong", "A3CModel", "FakePong-v0", 4, dict(env_id=5, rew_period=4, done_period=11), 3),
    ("gru_brk", "GRUModel", "FakeBreakout", 3, dict(env_id=6, rew_period=3, done_period=5), 4),
]
# checkpoints written by the reference's Updater.save_model after one update; resumed for a second one
CHECKPOINT_CASES = [
    # name, kind, state_shape, A, h, n_rollouts, T, optim, bptt
    ("fc_rms", "FCModel", (4, 4), 2, 24, 3, 8, "RMSprop", False),
    ("grufc_adam", "GRUFCModel", (4, 4), 2, 16, 3, 6, "Adam", True),
]
SAMPLE = 24   # elements sampled per tensor


def sample_idx(n):
    return (np.arange(SAMPLE, dtype=np.int64) * 2654435761 % max(n, 1)).astype(np.int64)




def atari_frame(seed):
    """210 x 160 x 3 uint8 frame shaped like ALE's: random pixels + patches of Pong's two background colours
    (144, 109: preprocessing.py:14-15) + black / white runs, so that every branch of pong_prep is exercised"""
    rng = np.random.default_rng(9000 + seed)
    f = rng.integers(0, 256, size=(210, 160, 3), dtype=np.uint8)
    f[35 + 2 * seed:75, 10:60, 0] = 144
    f[90:130, 40 + 4 * seed:120, 0] = 109
    f[140:150, :, 0] = 0
    f[150:160, 30:90] = 255
    return f


PREP_SEEDS = (0, 1, 2, 3)


class U8FakeEnv(O.FakeEnv):
    """O.FakeEnv with its (binary) frames as uint8, like pong_prep's output (preprocessing.py:11-17): the
    host pool then carries uint8 frames; values are identical to FakeEnv's float64 ones."""

    def _frame(self):
        return super()._frame().astype(np.uint8)


class RawAtariEnv(O.FakeEnv):
    """FakeEnv's schedules with RAW Atari-shaped observations ((210, 160, 3) uint8: random pixels and patches of Pong's
    background colours), optionally preprocessed on the HOST (prep = "pong_prep" / "breakout_prep", the reference's place
    for it: runner.py:61-69).  prep=None hands the raw frame on: the device preprocesses (a2c_frame_prep_u8)."""

    def __init__(self, prep=None, **kw):
        super().__init__(**kw)
        self.prep = prep

    def _frame(self):
        rng = np.random.default_rng(self.env_id * 100003 + self.t * 17 + self.n_resets * 7919)
        f = rng.integers(0, 256, size=(210, 160, 3), dtype=np.uint8)
        f[40 + self.t % 50:90, 10:70, 0] = 144
        f[100:150, 50 + self.env_id:130, 0] = 109
        f[150:170, :, 0] = 0
        if self.prep is None:
            return f
        return getattr(O, self.prep)(f)


class GreyFakeEnv(O.FakeEnv):
    """grey levels 0..255 as float64: what the reference's runner sees behind breakout_prep (preprocessing.py:19-23 hands
    the uint8 slice on unchanged, runner.py:199 casts it to float) -- the oracle's side of GreyU8FakeEnv"""

    def __init__(self, **kw):
        kw.pop("binary", None)
        super().__init__(binary=False, **kw)

    def _frame(self):
        return np.rint(super()._frame() * 255.0)      # formula_frames(binary=False) = b / 255 with b in 0..255: exact


class GreyU8FakeEnv(GreyFakeEnv):
    """the same frames as uint8: the host pool carries them over the uint8 transport (the packed one refuses them)"""

    def _frame(self):
        return super()._frame().astype(np.uint8)


class PongLikeEnv(U8FakeEnv):
    """80x80 binary uint8 frames = what pong_prep hands on (preprocessing.py:11-17); built as env_fn(j)"""

    def __init__(self, j=0):
        super().__init__(env_id=j, frame_shape=(1, 80, 80), rew_period=3 + j % 4, done_period=9 + j)


class F32FakeEnv(O.FakeEnv):
    """O.FakeEnv with float grey-level frames in [0, 1): the fp32 transport of the host pool.  (NOT what the reference's
    breakout_prep yields: its rgb2grey acts on an already 2-D uint8 slice and returns it unchanged -- uint8 0..255, see
    a2c_amd/preprocessing.py; this env only exercises the fp32 path.)"""

    def __init__(self, **kw):
        super().__init__(binary=False, **kw)


class FailingEnv(U8FakeEnv):
    """raises inside step() after `fail_at` steps (worker-failure propagation tests)"""

    def __init__(self, fail_at=3, **kw):
        super().__init__(**kw)
        self.fail_at = fail_at

    def step(self, action):
        if self.t >= self.fail_at:
            raise RuntimeError("env crashed (test)")
        return super().step(action)
