"""Synthetic host env of the benchmark and of the ingest tests: a
pre-generated tape of binary 84x84 uint8 frames (i.i.d. P(1) = 0.25; ``grey=True``: grey levels 0..255, what
breakout_prep hands on), rewards -1/0/+1 with
P = (.02, .96, .02) and real dones with P = 1/800, replayed in a loop; ``step`` ignores the action.
The env's own cost is deliberately nil (the benchmark measures the engine, env cost is excluded on
the GPU side and in the CPU baseline alike), but its frames are produced ON THE HOST, one env step
at a time, so they cross the host-device boundary exactly like a gym env's would.  numpy only:
the env workers import this module without torch."""
import numpy as np


class TapeEnv:
    """gym-style env (reset() -> obs, step(a) -> (obs, rew, done, info)) returning prepped (1,H,W) uint8 frames."""

    def __init__(self, env_id=0, length=129, frame_shape=(1, 84, 84), seed=1234, p_done=1.0 / 800, dtype="uint8", grey=False):
        rng = np.random.default_rng(seed + env_id)
        self.length = int(length)
        if grey:    # breakout_prep-like grey levels 0..255 (preprocessing.py:19-23: the uint8 slice travels unchanged)
            self.frames = np.floor(rng.random((self.length,) + tuple(frame_shape)) * 256.0).astype(dtype)
        else:       # pong_prep-like binary frames (preprocessing.py:11-17)
            self.frames = (rng.random((self.length,) + tuple(frame_shape)) < 0.25).astype(dtype)
        r = rng.random(self.length)
        self.rews = (r < 0.02).astype(np.float64) - (r > 0.98).astype(np.float64)
        self.dones = rng.random(self.length) < p_done
        self.t = 0

    def reset(self):
        return self.frames[self.t % self.length]

    def step(self, action):
        k = self.t % self.length
        self.t += 1
        return self.frames[self.t % self.length], float(self.rews[k]), bool(self.dones[k]), {}
