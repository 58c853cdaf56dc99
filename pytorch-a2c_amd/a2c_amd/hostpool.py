"""Parallel host env pool: env stepping on host CPUs, one region of pinned host memory between
the env workers and the MI355X (C layout and protocol: include/a2c_hostpool.h).

Reference: ``n_envs`` OS processes, each with its own gym env, each writing its state / reward /
done into shared tensors element by element (training.py:109-121, runner.py:199,208,222-226).
Here ``n_workers`` worker PROCESSES (``python -m a2c_amd.hostpool_worker``; numpy + ctypes only, they
never touch the GPU runtime) each own a contiguous block of envs.  A worker spins on the 8-byte
``cmd`` granule of each of its envs, steps the env with the action found there, writes the prepped
frame (uint8 when the preprocessor yields uint8 -- ``pong_prep`` does --, else fp32) into the env's
slot of the shared pinned region and publishes an 8-byte ``rec`` granule {step count, done, reward}.
Who writes ``cmd`` and who reads the frames depends on the ingest mode of the ``Runner``:

  zero-copy  the persistent rollout kernel (a2c_a3c_rollout) itself: it stores the sampled action
             into ``cmd`` and loads the frame straight from the pinned region over PCIe -- the host
             process is not in the loop at all during a rollout;
  memcpy     the GPU process: D2H of the actions -> ``post_actions`` -> ``wait_frames`` ->
             hipMemcpyAsync of the frames block into HBM.

This module imports neither torch nor the HIP library at import time (workers import it).
"""
import ctypes
import json
import mmap
import os
import pickle
import subprocess
import sys
import time
from ctypes import POINTER, Structure, c_double, c_float, c_int, c_int32, c_int64, c_size_t, c_uint32, c_uint64, c_void_p

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
POOL_LIB_PATH = os.path.join(_HERE, "liba2c_hostpool.so")
IDLE, ROLLOUT, SHUTDOWN = 0, 1, 2
FRAME_U8, FRAME_F32, FRAME_BITS = 0, 1, 2


class PoolHeader(Structure):
    """a2c_pool_header"""
    _fields_ = [("magic", c_uint64), ("version", c_uint32), ("n_envs", c_uint32), ("frame_bytes", c_uint32),
                ("frame_stride", c_uint32), ("frame_dtype", c_uint32), ("n_workers", c_uint32),
                ("off_cmd", c_uint64), ("off_rec", c_uint64), ("off_frames", c_uint64), ("total_bytes", c_uint64),
                ("phase", c_uint32), ("workers_ready", c_uint32), ("worker_error", c_uint32), ("ema_lock", c_uint32),
                ("rew_ema", c_double), ("episodes", c_uint64), ("frame_elems", c_uint32), ("seq_start", c_uint32),
                ("off_tagged", c_uint64), ("tagged_stride", c_uint32), ("tagged_chunks", c_uint32)]


P = c_void_p
POOL_SIGNATURES = {   # one entry per prototype in include/a2c_hostpool.h
    "a2c_pool_bytes": (c_size_t, [c_int, c_int]),
    "a2c_pool_bytes_tagged": (c_size_t, [c_int, c_int, c_uint32]),
    "a2c_pool_enable_tagged": (c_int, [P, c_size_t]),
    "a2c_pool_init": (c_int, [P, c_size_t, c_int, c_int, c_int, c_int, c_double]),
    "a2c_pool_set_frame_elems": (None, [P, c_uint32]),
    "a2c_pool_set_seq_start": (None, [P, c_uint32]),
    "a2c_pool_check": (c_int, [P]),
    "a2c_pool_set_phase": (None, [P, c_uint32]),
    "a2c_pool_phase": (c_uint32, [P]),
    "a2c_pool_poll": (c_int, [P, c_int, c_int, P, c_int64]),
    "a2c_pool_action": (c_int32, [P, c_int]),
    "a2c_pool_take": (c_int, [P, c_int, c_int, P, c_int64, P]),
    "a2c_pool_publish": (None, [P, c_int, P, c_uint32, c_float, c_int]),
    "a2c_pool_publish_bits": (c_int, [P, c_int, P, c_uint32, c_float, c_int]),
    "a2c_pool_episode": (None, [P, c_double]),
    "a2c_pool_worker_ready": (None, [P]),
    "a2c_pool_worker_failed": (None, [P, c_int]),
    "a2c_pool_post_actions": (None, [P, c_int, c_int, P, c_int64, c_uint32]),
    "a2c_pool_wait_frames": (c_int, [P, c_int, c_int, c_uint32, c_int64]),
    "a2c_pool_unpack": (None, [P, c_int, c_int, P, P]),
    "a2c_pool_rew_ema": (c_double, [P]),
    "a2c_pool_threads_start": (P, [P, c_int, P, P, c_int, c_int]),
    "a2c_pool_threads_start_push": (P, [P, c_int, P, P, c_int, c_int, P, P]),
    "a2c_pool_threads_stop": (None, [P]),
    "a2c_tape_env_create": (P, [P, P, P, c_int, c_int]),
    "a2c_tape_env_destroy": (None, [P]),
    "a2c_tape_env_vtable": (P, []),
}
_pool_lib = None


def pool_lib():
    global _pool_lib
    if _pool_lib is None:
        if not os.path.exists(POOL_LIB_PATH):
            raise ImportError(f"{POOL_LIB_PATH} not found: run make -C pytorch-a2c_amd/csrc (or __graft_entry__.build())")
        lib = ctypes.CDLL(POOL_LIB_PATH)
        for name, (res, args) in POOL_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _pool_lib = lib
    return _pool_lib


def usable_cpus():
    """CPUs this process can really burn: the affinity mask, capped by the cgroup CPU quota (cpu.max) --
    a container may show 256 CPUs and be throttled to 16; spinning workers beyond the quota get descheduled
    for whole scheduler periods, which stalls the device-side hand-shake."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def shm_path(name):
    return os.path.join("/dev/shm", name)


class Region:
    """A mapped pool region (creator or attached worker): numpy views of cmd / rec / frames."""

    def __init__(self, name, create_bytes=0):
        self.name = name
        path = shm_path(name)
        if create_bytes:
            fd = os.open(path, os.O_CREAT | os.O_EXCL | os.O_RDWR, 0o600)
            os.ftruncate(fd, create_bytes)
        else:
            fd = os.open(path, os.O_RDWR)
        try:
            self.size = os.fstat(fd).st_size
            self.mm = mmap.mmap(fd, self.size)
        finally:
            os.close(fd)
        self._buf = (ctypes.c_char * self.size).from_buffer(self.mm)
        self.base = ctypes.addressof(self._buf)
        self.created = bool(create_bytes)

    def bind(self):
        """views that need a formatted header"""
        h = self.header = PoolHeader.from_buffer(self.mm)
        arr = np.frombuffer(self.mm, dtype=np.uint8)
        n = h.n_envs
        self.cmd = arr[h.off_cmd:h.off_cmd + 8 * n].view(np.uint64)
        self.rec = arr[h.off_rec:h.off_rec + 8 * n].view(np.uint64)
        self.frames = arr[h.off_frames:h.off_frames + n * h.frame_stride].reshape(n, h.frame_stride)
        return self

    def close(self, unlink=False):
        for k in ("cmd", "rec", "frames", "header", "_buf"):
            self.__dict__.pop(k, None)
        try:
            self.mm.close()
        except BufferError:      # a numpy view is still alive somewhere: leave the mapping to process exit
            pass
        if unlink:
            try:
                os.unlink(shm_path(self.name))
            except FileNotFoundError:
                pass


def _dump_spec(spec):
    """the worker's spec, env factory included.  cloudpickle (optional dependency) carries lambdas / closures / classes
    of __main__; plain pickle only module-level callables"""
    try:
        import cloudpickle
    except ImportError:
        cloudpickle = None
    if cloudpickle is not None:
        return cloudpickle.dumps(spec)
    try:
        return pickle.dumps(spec)
    except Exception as e:      # noqa: BLE001
        raise TypeError("the env factory cannot be pickled for the env worker processes: install cloudpickle, pass a "
                        "module-level callable, or keep the envs in this process (hyps['env_pool'] = 'serial')") from e


class _PinnedPool:
    """What both pool kinds share: the pinned region and the GPU-process side of the protocol."""

    device_pool = False

    def _setup(self, n_envs, frame_shape, frame_dtype, n_workers, rew_ema0, register, frame_bits=False, seq_start=0):
        self.n_envs, self.n_workers = int(n_envs), int(n_workers)
        self.frame_shape = tuple(int(s) for s in frame_shape)
        self.frame_dtype = np.dtype(np.uint8 if np.dtype(frame_dtype) in (np.dtype(np.uint8), np.dtype(bool)) else np.float32)
        self.frame_elems = int(np.prod(self.frame_shape))
        # transport: what crosses the host link per frame.  "bits" = one bit per pixel for BINARY uint8 preprocessors
        # (pong_prep, preprocessing.py:15-16); the frames the kernels see are the same uint8 {0,1} planes
        self.frame_bits = bool(frame_bits)
        if self.frame_bits and self.frame_dtype != np.uint8:
            raise ValueError("frame_bits=True needs uint8 frames with values in {0, 1}")
        self.transport = "bits" if self.frame_bits else ("u8" if self.frame_dtype == np.uint8 else "f32")
        self.frame_bytes = (self.frame_elems + 7) // 8 if self.frame_bits else self.frame_elems * self.frame_dtype.itemsize
        self.rew_ema0, self.register = float(rew_ema0), bool(register)
        self.region = None
        self.dev_ptr = 0
        # env steps requested so far, per env (granules carry it modulo 2^32 / 2^31).  Lock-step blocks of envs share
        # one value; envs fall out of step with each other when n_rollouts is not a multiple of n_envs (runner.rollout)
        self.seq_start = int(seq_start)
        self.seq_env = np.full(self.n_envs, self.seq_start, dtype=np.int64)

    @property
    def seq(self):
        """the common step count of ALL envs (raises when they are out of step: use seq_of(env0, n))"""
        return self.seq_of(0, self.n_envs)

    def seq_of(self, env0, n):
        s = self.seq_env[env0:env0 + n]
        if not (s == s[0]).all():
            raise ValueError(f"envs {env0}..{env0 + n - 1} are not in lock-step: {sorted(set(s.tolist()))}")
        return int(s[0])

    def advance(self, env0, n, steps):
        self.seq_env[env0:env0 + n] += int(steps)

    def __len__(self):
        return self.n_envs

    def _pin_to_gpu_node(self):
        """Place this thread -- and with it the first touch of the pinned region and every worker thread / process
        it starts (they inherit the mask) -- on the NUMA node the GPU hangs off.  On the 2-socket MI355X box the
        zero-copy rollout runs at 41.8 GB/s of frame bytes from the GPU's node, 37.8 from the other one, and
        27-36 GB/s with multi-ms outliers when the scheduler is free to migrate the workers."""
        if not self.register or os.environ.get("A2C_NO_NUMA_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
            return
        from . import ops
        cpus = ops.device_numa_cpus()
        if cpus:
            allowed = set(cpus) & set(os.sched_getaffinity(0))
            if allowed:
                os.sched_setaffinity(0, allowed)

    def _create_region(self):
        lib = pool_lib()
        self._pin_to_gpu_node()
        # packed pools can keep a self-validating mirror of their frames (a2c_hostpool.h: tagged chunks): the ring
        # kernel then fetches poll + frame + record with ONE 16-byte load per lane (frames of up to 63 x 112 pixels)
        # OPT-IN (A2C_TAGGED=1): measured on MI355X the single 1 KB fetch is SLOWER than the 8-byte poll followed by the
        # frame load (2.38-2.44 vs 2.24-2.28 ms per 256 x 128 slot: its PCIe latency is longer and a miss costs a whole
        # second round trip) -- DESIGN.md section 7
        tagged = self.frame_bits and self.frame_elems <= 63 * 112 and os.environ.get("A2C_TAGGED") == "1"
        nbytes = lib.a2c_pool_bytes_tagged(self.n_envs, self.frame_bytes, self.frame_elems) if tagged else \
            lib.a2c_pool_bytes(self.n_envs, self.frame_bytes)
        self.name = f"a2c_pool_{os.getpid()}_{id(self) & 0xffffff:x}_{int(time.time() * 1e3) & 0xffffff:x}"
        reg = self.region = Region(self.name, create_bytes=nbytes)
        ctypes.memset(reg.base, 0, nbytes)          # first touch of every page from the (NUMA-placed) creating thread
        dt = FRAME_BITS if self.frame_bits else (FRAME_U8 if self.frame_dtype == np.uint8 else FRAME_F32)
        if lib.a2c_pool_init(reg.base, nbytes, self.n_envs, self.frame_bytes, dt, self.n_workers, self.rew_ema0):
            raise RuntimeError("a2c_pool_init failed")
        lib.a2c_pool_set_frame_elems(reg.base, self.frame_elems)
        lib.a2c_pool_set_seq_start(reg.base, self.seq_start & 0xffffffff)
        if tagged and lib.a2c_pool_enable_tagged(reg.base, nbytes):
            raise RuntimeError("a2c_pool_enable_tagged failed")
        reg.bind()
        if self.register:       # pin + map into the GPU's address space (no copy): hipHostRegister
            from . import ops
            self.dev_ptr = ops.pinned_register(reg.base, nbytes)
        return reg

    def _destroy_region(self):
        reg = self.region
        if self.dev_ptr:
            from . import ops
            ops.pinned_unregister(reg.base)
            self.dev_ptr = 0
        reg.close(unlink=True)
        self.region = None

    def _check_workers(self):
        h = self.region.header
        if h.worker_error:
            raise RuntimeError(f"env worker {h.worker_error - 1} died with an exception (see its stderr)")

    def __del__(self):
        try:
            self.close()
        except Exception:      # noqa: BLE001
            pass

    # ------------------------------------------------------------------ GPU-process side of the protocol
    @property
    def header(self):
        return self.region.header

    def set_phase(self, phase):
        pool_lib().a2c_pool_set_phase(self.region.base, phase)

    def post_actions(self, actions, env0=0, seq=None):
        """cmd[env0+i] = (seq, actions[i]) -- actions: contiguous int64 numpy array (e.g. a pinned staging view)"""
        a = np.ascontiguousarray(actions, dtype=np.int64)
        if seq is None:
            seq = self.seq_of(env0, a.shape[0])
        pool_lib().a2c_pool_post_actions(self.region.base, env0, a.shape[0], a.ctypes.data, 1, int(seq) & 0xffffffff)

    def wait_frames(self, seq, env0=0, n=None, timeout=30.0):
        rc = pool_lib().a2c_pool_wait_frames(self.region.base, env0, self.n_envs - env0 if n is None else n,
                                             int(seq) & 0xffffffff, int(timeout * 1e9))
        if rc == -3:
            self._check_workers()
            raise RuntimeError("an env worker failed")
        if rc:
            self._check_workers()
            raise TimeoutError(f"env workers did not deliver frame {seq} within {timeout}s")

    def unpack(self, rew, done, env0=0):
        """rec -> float32 numpy arrays rew[i], done[i] (real done = reset flag)"""
        pool_lib().a2c_pool_unpack(self.region.base, env0, rew.shape[0], rew.ctypes.data, done.ctypes.data)

    def rew_ema(self):
        return pool_lib().a2c_pool_rew_ema(self.region.base)

    def frames_view(self):
        """(n_envs, *frame_shape) numpy view of the current frames (no copy; the packed transport is unpacked into a copy)"""
        f = self.region.frames[:, :self.frame_bytes]
        if self.frame_bits:
            return np.unpackbits(f, axis=1, bitorder="little")[:, :self.frame_elems].reshape((self.n_envs,) + self.frame_shape)
        return f.view(self.frame_dtype).reshape((self.n_envs,) + self.frame_shape)

    # device addresses inside the registered region (zero-copy ingest)
    @property
    def dev_cmd(self):
        return self.dev_ptr + self.header.off_cmd

    @property
    def dev_phase(self):
        return self.dev_ptr + PoolHeader.phase.offset if self.dev_ptr else 0

    @property
    def dev_rec(self):
        """where device code polls the rec granules: the push mirror in HBM when the pool keeps one, else the pinned region"""
        if getattr(self, "push_ptr", 0):
            return self.push_ptr
        return self.dev_ptr + self.header.off_rec

    @property
    def dev_frames(self):
        if getattr(self, "push_ptr", 0):
            return self.push_ptr + self._push_rec_bytes
        return self.dev_ptr + self.header.off_frames

    @property
    def dev_tagged(self):
        """device address of the self-validating mirror of the packed frames (0: the pool keeps none)"""
        return self.dev_ptr + self.header.off_tagged if (self.dev_ptr and self.header.off_tagged) else 0

    def describe(self):
        h = self.header
        return json.dumps(dict(n_envs=h.n_envs, n_workers=h.n_workers, frame_bytes=h.frame_bytes,
                               frame_dtype={FRAME_U8: "u8", FRAME_F32: "f32", FRAME_BITS: "bits"}[h.frame_dtype]))


class ProcessEnvPool(_PinnedPool):
    """``n_envs`` host envs stepped by ``n_workers`` processes behind one pinned region.

    env_factory(**env_kwargs[j]) -> object with ``reset() -> obs`` and ``step(a) -> (obs, rew, done, info)``
    returning already prepped frames of shape (1, H, W) / (1, L) (what SequentialEnvironment returns).
    ``action_shift`` / ``pong`` are the hyps of runner.py:208,212-214 (the Pong done override only affects
    the episode-reward EMA here; the device applies it to the ``dones`` buffer).
    """

    def __init__(self, env_factory, n_envs, env_kwargs=None, n_workers=None, action_shift=0, pong=False,
                 frame_shape=None, frame_dtype=None, rew_ema0=-1.0, register=True, spin=True, sys_path=None,
                 probe_reset=False, frame_bits=False, seq_start=0):
        self.env_factory, self.n_envs = env_factory, int(n_envs)
        self.env_kwargs = list(env_kwargs) if env_kwargs is not None else [dict() for _ in range(n_envs)]
        assert len(self.env_kwargs) == self.n_envs
        if n_workers is None:
            # all usable CPUs but two (the GPU process and the OS); every worker spins during a rollout
            n_workers = max(1, min(self.n_envs, int(os.environ.get("A2C_ENV_WORKERS", "0")) or min(48, max(1, usable_cpus() - 2))))
        self.n_workers = int(n_workers)
        self.action_shift, self.pong = int(action_shift), bool(pong)
        # raw env objects: one extra reset() per env right after construction, which is what the reference's
        # SequentialEnvironment.__init__ does to read raw_shape (runner.py:45); wrappers that already do it pass False
        self.probe_reset = bool(probe_reset)
        if frame_shape is None or frame_dtype is None:
            # one probing env, like training.py:60-65 reads the observation shape from StatsRunner's env
            obs = np.asarray(env_factory(**self.env_kwargs[0]).reset())
            frame_shape = obs.shape if frame_shape is None else frame_shape
            frame_dtype = obs.dtype if frame_dtype is None else frame_dtype
        self._setup(n_envs, frame_shape, frame_dtype, n_workers, rew_ema0, register, frame_bits=frame_bits, seq_start=seq_start)
        self.spin = bool(spin)
        self.sys_path = list(sys_path) if sys_path is not None else [p for p in sys.path if p]
        self.procs = []

    # ------------------------------------------------------------------ life cycle
    def start(self, timeout=300.0):
        if self.region is not None:
            return self
        reg = self._create_region()
        per = -(-self.n_envs // self.n_workers)
        blocks = [(w, w * per, min(per, self.n_envs - w * per)) for w in range(self.n_workers) if w * per < self.n_envs]
        self.n_workers = len(blocks)
        reg.header.n_workers = self.n_workers
        env = dict(os.environ, PYTHONPATH=os.pathsep.join(self.sys_path + [os.environ.get("PYTHONPATH", "")]))
        for w, e0, n in blocks:
            spec = dict(shm=self.name, worker=w, env0=e0, n=n, factory=self.env_factory,
                        env_kwargs=self.env_kwargs[e0:e0 + n], action_shift=self.action_shift, pong=self.pong,
                        frame_shape=self.frame_shape, frame_dtype=self.frame_dtype.str, parent=os.getpid(),
                        probe_reset=self.probe_reset)
            p = subprocess.Popen([sys.executable, "-m", "a2c_amd.hostpool_worker"], stdin=subprocess.PIPE, env=env)
            p.stdin.write(_dump_spec(spec))
            p.stdin.close()
            self.procs.append(p)
        t0 = time.time()
        while reg.header.workers_ready < self.n_workers:
            self._check_workers()
            if time.time() - t0 > timeout:
                self.close()
                raise TimeoutError("env workers did not come up")
            time.sleep(0.005)
        if not self.spin:
            self.set_phase(IDLE)
        return self

    def _check_workers(self):
        super()._check_workers()
        for p in self.procs:
            if p.poll() is not None:
                raise RuntimeError(f"env worker pid {p.pid} exited with code {p.returncode}")

    def close(self):
        reg = self.region
        if reg is None:
            return
        pool_lib().a2c_pool_set_phase(reg.base, SHUTDOWN)
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()            # reap it: no zombie outlives the pool
        self.procs = []
        self._destroy_region()


class ThreadEnvPool(_PinnedPool):
    """Envs implemented natively (a2c_env_vtable, include/a2c_hostpool.h) stepped by ``n_threads`` pthreads
    of THIS process -- no Python in the loop; the env writes its observation straight into its pinned frame
    slot.  ``from_tape_envs`` wraps the synthetic benchmark env: the tapes of ``synthetic.TapeEnv`` objects
    are copied into native tape envs, so it produces byte-identical data to stepping those objects in
    worker processes.  Same device-facing protocol and API as ProcessEnvPool."""

    def __init__(self, vtable_ptr, env_ptrs, frame_shape, frame_dtype, n_threads=None, action_shift=0, pong=False,
                 rew_ema0=-1.0, register=True, destroy=None, frame_bits=False, seq_start=0):
        if n_threads is None:
            n_threads = max(1, min(len(env_ptrs), int(os.environ.get("A2C_ENV_THREADS", "0")) or min(8, max(1, usable_cpus() - 2))))
        self._setup(len(env_ptrs), frame_shape, frame_dtype, n_threads, rew_ema0, register, frame_bits=frame_bits,
                    seq_start=seq_start)
        self.vtable_ptr, self.env_ptrs, self._destroy_env = vtable_ptr, list(env_ptrs), destroy
        self.action_shift, self.pong = int(action_shift), bool(pong)
        self.handle = None

    @classmethod
    def from_tape_envs(cls, envs, **kw):
        lib = pool_lib()
        ptrs = []
        for e in envs:
            fr = np.ascontiguousarray(e.frames)
            rw = np.ascontiguousarray(e.rews, dtype=np.float64)
            dn = np.ascontiguousarray(e.dones, dtype=np.uint8)
            ptr = lib.a2c_tape_env_create(fr.ctypes.data, rw.ctypes.data, dn.ctypes.data, e.length, fr[0].nbytes)
            if not ptr:
                raise RuntimeError("a2c_tape_env_create failed")
            ptrs.append(ptr)
        e0 = envs[0]
        return cls(lib.a2c_tape_env_vtable(), ptrs, e0.frames.shape[1:], e0.frames.dtype, destroy=lib.a2c_tape_env_destroy, **kw)

    def start(self, timeout=None):
        if self.region is not None:
            return self
        reg = self._create_region()
        self._env_array = (c_void_p * self.n_envs)(*self.env_ptrs)
        # Push mirror (uint8 / packed frames, a registered pool = a GPU process): the worker threads ALSO write every answer
        # straight into fine-grained device memory (a2c_push_buffer_alloc; large BAR, checked with a guarded probe write), so
        # the rollout kernels poll rec and fetch the frame from HBM instead of over PCIe (two dependent round trips per env
        # step).  Default since round 5 (A2C_PUSH=0 switches it off): while the ring kernel's step was bound by its fp32
        # matrix work the shorter turn-around (7.1 -> 5.7 us per env step) bought nothing; with conv1 on the bf16 pipe the
        # step waits on the answer, and the same change is worth 0.3-0.5 ms per 256 x 128 slot.
        # A platform without a host-writable device buffer keeps the pinned host region as the only copy.
        self.push_ptr = 0
        if self.register and self.frame_dtype == np.uint8 and os.environ.get("A2C_PUSH", "1") != "0":
            from . import ops
            h = reg.header
            self._push_rec_bytes = (8 * self.n_envs + 255) // 256 * 256
            self.push_ptr = ops.push_buffer_alloc(self._push_rec_bytes + self.n_envs * int(h.frame_stride))
        if self.push_ptr:
            self.handle = pool_lib().a2c_pool_threads_start_push(reg.base, self.n_workers, self.vtable_ptr, self._env_array,
                                                                 self.action_shift, int(self.pong), self.push_ptr,
                                                                 self.push_ptr + self._push_rec_bytes)
        else:
            self.handle = pool_lib().a2c_pool_threads_start(reg.base, self.n_workers, self.vtable_ptr, self._env_array,
                                                            self.action_shift, int(self.pong))
        if not self.handle:
            self._free_push()
            self._destroy_region()
            raise RuntimeError("a2c_pool_threads_start failed")
        self.n_workers = int(reg.header.n_workers)
        return self

    def _free_push(self):
        if getattr(self, "push_ptr", 0):
            from . import ops
            ops.push_buffer_free(self.push_ptr)
            self.push_ptr = 0

    def close(self):
        if self.region is None:
            return
        if self.handle:
            pool_lib().a2c_pool_threads_stop(self.handle)
            self.handle = None
        self._free_push()
        self._destroy_region()
        if self._destroy_env is not None:
            for p in self.env_ptrs:
                self._destroy_env(p)
        self.env_ptrs = []
