"""Tensor-level wrappers of the C ABI (include/a2c_mi355x.h).

Every function takes torch tensors that already live in HBM, checks device / dtype /
layout, and enqueues the HIP kernels on torch's current stream.  Nothing here computes
on the CPU: a non-CUDA tensor raises.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ConvDesc, check

_F32 = torch.float32


def lib():
    """the C-ABI library (ctypes) -- or, with A2C_TORCH_OPS=1, the same entry points reached through
    torch.ops.a2c_mi355x.abi_* (TorchAbi below): the launches of the hot loop then go through the PyTorch dispatcher"""
    if _TORCH_ABI_ON():
        return torch_abi()
    return _lib.load()


def _TORCH_ABI_ON():
    import os
    return os.environ.get("A2C_TORCH_OPS") == "1"


class TorchAbi:
    """``lib()`` when A2C_TORCH_OPS=1: attribute access returns, for every kernel-launching entry point of
    include/a2c_mi355x.h that has a generated op (csrc/torch_ops_abi.inc; tools/gen_torch_abi_ops.py), a callable with the
    ctypes binding's positional signature that forwards the call to ``torch.ops.a2c_mi355x.abi_<name>(bufs, offs, ints,
    floats)``: every pointer argument is replaced by (the torch tensor that owns that memory, byte offset into it) -- the
    tensors this module has seen (workspaces, arenas, rollout buffers: ``note_tensor``) are looked up by address --, the
    stream argument must be torch's current stream (it is what the op uses).  Same C function, same arguments: results are
    bit-identical to the ctypes path (tests/test_gpu_system.py).  What cannot be expressed -- argument blocks
    (a2c_a3c_step / a2c_a3c_rollout), pointer tables, addresses outside any torch tensor (the pinned pool region), a foreign
    stream -- goes to the ctypes library and is counted in ``stats['ctypes']``."""

    def __init__(self):
        import importlib.util
        import os
        self._c = _lib.load()
        self._ops = load_torch_ops()
        root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        # the parameter classes come from the same header parse that generated the C++ side
        spec = importlib.util.spec_from_file_location("_a2c_gen_abi", os.path.join(root, "tools", "gen_torch_abi_ops.py"))
        gen = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gen)
        self._layout = {n: list(zip(kinds, [t for t, _ in params])) for n, params, kinds in gen.plan()[0]}
        self._bases, self._refs = [], {}          # sorted storage base addresses -> weakref(tensor), nbytes
        self._misses = set()                      # addresses no live tensor contains (looked for once)
        self._null = {}
        self.stats = {"torch_ops": 0, "ctypes": 0, "by_name": {}, "unresolved": {}}
        self._fns = {}

    # -- address book
    def note(self, t):
        import bisect
        import weakref
        if t is None or not t.is_cuda:
            return
        st = t.untyped_storage()
        base, nb = st.data_ptr(), st.nbytes()
        if not nb:
            return
        ent = self._refs.get(base)
        if ent is not None and ent[1] >= nb and ent[0]() is not None:
            return
        if ent is None:
            bisect.insort(self._bases, base)
        flat = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st, 0, (nb,))      # a byte view of the whole storage
        self._keep = getattr(self, "_keep", {})
        self._keep[base] = flat             # (the view keeps the storage alive: opt-in debugging / boundary mode only)
        self._refs[base] = (weakref.ref(flat), nb)

    def _lookup(self, addr):
        import bisect
        i = bisect.bisect_right(self._bases, addr) - 1
        if i < 0:
            return None
        base = self._bases[i]
        ref, nb = self._refs[base]
        t = ref()
        if t is None or addr >= base + nb:
            return None
        return t, addr - base

    def _find(self, addr):
        hit = self._lookup(addr)
        if hit is None and addr in self._misses:
            return None                     # (an address of the pinned pool region: asked for at every relay step, never a tensor)
        if hit is None:
            # an address this module was never handed as a tensor (a buffer a model allocated itself and passes as
            # data_ptr() + offset): find the live HIP tensor whose storage contains it -- once per storage, then it is noted
            import gc
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")       # (isinstance on lazily deprecated module attributes warns)
                for obj in gc.get_objects():
                    try:
                        if isinstance(obj, torch.Tensor) and obj.is_cuda:
                            st = obj.untyped_storage()
                            b = st.data_ptr()
                            if b <= addr < b + st.nbytes():
                                self.note(obj)
                                break
                    except Exception:      # noqa: BLE001  (tensors without storage, objects that refuse isinstance)
                        continue
            hit = self._lookup(addr)
            if hit is None and len(self._misses) < 4096:
                self._misses.add(addr)
        return hit

    def _nul(self, dev):
        t = self._null.get(dev)
        if t is None:
            t = self._null[dev] = torch.empty(0, device=dev)
        return t

    def __getattr__(self, name):
        if name.startswith("_") or name in ("stats", "note"):
            raise AttributeError(name)
        fn = self._fns.get(name)
        if fn is None:
            fn = self._fns[name] = self._make(name)
        return fn

    def _make(self, name):
        cfn = getattr(self._c, name)
        lay = self._layout.get(name)
        if lay is None:
            return cfn                      # queries (_supported, _ws_bytes ...), argument blocks, host-side entry points
        op = getattr(self._ops, "abi_" + name[4:])
        stats = self.stats

        def call(*args):
            st = args[-1]
            st = int(st.value or 0) if hasattr(st, "value") else int(st or 0)
            bufs, offs, ints, flts = [], [], [], []
            ok = st == stream()
            dev = torch.device("cuda", torch.cuda.current_device())
            if ok:
                for (kind, _), a in zip(lay, args[:-1]):
                    if kind == "ptr":
                        addr = int(getattr(a, "value", a) or 0)
                        if addr == 0:
                            bufs.append(self._nul(dev)); offs.append(0)
                            continue
                        hit = self._find(addr)
                        if hit is None:
                            stats["unresolved"][name] = stats["unresolved"].get(name, 0) + 1
                            ok = False
                            break
                        bufs.append(hit[0]); offs.append(hit[1])
                    elif kind == "int":
                        ints.append(int(a))
                    elif kind == "float":
                        flts.append(float(a))
                    else:                   # a2c_conv_desc*: ctypes.byref(desc)
                        d = a._obj
                        ints.extend(int(getattr(d, f)) for f in ("Cin", "H", "W", "Cout", "ks", "stride", "pad", "OH", "OW"))
            if not ok:
                stats["ctypes"] += 1
                return cfn(*args)
            stats["torch_ops"] += 1
            stats["by_name"][name] = stats["by_name"].get(name, 0) + 1
            try:
                op(bufs, offs, ints, flts)
            except RuntimeError as e:       # the op raises where the launcher returned an error code
                msg = str(e)
                for code in (-1, -2, -3):
                    if self._c.a2c_error_string(code).decode() in msg:
                        return code
                raise
            return 0
        return call


_torch_abi = None


def torch_abi():
    global _torch_abi
    if _torch_abi is None:
        _torch_abi = TorchAbi()
    return _torch_abi


def note_tensor(t):
    """A2C_TORCH_OPS=1: remember which tensor owns this device memory (pointer arguments are looked up by address)"""
    if _TORCH_ABI_ON():
        torch_abi().note(t)
    return t


def load_torch_ops():
    """Registers ``torch.ops.a2c_mi355x.*`` (TORCH_LIBRARY shim over the C ABI, csrc/torch_ops.cpp): discount,
    gae_returns, softmax_sample, frame_stack_push, loss_fwd_bwd, linear, clip_rmsprop_.  HIP dispatch key only."""
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liba2c_torch_ops.so")
    if not os.path.exists(path):
        raise ImportError(f"{path} not found: run make -C pytorch-a2c_amd/csrc")
    _lib.load()
    torch.ops.load_library(path)
    return torch.ops.a2c_mi355x


def stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    if t is None:
        return 0
    if _TORCH_ABI_ON():
        torch_abi().note(t)
    return t.data_ptr()


def _chk(t, name, dtype=_F32, contig=True):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"a2c_amd: `{name}` must be a CUDA (HIP) tensor; the MI355X kernels have no CPU fallback")
    if t.dtype != dtype:
        raise TypeError(f"a2c_amd: `{name}` must be {dtype}, got {t.dtype}")
    if contig and not t.is_contiguous():
        raise ValueError(f"a2c_amd: `{name}` must be contiguous")


class KernelTimers:
    """Optional HIP-event timing of named launch sites, recorded on the stream the kernels are
    launched on (bench.py's roofline figures).  Disabled (None) unless a caller installs one."""

    def __init__(self):
        self.events = {}

    def span(self, name):
        return _Span(self, name)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for name, pairs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in pairs]
            out[name] = dict(launches=len(ms), avg_ms=sum(ms) / max(len(ms), 1), total_ms=sum(ms))
        return out


class _Span:
    def __init__(self, timers, name):
        self.t, self.name = timers, name

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()
        return self

    def __exit__(self, *exc):
        self.b.record()
        self.t.events.setdefault(self.name, []).append((self.a, self.b))
        return False


class _NoSpan:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


TIMERS = None
_NOSPAN = _NoSpan()


def span(name):
    return TIMERS.span(name) if TIMERS is not None else _NOSPAN


# ---------------------------------------------------------------- side branches (launch-latency-bound chains next to big kernels)
import threading as _threading

_SIDE_STREAMS = {}
_TLS = _threading.local()       # open branches per THREAD: a Runner thread's _prep must not join (and clear) the updater's branch


def _open_branches():
    b = getattr(_TLS, "branches", None)
    if b is None:
        b = _TLS.branches = []
    return b


class side_branch:
    """``with ops.side_branch(i):`` -- what is enqueued inside runs on side stream i, forked from the current stream at entry
    (it sees everything enqueued before); ``ops.join_branches()`` makes the current stream wait for every open branch.
    For chains of small dependent kernels that do not feed the big kernel behind them (the head / projection gradients of
    A3CModel next to its conv backward; the re-derivation of the inference weights): as nodes of ONE stream each costs its
    ~4 us launch latency on the critical path of the epoch, as a parallel branch of the (captured) graph it costs nothing.
    Same kernels, same arguments: results are bit-identical to the single-stream order.  The side streams are created on
    the first EAGER use (every captured update is preceded by an eager one).  A2C_NO_BRANCH=1: plain single-stream order."""

    def __init__(self, idx=0):
        self.idx, self.ctx = idx, None

    def __enter__(self):
        import os
        if os.environ.get("A2C_NO_BRANCH") == "1" or not torch.cuda.is_available():
            return self
        main = torch.cuda.current_stream()
        key = (main.device.index, self.idx)
        s = _SIDE_STREAMS.get(key)
        if s is None:
            if torch.cuda.is_current_stream_capturing():
                return self                       # (no stream creation inside a capture: stay on the main stream)
            s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=main.device)
        s.wait_stream(main)
        self.ctx = torch.cuda.stream(s)
        self.ctx.__enter__()
        _open_branches().append(s)
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.ctx = None
        return False


def join_branches():
    """the current stream waits for every branch THIS THREAD opened since its last join (call it in a ``finally``: a branch
    left open by an exception would otherwise stay an unjoined stream of a capture)"""
    br = _open_branches()
    if br:
        main = torch.cuda.current_stream()
        try:
            for s in br:
                if s != main:
                    main.wait_stream(s)
        finally:
            br.clear()


class graph_capture:
    """``torch.cuda.graph(g, capture_error_mode="thread_local")`` with the garbage collector held off: a cyclic-GC pass in
    the middle of a capture may run the destructor of an OLD hipGraph / pinned pool (hipGraphDestroy, hipFree,
    hipHostUnregister -- none of them legal while a stream is capturing) and aborts the process.  Pending garbage is
    collected BEFORE the capture starts."""

    def __init__(self, g):
        self.g = g

    def __enter__(self):
        import gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()
        self._ctx = torch.cuda.graph(self.g, capture_error_mode="thread_local")
        try:
            return self._ctx.__enter__()
        except BaseException:
            if self._was:
                gc.enable()
            raise

    def __exit__(self, *exc):
        import gc
        try:
            return self._ctx.__exit__(*exc)
        finally:
            if self._was:
                gc.enable()


class Workspace:
    """Named, persistent device buffers (no allocation inside steady-state steps, so the
    launch sequences are hipGraph-capturable and pointers are stable across updates)."""

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def get(self, name, shape, dtype=_F32, zero=False):
        shape = tuple(int(s) for s in shape)
        key = (name, dtype)
        n = 1
        for s in shape:
            n *= s
        buf = self._bufs.get(key)
        if buf is None or buf.numel() < n:
            buf = torch.empty(max(n, 1), dtype=dtype, device=self.device)
            self._bufs[key] = buf
            if zero:
                buf.zero_()
        note_tensor(buf)
        return buf[:n].view(shape)

    def bytes(self, name, nbytes):
        n = (int(nbytes) + 3) // 4
        return self.get(name, (max(n, 1),))


# ---------------------------------------------------------------- scans / normalisation
def discount_rows(x, dones, factor, n_seg, T, out=None, err=None, st=None):
    _chk(x, "x"); _chk(dones, "dones")
    if out is None:
        out = torch.empty_like(x)
    check(lib().a2c_discount_scan(_p(x), _p(dones), _p(out), n_seg, T, float(factor), _p(err),
                                  st if st is not None else stream()), "a2c_discount_scan")
    return out


def gae_returns(deltas, rewards, dones, g_adv, g_ret, n_seg, T, advs, rets, err=None, st=None):
    for n, t in (("deltas", deltas), ("rewards", rewards), ("dones", dones), ("advs", advs), ("rets", rets)):
        _chk(t, n)
    check(lib().a2c_gae_returns_fused(_p(deltas), _p(rewards), _p(dones), _p(advs), _p(rets), n_seg, T,
                                      float(g_adv), float(g_ret), _p(err), st if st is not None else stream()),
          "a2c_gae_returns_fused")


REDUCE_SCRATCH_DOUBLES = 8 + 3 * 1024        # A2C_REDUCE_SCRATCH_DOUBLES of include/a2c_mi355x.h
_reduce_scratch = {}


def new_reduce_scratch(device):
    """zero-initialised scratch (ticket word + per-workgroup partials) of the deterministic scalar reductions (a2c_moments,
    a2c_loss_fwd_bwd, a2c_gradnorm_sq).  ONE buffer per OWNER (an Updater, an optimiser): the kernels an owner enqueues run
    one after the other on its stream and each leaves the ticket word zero, so the address is stable across hipGraph
    captures -- and two owners (two captured updates replayed on different streams) never share a ticket.  Allocated
    outside any capture: a buffer born inside one would live in that graph's private pool."""
    if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
        raise RuntimeError("reduce scratch requested during a hipGraph capture: run the update eagerly once first")
    return torch.zeros(REDUCE_SCRATCH_DOUBLES, dtype=torch.float64, device=device)


def reduce_scratch(device, st):
    """fallback for callers without an owner (the torch.ops side door, kernel tests): one buffer per (device, stream)"""
    key = (str(device), int(st))
    buf = _reduce_scratch.get(key)
    if buf is None:
        buf = _reduce_scratch[key] = new_reduce_scratch(device)
    return buf


def moments(x, sums, st=None, scratch=None):
    _chk(x, "x"); _chk(sums, "sums", torch.float64)
    st = st if st is not None else stream()
    scratch = scratch if scratch is not None else reduce_scratch(x.device, st)
    check(lib().a2c_moments(_p(x), x.numel(), _p(sums), _p(scratch), st), "a2c_moments")


def normalize(x, y, sums, n_global, eps=1e-6, st=None):
    _chk(x, "x"); _chk(y, "y"); _chk(sums, "sums", torch.float64)
    check(lib().a2c_normalize(_p(x), _p(y), x.numel(), _p(sums), n_global, eps, st if st is not None else stream()),
          "a2c_normalize")


def add(a, b, y, st=None):
    _chk(a, "a"); _chk(b, "b"); _chk(y, "y")
    check(lib().a2c_add(_p(a), _p(b), _p(y), a.numel(), st if st is not None else stream()), "a2c_add")


# ---------------------------------------------------------------- rollout step kernels
def frame_stack_push(frame_new, reset_mask, prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, st=None):
    check(lib().a2c_frame_stack_push(_p(frame_new), _p(reset_mask), prev_ptr, prev_stride, out_ptr, out_stride,
                                     B, C, HW, st if st is not None else stream()), "a2c_frame_stack_push")


def pool_publish_actions(cmd_ptr, actions_ptr, act_stride, n, seq_base, seq_off, st=None):
    """device side of a2c_pool_post_actions: cmd granules of n envs written from the stream (include/a2c_mi355x.h g1)"""
    check(lib().a2c_pool_publish_actions(cmd_ptr, actions_ptr, act_stride, n, _p(seq_base), seq_off,
                                         st if st is not None else stream()), "a2c_pool_publish_actions")


def pool_ingest(rec_ptr, frames_ptr, frame_stride, frame_bytes, n, seq_base, seq_off, timeout_ticks, err, rew, done,
                frames_out_ptr, out_stride, st=None):
    """device side of a2c_pool_wait_frames + unpack + the H2D copy of the frames block"""
    check(lib().a2c_pool_ingest(rec_ptr, frames_ptr, frame_stride, frame_bytes, n, _p(seq_base), seq_off, timeout_ticks,
                                _p(err), _p(rew), _p(done), frames_out_ptr, out_stride,
                                st if st is not None else stream()), "a2c_pool_ingest")


def pool_ingest_bits(rec_ptr, frames_ptr, frame_stride, n_pixels, n, seq_base, seq_off, timeout_ticks, err, rew, done,
                     frames_out_ptr, out_stride, st=None):
    """pool_ingest for the packed transport (one bit per pixel over the host link; uint8 pixels land in HBM)"""
    check(lib().a2c_pool_ingest_bits(rec_ptr, frames_ptr, frame_stride, n_pixels, n, _p(seq_base), seq_off, timeout_ticks,
                                     _p(err), _p(rew), _p(done), frames_out_ptr, out_stride,
                                     st if st is not None else stream()), "a2c_pool_ingest_bits")


def pool_ingest_post(packed_bits, rec_ptr, frames_ptr, frame_stride, frame_elems, n, seq_base, seq_off, timeout_ticks, err, rew,
                     done, frames_out_ptr, out_stride, val_ptr, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma,
                     pong, done_eff, h, h_rows_ptr, h_rows_stride, h_src_ptr, nvalid_rows, nvalid_carry_ptr, st=None):
    """pool_ingest[_bits] + rollout_post_frames of the same env step in one launch (the workgroup that fetched an env's answer
    does that env's bookkeeping and hidden row)"""
    check(lib().a2c_pool_ingest_post(int(bool(packed_bits)), rec_ptr, frames_ptr, frame_stride, frame_elems, n, _p(seq_base),
                                     seq_off, timeout_ticks, _p(err), _p(rew), _p(done), frames_out_ptr, out_stride, val_ptr,
                                     val_stride, _p(val_prev), _p(rewards), _p(dones), _p(deltas), T, t, slot0, float(gamma),
                                     int(bool(pong)), _p(done_eff), _p(h), 0 if h is None else h.shape[1], h_rows_ptr,
                                     h_rows_stride, h_src_ptr, _p(nvalid_rows), nvalid_carry_ptr,
                                     st if st is not None else stream()), "a2c_pool_ingest_post")


def store_u32_system(dev_ptr, value, st=None):
    """one system-scope 4-byte store from the stream (the host pool's phase word)"""
    check(lib().a2c_store_u32_system(dev_ptr, value, st if st is not None else stream()), "a2c_store_u32_system")


def unpack_bits(src_ptr, src_stride, dst_ptr, dst_stride, n, n_pixels, st=None):
    """packed frames staged in HBM -> uint8 frames (memcpy ingest of a frame_bits pool)"""
    check(lib().a2c_unpack_bits(src_ptr, src_stride, dst_ptr, dst_stride, n, n_pixels,
                                st if st is not None else stream()), "a2c_unpack_bits")


def frame_stack_push_u8(frame_u8_ptr, frame_stride, reset_mask, prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, st=None):
    """frame_stack_push with the new frame as uint8 pixels at frame_u8_ptr + b*frame_stride (device address)"""
    check(lib().a2c_frame_stack_push_u8(frame_u8_ptr, frame_stride, _p(reset_mask), prev_ptr, prev_stride, out_ptr,
                                        out_stride, B, C, HW, st if st is not None else stream()), "a2c_frame_stack_push_u8")


def softmax_sample(logits, u, actions_ptr, act_stride, B, A, probs=None, st=None):
    _chk(logits, "logits", contig=False); _chk(u, "u")
    check(lib().a2c_softmax_sample(_p(logits), logits.stride(0), _p(u), actions_ptr, act_stride, _p(probs), B, A,
                                   st if st is not None else stream()), "a2c_softmax_sample")


def sample_probs(probs, u, out, st=None):
    _chk(probs, "probs"); _chk(u, "u"); _chk(out, "out")
    A = probs.shape[-1]
    check(lib().a2c_sample_probs(_p(probs), _p(u), _p(out), probs.numel() // A, A,
                                 st if st is not None else stream()), "a2c_sample_probs")


def rollout_record(rew, done, val_ptr, val_stride, val_prev, rewards, dones, deltas, done_eff, h, B, T, t, slot0,
                   gamma, pong, st=None):
    check(lib().a2c_rollout_record(_p(rew), _p(done), val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones), _p(deltas),
                                   _p(done_eff), _p(h), 0 if h is None else h.shape[-1], B, T, t, slot0,
                                   float(gamma), int(bool(pong)), st if st is not None else stream()),
          "a2c_rollout_record")


def rollout_post(rew, done, val_ptr, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, frame_new,
                 reset_mask, prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, st=None):
    """record + frame_stack_push of one env step in one launch (feed-forward nets)"""
    check(lib().a2c_rollout_post(_p(rew), _p(done), val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones),
                                 _p(deltas), T, t, slot0, float(gamma), int(bool(pong)), _p(frame_new), _p(reset_mask),
                                 prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW,
                                 st if st is not None else stream()), "a2c_rollout_post")


def rollout_post_u8(rew, done, val_ptr, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, frame_u8_ptr,
                    frame_stride, reset_mask, prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, st=None):
    check(lib().a2c_rollout_post_u8(_p(rew), _p(done), val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones),
                                    _p(deltas), T, t, slot0, float(gamma), int(bool(pong)), frame_u8_ptr, frame_stride,
                                    _p(reset_mask), prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW,
                                    st if st is not None else stream()), "a2c_rollout_post_u8")


def rollout_post_rec(rew, done, val_ptr, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, frame32_ptr,
                     frame_u8_ptr, frame_stride, reset_mask, prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, done_eff, h,
                     h_rows_ptr, h_rows_stride, h_src_ptr=0, st=None):
    """rollout_post[_u8] + the recurrent nets' hidden-state reset / h_states row in the same launch"""
    check(lib().a2c_rollout_post_rec(_p(rew), _p(done), val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones), _p(deltas),
                                     T, t, slot0, float(gamma), int(bool(pong)), frame32_ptr, frame_u8_ptr, frame_stride,
                                     _p(reset_mask), prev_ptr, prev_stride, out_ptr, out_stride, B, C, HW, _p(done_eff),
                                     _p(h), h.shape[1], h_rows_ptr, h_rows_stride, h_src_ptr,
                                     st if st is not None else stream()), "a2c_rollout_post_rec")


def compose_heads(Wh, bh, Wp, bp, Wc, bc, st=None):
    """Wc = Wh . Wp, bc = Wh . bp + bh (inference-only composition of two activation-free linear layers)"""
    for t, n in ((Wh, "Wh"), (bh, "bh"), (Wp, "Wp"), (bp, "bp"), (Wc, "Wc"), (bc, "bc")):
        _chk(t, n)
    N, H = Wh.shape
    F = Wp.shape[1]
    check(lib().a2c_compose_heads(_p(Wh), _p(bh), _p(Wp), _p(bp), _p(Wc), _p(bc), N, H, F,
                                  st if st is not None else stream()), "a2c_compose_heads")


def a3c_step_supported(C, H, W, n_actions):
    return bool(lib().a2c_a3c_step_supported(int(C), int(H), int(W), int(n_actions)))


def a3c_ring_supported(B, C, H, W, n_actions, conv1_weight_ptr):
    """a2c_a3c_rollout would run the ring kernel (the body that can leave the fp32 state rows out) for this call"""
    return bool(lib().a2c_a3c_ring_supported(int(B), int(C), int(H), int(W), int(n_actions), int(conv1_weight_ptr)))


def a3c_step(st=None, **kw):
    """One whole rollout step of the A3CModel-shaped policy in one launch (see a2c_a3c_step in the
    header).  Keyword arguments are the fields of a2c_a3c_step_args (pointers as ints, 0 = NULL)."""
    args = _lib.A3CStepArgs()
    for k, v in kw.items():
        setattr(args, k, v)
    check(lib().a2c_a3c_step(ctypes.byref(args), st if st is not None else stream()), "a2c_a3c_step")


def a3c_rollout(st=None, **kw):
    """A whole rollout slot in one persistent launch fed by the host env pool (a2c_a3c_rollout in the header).
    Keyword arguments are the fields of a2c_a3c_rollout_args."""
    args = _lib.A3CRolloutArgs()
    for k, v in kw.items():
        setattr(args, k, v)
    check(lib().a2c_a3c_rollout(ctypes.byref(args), st if st is not None else stream()), "a2c_a3c_rollout")


# ---------------------------------------------------------------- pinned staging / async copies
def pinned_register(host_addr, nbytes):
    """hipHostRegister(mapped): pins [host_addr, +nbytes) and returns the device address of the range"""
    dev = ctypes.c_void_p()
    check(lib().a2c_pinned_register(host_addr, nbytes, ctypes.byref(dev)), "a2c_pinned_register")
    return int(dev.value)


def pinned_unregister(host_addr):
    check(lib().a2c_pinned_unregister(host_addr), "a2c_pinned_unregister")


def push_buffer_alloc(nbytes):
    """fine-grained DEVICE memory the host writes into directly (large BAR): the address, valid on both sides, or 0 when
    the platform refuses it (the caller then keeps the pinned host region as the only copy)"""
    ptr = ctypes.c_void_p(0)
    if lib().a2c_push_buffer_alloc(nbytes, ctypes.byref(ptr)) != 0:
        return 0
    return int(ptr.value or 0)


def push_buffer_free(ptr):
    if ptr:
        lib().a2c_push_buffer_free(ptr)


def device_numa_cpus():
    """CPUs of the NUMA node the current HIP device hangs off, or None when that cannot be read"""
    buf = ctypes.create_string_buffer(64)
    if lib().a2c_device_pci_bus_id(buf, 64) != 0:
        return None
    bdf = buf.value.decode().lower()
    try:
        node = int(open(f"/sys/bus/pci/devices/{bdf}/numa_node").read())
        if node < 0:
            return None
        cpus = []
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            a, _, b = part.partition("-")
            cpus += list(range(int(a), int(b or a) + 1))
        return cpus or None
    except (OSError, ValueError):
        return None


H2D, D2H, D2D = 1, 2, 3


def memcpy_async(dst_ptr, src_ptr, nbytes, kind, st=None):
    check(lib().a2c_memcpy_async(dst_ptr, src_ptr, nbytes, kind, st if st is not None else stream()), "a2c_memcpy_async")


def rollout_bootstrap(val_ptr, val_stride, val_prev, rewards, dones, deltas, B, T, slot0, gamma, st=None):
    check(lib().a2c_rollout_bootstrap(val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones), _p(deltas), B, T, slot0,
                                      float(gamma), st if st is not None else stream()), "a2c_rollout_bootstrap")


def copy_rows(src_ptr, src_stride, dst_ptr, dst_stride, B, n, st=None):
    check(lib().a2c_copy_rows(src_ptr, src_stride, dst_ptr, dst_stride, B, n, st if st is not None else stream()),
          "a2c_copy_rows")


def mask_rows(x, dones_ptr, done_stride, st=None):
    _chk(x, "x")
    check(lib().a2c_mask_rows(_p(x), x.stride(0), dones_ptr, done_stride, x.shape[0], x.shape[1],
                              st if st is not None else stream()), "a2c_mask_rows")


def permute_rows(src, dst, R, T, n, st=None):
    check(lib().a2c_permute_rows(_p(src), _p(dst), R, T, n, st if st is not None else stream()), "a2c_permute_rows")


# ---------------------------------------------------------------- loss
def loss_fwd_bwd(logits, vals, actions, advs, returns, adv_sums, n_global, pi_coef, val_coef, entr_coef, dlogits,
                 dvals, loss_sums, st=None, scratch=None):
    """logits (N,A) / dlogits (N,A) may be column slices of wider (N,A+1) buffers and vals / dvals
    their last column: only stride(0) is used for addressing."""
    _chk(logits, "logits", contig=False); _chk(vals, "vals", contig=False)
    _chk(actions, "actions", torch.int64)
    _chk(advs, "advs"); _chk(returns, "returns"); _chk(dlogits, "dlogits", contig=False)
    _chk(dvals, "dvals", contig=False); _chk(loss_sums, "loss_sums", torch.float64)
    n, A = logits.shape
    st = st if st is not None else stream()
    check(lib().a2c_loss_fwd_bwd(_p(logits), logits.stride(0), _p(vals), vals.stride(0), _p(actions), _p(advs),
                                 _p(returns), _p(adv_sums), n, n_global, A, float(pi_coef), float(val_coef),
                                 float(entr_coef), _p(dlogits), dlogits.stride(0), _p(dvals), dvals.stride(0),
                                 _p(loss_sums), _p(scratch if scratch is not None else reduce_scratch(logits.device, st)), st),
          "a2c_loss_fwd_bwd")


# ---------------------------------------------------------------- dense
def pick_splitk(M, N, K, target_wgs=512, min_k=32):
    min_k = int(os.environ.get("A2C_SPLITK_MIN_K", min_k))
    target_wgs = int(os.environ.get("A2C_SPLITK_TARGET", target_wgs))
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles >= target_wgs // 2:
        return 1
    if tiles <= 4 and K <= 4096 and "A2C_SPLITK_TARGET" not in os.environ:
        # a handful of tiles over a short K (GRUModel's resize_emb at rollout batch: 256 x 256 x 2304): the slabs the reduce reads
        # back weigh more than the last workgroups fill -- 64 slabs 15.2 + 5.0 us, 72 (the rule below) 16.2 + 5.7, 32: 17.0 + 5.2
        target_wgs = min(target_wgs, 256)
    s = max(1, min(target_wgs // tiles, K // min_k))
    return int(s)


def gemm(transA, transB, M, N, K, A_ptr, lda, B_ptr, ldb, C_ptr, ldc, bias=None, relu=False, mask_ptr=0, ldmask=0,
         accumulate=False, splitk=1, ws=None, st=None):
    ws_ptr, ws_bytes = (0, 0) if ws is None else (ws.data_ptr(), ws.numel() * ws.element_size())
    check(lib().a2c_gemm_f32(int(transA), int(transB), M, N, K, A_ptr, lda, B_ptr, ldb, C_ptr, ldc, _p(bias),
                             int(bool(relu)), mask_ptr, ldmask, int(bool(accumulate)), splitk, ws_ptr, ws_bytes,
                             st if st is not None else stream()), "a2c_gemm_f32")


def gemm_x6_image_bytes(rows, K):
    return lib().a2c_gemm_x6_image_bytes(rows, K)


def gemm_x6_split(src_ptr, ld, rows, K, k_contiguous, image, st=None):
    """three-piece bf16 panel image of a rows x K fp32 operand (include/a2c_mi355x.h: a2c_gemm_x6_split)"""
    check(lib().a2c_gemm_x6_split(src_ptr, ld, rows, K, int(bool(k_contiguous)), _p(image), st if st is not None else stream()),
          "a2c_gemm_x6_split")


def gemm_x6_images(M, N, K, image_a, image_b, C_ptr, ldc, bias=None, relu=False, mask_ptr=0, ldmask=0, accumulate=False,
                   splitk=1, ws=None, st=None):
    """C = A B^T from two prebuilt images: six exact bf16 piece products per element pair, fp32 sums"""
    ws_ptr, ws_bytes = (0, 0) if ws is None else (ws.data_ptr(), ws.numel() * ws.element_size())
    if ws is not None:
        note_tensor(ws)
    check(lib().a2c_gemm_x6_images(M, N, K, _p(image_a), _p(image_b), C_ptr, ldc, _p(bias), int(bool(relu)), mask_ptr, ldmask,
                                   int(bool(accumulate)), splitk, ws_ptr, ws_bytes, st if st is not None else stream()),
          "a2c_gemm_x6_images")


def gemm_partial(transA, transB, M, N, K, A_ptr, lda, B_ptr, ldb, splitk, ws, st=None):
    """split-K phase only; returns the number of [M][N] slabs written to ws"""
    check(lib().a2c_gemm_f32_partial(int(transA), int(transB), M, N, K, A_ptr, lda, B_ptr, ldb, splitk, ws.data_ptr(),
                                     ws.numel() * ws.element_size(), st if st is not None else stream()),
          "a2c_gemm_f32_partial")
    return lib().a2c_gemm_splits(K, splitk)


def heads_fused(xs_ptr, nslab, slab_stride, ldx, bias_in, relu_in, emb_out, W, b, heads, M, u=None, n_logits=0,
                actions_ptr=0, act_stride=0, st=None, publish=None):
    """publish = (cmd_ptr, seq_base tensor, seq_off): the sampling thread also stores the action's cmd granule (device relay)"""
    N, K = W.shape
    cmd, seq_base, seq_off = publish if publish is not None else (0, None, 0)
    check(lib().a2c_heads_fused_publish(xs_ptr, nslab, slab_stride, ldx, _p(bias_in), int(bool(relu_in)), _p(emb_out),
                                        0 if emb_out is None else emb_out.stride(0), _p(W), _p(b), _p(heads), heads.stride(0),
                                        M, N, K, _p(u), n_logits, actions_ptr, act_stride, cmd, _p(seq_base), seq_off,
                                        st if st is not None else stream()),
          "a2c_heads_fused_publish")


def gemm_ws_bytes(M, N, splitk, K=None):
    """split-K slabs; with K given, plus the bf16 images of the x 9 path of large products behind them (a2c_gemm_x9_ws_bytes)"""
    base = lib().a2c_gemm_ws_bytes(M, N, splitk)
    if K is None:
        return base
    x9 = lib().a2c_gemm_x9_ws_bytes(M, N, K)
    return base if not x9 else (base + 255) // 256 * 256 + x9


def colsum(x_ptr, ld, M, N, out, ws, st=None):
    check(lib().a2c_colsum(x_ptr, ld, M, N, _p(out), ws.data_ptr(), ws.numel() * ws.element_size(),
                           st if st is not None else stream()), "a2c_colsum")


def colsum_ws_bytes(N):
    return lib().a2c_colsum_ws_bytes(N)


# ---------------------------------------------------------------- conv
def conv_desc(Cin, H, W, Cout, ks, stride, pad):
    OH = (H - ks + 2 * pad) // stride + 1
    OW = (W - ks + 2 * pad) // stride + 1
    return ConvDesc(Cin, H, W, Cout, ks, stride, pad, OH, OW)


def conv_prep_floats(d, kind):
    return lib().a2c_conv2d_prep_floats(ctypes.byref(d), kind)


def conv_prep(d, kind, weight, wprep, st=None):
    check(lib().a2c_conv2d_prep_weights(ctypes.byref(d), kind, _p(weight), _p(wprep),
                                        st if st is not None else stream()), "a2c_conv2d_prep_weights")


def conv_fwd(d, in_ptr, in_bstride, wprep, bias, relu, out, B, st=None, out_bstride=None):
    """out: tensor, or a raw device address together with out_bstride (rows of a larger buffer)"""
    out_ptr = out if isinstance(out, int) else _p(out)
    check(lib().a2c_conv2d_fwd(ctypes.byref(d), in_ptr, in_bstride, _p(wprep), _p(bias), int(bool(relu)), out_ptr,
                               d.Cout * d.OH * d.OW if out_bstride is None else out_bstride, B,
                               st if st is not None else stream()), "a2c_conv2d_fwd")


def conv_bwd_data(d, dout, wprep_bwd, mask, din, B, st=None):
    check(lib().a2c_conv2d_bwd_data(ctypes.byref(d), _p(dout), _p(wprep_bwd), _p(mask), _p(din), B,
                                    st if st is not None else stream()), "a2c_conv2d_bwd_data")


def conv_sign_words(d):
    """words per sample of the sign-word image of d's OUTPUT (0: d's forward cannot write one)"""
    return int(lib().a2c_conv2d_sign_words(ctypes.byref(d)))


def conv_fwd_signs(d, in_ptr, in_bstride, wprep, bias, relu, out, signs_ptr, signs_bstride, B, st=None, out_bstride=None):
    """conv_fwd that also leaves (out > 0) as one bit per activation at signs_ptr (rows of signs_bstride uint32 words)"""
    out_ptr = out if isinstance(out, int) else _p(out)
    check(lib().a2c_conv2d_fwd_signs(ctypes.byref(d), in_ptr, in_bstride, _p(wprep), _p(bias), int(bool(relu)), out_ptr,
                                     d.Cout * d.OH * d.OW if out_bstride is None else out_bstride, signs_ptr, signs_bstride,
                                     B, st if st is not None else stream()), "a2c_conv2d_fwd_signs")


def conv_bwd_data_signs_supported(d):
    return bool(lib().a2c_conv2d_bwd_data_signs_supported(ctypes.byref(d)))


def conv_bwd_data_signs(d, dout, wprep_bwd, signs, din, B, st=None):
    """conv_bwd_data with the ReLU mask given as the sign words of the layer's input: (B, words) int32 tensor or None"""
    check(lib().a2c_conv2d_bwd_data_signs(ctypes.byref(d), _p(dout), _p(wprep_bwd), _p(signs),
                                          0 if signs is None else signs.stride(0), _p(din), B,
                                          st if st is not None else stream()), "a2c_conv2d_bwd_data_signs")


def small_n_bwd_data_bits(dy, ldy, W, dx, ldx, maskbits, M, N, K, st=None):
    """dx[m, :K] = (dy[m, :N] . W[:N, :K]) * bit(m, k) with the ReLU mask as one bit per activation ((M, K/8) uint8 rows):
    a2c_small_n_bwd_data_bits (A3CModel's da2 from the ring kernel's a2 mask bits)"""
    _chk(dy, "dy", contig=False); _chk(W, "W"); _chk(dx, "dx"); _chk(maskbits, "maskbits", torch.uint8)
    check(lib().a2c_small_n_bwd_data_bits(_p(dy), ldy, _p(W), _p(dx), ldx, _p(maskbits), maskbits.stride(0), M, N, K,
                                          st if st is not None else stream()), "a2c_small_n_bwd_data_bits")


def lanemask_from_act(act, lanemask, st=None):
    """lane masks (include/a2c_mi355x.h: a2c_conv2d_bwd_data_lanemask) of a float activation tensor -> (rows, n/64) int64"""
    _chk(act, "act"); _chk(lanemask, "lanemask", torch.int64)
    check(lib().a2c_lanemask_from_act(_p(act), _p(lanemask), act.numel(), st if st is not None else stream()),
          "a2c_lanemask_from_act")


def conv_bwd_data_lanemask_supported(d, B):
    return bool(lib().a2c_conv2d_bwd_data_lanemask_supported(ctypes.byref(d), int(B)))


def conv_bwd_data_lanemask(d, dout, wprep_bwd, lanemask, din, B, st=None):
    """conv_bwd_data with the ReLU mask given as the lane masks of the layer's input ((B, Cin*H*W/64) int64)"""
    _chk(lanemask, "lanemask", torch.int64)
    check(lib().a2c_conv2d_bwd_data_lanemask(ctypes.byref(d), _p(dout), _p(wprep_bwd), _p(lanemask), _p(din), B,
                                             st if st is not None else stream()), "a2c_conv2d_bwd_data_lanemask")


def conv_bwd_data_w1_frames_ws_bytes(d2, d1, B):
    """> 0 when layer d2's sign-word backward-data can absorb the first layer d1's weight gradient from the uint8 frame store"""
    return lib().a2c_conv2d_bwd_data_w1_frames_ws_bytes(ctypes.byref(d2), ctypes.byref(d1), B)


def conv_bwd_data_w1_frames(d2, dout, wprep_bwd, signs, d1, fstore, slot_stride, T, nvalid, dW1, db1, B, ws, st=None):
    _chk(fstore, "frame_store", torch.uint8); _chk(nvalid, "nvalid", torch.int32); _chk(signs, "signs", torch.int32)
    check(lib().a2c_conv2d_bwd_data_w1_frames(ctypes.byref(d2), _p(dout), _p(wprep_bwd), _p(signs), signs.stride(0), ctypes.byref(d1),
                                              _p(fstore), slot_stride, T, _p(nvalid), _p(dW1), _p(db1), B, ws.data_ptr(),
                                              ws.numel() * ws.element_size(), st if st is not None else stream()),
          "a2c_conv2d_bwd_data_w1_frames")


def conv_bwd_weight_ws_bytes(d, B):
    return lib().a2c_conv2d_bwd_weight_ws_bytes(ctypes.byref(d), B)


def conv_bwd_data_w1_ws_bytes(d2, d1, B):
    """> 0 when layer d2's backward-data can absorb the weight gradient of the first layer d1 below it"""
    return lib().a2c_conv2d_bwd_data_w1_ws_bytes(ctypes.byref(d2), ctypes.byref(d1), B)


def conv_bwd_data_w1(d2, dout, wprep_bwd, mask, d1, x_ptr, x_bstride, dW1, db1, B, ws, st=None):
    check(lib().a2c_conv2d_bwd_data_w1(ctypes.byref(d2), _p(dout), _p(wprep_bwd), _p(mask), ctypes.byref(d1), x_ptr,
                                       x_bstride, _p(dW1), _p(db1), B, ws.data_ptr(), ws.numel() * ws.element_size(),
                                       st if st is not None else stream()), "a2c_conv2d_bwd_data_w1")


def conv_bwd_weight(d, in_ptr, in_bstride, dout, dW, db, B, ws, st=None):
    check(lib().a2c_conv2d_bwd_weight(ctypes.byref(d), in_ptr, in_bstride, _p(dout), _p(dW), _p(db), B,
                                      ws.data_ptr(), ws.numel() * ws.element_size(),
                                      st if st is not None else stream()), "a2c_conv2d_bwd_weight")


def conv_bwd_rank_supported(d, n_logits, B):
    """both backward passes of the layer below a rank-n_logits head can form their dOut in the kernels (A3CModel's conv2)"""
    return bool(lib().a2c_conv2d_bwd_rank_supported(ctypes.byref(d), n_logits, B))


def conv_bwd_data_lanemask_rank(d, dl, ld_dl, n_logits, Wc, maskbits, mask_row_bytes, wprep_bwd, lanemask, din, B, st=None):
    _chk(maskbits, "maskbits", torch.uint8); _chk(lanemask, "lanemask", torch.int64)
    check(lib().a2c_conv2d_bwd_data_lanemask_rank(ctypes.byref(d), _p(dl), ld_dl, n_logits, _p(Wc), _p(maskbits), mask_row_bytes,
                                                  _p(wprep_bwd), _p(lanemask), _p(din), B, st if st is not None else stream()),
          "a2c_conv2d_bwd_data_lanemask_rank")


def conv_bwd_weight_rank(d, in_ptr, in_bstride, dl, ld_dl, n_logits, Wc, maskbits, mask_row_bytes, dW, db, B, ws, st=None):
    _chk(maskbits, "maskbits", torch.uint8)
    check(lib().a2c_conv2d_bwd_weight_rank(ctypes.byref(d), in_ptr, in_bstride, _p(dl), ld_dl, n_logits, _p(Wc), _p(maskbits),
                                           mask_row_bytes, _p(dW), _p(db), B, ws.data_ptr(), ws.numel() * ws.element_size(),
                                           st if st is not None else stream()), "a2c_conv2d_bwd_weight_rank")


def conv_bwd_weight_frames(d, fstore, slot_stride, T, nvalid, dout, dW, db, B, ws, st=None):
    """first-layer weight gradient from the rollout's single-frame uint8 store (stack-on-load)"""
    _chk(fstore, "frame_store", torch.uint8); _chk(nvalid, "nvalid", torch.int32)
    check(lib().a2c_conv2d_bwd_weight_frames(ctypes.byref(d), _p(fstore), slot_stride, T, _p(nvalid), _p(dout), _p(dW), _p(db),
                                             B, ws.data_ptr(), ws.numel() * ws.element_size(),
                                             st if st is not None else stream()), "a2c_conv2d_bwd_weight_frames")


def conv_fwd_frames_supported(d):
    return bool(lib().a2c_conv2d_fwd_frames_supported(ctypes.byref(d)))


def conv_fwd_frames(d, fstore_ptr, sample_stride, T, nvalid_ptr, nvalid_stride, wprep, bias, relu, out, B, st=None,
                    out_bstride=None, signs=None):
    """first-layer forward with the input stacked on load from the single-frame uint8 store (raw device addresses:
    the window of sample b starts at fstore_ptr + (b // T) * sample_stride + (b % T) * H * W); signs = (ptr, row stride)"""
    out_ptr = out if isinstance(out, int) else _p(out)
    sg, sgs = signs if signs is not None else (0, 0)
    check(lib().a2c_conv2d_fwd_frames(ctypes.byref(d), fstore_ptr, sample_stride, T, nvalid_ptr, nvalid_stride, _p(wprep), _p(bias),
                                      int(bool(relu)), out_ptr, d.Cout * d.OH * d.OW if out_bstride is None else out_bstride,
                                      sg, sgs, B, st if st is not None else stream()), "a2c_conv2d_fwd_frames")


class ConvChain:
    """argument block of a2c_conv2d_fwd_chain for a fixed run of layers (descriptor array + pointer tables, built once)"""

    def __init__(self, descs):
        n = len(descs)
        self.n = n
        self.descs = (ConvDesc * n)(*descs)
        self.ok = bool(lib().a2c_conv2d_fwd_chain_supported(self.descs, n))
        self.wprep, self.bias, self.out = (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)(), (ctypes.c_void_p * n)()
        self.obs, self.sg, self.sgs = (ctypes.c_int64 * n)(), (ctypes.c_void_p * n)(), (ctypes.c_int64 * n)()

    def fwd(self, in_ptr, in_bstride, wpreps, biases, outs, out_bstrides, B, st=None, relu=True, signs=None):
        """outs: raw device addresses; signs = {layer index: (ptr, row stride in words)} (layer 0, or layers 0 and 1) or None"""
        for i in range(self.n):
            self.wprep[i], self.bias[i], self.out[i], self.obs[i] = _p(wpreps[i]), _p(biases[i]), outs[i], out_bstrides[i]
            self.sg[i], self.sgs[i] = (signs or {}).get(i, (0, 0))
        check(lib().a2c_conv2d_fwd_chain(self.descs, self.n, in_ptr, in_bstride, self.wprep, self.bias, int(bool(relu)), self.out,
                                         self.obs, self.sg, self.sgs, B, st if st is not None else stream()), "a2c_conv2d_fwd_chain")


# ---------------------------------------------------------------- f4: device preprocessing, single-frame store
PREP_SPECS = {"pong_prep": (35, 195, 0, None, 2, 1), "breakout_prep": (35, 195, 8, -8, 2, 0)}     # preprocessing.py:11-23


def prep_out_shape(name, H, W):
    y0, y1, x0, x1, step, _ = PREP_SPECS[name]
    x1 = W if x1 is None else (W + x1 if x1 < 0 else x1)
    return (1, (y1 - y0 + step - 1) // step, (x1 - x0 + step - 1) // step)


def frame_prep_u8(name, raw_ptr, raw_stride, H, W, C, out_ptr, out_stride, n, st=None):
    """pong_prep / breakout_prep on n raw (H, W, C) uint8 frames in HBM -> prepped uint8 frames"""
    y0, y1, x0, x1, step, binarise = PREP_SPECS[name]
    x1 = W if x1 is None else (W + x1 if x1 < 0 else x1)
    check(lib().a2c_frame_prep_u8(raw_ptr, raw_stride, H, W, C, y0, y1, x0, x1, step, binarise, out_ptr, out_stride, n,
                                  st if st is not None else stream()), "a2c_frame_prep_u8")


def frame_store_begin(fstore_ptr, slot_stride, T, C, HW, nvalid_rows_ptr, nvalid_carry_ptr, B, st=None):
    check(lib().a2c_frame_store_begin(fstore_ptr, slot_stride, T, C, HW, nvalid_rows_ptr, nvalid_carry_ptr, B,
                                      st if st is not None else stream()), "a2c_frame_store_begin")


def rollout_post_frames(rew, done, val_ptr, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, B, done_eff,
                        h, h_rows_ptr, h_rows_stride, h_src_ptr, nvalid_rows, nvalid_carry_ptr, st=None):
    """bookkeeping of env step t for the single-frame store (no frame-stack copy), see the header"""
    check(lib().a2c_rollout_post_frames(_p(rew), _p(done), val_ptr, val_stride, _p(val_prev), _p(rewards), _p(dones), _p(deltas),
                                        T, t, slot0, float(gamma), int(bool(pong)), B, _p(done_eff), _p(h),
                                        0 if h is None else h.shape[1], h_rows_ptr, h_rows_stride, h_src_ptr, _p(nvalid_rows),
                                        nvalid_carry_ptr, st if st is not None else stream()), "a2c_rollout_post_frames")


def frames_to_states(fstore_ptr, slot_stride, nvalid_ptr, nvalid_slot_stride, out_ptr, out_slot_stride, R, nt, C, HW, st=None):
    check(lib().a2c_frames_to_states(fstore_ptr, slot_stride, nvalid_ptr, nvalid_slot_stride, out_ptr, out_slot_stride, R, nt, C,
                                     HW, st if st is not None else stream()), "a2c_frames_to_states")


# ---------------------------------------------------------------- GRU / LayerNorm
def gru_gates(gx, gh, b, h, z, r, rh, st=None):
    B, hd = h.shape
    check(lib().a2c_gru_gates(_p(gx), _p(gh), _p(b), _p(h), _p(z), _p(r), _p(rh), B, hd,
                              st if st is not None else stream()), "a2c_gru_gates")


def gru_out(gx, rh_u, b, h, z, c, h_new, st=None):
    B, hd = h.shape
    check(lib().a2c_gru_out(_p(gx), _p(rh_u), _p(b), _p(h), _p(z), _p(c), _p(h_new), B, hd,
                            st if st is not None else stream()), "a2c_gru_out")


def gru_cell_fwd(x, h, WxC, WhC, Wh2, b, gx, z, r, rh, c, h_new, st=None):
    """the whole cell forward of a rollout step in two launches (a2c_gru_cell_fwd); x may be a row-strided view"""
    B, hd = h.shape
    xs = x.shape[1]
    check(lib().a2c_gru_cell_fwd(_p(x), x.stride(0), _p(h), _p(WxC), _p(WhC), _p(Wh2), _p(b), _p(gx), _p(z), _p(r), _p(rh),
                                 _p(c), _p(h_new), B, xs, hd, st if st is not None else stream()), "a2c_gru_cell_fwd")


def gru_cell_bwd(dh_new, carry, dones_ptr, done_stride, h, z, r, c, Wh, dc_pre, dz, dz_pre, dr_pre, dh, st=None):
    """one BPTT backward step of the cell in two launches (a2c_gru_cell_bwd); carry (or None) must not alias dh"""
    B, hd = h.shape
    check(lib().a2c_gru_cell_bwd(_p(dh_new), _p(carry), dones_ptr, done_stride, _p(h), _p(z), _p(r), _p(c), _p(Wh), _p(dc_pre),
                                 _p(dz), _p(dz_pre), _p(dr_pre), _p(dh), B, hd, st if st is not None else stream()),
          "a2c_gru_cell_bwd")


def gru_out_bwd(dh_new, h, z, c, dc_pre, dz, dh, st=None):
    B, hd = h.shape
    check(lib().a2c_gru_out_bwd(_p(dh_new), _p(h), _p(z), _p(c), _p(dc_pre), _p(dz), _p(dh), B, hd,
                                st if st is not None else stream()), "a2c_gru_out_bwd")


def gru_out_bwd_carry(dh_new, carry, dones_ptr, done_stride, h, z, c, dc_pre, dz, dh, st=None):
    """gru_out_bwd with dh_new + carry * (1 - done) as the incoming gradient (BPTT: the carry of the next time step)"""
    B, hd = h.shape
    check(lib().a2c_gru_out_bwd_carry(_p(dh_new), _p(carry), dones_ptr, done_stride, _p(h), _p(z), _p(c), _p(dc_pre), _p(dz),
                                      _p(dh), B, hd, st if st is not None else stream()), "a2c_gru_out_bwd_carry")


def gru_gates_bwd(d_rh, dz, h, z, r, dz_pre, dr_pre, dh, st=None):
    B, hd = h.shape
    check(lib().a2c_gru_gates_bwd(_p(d_rh), _p(dz), _p(h), _p(z), _p(r), _p(dz_pre), _p(dr_pre), _p(dh), B, hd,
                                  st if st is not None else stream()), "a2c_gru_gates_bwd")


def layernorm_fwd(x, w, b, y, mean, rstd, st=None):
    rows, n = x.shape
    check(lib().a2c_layernorm_fwd(_p(x), _p(w), _p(b), _p(y), _p(mean), _p(rstd), rows, n,
                                  st if st is not None else stream()), "a2c_layernorm_fwd")


def layernorm_bwd(dy, x, w, mean, rstd, dx, dw_rows, accumulate=False, st=None):
    rows, n = x.shape
    check(lib().a2c_layernorm_bwd(_p(dy), _p(x), _p(w), _p(mean), _p(rstd), _p(dx), _p(dw_rows), rows, n,
                                  int(bool(accumulate)), st if st is not None else stream()), "a2c_layernorm_bwd")


# ---------------------------------------------------------------- clip + optimiser
def gradnorm_sq(grads, sumsq, st=None, scratch=None):
    st = st if st is not None else stream()
    scratch = scratch if scratch is not None else reduce_scratch(grads.device, st)
    check(lib().a2c_gradnorm_sq(_p(grads), grads.numel(), _p(sumsq), _p(scratch), st), "a2c_gradnorm_sq")


def clip_rmsprop(params, grads, square_avg, sumsq, max_norm, lr, alpha, eps, norm_out, st=None):
    check(lib().a2c_clip_rmsprop(_p(params), _p(grads), _p(square_avg), params.numel(), _p(sumsq), max_norm, lr,
                                 alpha, eps, _p(norm_out), st if st is not None else stream()), "a2c_clip_rmsprop")


def clip_adam(params, grads, exp_avg, exp_avg_sq, sumsq, max_norm, lr, beta1, beta2, eps, step, norm_out, st=None):
    check(lib().a2c_clip_adam(_p(params), _p(grads), _p(exp_avg), _p(exp_avg_sq), params.numel(), _p(sumsq),
                              max_norm, lr, beta1, beta2, eps, step, _p(norm_out),
                              st if st is not None else stream()), "a2c_clip_adam")
