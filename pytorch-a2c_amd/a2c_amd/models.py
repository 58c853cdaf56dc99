"""Drop-in for the reference's ``a2c/models.py``: same class names, constructor arguments,
``forward`` signatures / return values, attributes (``is_recurrent``, ``h_size``,
``req_grads``) and ``state_dict`` keys -- but forward AND backward run on the MI355X kernels
of liba2c_mi355x.so (LDS-tiled matrix-core convolutions, fp32 MFMA GEMMs, fused GRU gates).

The torch ``nn`` sub-modules created here are only parameter containers (they give the
reference's exact key names, parameter order and default initialisation); their own
``forward`` is never called.  All parameters live in one flat HBM arena (engine.Arena).

Quirks of the reference that change results and are reproduced:
  * A3CModel: no activation after ``proj_matrx`` (models.py:73); the value head reads a
    DETACHED embedding (models.py:85) so the value loss never reaches the encoder;
    ``emb_bnorm`` exists but is unused -> its parameters never get a gradient.
  * GRUModel ``conv_block(activation="lerelu")`` resolves to ReLU (models.py:681-689).
  * GRU cell: h = z*h_old + (1-z)*tanh(x Wx2 + (r*h_old) Wh2 + b2), weights (3,in,h) used as
    ``x.mm(W)`` (models.py:472-475).
  * FCModel / GRUFCModel value head = LayerNorm -> Linear(h,1) -> Linear(1,1) (models.py:392-394).
Out of scope: ``bnorm=True`` paths, continuous actions, the unused noise helpers.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .engine import Arena, ConvLayer, linear_bwd_data, linear_bwd_weight, linear_fwd


def _conv_block(cin, cout, ks, stride, pad):
    return nn.Sequential(nn.Conv2d(cin, cout, ks, stride=stride, padding=pad), nn.ReLU())


def _sampler(sampler):
    """sampler = (u, actions_ptr, act_stride[, publish]); publish = (cmd_ptr, seq_base, seq_off) or None (ops.heads_fused)"""
    return (*sampler[:3], sampler[3] if len(sampler) > 3 else None)


class _HipNet(nn.Module):
    """Machinery shared by every model: arena, workspaces, public forward, autograd bridge."""

    is_recurrent = False
    _unused_params = ()          # parameter names that never receive a gradient
    _packed_after = ()           # parameter names laid out right behind their predecessor in the arena

    # ------------------------------------------------------------------ arena / device
    def _post_init(self):
        self._arena = None
        self._ws = {}
        self._dirty = True
        self._saved = {}
        self._stash = None
        self._stash_frames = None

    def _arena_order(self):
        return [n for n, _ in self.named_parameters()]

    def _ensure_device(self):
        if self._arena is not None:
            return
        if not torch.cuda.is_available():
            raise RuntimeError("a2c_amd models need a HIP device: there is no CPU fallback")
        named = dict(self.named_parameters())
        dev = torch.device("cuda", torch.cuda.current_device())
        for b_name, b in self.named_buffers():
            b.data = b.data.to(dev)
        order = self._arena_order()
        trainable = [n for n in order if n not in self._unused_params]
        self._arena = Arena([(n, named[n]) for n in order], set(trainable), dev, set(self._packed_after))
        self._dev = dev
        self._dirty = True
        self._bind_params()

    def _apply(self, fn, *a, **k):
        # .cuda()/.to()/.share_memory() rebuild per-parameter storage: drop the arena, it is
        # rebuilt (from the moved parameters) at the next kernel call.
        out = super()._apply(fn, *a, **k)
        self._arena = None
        self._ws = {}
        self.mark_dirty()
        return out

    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.mark_dirty()
        return out

    def mark_dirty(self):
        """Tell the net its parameters changed (re-derives the conv weight fragments; a rollout's stashed
        activations no longer belong to these weights)."""
        self._dirty = True
        self._stash = None
        self._stash_frames = None
        self._stash_lm = False

    # ---- rollout -> update activation stash (models with a one-launch rollout step)
    def _conv_chain(self):
        """ops.ConvChain over layers 1 .. n-1, or None when the library has no chain kernel for them"""
        ch = getattr(self, "_chain", 0)
        if ch == 0:
            ch = None
            if len(self._cl) >= 2 and not any(l.padded for l in self._cl[1:]):
                c = ops.ConvChain([l.d for l in self._cl[1:]])
                ch = c if c.ok else None
            self._chain = ch
        if ch is not None and os.environ.get("A2C_NO_CHAIN") == "1":
            return None
        return ch

    def stash_rows(self, states, n_rows, T=None):
        """Buffers a rollout may fill with the conv activations of its states (row e of the rollout buffer ->
        row e of the "train" workspace), or None when the model has no stash / it is disabled.  T = n_tsteps."""
        return None

    def stash_commit(self, states, n_rows, frames=None):
        """The rollout that just ran covered ALL n_rows rows of `states` with the current weights.
        ``frames`` = (frame_store uint8 (slots, T+4, HW), nvalid int32 (N,), T): the single-frame store the
        persistent rollout kernel filled (the first layer's weight gradient stacks the frames on load)."""
        self._stash = (states.data_ptr(), int(n_rows))
        self._stash_frames = frames

    def _stash_valid(self, x_ptr, n_rows):
        return getattr(self, "_stash", None) == (x_ptr, int(n_rows))

    def _need_states(self):
        """about to read the fp32 `states` rows: a rollout that kept the single-frame store only (hyps['lazy_states'])
        materialises them now (Runner.materialize_states)"""
        cb = getattr(self, "_materialize_states", None)
        if cb is not None:
            cb()

    def ws(self, tag):
        w = self._ws.get(tag)
        if w is None:
            w = self._ws[tag] = ops.Workspace(self._dev)
        return w

    def _refresh(self, st):
        if self._dirty:
            self._prep(st)
            self._dirty = False

    def _prep(self, st):
        pass

    def _prep_sig(self):
        """what _prep derives: a captured update whose tail ran a different _prep leaves the net dirty after its replay"""
        return ""

    def P(self, name):
        return self._pd[name]

    def _bind_params(self):
        self._pd = dict(self.named_parameters())

    def G(self, name):
        return self._arena.grad(name)

    def req_grads(self, calc_bool):
        for p in self.parameters():
            p.requires_grad = calc_bool

    def check_grads(self):
        for p in self.parameters():
            if torch.sum(p.data != p.data) > 0:
                print("NaNs in Grad!")

    # ------------------------------------------------------------------ public forward
    def _to_dev(self, t):
        self._ensure_device()
        return t.detach().to(device=self._dev, dtype=torch.float32).contiguous()

    def forward(self, x, old_h=None):
        """(val (B,1), logits (B,A)[, h (B,h_size)]) like the reference.  With autograd enabled
        and parameters requiring grad, the result is differentiable (HIP backward kernels)."""
        self._ensure_device()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        xd = self._to_dev(x)
        hd = self._to_dev(old_h) if self.is_recurrent else None
        if self.is_recurrent and old_h is None:
            raise TypeError("recurrent model: forward(x, old_h)")
        if not need_grad:
            out = self._run_forward(xd, hd, "pub", save=False)
            return tuple(o.clone() for o in out)
        params = [p for n, p in self.named_parameters() if n not in self._unused_params]
        h_arg = old_h if (self.is_recurrent and torch.is_tensor(old_h) and old_h.requires_grad) else None
        return _NetFunction.apply(self, xd, hd, h_arg, *params)

    def _run_forward(self, xd, hd, tag, save):
        B = xd.shape[0]
        st = ops.stream()
        self._refresh(st)
        bstride = xd[0].numel()
        if self.is_recurrent:
            o = self._fwd(xd.data_ptr(), bstride, B, tag, st, save, h_in=hd)
            return o["vals"].view(B, 1), o["logits"], o["h"]
        o = self._fwd(xd.data_ptr(), bstride, B, tag, st, save)
        return o["vals"].view(B, 1), o["logits"]

    # heads buffer: (B, A+1) = [logits | value]; gradient buffer has the same layout
    def _heads(self, tag, B):
        A = self.output_space
        hb = self.ws(tag).get("heads", (B, A + 1))
        return hb, hb[:, :A], hb[:, A]

    def dheads(self, tag, B):
        A = self.output_space
        db = self.ws(tag).get("dheads", (B, A + 1))
        return db, db[:, :A], db[:, A]


class _NetFunction(torch.autograd.Function):
    """Autograd bridge for the public ``net(x)`` call: backward runs the HIP backward pass."""

    @staticmethod
    def forward(ctx, net, xd, hd, h_arg, *params):
        out = net._run_forward(xd, hd, "ag", save=True)
        ctx.net, ctx.xd, ctx.hd = net, xd, hd
        ctx.has_h = h_arg is not None
        return tuple(o.clone() for o in out)

    @staticmethod
    def backward(ctx, *gouts):
        net, xd = ctx.net, ctx.xd
        B = xd.shape[0]
        st = ops.stream()
        A = net.output_space
        db, dl, dv = net.dheads("ag", B)
        db.zero_()
        if gouts[1] is not None:
            dl.copy_(gouts[1])
        if gouts[0] is not None:
            dv.copy_(gouts[0].reshape(B))
        dh_next = None
        if net.is_recurrent and len(gouts) > 2 and gouts[2] is not None:
            dh_next = gouts[2].contiguous()
        dh_in = net._bwd(xd.data_ptr(), xd[0].numel(), B, "ag", st, dh_next=dh_next) if net.is_recurrent else \
            net._bwd(xd.data_ptr(), xd[0].numel(), B, "ag", st)
        grads = [net.G(n).clone() for n, _ in net.named_parameters() if n not in net._unused_params]
        gh = dh_in.clone() if (ctx.has_h and dh_in is not None) else None
        return (None, None, None, gh, *grads)


# ====================================================================== A3CModel
class A3CModel(_HipNet):
    """models.py:7-175."""
    _unused_params = ("emb_bnorm.weight", "emb_bnorm.bias")
    _packed_after = ("value.weight", "value.bias")     # [pi.weight; value.weight], [pi.bias | value.bias]

    def __init__(self, input_space, output_space, h_size=256, bnorm=False, is_discrete=True, **kwargs):
        super().__init__()
        if bnorm or not is_discrete:
            raise NotImplementedError("a2c_amd: bnorm / continuous actions are out of scope (see DESIGN.md)")
        self.is_recurrent = False
        self.input_space, self.output_space, self.h_size, self.is_discrete = input_space, output_space, h_size, True
        C, H, W = input_space[-3:]
        self.convs = nn.ModuleList([])
        self.conv1 = _conv_block(C, 16, 8, 4, 0)
        self.convs.append(self.conv1)
        self.conv2 = _conv_block(16, 32, 4, 2, 0)
        self.convs.append(self.conv2)
        self.features = nn.Sequential(*self.convs)
        self._c1 = ConvLayer(C, H, W, 16, 8, 4, 0, need_bwd_data=False, name="conv1")
        _, h1, w1 = self._c1.out_shape
        self._c2 = ConvLayer(16, h1, w1, 32, 4, 2, 0, need_bwd_data=True, name="conv2")
        self.flat_size = int(np.prod(self._c2.out_shape))
        self.proj_matrx = nn.Linear(self.flat_size, h_size)
        self.emb_bnorm = nn.BatchNorm1d(h_size)
        self.pi = nn.Linear(h_size, output_space)
        self.value = nn.Linear(h_size, 1)
        self._post_init()

    def _arena_order(self):
        return ["convs.0.0.weight", "convs.0.0.bias", "convs.1.0.weight", "convs.1.0.bias", "proj_matrx.weight",
                "proj_matrx.bias", "pi.weight", "value.weight", "pi.bias", "value.bias", "emb_bnorm.weight",
                "emb_bnorm.bias"]

    def _prep(self, st):
        # two independent chains of small launches (the conv fragments, the composed heads) at the tail of every update, on the
        # path to the next rollout: a parallel branch (ops.side_branch), not ~7 launch latencies in a row
        with ops.side_branch(1):         # (one branch: every fork / join is itself ~5-10 us of graph edge)
            self._c1.prep(self.P("convs.0.0.weight"), ops.stream())
            self._c2.prep(self.P("convs.1.0.weight"), ops.stream())
        try:
            self._prep_heads(st)
        finally:
            ops.join_branches()

    def _prep_heads(self, st):
        # Inference-only composition: proj_matrx has NO activation (models.py:73), so
        #   [logits | value] = (f Wp^T + bp) Wh^T + bh = f (Wh Wp)^T + (Wh bp + bh).
        # Wc = Wh Wp ((A+1) x flat) lets a rollout step skip the h x flat GEMM entirely; the update
        # (which needs the embedding for its backward pass) keeps the exact two-step path.
        A, h, F = self.output_space, self.h_size, self.flat_size
        ar = self._arena
        Wh = ar.params[ar.offsets["pi.weight"][0]:][:(A + 1) * h].view(A + 1, h)
        bh = ar.params[ar.offsets["pi.bias"][0]:][:A + 1]
        if A + 1 > 8:               # wide action spaces: no skinny-head kernels, plain GEMMs + the sampling kernel
            return
        if getattr(self, "_Wc", None) is None:
            self._Wc = torch.empty(A + 1, F, device=self._dev)
            self._bc = torch.empty(A + 1, device=self._dev)
        ops.compose_heads(Wh, bh, self.P("proj_matrx.weight").data, self.P("proj_matrx.bias").data, self._Wc, self._bc, st)

    @property
    def _fused_sampling(self):
        """_fwd(..., sampler=(u, actions_ptr, act_stride)) samples inside the heads kernel (up to 7 actions)"""
        return self.output_space + 1 <= 8

    def stash_rows(self, states, n_rows, T=None):
        """(a1, a2) buffers of the update's workspace for a rollout over all n_rows rows of `states`: the one-launch
        rollout step writes each state's conv activations there, and update_model's forward (updater.py:80; same
        weights, same states: training.py:150-165) reads them instead of recomputing 45 % of its FLOPs."""
        if os.environ.get("A2C_NO_STASH") == "1" or not self._step_supported():
            return None
        ws = self.ws("train")
        # + the heads [logits | value] of every state: the rollout computed them through the composed matrix
        # Wc = [pi;value].proj_matrx, so the update needs neither the 2592 -> 256 forward GEMM nor the embedding
        return (ws.get("a1", (n_rows,) + self._c1.out_shape), ws.get("a2", (n_rows,) + self._c2.out_shape),
                self._heads("train", n_rows)[0], self._a1_lanemask_rows(n_rows), self._a2_maskbit_rows(n_rows))

    def _a2_maskbit_rows(self, n_rows):
        """(n_rows, F/8) uint8: mask bits of the a2 stash rows (include/a2c_mi355x.h: a2_maskbit_rows), written by the ring
        kernel; da2 = (dl . Wc) * (a2 > 0) then reads 324 B per sample instead of the 10.4 KB row (a2c_small_n_bwd_data_bits)"""
        F = self.flat_size
        if os.environ.get("A2C_NO_LANEMASK") == "1" or F % 8 or self.output_space + 1 > 8:
            return None
        return self.ws("train").get("a2_mb", (n_rows, F // 8), dtype=torch.uint8)

    def _a1_lanemask_rows(self, n_rows):
        """(n_rows, 16*OH1*OW1/64) int64: lane masks of the a1 stash rows (include/a2c_mi355x.h: a1_lanemask_rows), which the
        ring kernel writes beside them -- conv2's backward-data then reads 800 B per sample as its ReLU mask instead of the
        25.6 KB row.  None when the library's streaming kernel cannot take them at this batch (or A2C_NO_LANEMASK=1)."""
        n = int(np.prod(self._c1.out_shape))
        if os.environ.get("A2C_NO_LANEMASK") == "1" or n % 256 or not ops.conv_bwd_data_lanemask_supported(self._c2.d, n_rows):
            return None
        return self.ws("train").get("a1_lm", (n_rows, n // 64), dtype=torch.int64)

    def stash_commit(self, states, n_rows, frames=None, lanemask=False):
        super().stash_commit(states, n_rows, frames=frames)
        self._stash_lm = bool(lanemask)

    def _fwd(self, x_ptr, bstride, B, tag, st, save, sampler=None):
        ws, P = self.ws(tag), self.P
        A, h = self.output_space, self.h_size
        a1 = ws.get("a1", (B,) + self._c1.out_shape)
        a2 = ws.get("a2", (B,) + self._c2.out_shape)
        emb = ws.get("emb", (B, h))
        stashed = save and tag == "train" and self._stash_valid(x_ptr, B)
        hb, logits, vals = self._heads(tag, B)
        self._emb_free = False
        self._bwd_frames = self._stash_frames if (stashed and os.environ.get("A2C_NO_FRAME_STORE") != "1") else None
        if stashed and os.environ.get("A2C_NO_HEADS_STASH") != "1":
            # a1, a2 AND [logits | value] of every row were left here by the rollout: no forward work at all; the
            # backward pass gets dWh = db^T emb without the embedding (see _bwd)
            self._emb_free = True
            return dict(logits=logits, vals=vals, sampled=False)
        if not stashed:
            if save and tag == "train":
                self._need_states()
            self._c1.fwd(x_ptr, bstride, P("convs.0.0.bias"), a1, B, st)
            self._c2.fwd(a1.data_ptr(), a1[0].numel(), P("convs.1.0.bias"), a2, B, st)
        # [pi.weight; value.weight] and [pi.bias | value.bias] are adjacent in the arena: one (A+1)-wide head
        Wh = self._arena.params[self._arena.offsets["pi.weight"][0]:][:(A + 1) * h].view(A + 1, h)
        bh = self._arena.params[self._arena.offsets["pi.bias"][0]:][:A + 1]
        u, a_ptr, a_stride, pub = _sampler(sampler) if sampler is not None else (None, 0, 0, None)
        if A + 1 > 8:               # wide heads: two plain GEMMs; the runner samples with a2c_softmax_sample
            linear_fwd(ws, a2.data_ptr(), self.flat_size, P("proj_matrx.weight"), P("proj_matrx.bias"), emb, B, st)
            linear_fwd(ws, emb.data_ptr(), h, Wh, bh, hb, B, st)
            return dict(logits=logits, vals=vals, sampled=False)
        if sampler is not None and not save and os.environ.get("A2C_NO_COMPOSED_HEADS") != "1":
            # rollout step: heads straight from the conv features through the composed matrix
            ops.heads_fused(a2.data_ptr(), 1, 0, self.flat_size, None, False, None, self._Wc, self._bc, hb, B, u, A,
                            a_ptr, a_stride, st, publish=pub)
            return dict(logits=logits, vals=vals, sampled=True, published=pub is not None)
        Wp = P("proj_matrx.weight")
        sk = ops.pick_splitk(B, h, self.flat_size)
        if sk > 1 and os.environ.get("A2C_NO_FUSED_TAIL") != "1":
            # small batch (rollout): split-K slabs are summed by the heads kernel itself -- proj_matrx
            # epilogue (fixed-order slab sum + bias), both heads and the action sampling in ONE node
            buf = ws.bytes("gemm_ws", ops.gemm_ws_bytes(B, h, sk))
            with ops.span(f"linear.fwd {h}x{self.flat_size}"):
                nslab = ops.gemm_partial(0, 1, B, h, self.flat_size, a2.data_ptr(), self.flat_size, Wp.data_ptr(),
                                         self.flat_size, sk, buf, st)
            ops.heads_fused(buf.data_ptr(), nslab, B * h, h, P("proj_matrx.bias"), False, emb, Wh, bh, hb, B, u, A,
                            a_ptr, a_stride, st, publish=pub if u is not None else None)
        else:
            linear_fwd(ws, a2.data_ptr(), self.flat_size, Wp, P("proj_matrx.bias"), emb, B, st)
            if u is None:         # update: plain skinny layer (wave per row); nothing to sample
                linear_fwd(ws, emb.data_ptr(), h, Wh, bh, hb, B, st)
            else:
                ops.heads_fused(emb.data_ptr(), 1, 0, h, None, False, None, Wh, bh, hb, B, u, A, a_ptr, a_stride, st, publish=pub)
        return dict(logits=logits, vals=vals, sampled=u is not None, published=u is not None and pub is not None)

    def _step_supported(self):
        """True when the one-launch rollout step (a2c_a3c_step) covers this net's shapes."""
        if os.environ.get("A2C_NO_FUSED_STEP") == "1" or os.environ.get("A2C_NO_COMPOSED_HEADS") == "1":
            return False
        C, H, W = self.input_space[-3:]
        return self.output_space + 1 <= 8 and ops.a3c_step_supported(C, H, W, self.output_space)

    def _step(self, B, st, **kw):
        """[bookkeeping] + [frame stack] + forward + [sample] + [bootstrap] of one rollout step in ONE
        launch; ``kw`` = the state/bookkeeping fields of a2c_a3c_step_args."""
        C, H, W = self.input_space[-3:]
        hb, logits, vals = self._heads("roll", B)
        P = self.P
        ops.a3c_step(st, B=B, C=C, H=H, W=W, n_actions=self.output_space, wfrag1=self._c1.wf.data_ptr(),
                     bias1=P("convs.0.0.bias").data_ptr(), wfrag2=self._c2.wf.data_ptr(),
                     bias2=P("convs.1.0.bias").data_ptr(), Wc=self._Wc.data_ptr(), bc=self._bc.data_ptr(),
                     heads=hb.data_ptr(), ldh=hb.stride(0), **kw)
        return dict(logits=logits, vals=vals, sampled=True)

    def _bwd(self, x_ptr, bstride, B, tag, st):
        ws, P, G = self.ws(tag), self.P, self.G
        A, h = self.output_space, self.h_size
        a1 = ws.get("a1", (B,) + self._c1.out_shape)
        a2 = ws.get("a2", (B,) + self._c2.out_shape)
        emb = ws.get("emb", (B, h))
        db, dl, dv = self.dheads(tag, B)
        ar = self._arena
        dWh = ar.grads[ar.offsets["pi.weight"][0]:][:(A + 1) * h].view(A + 1, h)
        dbh = ar.grads[ar.offsets["pi.bias"][0]:][:A + 1]
        F = self.flat_size
        Wp = P("proj_matrx.weight")
        emb_free = getattr(self, "_emb_free", False)
        # Rank-A backward.  proj_matrx has no activation (models.py:73) and the value head reads a DETACHED embedding
        # (models.py:85), so demb = dl . W_pi has rank <= A and neither N x 256 x 2592 GEMM of the textbook backward is
        # needed:   G(proj_matrx.weight) = demb^T a2 = W_pi^T . S[:A]       with S = db^T a2 ((A+1) x F, skinny reduction)
        #           G(proj_matrx.bias)   = colsum(demb) = W_pi^T . colsum(dl)
        #           da2 = (demb . Wp) * (a2 > 0) = (dl . Wc[:A]) * (a2 > 0)  with Wc = [pi; value] . proj_matrx (_prep)
        # The same fp32 products as the reference's, re-associated.  A + 1 > 8 keeps the two GEMMs.
        rank_bwd = A + 1 <= 8 and getattr(self, "_Wc", None) is not None and os.environ.get("A2C_NO_RANK_BWD") != "1"
        da2 = ws.get("da2", (B,) + self._c2.out_shape)

        def head_grads(st):
            """gradients of the heads and of proj_matrx: a chain of ~12 small launches that feeds nothing below"""
            if emb_free or rank_bwd:
                S = ws.get("dWh_S", (A + 1, F))      # S = db^T a2: ONE skinny reduction over the batch (column chunks inside the launch)
                linear_bwd_weight(ws, db, a2.data_ptr(), F, S, dbh if emb_free else None, B, st)
            if emb_free:
                # dWh = db^T emb with emb = a2 Wp^T + 1 bp^T never materialised:  dWh = S Wp^T + colsum(db) bp^T
                # (K in chunks of <= 1024: the one-launch small-product kernel; the block-tiled GEMM takes 72 us for these 5 MFLOP)
                nch = next(n for n in range(max(1, -(-F // 1024)), F + 1) if F % n == 0 and (F // n) % 8 == 0)
                Kc = F // nch
                for c in range(nch):
                    ops.gemm(0, 1, A + 1, h, Kc, S.data_ptr() + 4 * c * Kc, F, Wp.data_ptr() + 4 * c * Kc, F, dWh.data_ptr(), h,
                             accumulate=(c > 0), st=st)
                ops.gemm(0, 0, A + 1, h, 1, dbh.data_ptr(), 1, P("proj_matrx.bias").data_ptr(), h, dWh.data_ptr(), h,
                         accumulate=True, st=st)
            else:
                linear_bwd_weight(ws, db, emb.data_ptr(), h, dWh, dbh, B, st)
            if rank_bwd:
                Wpi, dWp = P("pi.weight"), G("proj_matrx.weight")
                with ops.span("rank_bwd proj_matrx grads"):
                    ops.gemm(1, 0, h, F, A, Wpi.data_ptr(), h, S.data_ptr(), F, dWp.data_ptr(), F, st=st)
                    ops.gemm(1, 0, h, 1, A, Wpi.data_ptr(), h, dbh.data_ptr(), 1, G("proj_matrx.bias").data_ptr(), 1, st=st)

        fuse_da2, mb = False, None
        try:
            if rank_bwd:
                # da2 needs only dl and the composed matrix: the conv backward (2.5 of the update's 2.9 ms, four big launches)
                # starts at once; the head / projection gradients run as a parallel branch beside it (ops.side_branch)
                with ops.side_branch(0):
                    head_grads(ops.stream())
                mb = None
                if (tag == "train" and getattr(self, "_stash_lm", False) and self._stash_valid(x_ptr, B)
                        and os.environ.get("A2C_NO_LANEMASK") != "1"):
                    mb = self._a2_maskbit_rows(B)       # the ring kernel left (a2 > 0) as bits beside the stash
                # da2 = (dl . Wc) * (a2 > 0) formed INSIDE conv2's two backward kernels while they stage a sample (340 MB less
                # written and 680 MB less read per update, one launch fewer): A2C_NO_RANK_FUSE=1 keeps the tensor
                fuse_da2 = (mb is not None and os.environ.get("A2C_NO_RANK_FUSE") != "1" and a1[0].numel() % 4 == 0
                            and ops.conv_bwd_rank_supported(self._c2.d, A, B))
                if not fuse_da2:
                    with ops.span("rank_bwd da2"):
                        if mb is not None:
                            ops.small_n_bwd_data_bits(dl, dl.stride(0), self._Wc, da2, F, mb, B, A, F, st)
                        else:
                            ops.gemm(0, 0, B, F, A, dl.data_ptr(), dl.stride(0), self._Wc.data_ptr(), F, da2.data_ptr(), F,
                                     mask_ptr=a2.data_ptr(), ldmask=F, st=st)
            else:
                head_grads(st)
                demb = ws.get("demb", (B, h))
                linear_bwd_data(ws, db, P("pi.weight"), demb, B, st, n_cols=A)        # value head is detached
                linear_bwd_weight(ws, demb, a2.data_ptr(), F, G("proj_matrx.weight"), G("proj_matrx.bias"), B, st)
                linear_bwd_data(ws, demb, Wp, da2.view(B, -1), B, st, mask=a2)
            if fuse_da2:
                buf = ws.bytes("conv_wgrad_ws", ops.conv_bwd_weight_ws_bytes(self._c2.d, B))
                with ops.span("conv2.bwd_weight"):
                    ops.conv_bwd_weight_rank(self._c2.d, a1.data_ptr(), a1[0].numel(), dl, dl.stride(0), A, self._Wc, mb, mb.stride(0),
                                             G("convs.1.0.weight"), G("convs.1.0.bias"), B, buf, st)
            else:
                self._c2.bwd_weight(a1.data_ptr(), a1[0].numel(), da2, G("convs.1.0.weight"), G("convs.1.0.bias"), B, ws, st)
            da1 = ws.get("da1", (B,) + self._c1.out_shape)
            # the ring kernel left a1's lane masks beside the stash: 800 B per sample instead of the 25.6 KB row as the mask
            lm = None
            if tag == "train" and getattr(self, "_stash_lm", False) and self._stash_valid(x_ptr, B) and os.environ.get("A2C_NO_LANEMASK") != "1":
                lm = self._a1_lanemask_rows(B)
            if fuse_da2 and lm is not None:
                with ops.span("conv2.bwd_data"):
                    ops.conv_bwd_data_lanemask_rank(self._c2.d, dl, dl.stride(0), A, self._Wc, mb, mb.stride(0), self._c2.wb, lm, da1, B, st)
            else:
                if fuse_da2:      # (no lane masks after all: the tensor is needed)
                    ops.small_n_bwd_data_bits(dl, dl.stride(0), self._Wc, da2, F, mb, B, A, F, st)
                self._c2.bwd_data(da2, a1, da1, B, st, lanemask=lm)
            fr = getattr(self, "_bwd_frames", None)
            if fr is not None:      # stack-on-load from the single-frame uint8 store: 28 KB instead of 113 KB per sample
                fstore, nvalid, T = fr
                buf = ws.bytes("conv_wgrad_ws", ops.conv_bwd_weight_ws_bytes(self._c1.d, B))
                with ops.span("conv1.bwd_weight"):
                    ops.conv_bwd_weight_frames(self._c1.d, fstore, fstore.stride(0), T, nvalid, da1, G("convs.0.0.weight"),
                                               G("convs.0.0.bias"), B, buf, st)
            else:
                if tag == "train":
                    self._need_states()
                self._c1.bwd_weight(x_ptr, bstride, da1, G("convs.0.0.weight"), G("convs.0.0.bias"), B, ws, st)
        finally:
            ops.join_branches()       # also on an exception: no branch stays open (or unjoined inside a capture)


def h_is_lockstep(stash, B, R, T):
    """stash = (bufs, row0, row_stride) of a rollout step that plays ALL R slots at once: rows t, t+T, t+2T, ..."""
    return B == R and stash[2] == T and 0 <= stash[1] < T


# ====================================================================== conv-stack models
class _ConvStackNet(_HipNet):
    """Shared by ConvModel and GRUModel: a stack of 3x3 conv+ReLU blocks."""

    def _build_convs(self, input_space, specs):
        C, H, W = input_space[-3:]
        self.convs = nn.ModuleList([])
        self._cl = []
        for i, (co, ks, s, p) in enumerate(specs):
            self.convs.append(_conv_block(C, co, ks, s, p))
            layer = ConvLayer(C, H, W, co, ks, s, p, need_bwd_data=(i > 0), name=f"conv{i+1}")
            self._cl.append(layer)
            C, H, W = layer.out_shape
        self.features = nn.Sequential(*self.convs)
        self.flat_size = C * H * W

    def _prep(self, st):
        for i, l in enumerate(self._cl):
            l.prep(self.P(f"convs.{i}.0.weight"), st)
        if hasattr(self, "gru"):
            self._gru_prep(st)

    def _sign_layers(self):
        """{i: words per sample}: layers whose forward leaves the SIGN WORDS of its output (one bit per activation) for the
        backward-data of layer i + 1, which then reads those instead of the float activation as its ReLU mask
        (a2c_conv2d_fwd_signs / a2c_conv2d_bwd_data_signs): 1/32 of the mask's HBM reads in the update."""
        sl = getattr(self, "_sign_l", None)
        if sl is None:
            sl = {}
            if os.environ.get("A2C_NO_SIGNS") != "1":
                for i in range(len(self._cl) - 1):
                    if self._cl[i].sign_words and self._cl[i + 1].bwd_reads_signs:
                        sl[i] = self._cl[i].sign_words
            self._sign_l = sl
            self._signs_ok = {}
        return sl

    def _convs_fwd(self, x_ptr, bstride, B, ws, st, stash=None, train=False):
        """-> [(ptr, batch stride in floats)] of every layer's activation.  ``stash`` = (bufs, row0, row_stride): the
        activations of sample b go to row row0 + b*row_stride of the update's (N, ...) buffers instead of the
        per-tag workspace (a rollout step of all envs fills rows slot*T + t).  With a stash, or train=True (the update's
        own forward), the sign words of _sign_layers go beside them (rows of the "train" workspace's sg{i})."""
        acts = []
        ptr, bs = x_ptr, bstride
        sl = self._sign_layers() if (stash is not None or train) else {}
        fsrc = getattr(self, "_frames_src", None)       # rollout step: layer 0 stacks its input on load (runner, row f4)
        # layers 1 .. n-1 of a rollout-sized batch as ONE launch where the library has a chain kernel for them (GRUModel's four
        # stride-2 layers: a workgroup walks one sample through all of them; bit-identical to the per-layer launches)
        chain = self._conv_chain() if (not train and 64 < B <= 4096 and all(i in (1, 2, 3) for i in sl if i >= 1)
                                       and (2 not in sl or 1 in sl) and (3 not in sl or 2 in sl)) else None
        for i, l in enumerate(self._cl):
            n = int(np.prod(l.out_shape))
            if chain is not None and i == 1:
                nl = len(self._cl)
                csl = [k for k in (1, 2, 3) if k in sl]                   # layers of the chain that leave sign words
                if stash is None:
                    outs = [ws.get(f"a{k}", (B,) + self._cl[k].out_shape) for k in range(1, nl)]
                    optrs, obss = [o.data_ptr() for o in outs], [int(np.prod(self._cl[k].out_shape)) for k in range(1, nl)]
                    sg = {k - 1: (ws.get(f"sg{k}", (B, sl[k]), dtype=torch.int32).data_ptr(), sl[k]) for k in csl}
                else:
                    bufs, row0, rstride = stash
                    ns = [int(np.prod(self._cl[k].out_shape)) for k in range(1, nl)]
                    optrs = [bufs[k].data_ptr() + 4 * row0 * ns[k - 1] for k in range(1, nl)]
                    obss = [rstride * v for v in ns]
                    sg = {k - 1: (self._sign_bufs[k].data_ptr() + 4 * row0 * sl[k], rstride * sl[k]) for k in csl}
                # a2c_conv2d_fwd_chain wants 16-byte pointers and strides that are multiples of 4 floats (stash rows
                # bufs[k] + 4*row0*n are 16-byte aligned only when n % 4 == 0); the per-layer launches below take anything
                if ptr % 16 == 0 and bs % 4 == 0 and all(q % 16 == 0 for q in optrs) and all(o % 4 == 0 for o in obss):
                    with ops.span("conv2-5.fwd_chain"):
                        chain.fwd(ptr, bs, [c.wf for c in self._cl[1:]], [self.P(f"convs.{k}.0.bias") for k in range(1, nl)],
                                  optrs, obss, B, st, signs=sg or None)
                    for k in csl:
                        self._signs_ok[k] = (self._signs_ok.get(k, True) if (stash is not None and stash[1]) else True)
                    for k in range(1, nl):
                        acts.append((optrs[k - 1], obss[k - 1]))
                    return acts
                chain = None
            if stash is None:
                a = ws.get(f"a{i}", (B,) + l.out_shape)
                sg = None
                if i in sl:
                    sg = (ws.get(f"sg{i}", (B, sl[i]), dtype=torch.int32).data_ptr(), sl[i])
                if i == 0 and fsrc is not None:
                    done = l.fwd_frames(fsrc, self.P("convs.0.0.bias"), a, B, st, signs=sg)
                else:
                    done = l.fwd(ptr, bs, self.P(f"convs.{i}.0.bias"), a, B, st, signs=sg)
                if i in sl:
                    self._signs_ok[i] = bool(done)
                ptr, bs = a.data_ptr(), n
            else:
                bufs, row0, rstride = stash
                optr = bufs[i].data_ptr() + 4 * row0 * n
                sg = None
                if i in sl:
                    sgb = self._sign_bufs[i]
                    sg = (sgb.data_ptr() + 4 * row0 * sl[i], rstride * sl[i])
                if i == 0 and fsrc is not None:
                    done = l.fwd_frames(fsrc, self.P("convs.0.0.bias"), optr, B, st, out_bstride=rstride * n, signs=sg)
                else:
                    done = l.fwd(ptr, bs, self.P(f"convs.{i}.0.bias"), optr, B, st, out_bstride=rstride * n, signs=sg)
                if i in sl:
                    self._signs_ok[i] = bool(done) and self._signs_ok.get(i, True) if row0 else bool(done)
                ptr, bs = optr, rstride * n
            acts.append((ptr, bs))
        return acts

    def stash_rows(self, states, n_rows, T=None):
        """per-layer (n_rows, C, H, W) activation buffers of the update's workspace: a rollout that covers all rows
        writes each state's conv activations straight into them and update_model's forward (updater.py:80; same
        weights, same states) skips the conv stack."""
        if os.environ.get("A2C_NO_STASH") == "1":
            return None
        ws = self.ws("train")
        bufs = [ws.get(f"a{i}", (n_rows,) + l.out_shape) for i, l in enumerate(self._cl)]
        self._sign_bufs = {i: ws.get(f"sg{i}", (n_rows, nw), dtype=torch.int32) for i, nw in self._sign_layers().items()}
        if self._E_STASH and os.environ.get("A2C_NO_EMB_STASH") != "1":
            # + the embedding e = relu(resize_emb(features)) of every state (the rollout computed it too): the update's
            # forward then also skips the flat_size -> e GEMM (ConvModel: 28224 -> 2000, 2.6 of its 16 ms)
            bufs.append(ws.get("e", (n_rows, self._E_STASH())))
        self._cells_T = None
        if T and hasattr(self, "gru") and hasattr(self, "_tm_bufs") and n_rows % T == 0 and os.environ.get("A2C_NO_CELL_STASH") != "1" \
                and len(bufs) > len(self._cl) and self.output_space + 1 <= 8:
            # BPTT (updater.py:139-169) re-runs every GRU cell of the slot with the weights and inputs the rollout
            # just used: with all R slots played in lock-step the rollout's step t IS row t of the update's time-major
            # buffers, so the cells (and the heads) write there and bptt_forward has nothing left to compute
            self._cells_T, self._cells_R, self._cells_done = int(T), n_rows // int(T), -1
            self._tm_bufs(ws, self._cells_R, int(T))
        return bufs

    _E_STASH = None      # subclasses: callable -> width of the "e" workspace rows

    def _e_out(self, ws, B, stash):
        """(B, width) tensor the rollout writes e into: rows of the update's workspace when the stash carries them"""
        width = self._E_STASH()
        if stash is not None and len(stash[0]) > len(self._cl):
            bufs, row0, rstride = stash
            return bufs[len(self._cl)][row0::rstride][:B]
        return ws.get("e", (B, width))

    def _e_stashed(self, x_ptr, B):
        return (self._E_STASH is not None and os.environ.get("A2C_NO_EMB_STASH") != "1" and os.environ.get("A2C_NO_STASH") != "1"
                and self._stash_valid(x_ptr, B))

    def _convs_train(self, x_ptr, bstride, B, ws, st):
        """conv stack of the update's forward: recomputed, or taken from the rollout's stash"""
        if self._stash_valid(x_ptr, B):
            return [(ws.get(f"a{i}", (B,) + l.out_shape).data_ptr(), int(np.prod(l.out_shape))) for i, l in enumerate(self._cl)]
        self._need_states()
        return self._convs_fwd(x_ptr, bstride, B, ws, st, train=True)

    def _convs_bwd(self, x_ptr, bstride, B, ws, st, d_last):
        """d_last: gradient wrt the last conv's pre-activation output (ReLU mask already applied)."""
        n = len(self._cl)
        acts = [ws.get(f"a{i}", (B,) + l.out_shape) for i, l in enumerate(self._cl)]
        d = d_last
        for i in range(n - 1, -1, -1):
            l = self._cl[i]
            in_ptr, in_bs = (x_ptr, bstride) if i == 0 else (acts[i - 1].data_ptr(), acts[i - 1][0].numel())
            fr = self._stash_frames if (i == 0 and l.frames_ok and self._stash_valid(x_ptr, B)
                                       and os.environ.get("A2C_NO_FRAME_STORE") != "1") else None
            if fr is not None:
                # the rollout kept ONE uint8 frame per env step: the first layer's weight gradient stacks them on load
                # (28 KB instead of the 113 KB fp32 state per sample; same values, same summation order)
                fstore, nvalid, T = fr
                buf = ws.bytes("conv_wgrad_ws", ops.conv_bwd_weight_ws_bytes(l.d, B))
                with ops.span(l.name + ".bwd_weight"):
                    ops.conv_bwd_weight_frames(l.d, fstore, fstore.stride(0), T, nvalid, d, self.G("convs.0.0.weight"),
                                               self.G("convs.0.0.bias"), B, buf, st)
                continue
            if i == 0:
                self._need_states()
            l.bwd_weight(in_ptr, in_bs, d, self.G(f"convs.{i}.0.weight"), self.G(f"convs.{i}.0.bias"), B, ws, st)
            if i == 1 and self._stash_frames is None and not self._cl[0].padded and not l.padded and bstride % 4 == 0 and x_ptr % 16 == 0:
                # layer 2's input gradient only feeds layer 1's weight gradient: one fused pass, da0 never reaches HBM
                l0 = self._cl[0]
                nb = ops.conv_bwd_data_w1_ws_bytes(l.d, l0.d, B)
                if nb:
                    buf = ws.bytes("conv_w1_ws", nb)
                    with ops.span("conv2.bwd_data+conv1.bwd_weight"):
                        ops.conv_bwd_data_w1(l.d, d, l.wb, acts[0], l0.d, x_ptr, bstride, self.G("convs.0.0.weight"),
                                             self.G("convs.0.0.bias"), B, buf, st)
                    return
            if i == 1 and os.environ.get("A2C_NO_W1_FRAMES") != "1":
                # Round 6: layer 2's input gradient (16 x 84 x 84 floats per sample: 14.8 GB at N = 32,768, written by one launch
                # and read back by the next) only feeds layer 1's weight gradient.  With the single-frame store and layer 1's
                # sign words at hand the fused pass keeps it in LDS and multiplies it against the uint8 frames on the bf16 pipe
                # (a2c_conv2d_bwd_data_w1_frames): the same sums, re-associated; da0 never reaches HBM.
                l0, sl = self._cl[0], self._sign_layers()
                fr0 = self._stash_frames if (l0.frames_ok and self._stash_valid(x_ptr, B)
                                             and os.environ.get("A2C_NO_FRAME_STORE") != "1") else None
                if fr0 is not None and 0 in sl and self._signs_ok.get(0, False) and not l.padded:
                    nb = ops.conv_bwd_data_w1_frames_ws_bytes(l.d, l0.d, B)
                    if nb:
                        fstore, nvalid, T = fr0
                        buf = ws.bytes("conv_w1f_ws", nb)
                        with ops.span("conv2.bwd_data+conv1.bwd_weight"):
                            ops.conv_bwd_data_w1_frames(l.d, d, l.wb, ws.get("sg0", (B, sl[0]), dtype=torch.int32), l0.d, fstore,
                                                        fstore.stride(0), T, nvalid, self.G("convs.0.0.weight"),
                                                        self.G("convs.0.0.bias"), B, buf, st)
                        return
            if i > 0:
                dprev = ws.get(f"da{i-1}", (B,) + self._cl[i - 1].out_shape)
                sl = self._sign_layers()
                if (i - 1) in sl and self._signs_ok.get(i - 1, False):     # the forward that filled acts[i-1] left its sign words
                    l.bwd_data(d, None, dprev, B, st, signs=ws.get(f"sg{i-1}", (B, sl[i - 1]), dtype=torch.int32))
                else:
                    l.bwd_data(d, acts[i - 1], dprev, B, st)
                d = dprev


class ConvModel(_ConvStackNet):
    """models.py:177-365."""
    _packed_after = ("value.0.weight", "value.0.bias")
    SPECS = [(16, 3, 1, 1), (24, 3, 1, 1), (32, 3, 2, 1), (64, 3, 2, 1)]
    CONV_H = 2000

    def _E_STASH(self):
        return self.CONV_H

    def __init__(self, input_space, output_space, h_size=288, bnorm=False, is_discrete=True, **kwargs):
        super().__init__()
        if bnorm or not is_discrete:
            raise NotImplementedError("a2c_amd: bnorm / continuous actions are out of scope (see DESIGN.md)")
        self.is_recurrent = False
        self.input_space, self.output_space, self.h_size = input_space, output_space, h_size
        self.bnorm, self.is_discrete = False, True
        self._build_convs(input_space, self.SPECS)
        print("Flat Features Size:", self.flat_size)
        ch = self.CONV_H
        self.resize_emb = nn.Sequential(nn.Linear(self.flat_size, ch), nn.ReLU())
        self.pi = nn.Sequential(nn.Linear(ch, h_size), nn.ReLU(), nn.Linear(h_size, output_space))
        self.value = nn.Sequential(nn.Linear(ch, h_size), nn.ReLU(), nn.Linear(h_size, 1))
        if h_size % 4:
            raise ValueError("a2c_amd ConvModel: h_size must be a multiple of 4")
        self._post_init()

    def _arena_order(self):
        names = []
        for i in range(4):
            names += [f"convs.{i}.0.weight", f"convs.{i}.0.bias"]
        # pi.0 / value.0 adjacent: the two hidden layers run as ONE (2h x 2000) GEMM
        return names + ["resize_emb.0.weight", "resize_emb.0.bias", "pi.0.weight", "value.0.weight", "pi.0.bias",
                        "value.0.bias", "pi.2.weight", "pi.2.bias", "value.2.weight", "value.2.bias"]

    def _cat(self, buf, name, rows, cols=None):
        o = self._arena.offsets[name][0]
        t = buf[o:o + rows * (cols or 1)]
        return t.view(rows, cols) if cols else t

    @property
    def _fused_sampling(self):
        """_fwd(..., sampler=(u, actions_ptr, act_stride)) samples inside the heads kernel (up to 7 actions)"""
        return self.output_space + 1 <= 8 and os.environ.get("A2C_NO_FUSED_TAIL") != "1"

    def _set_rollout_batch(self, B):
        """Runner, before a rollout block of B envs: at 128..256 rows the resize_emb product is matrix-bound (29 GFLOP per env
        step) and runs on the bf16 pipe from a piece image of the weights that _prep keeps current from then on"""
        if (128 <= B <= 256 and getattr(self, "_W6", None) is None and self.flat_size * self.CONV_H >= (1 << 24)
                and os.environ.get("A2C_NO_X6_FWD") != "1" and os.environ.get("A2C_GEMM_X9", "") != "0"):
            self._ensure_device()
            self._W6 = torch.empty(ops.gemm_x6_image_bytes(self.CONV_H, self.flat_size) // 2, dtype=torch.int16, device=self._dev)
            self._dirty = True

    def _prep_sig(self):
        return "x6" if getattr(self, "_W6", None) is not None else ""

    def _prep(self, st):
        super()._prep(st)
        if getattr(self, "_W6", None) is not None:
            W = self.P("resize_emb.0.weight")
            ops.gemm_x6_split(W.data_ptr(), self.flat_size, self.CONV_H, self.flat_size, True, self._W6, st)
        # Inference-only: pi.2 and value.2 read the two halves of ONE hidden row (pi.0 | value.0 run as one GEMM), so
        # both heads are one (A+1) x 2h skinny layer with zero blocks: [[pi.2.weight, 0], [0, value.2.weight]]
        A, h = self.output_space, self.h_size
        if A + 1 > 8:
            return
        if getattr(self, "_Whd", None) is None:
            self._Whd = torch.zeros(A + 1, 2 * h, device=self._dev)
            self._bhd = torch.zeros(A + 1, device=self._dev)
        ops.copy_rows(self.P("pi.2.weight").data_ptr(), h, self._Whd.data_ptr(), 2 * h, A, h, st)
        ops.copy_rows(self.P("value.2.weight").data_ptr(), h, self._Whd.data_ptr() + 4 * (A * 2 * h + h), 2 * h, 1, h, st)
        ops.copy_rows(self.P("pi.2.bias").data_ptr(), A, self._bhd.data_ptr(), A, 1, A, st)
        ops.copy_rows(self.P("value.2.bias").data_ptr(), 1, self._bhd.data_ptr() + 4 * A, 1, 1, 1, st)

    def _fwd(self, x_ptr, bstride, B, tag, st, save, stash=None, sampler=None):
        ws, P = self.ws(tag), self.P
        A, h, ch = self.output_space, self.h_size, self.CONV_H
        acts = self._convs_train(x_ptr, bstride, B, ws, st) if (save and tag == "train") else \
            self._convs_fwd(x_ptr, bstride, B, ws, st, stash=stash)
        if save and tag == "train" and self._e_stashed(x_ptr, B):
            e = ws.get("e", (B, ch))                 # left here, row by row, by the rollout
        else:
            e = self._e_out(ws, B, stash)
            if not save and getattr(self, "_W6", None) is not None and 128 <= B <= 256:
                # rollout batch against the 226 MB layer: the bf16 piece image of the weights is rebuilt once per update
                # (_prep), the activation rows are split per step; six exact piece products, fp32 sums (gemm_x6_kernel)
                F = self.flat_size
                img = ws.bytes("x6_act", ops.gemm_x6_image_bytes(B, F))
                sk = 32
                buf = ws.bytes("gemm_ws", 4 * sk * B * ch)
                with ops.span(f"linear.fwd {ch}x{F}"):
                    ops.gemm_x6_split(acts[-1][0], acts[-1][1], B, F, True, img, st)
                    ops.gemm_x6_images(B, ch, F, img, self._W6, e.data_ptr(), e.stride(0), bias=P("resize_emb.0.bias"), relu=True,
                                       splitk=sk, ws=buf, st=st)
            else:
                linear_fwd(ws, acts[-1][0], acts[-1][1], P("resize_emb.0.weight"), P("resize_emb.0.bias"), e, B, st, relu=True)
        lde = e.stride(0)
        hid = ws.get("hid", (B, 2 * h))
        W0, b0 = self._cat(self._arena.params, "pi.0.weight", 2 * h, ch), self._cat(self._arena.params, "pi.0.bias", 2 * h)
        hb, logits, vals = self._heads(tag, B)
        if sampler is not None and not save and self._fused_sampling:
            # rollout step: the split-K slabs of the hidden layer are summed by the heads kernel itself -- slab sum +
            # bias + ReLU, both heads (block matrix above) and the action sampling in ONE node instead of four
            u, a_ptr, a_stride, pub = _sampler(sampler)
            sk = ops.pick_splitk(B, 2 * h, ch)
            if sk > 1:
                buf = ws.bytes("gemm_ws", ops.gemm_ws_bytes(B, 2 * h, sk))
                with ops.span(f"linear.fwd {2 * h}x{ch}"):
                    nslab = ops.gemm_partial(0, 1, B, 2 * h, ch, e.data_ptr(), lde, W0.data_ptr(), ch, sk, buf, st)
                ops.heads_fused(buf.data_ptr(), nslab, B * 2 * h, 2 * h, b0, True, None, self._Whd, self._bhd, hb, B, u, A,
                                a_ptr, a_stride, st, publish=pub)
            else:
                linear_fwd(ws, e.data_ptr(), lde, W0, b0, hid, B, st, relu=True)
                ops.heads_fused(hid.data_ptr(), 1, 0, 2 * h, None, False, None, self._Whd, self._bhd, hb, B, u, A, a_ptr,
                                a_stride, st, publish=pub)
            return dict(logits=logits, vals=vals, sampled=True, published=pub is not None)
        linear_fwd(ws, e.data_ptr(), lde, W0, b0, hid, B, st, relu=True)
        linear_fwd(ws, hid.data_ptr(), 2 * h, P("pi.2.weight"), P("pi.2.bias"), logits, B, st)
        linear_fwd(ws, hid.data_ptr() + 4 * h, 2 * h, P("value.2.weight"), P("value.2.bias"), hb[:, A:], B, st)
        return dict(logits=logits, vals=vals)

    def _bwd(self, x_ptr, bstride, B, tag, st):
        ws, P, G = self.ws(tag), self.P, self.G
        A, h, ch = self.output_space, self.h_size, self.CONV_H
        acts = [ws.get(f"a{i}", (B,) + l.out_shape) for i, l in enumerate(self._cl)]
        e = ws.get("e", (B, ch))
        hid = ws.get("hid", (B, 2 * h))
        db, dl, dv = self.dheads(tag, B)
        dvm = db[:, A:]
        linear_bwd_weight(ws, dl, hid.data_ptr(), 2 * h, G("pi.2.weight"), G("pi.2.bias"), B, st)
        linear_bwd_weight(ws, dvm, hid.data_ptr() + 4 * h, 2 * h, G("value.2.weight"), G("value.2.bias"), B, st)
        dhid = ws.get("dhid", (B, 2 * h))
        ops.gemm(0, 0, B, h, A, dl.data_ptr(), A + 1, P("pi.2.weight").data_ptr(), h, dhid.data_ptr(), 2 * h,
                 mask_ptr=hid.data_ptr(), ldmask=2 * h, st=st)
        ops.gemm(0, 0, B, h, 1, dvm.data_ptr(), A + 1, P("value.2.weight").data_ptr(), h, dhid.data_ptr() + 4 * h,
                 2 * h, mask_ptr=hid.data_ptr() + 4 * h, ldmask=2 * h, st=st)
        linear_bwd_weight(ws, dhid, e.data_ptr(), ch, self._cat(self._arena.grads, "pi.0.weight", 2 * h, ch),
                          self._cat(self._arena.grads, "pi.0.bias", 2 * h), B, st)
        de = ws.get("de", (B, ch))
        linear_bwd_data(ws, dhid, self._cat(self._arena.params, "pi.0.weight", 2 * h, ch), de, B, st, mask=e)
        linear_bwd_weight(ws, de, acts[-1].data_ptr(), self.flat_size, G("resize_emb.0.weight"),
                          G("resize_emb.0.bias"), B, st)
        dlast = ws.get(f"da{len(acts)-1}", (B,) + self._cl[-1].out_shape)
        linear_bwd_data(ws, de, P("resize_emb.0.weight"), dlast.view(B, -1), B, st, mask=acts[-1])
        self._convs_bwd(x_ptr, bstride, B, ws, st, dlast)


# ====================================================================== GRU cell helper
class GRU(nn.Module):
    """Parameter container with the reference's names / shapes / init (models.py:426-481)."""

    def __init__(self, x_size=256, h_size=256, layer_norm=False, **kwargs):
        super().__init__()
        self.x_size, self.h_size, self.n_state_vars = x_size, h_size, 1
        std = 1 / float(np.sqrt(h_size))
        self.W_x = nn.Parameter(torch.normal(torch.zeros(3, x_size, h_size), std=std), requires_grad=True)
        self.W_h = nn.Parameter(torch.normal(torch.zeros(3, h_size, h_size), std=std), requires_grad=True)
        self.b = nn.Parameter(torch.zeros(3, 1, h_size), requires_grad=True)

    def forward(self, x, old_h):
        raise RuntimeError("a2c_amd.GRU is a parameter container; call the owning model")


class _GruMixin:
    """GRU cell forward / backward over B rows on the HIP kernels (6 GEMMs + 2 fused gate
    kernels forward).  Buffers are (B, h) contiguous; `pre` prefixes workspace names so the
    BPTT unroll can keep one set per time step (time-major)."""

    @staticmethod
    def _mm(ws, ta, tb, M, N, K, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, st, accumulate=False):
        """One of the cell's h x h products.  At rollout batch these are 2 x 2 tiles of 128 x 128: split K
        (deterministic slab reduce) so that more than 4 workgroups share the work."""
        sk = ops.pick_splitk(M, N, K)
        buf = ws.bytes("gemm_ws", ops.gemm_ws_bytes(M, N, sk)) if sk > 1 else None
        ops.gemm(ta, tb, M, N, K, a_ptr, lda, b_ptr, ldb, c_ptr, ldc, accumulate=accumulate, splitk=sk, ws=buf, st=st)

    def _gru_prep(self, st):
        """Inference/forward-only derived weights (rebuilt when the parameters change): the gate matrices side by
        side, [Wx0|Wx1|Wx2] (in, 3h) and [Wh0|Wh1] (h, 2h), so that a cell's x-side and h-side gate products
        are ONE GEMM each instead of three / two (models.py:472-475 multiplies gate by gate)."""
        hd, xs = self.h_size, self.gru.x_size
        Wx, Wh = self.P("gru.W_x"), self.P("gru.W_h")
        if getattr(self, "_WxC", None) is None:
            self._WxC = torch.empty(xs, 3 * hd, device=self._dev)
            self._WhC = torch.empty(hd, 2 * hd, device=self._dev)
        for g in range(3):
            ops.copy_rows(Wx[g].data_ptr(), hd, self._WxC.data_ptr() + 4 * g * hd, 3 * hd, xs, hd, st)
        for g in range(2):
            ops.copy_rows(Wh[g].data_ptr(), hd, self._WhC.data_ptr() + 4 * g * hd, 2 * hd, hd, hd, st)

    def _gru_fwd(self, ws, x, h_in, B, st, bufs):
        hd, xs = self.h_size, self.gru.x_size
        Wx, Wh, b = self.P("gru.W_x"), self.P("gru.W_h"), self.P("gru.b")
        gx, gh, z, r, rh, rhu, c, hn = (bufs[k] for k in ("gx", "gh", "z", "r", "rh", "rhu", "c", "hn"))
        cat = getattr(self, "_WxC", None) is not None and os.environ.get("A2C_NO_GRU_CAT") != "1"
        if (cat and x is not None and B <= 1280 and hd % 32 == 0 and xs % 8 == 0 and x.stride(0) % 4 == 0
                and x.data_ptr() % 16 == 0 and h_in.data_ptr() % 16 == 0 and os.environ.get("A2C_NO_GRU_FUSE") != "1"):
            # a step at rollout batch: the five launches below (5-6 us each, launch-bound) as two, bit-identical
            with ops.span("gru.cell_fwd"):
                ops.gru_cell_fwd(x, h_in, self._WxC, self._WhC, Wh[2], b, gx, z, r, rh, c, hn, st)
            return
        if x is not None:   # gx may have been precomputed for all time steps at once
            if cat:
                self._mm(ws, 0, 0, B, 3 * hd, xs, x.data_ptr(), x.stride(0), self._WxC.data_ptr(), 3 * hd, gx.data_ptr(), 3 * hd, st)
            else:
                for g in range(3):
                    self._mm(ws, 0, 0, B, hd, xs, x.data_ptr(), x.stride(0), Wx[g].data_ptr(), hd, gx.data_ptr() + 4 * g * hd, 3 * hd, st)
        if cat:
            self._mm(ws, 0, 0, B, 2 * hd, hd, h_in.data_ptr(), hd, self._WhC.data_ptr(), 2 * hd, gh.data_ptr(), 2 * hd, st)
        else:
            for g in range(2):
                self._mm(ws, 0, 0, B, hd, hd, h_in.data_ptr(), hd, Wh[g].data_ptr(), hd, gh.data_ptr() + 4 * g * hd, 2 * hd, st)
        ops.gru_gates(gx, gh, b, h_in, z, r, rh, st)
        self._mm(ws, 0, 0, B, hd, hd, rh.data_ptr(), hd, Wh[2].data_ptr(), hd, rhu.data_ptr(), hd, st)
        ops.gru_out(gx, rhu, b, h_in, z, c, hn, st)

    def _gru_bwd_step(self, ws, dhn, h_in, B, st, bufs, dbufs, carry=None):
        """dhn (B,h) total gradient wrt this step's h_new -- plus, inside the BPTT unroll, carry = (tensor, dones_ptr,
        done_stride): the gradient that reached the next step's h_in = h_new * (1 - done), folded into the same launch.
        Fills dbufs dz_pre/dr_pre/dc_pre and returns dh_in (B,h) in dbufs['dh']."""
        hd = self.h_size
        Wh = self.P("gru.W_h")
        z, r, c = bufs["z"], bufs["r"], bufs["c"]
        dzp, drp, dcp, dz, drh, dh = (dbufs[k] for k in ("dz_pre", "dr_pre", "dc_pre", "dz", "d_rh", "dh"))
        if (B <= 1280 and hd % 32 == 0 and (carry is None or carry[0].data_ptr() != dh.data_ptr())
                and os.environ.get("A2C_NO_GRU_FUSE") != "1"):
            # the step's five launches (~4.5 us each, strictly serial over the unroll) as two, bit-identical
            with ops.span("gru.cell_bwd"):
                if carry is not None:
                    ops.gru_cell_bwd(dhn, carry[0], carry[1], carry[2], h_in, z, r, c, Wh, dcp, dz, dzp, drp, dh, st)
                else:
                    ops.gru_cell_bwd(dhn, None, 0, 1, h_in, z, r, c, Wh, dcp, dz, dzp, drp, dh, st)
            return dh
        if carry is not None:
            ops.gru_out_bwd_carry(dhn, carry[0], carry[1], carry[2], h_in, z, c, dcp, dz, dh, st)
        else:
            ops.gru_out_bwd(dhn, h_in, z, c, dcp, dz, dh, st)
        self._mm(ws, 0, 1, B, hd, hd, dcp.data_ptr(), hd, Wh[2].data_ptr(), hd, drh.data_ptr(), hd, st)   # dc_pre Wh2^T
        ops.gru_gates_bwd(drh, dz, h_in, z, r, dzp, drp, dh, st)
        self._mm(ws, 0, 1, B, hd, hd, dzp.data_ptr(), hd, Wh[0].data_ptr(), hd, dh.data_ptr(), hd, st, accumulate=True)
        self._mm(ws, 0, 1, B, hd, hd, drp.data_ptr(), hd, Wh[1].data_ptr(), hd, dh.data_ptr(), hd, st, accumulate=True)
        return dh

    def _gru_bwd_weights(self, ws, x, h_in, rh, dzp, drp, dcp, dx, M, st, x_mask=None):
        """Batched over M rows (all time steps at once): dW_x, dW_h, db and dx = sum_g dG_g Wx_g^T."""
        hd, xs = self.h_size, self.gru.x_size
        Wx = self.P("gru.W_x")
        dWx, dWh, dbg = self.G("gru.W_x"), self.G("gru.W_h"), self.G("gru.b")
        dgs = (dzp, drp, dcp)
        lefts = (h_in, h_in, rh)
        cs = ws.bytes("colsum_ws", ops.colsum_ws_bytes(hd))
        for g in range(3):
            sk = ops.pick_splitk(xs, hd, M)
            buf = ws.bytes("gemm_ws", ops.gemm_ws_bytes(xs, hd, sk)) if sk > 1 else None
            ops.gemm(1, 0, xs, hd, M, x.data_ptr(), xs, dgs[g].data_ptr(), hd, dWx[g].data_ptr(), hd, splitk=sk, ws=buf,
                     st=st)
            sk = ops.pick_splitk(hd, hd, M)
            buf = ws.bytes("gemm_ws", ops.gemm_ws_bytes(hd, hd, sk)) if sk > 1 else None
            ops.gemm(1, 0, hd, hd, M, lefts[g].data_ptr(), hd, dgs[g].data_ptr(), hd, dWh[g].data_ptr(), hd, splitk=sk,
                     ws=buf, st=st)
            ops.colsum(dgs[g].data_ptr(), hd, M, hd, dbg[g], cs, st)
        for g in range(3):
            last = g == 2
            ops.gemm(0, 1, M, xs, hd, dgs[g].data_ptr(), hd, Wx[g].data_ptr(), hd, dx.data_ptr(), xs,
                     mask_ptr=x_mask.data_ptr() if (last and x_mask is not None) else 0, ldmask=xs,
                     accumulate=(g > 0), st=st)

    # ---- the recurrent part of the BPTT unroll (updater.py:161-166), time-major (T,R,.) buffers
    def _tm_bufs(self, ws, R, T):
        h = self.h_size
        return {k: ws.get("tm_" + k, (T, R, s)) for k, s in
                dict(gx=3 * h, gh=2 * h, z=h, r=h, rh=h, rhu=h, c=h, hn=h, h_in=h).items()}

    def _bptt_cells_fwd(self, ws, x_tm, h_states, dones, R, T, st):
        h, xs = self.h_size, self.gru.x_size
        N = R * T
        Wx = self.P("gru.W_x")
        tm = self._tm_bufs(ws, R, T)
        if getattr(self, "_WxC", None) is not None and os.environ.get("A2C_NO_GRU_CAT") != "1":
            ops.gemm(0, 0, N, 3 * h, xs, x_tm.data_ptr(), xs, self._WxC.data_ptr(), 3 * h, tm["gx"].data_ptr(), 3 * h, st=st)
        else:
            for g in range(3):   # x-side gate products for all steps at once
                ops.gemm(0, 0, N, h, xs, x_tm.data_ptr(), xs, Wx[g].data_ptr(), h, tm["gx"].data_ptr() + 4 * g * h, 3 * h,
                         st=st)
        ops.copy_rows(h_states.data_ptr(), T * h, tm["h_in"][0].data_ptr(), h, R, h, st)   # hs = h_states[:,0]
        for t in range(T):
            bufs = {k: tm[k][t] for k in ("gx", "gh", "z", "r", "rh", "rhu", "c", "hn")}
            self._gru_fwd(ws, None, tm["h_in"][t], R, st, bufs)
            if t + 1 < T:     # hs = hs * (1 - dones[:, t])
                nxt = tm["h_in"][t + 1]
                ops.copy_rows(tm["hn"][t].data_ptr(), h, nxt.data_ptr(), h, R, h, st)
                ops.mask_rows(nxt, dones.data_ptr() + 4 * t, T, st)
        return tm

    def _bptt_cells_bwd(self, ws, x_tm, dhn_tm, dones, R, T, st, x_mask):
        """dhn_tm (T,R,h): gradient wrt every step's h_new coming from the heads.  Returns the
        gradient wrt x_tm (time-major) and fills the GRU parameter gradients."""
        h = self.h_size
        N = R * T
        tm = self._tm_bufs(ws, R, T)
        dtm = {k: ws.get("tm_" + k, (T, R, h)) for k in ("dz_pre", "dr_pre", "dc_pre")}
        scratch = {k: ws.get("bp_" + k, (R, h)) for k in ("dz", "d_rh", "dh")}
        dh_alt = ws.get("bp_dh2", (R, h))            # the fused step reads the carry while it writes dh: two buffers, in turn
        carry = None
        fused = os.environ.get("A2C_NO_GRU_CARRY") != "1"
        for t in range(T - 1, -1, -1):
            dhn = dhn_tm[t]
            if carry is not None and not fused:
                ops.add(dhn, carry, dhn, st)
            bufs = {k: tm[k][t] for k in ("z", "r", "c")}
            dbufs = dict(dz_pre=dtm["dz_pre"][t], dr_pre=dtm["dr_pre"][t], dc_pre=dtm["dc_pre"][t], **scratch)
            if carry is not None and carry.data_ptr() == dbufs["dh"].data_ptr():
                dbufs["dh"] = dh_alt
            # h_in[t+1] = hn[t] * (1 - dones[:, t]): the gradient that reached it comes back masked by the same factor --
            # inside the first launch of this step (was: mask_rows + add, two launches of 3.8 us per time step)
            dh = self._gru_bwd_step(ws, dhn, tm["h_in"][t], R, st, bufs, dbufs,
                                    carry=(carry, dones.data_ptr() + 4 * t, T) if (carry is not None and fused) else None)
            if t > 0:
                if not fused:
                    ops.mask_rows(dh, dones.data_ptr() + 4 * (t - 1), T, st)
                carry = dh
        dx_tm = ws.get("dx_tm", (T, R, self.gru.x_size))
        flat = lambda t_: t_.view(N, -1)
        self._gru_bwd_weights(ws, flat(x_tm), flat(tm["h_in"]), flat(tm["rh"]), flat(dtm["dz_pre"]),
                              flat(dtm["dr_pre"]), flat(dtm["dc_pre"]), flat(dx_tm), N, st,
                              x_mask=None if x_mask is None else flat(x_mask))
        return dx_tm

    def _cell_bufs(self, ws, B, pre=""):
        hd = self.h_size
        shapes = dict(gx=(B, 3 * hd), gh=(B, 2 * hd), z=(B, hd), r=(B, hd), rh=(B, hd), rhu=(B, hd), c=(B, hd),
                      hn=(B, hd))
        return {k: ws.get(pre + k, s) for k, s in shapes.items()}

    def _cell_dbufs(self, ws, B, pre=""):
        hd = self.h_size
        return {k: ws.get(pre + k, (B, hd)) for k in ("dz_pre", "dr_pre", "dc_pre", "dz", "d_rh", "dh")}


class _LNValueMixin:
    """value_out = LayerNorm -> Linear(h,1) -> Linear(1,1) (models.py:392-394)."""

    def _value_fwd(self, ws, feat, B, st, vals_col):
        P, h = self.P, self.h_size
        ln, mean, rstd, v1 = ws.get("ln", (B, h)), ws.get("ln_mean", (B,)), ws.get("ln_rstd", (B,)), ws.get("v1", (B, 1))
        ops.layernorm_fwd(feat, P("value_out.0.weight"), P("value_out.0.bias"), ln, mean, rstd, st)
        linear_fwd(ws, ln.data_ptr(), h, P("value_out.1.weight"), P("value_out.1.bias"), v1, B, st)
        linear_fwd(ws, v1.data_ptr(), 1, P("value_out.2.weight"), P("value_out.2.bias"), vals_col, B, st)

    def _value_bwd(self, ws, feat, dvm, dfeat, B, st):
        """dvm (B,1) strided column of the dheads buffer; writes dfeat (overwrite)."""
        P, G, h = self.P, self.G, self.h_size
        ln, mean, rstd, v1 = ws.get("ln", (B, h)), ws.get("ln_mean", (B,)), ws.get("ln_rstd", (B,)), ws.get("v1", (B, 1))
        linear_bwd_weight(ws, dvm, v1.data_ptr(), 1, G("value_out.2.weight"), G("value_out.2.bias"), B, st)
        dv1 = ws.get("dv1", (B, 1))
        linear_bwd_data(ws, dvm, P("value_out.2.weight"), dv1, B, st)
        linear_bwd_weight(ws, dv1, ln.data_ptr(), h, G("value_out.1.weight"), G("value_out.1.bias"), B, st)
        dln = ws.get("dln", (B, h))
        linear_bwd_data(ws, dv1, P("value_out.1.weight"), dln, B, st)
        dwr = ws.get("ln_dw_rows", (B, h))
        ops.layernorm_bwd(dln, feat, P("value_out.0.weight"), mean, rstd, dfeat, dwr, accumulate=False, st=st)
        cs = ws.bytes("colsum_ws", ops.colsum_ws_bytes(h))
        ops.colsum(dwr.data_ptr(), h, B, h, G("value_out.0.weight"), cs, st)
        ops.colsum(dln.data_ptr(), h, B, h, G("value_out.0.bias"), cs, st)


# ====================================================================== GRUModel
class GRUModel(_ConvStackNet, _GruMixin):
    """models.py:546-713."""
    _packed_after = ("value.weight", "value.bias")
    SPECS = [(16, 3, 1, 1), (24, 3, 2, 1), (32, 3, 2, 1), (48, 3, 2, 1), (64, 3, 2, 1)]

    def __init__(self, input_space, output_space, h_size=288, bnorm=False, is_discrete=True, **kwargs):
        super().__init__()
        if bnorm or not is_discrete:
            raise NotImplementedError("a2c_amd: bnorm / continuous actions are out of scope (see DESIGN.md)")
        self.is_recurrent = True
        self.input_space, self.output_space, self.h_size = input_space, output_space, h_size
        self.bnorm, self.is_discrete = False, True
        self._build_convs(input_space, self.SPECS)
        print("Flat Features Size:", self.flat_size)
        self.resize_emb = nn.Sequential(nn.Linear(self.flat_size, h_size), nn.ReLU())
        self.gru = GRU(x_size=h_size, h_size=h_size)
        self.pi = nn.Linear(h_size, output_space)
        self.value = nn.Linear(h_size, 1)
        if h_size % 4:
            raise ValueError("a2c_amd GRUModel: h_size must be a multiple of 4")
        self._post_init()

    def _arena_order(self):
        names = []
        for i in range(5):
            names += [f"convs.{i}.0.weight", f"convs.{i}.0.bias"]
        return names + ["resize_emb.0.weight", "resize_emb.0.bias", "gru.W_x", "gru.W_h", "gru.b", "pi.weight",
                        "value.weight", "pi.bias", "value.bias"]

    def _head_w(self, buf):
        A, h = self.output_space, self.h_size
        ar = self._arena
        return (buf[ar.offsets["pi.weight"][0]:][:(A + 1) * h].view(A + 1, h), buf[ar.offsets["pi.bias"][0]:][:A + 1])

    def _embed_fwd(self, x_ptr, bstride, B, ws, st, train=False, stash=None):
        acts = self._convs_train(x_ptr, bstride, B, ws, st) if train else self._convs_fwd(x_ptr, bstride, B, ws, st, stash=stash)
        if train and self._e_stashed(x_ptr, B):
            return ws.get("e", (B, self.h_size))     # left here, row by row, by the rollout
        e = self._e_out(ws, B, None if train else stash)
        linear_fwd(ws, acts[-1][0], acts[-1][1], self.P("resize_emb.0.weight"), self.P("resize_emb.0.bias"), e, B, st,
                   relu=True)
        return e

    def _E_STASH(self):
        return self.h_size

    def _cell_stash_step(self, stash, B):
        """time step whose cells a rollout step of B envs may write into the update's time-major buffers, or None"""
        T = getattr(self, "_cells_T", None)
        if T is None or stash is None or not getattr(self, "_cell_stash_ok", True) or not h_is_lockstep(stash, B, self._cells_R, T):
            return None
        return stash[1]

    def _roll_outputs(self, stash, B):
        """(vals_ptr, vals_ld, h_new_ptr) a rollout step with this stash tuple leaves behind, or None: the runner's next
        segment reads the values / hidden rows from there (pointers are a pure function of the step: graph-safe)"""
        t = self._cell_stash_step(stash, B)
        if t is None:
            return None
        A, h = self.output_space, self.h_size
        hb = self._heads("train", self._cells_R * self._cells_T)[0]
        tm = self._tm_bufs(self.ws("train"), self._cells_R, self._cells_T)
        return (hb.data_ptr() + 4 * (stash[1] * (A + 1) + A), stash[2] * (A + 1), tm["hn"][t].data_ptr())

    def _cells_stashed(self, states, R, T):
        return (getattr(self, "_cells_T", None) == T and getattr(self, "_cells_R", None) == R and
                getattr(self, "_cells_done", -1) == T - 1 and self._stash_valid(states.data_ptr(), R * T))

    def _embed_bwd(self, x_ptr, bstride, B, ws, st, de):
        """de: gradient wrt e (ReLU mask already applied)."""
        last = self._cl[-1]
        a_last = ws.get(f"a{len(self._cl)-1}", (B,) + last.out_shape)
        linear_bwd_weight(ws, de, a_last.data_ptr(), self.flat_size, self.G("resize_emb.0.weight"),
                          self.G("resize_emb.0.bias"), B, st)
        dlast = ws.get(f"da{len(self._cl)-1}", (B,) + last.out_shape)
        linear_bwd_data(ws, de, self.P("resize_emb.0.weight"), dlast.view(B, -1), B, st, mask=a_last)
        self._convs_bwd(x_ptr, bstride, B, ws, st, dlast)

    @property
    def _fused_sampling(self):
        return self.output_space + 1 <= 8 and os.environ.get("A2C_NO_FUSED_TAIL") != "1"

    def _fwd(self, x_ptr, bstride, B, tag, st, save, h_in, stash=None, sampler=None):
        ws = self.ws(tag)
        e = self._embed_fwd(x_ptr, bstride, B, ws, st, train=(save and tag == "train"), stash=stash)
        bufs = self._cell_bufs(ws, B)
        t_cell = self._cell_stash_step(stash, B) if (sampler is not None and not save) else None
        if t_cell is not None:
            # this step of ALL slots = row t of the update's time-major buffers (see stash_rows)
            tm = self._tm_bufs(self.ws("train"), self._cells_R, self._cells_T)
            bufs = dict(bufs, **{k: tm[k][t_cell] for k in ("z", "r", "rh", "c", "hn")})
            self._gru_fwd(ws, e, h_in, B, st, bufs)
            hb_t = self._heads("train", self._cells_R * self._cells_T)[0][stash[1]::stash[2]][:B]
            Wh, bh = self._head_w(self._arena.params)
            u, a_ptr, a_stride, pub = _sampler(sampler)
            ops.heads_fused(bufs["hn"].data_ptr(), 1, 0, self.h_size, None, False, None, Wh, bh, hb_t, B, u, self.output_space,
                            a_ptr, a_stride, st, publish=pub)
            return dict(logits=hb_t[:, :self.output_space], vals=hb_t[:, self.output_space], h=bufs["hn"], sampled=True,
                        h_next_src=bufs["hn"], published=pub is not None)
        if save or not h_in.is_contiguous():       # the backward pass re-reads h_in from the workspace
            hin = ws.get("h_in", (B, self.h_size))
            hin.copy_(h_in)
        else:                                      # rollout / eval: the caller's rows are read in place ...
            hin = h_in
            if sampler is not None:                # ... and UPDATED in place by an action step (gru_out is elementwise
                bufs = dict(bufs, hn=h_in)         # in h and nothing reads the old h after it); the bootstrap forward
                                                   # (no sampler) must leave the hidden state alone (runner.py:236-245)
        self._gru_fwd(ws, e, hin, B, st, bufs)
        hb, logits, vals = self._heads(tag, B)
        Wh, bh = self._head_w(self._arena.params)
        if sampler is not None and not save and self._fused_sampling:      # rollout step: heads + sampling in one node
            u, a_ptr, a_stride, pub = _sampler(sampler)
            ops.heads_fused(bufs["hn"].data_ptr(), 1, 0, self.h_size, None, False, None, Wh, bh, hb, B, u, self.output_space,
                            a_ptr, a_stride, st, publish=pub)
            return dict(logits=logits, vals=vals, h=bufs["hn"], sampled=True, published=pub is not None)
        linear_fwd(ws, bufs["hn"].data_ptr(), self.h_size, Wh, bh, hb, B, st)
        return dict(logits=logits, vals=vals, h=bufs["hn"])

    def _bwd(self, x_ptr, bstride, B, tag, st, dh_next=None):
        ws = self.ws(tag)
        h = self.h_size
        bufs, dbufs = self._cell_bufs(ws, B), self._cell_dbufs(ws, B)
        hin, e = ws.get("h_in", (B, h)), ws.get("e", (B, h))
        db, dl, dv = self.dheads(tag, B)
        dWh, dbh = self._head_w(self._arena.grads)
        linear_bwd_weight(ws, db, bufs["hn"].data_ptr(), h, dWh, dbh, B, st)
        dhn = ws.get("dhn", (B, h))
        Wh, _ = self._head_w(self._arena.params)
        linear_bwd_data(ws, db, Wh, dhn, B, st)
        if dh_next is not None:
            ops.add(dhn, dh_next, dhn, st)
        dh = self._gru_bwd_step(ws, dhn, hin, B, st, bufs, dbufs)
        de = ws.get("de", (B, h))
        self._gru_bwd_weights(ws, e, hin, bufs["rh"], dbufs["dz_pre"], dbufs["dr_pre"], dbufs["dc_pre"], de, B, st,
                              x_mask=e)
        self._embed_bwd(x_ptr, bstride, B, ws, st, de)
        return dh

    # ---- BPTT unroll (Updater.bptt, updater.py:139-169): R slots x T steps, time-major scratch
    def bptt_forward(self, states, h_states, dones, R, T, tag, st):
        ws, h, A = self.ws(tag), self.h_size, self.output_space
        N = R * T
        self._refresh(st)
        e = self._embed_fwd(states.data_ptr(), states[0].numel(), N, ws, st, train=True)          # rollout-major (N,h)
        e_tm = ws.get("e_tm", (T, R, h))
        ops.permute_rows(e, e_tm, R, T, h, st)
        if tag == "train" and self._cells_stashed(states, R, T):
            # the rollout left z, r, rh, c, hn of every step in the time-major buffers and [logits | value] of every state
            # in the heads rows; h_in[t] = the masked hidden state the rollout recorded in h_states (runner.py:201)
            tm = self._tm_bufs(ws, R, T)
            ops.permute_rows(h_states, tm["h_in"].view(N, h), R, T, h, st)
            hb, logits, vals = self._heads(tag, N)
            return vals, logits
        tm = self._bptt_cells_fwd(ws, e_tm, h_states, dones, R, T, st)
        heads_tm = ws.get("heads_tm", (N, A + 1))
        Wh, bh = self._head_w(self._arena.params)
        linear_fwd(ws, tm["hn"].data_ptr(), h, Wh, bh, heads_tm, N, st)
        hb, logits, vals = self._heads(tag, N)
        ops.permute_rows(heads_tm, hb, T, R, A + 1, st)          # back to rollout-major
        return vals, logits

    def bptt_backward(self, states, dones, R, T, tag, st):
        ws, h, A = self.ws(tag), self.h_size, self.output_space
        N = R * T
        e_tm = ws.get("e_tm", (T, R, h))
        hn_tm = ws.get("tm_hn", (T, R, h))
        db, dl, dv = self.dheads(tag, N)
        db_tm = ws.get("dheads_tm", (N, A + 1))
        ops.permute_rows(db, db_tm, R, T, A + 1, st)
        dWh, dbh = self._head_w(self._arena.grads)
        linear_bwd_weight(ws, db_tm, hn_tm.data_ptr(), h, dWh, dbh, N, st)
        dhn_tm = ws.get("dhn_tm", (T, R, h))
        Wh, _ = self._head_w(self._arena.params)
        linear_bwd_data(ws, db_tm, Wh, dhn_tm.view(N, h), N, st)
        de_tm = self._bptt_cells_bwd(ws, e_tm, dhn_tm, dones, R, T, st, x_mask=e_tm)
        de = ws.get("de", (N, h))
        ops.permute_rows(de_tm, de, T, R, h, st)
        self._embed_bwd(states.data_ptr(), states[0].numel(), N, ws, st, de)


# ====================================================================== FCModel / GRUFCModel
class _FCBase(_HipNet, _LNValueMixin):
    def _build_base(self, input_shape, output_space, h_size):
        self.flat_size = int(np.prod(input_shape[-3:]))
        self.input_shape, self.output_space, self.h_size = input_shape, output_space, h_size
        self.base = nn.Sequential(nn.Linear(self.flat_size, h_size), nn.ReLU(), nn.Linear(h_size, h_size))

    def _build_heads(self, output_space, h_size):
        self.action_out = nn.Linear(h_size, output_space)
        self.value_out = nn.Sequential(nn.LayerNorm(h_size), nn.Linear(h_size, 1), nn.Linear(1, 1))

    def _base_fwd(self, x_ptr, bstride, B, ws, st):
        P, h = self.P, self.h_size
        t1, fx = ws.get("t1", (B, h)), ws.get("fx", (B, h))
        linear_fwd(ws, x_ptr, bstride, P("base.0.weight"), P("base.0.bias"), t1, B, st, relu=True)
        linear_fwd(ws, t1.data_ptr(), h, P("base.2.weight"), P("base.2.bias"), fx, B, st)
        return fx

    def _base_bwd(self, x_ptr, bstride, B, ws, st, dfx):
        P, G, h = self.P, self.G, self.h_size
        t1 = ws.get("t1", (B, h))
        linear_bwd_weight(ws, dfx, t1.data_ptr(), h, G("base.2.weight"), G("base.2.bias"), B, st)
        dt1 = ws.get("dt1", (B, h))
        linear_bwd_data(ws, dfx, P("base.2.weight"), dt1, B, st, mask=t1)
        linear_bwd_weight(ws, dt1, x_ptr, bstride, G("base.0.weight"), G("base.0.bias"), B, st)


class FCModel(_FCBase):
    """models.py:367-424."""

    def __init__(self, input_shape, output_space, h_size=200, bnorm=False, is_discrete=True, **kwargs):
        super().__init__()
        if bnorm or not is_discrete:
            raise NotImplementedError("a2c_amd: bnorm / continuous actions are out of scope (see DESIGN.md)")
        self.is_discrete, self.is_recurrent = True, False
        self._build_base(input_shape, output_space, h_size)
        self._build_heads(output_space, h_size)
        self._post_init()

    def _fwd(self, x_ptr, bstride, B, tag, st, save):
        ws, P, h = self.ws(tag), self.P, self.h_size
        fx = self._base_fwd(x_ptr, bstride, B, ws, st)
        hb, logits, vals = self._heads(tag, B)
        linear_fwd(ws, fx.data_ptr(), h, P("action_out.weight"), P("action_out.bias"), logits, B, st)
        self._value_fwd(ws, fx, B, st, hb[:, self.output_space:])
        return dict(logits=logits, vals=vals)

    def _bwd(self, x_ptr, bstride, B, tag, st):
        ws, P, G, h, A = self.ws(tag), self.P, self.G, self.h_size, self.output_space
        fx = ws.get("fx", (B, h))
        db, dl, dv = self.dheads(tag, B)
        dfx = ws.get("dfx", (B, h))
        self._value_bwd(ws, fx, db[:, A:], dfx, B, st)
        linear_bwd_weight(ws, dl, fx.data_ptr(), h, G("action_out.weight"), G("action_out.bias"), B, st)
        ops.gemm(0, 0, B, h, A, dl.data_ptr(), A + 1, P("action_out.weight").data_ptr(), h, dfx.data_ptr(), h,
                 accumulate=True, st=st)
        self._base_bwd(x_ptr, bstride, B, ws, st, dfx)


class GRUFCModel(_FCBase, _GruMixin):
    """models.py:483-543."""

    def _prep(self, st):
        self._gru_prep(st)

    def __init__(self, input_shape, output_space, h_size=200, bnorm=False, is_discrete=True, **kwargs):
        super().__init__()
        if bnorm or not is_discrete:
            raise NotImplementedError("a2c_amd: bnorm / continuous actions are out of scope (see DESIGN.md)")
        self.is_discrete, self.is_recurrent = True, True
        self._build_base(input_shape, output_space, h_size)
        self.gru = GRU(x_size=h_size, h_size=h_size)
        self._build_heads(output_space, h_size)
        self._post_init()

    def _heads_fwd(self, ws, hn, B, hb, st):
        A = self.output_space
        linear_fwd(ws, hn.data_ptr(), self.h_size, self.P("action_out.weight"), self.P("action_out.bias"), hb[:, :A], B,
                   st)
        self._value_fwd(ws, hn, B, st, hb[:, A:])

    def _heads_bwd(self, ws, hn, B, db, dhn, st):
        A, h = self.output_space, self.h_size
        self._value_bwd(ws, hn, db[:, A:], dhn, B, st)
        dl = db[:, :A]
        linear_bwd_weight(ws, dl, hn.data_ptr(), h, self.G("action_out.weight"), self.G("action_out.bias"), B, st)
        ops.gemm(0, 0, B, h, A, dl.data_ptr(), A + 1, self.P("action_out.weight").data_ptr(), h, dhn.data_ptr(), h,
                 accumulate=True, st=st)

    def _fwd(self, x_ptr, bstride, B, tag, st, save, h_in):
        ws = self.ws(tag)
        fx = self._base_fwd(x_ptr, bstride, B, ws, st)
        bufs = self._cell_bufs(ws, B)
        hin = ws.get("h_in", (B, self.h_size))
        hin.copy_(h_in)
        self._gru_fwd(ws, fx, hin, B, st, bufs)
        hb, logits, vals = self._heads(tag, B)
        self._heads_fwd(ws, bufs["hn"], B, hb, st)
        return dict(logits=logits, vals=vals, h=bufs["hn"])

    def _bwd(self, x_ptr, bstride, B, tag, st, dh_next=None):
        ws, h = self.ws(tag), self.h_size
        bufs, dbufs = self._cell_bufs(ws, B), self._cell_dbufs(ws, B)
        hin, fx = ws.get("h_in", (B, h)), ws.get("fx", (B, h))
        dhn = ws.get("dhn", (B, h))
        self._heads_bwd(ws, bufs["hn"], B, self.dheads(tag, B)[0], dhn, st)
        if dh_next is not None:
            ops.add(dhn, dh_next, dhn, st)
        dh = self._gru_bwd_step(ws, dhn, hin, B, st, bufs, dbufs)
        dfx = ws.get("dfx", (B, h))
        self._gru_bwd_weights(ws, fx, hin, bufs["rh"], dbufs["dz_pre"], dbufs["dr_pre"], dbufs["dc_pre"], dfx, B, st)
        self._base_bwd(x_ptr, bstride, B, ws, st, dfx)
        return dh

    # BPTT (updater.py:139-169): base MLP batched rollout-major, GRU cells + heads time-major
    def bptt_forward(self, states, h_states, dones, R, T, tag, st):
        ws, h, A = self.ws(tag), self.h_size, self.output_space
        N = R * T
        self._refresh(st)
        fx = self._base_fwd(states.data_ptr(), states[0].numel(), N, ws, st)
        fx_tm = ws.get("fx_tm", (T, R, h))
        ops.permute_rows(fx, fx_tm, R, T, h, st)
        tm = self._bptt_cells_fwd(ws, fx_tm, h_states, dones, R, T, st)
        heads_tm = ws.get("heads_tm", (N, A + 1))
        self._heads_fwd(ws, tm["hn"].view(N, h), N, heads_tm, st)
        hb, logits, vals = self._heads(tag, N)
        ops.permute_rows(heads_tm, hb, T, R, A + 1, st)
        return vals, logits

    def bptt_backward(self, states, dones, R, T, tag, st):
        ws, h, A = self.ws(tag), self.h_size, self.output_space
        N = R * T
        fx_tm = ws.get("fx_tm", (T, R, h))
        hn_tm = ws.get("tm_hn", (T, R, h))
        db, dl, dv = self.dheads(tag, N)
        db_tm = ws.get("dheads_tm", (N, A + 1))
        ops.permute_rows(db, db_tm, R, T, A + 1, st)
        dhn_tm = ws.get("dhn_tm", (T, R, h))
        self._heads_bwd(ws, hn_tm.view(N, h), N, db_tm, dhn_tm.view(N, h), st)
        dfx_tm = self._bptt_cells_bwd(ws, fx_tm, dhn_tm, dones, R, T, st, x_mask=None)
        dfx = ws.get("dfx", (N, h))
        ops.permute_rows(dfx_tm, dfx, T, R, h, st)
        self._base_bwd(states.data_ptr(), states[0].numel(), N, ws, st, dfx)
