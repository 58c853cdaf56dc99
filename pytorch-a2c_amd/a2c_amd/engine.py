"""Host-side building blocks shared by the model classes: the flat parameter arena, conv /
linear layer launch helpers with their backward passes.  Everything here only enqueues HIP
kernels through ``ops`` (the C ABI); no arithmetic happens on the host."""
import torch

from . import ops

ALIGN = 4  # floats: every tensor starts 16 B aligned in the arena


class Arena:
    """All parameters of a net in ONE contiguous fp32 device buffer (+ a parallel gradient
    buffer).  Parameters that receive gradients come first (``[0, n_train)``): that prefix is
    what clip+optimiser kernels sweep and what RCCL all-reduces.  ``nn.Parameter.data`` of every
    parameter is re-pointed to a view of the arena, so ``state_dict`` / ``load_state_dict`` /
    checkpoints keep working unchanged."""

    def __init__(self, named_params, trainable_names, device, packed_after=()):
        """``packed_after``: names that must start IMMEDIATELY after the previous tensor (no
        alignment padding), so that e.g. [pi.bias | value.bias] is one contiguous vector."""
        train = [(n, p) for n, p in named_params if n in trainable_names]
        rest = [(n, p) for n, p in named_params if n not in trainable_names]
        self.offsets = {}
        off = 0
        end = 0                      # end of the previous tensor (unpadded)
        for group in (train, rest):
            for n, p in group:
                if n in packed_after:
                    off = end
                self.offsets[n] = (off, p.numel(), tuple(p.shape))
                end = off + p.numel()
                off = (end + ALIGN - 1) // ALIGN * ALIGN
            if group is train:
                self.n_train = off
                # ALIGN floats right behind the trainable prefix that travel WITH the gradient all-reduce of a sharded
                # update (the three loss sums the reference reports, updater.py:134-136): one collective fewer
                off += ALIGN
                end = off
        self.size = off
        self.params = torch.zeros(off, dtype=torch.float32, device=device)
        self.grads = torch.zeros(off, dtype=torch.float32, device=device)
        self.trainable = [n for n, _ in train]
        with torch.no_grad():
            for n, p in train + rest:
                o, k, shp = self.offsets[n]
                view = self.params[o:o + k].view(shp)
                view.copy_(p.data)
                p.data = view
        self._named = dict(train + rest)
        self.attach_grads()

    def attach_grads(self):
        """Point ``p.grad`` of every trainable parameter at its slice of the gradient arena."""
        for n in self.trainable:
            o, k, shp = self.offsets[n]
            self._named[n].grad = self.grads[o:o + k].view(shp)

    def grad(self, name):
        o, k, shp = self.offsets[name]
        return self.grads[o:o + k].view(shp)

    def train_params(self):
        return self.params[:self.n_train]

    def train_grads(self):
        return self.grads[:self.n_train]

    def reduce_span(self):
        """what one sharded update all-reduces: the gradients + the tail slot"""
        return self.grads[:self.n_train + ALIGN]

    def grad_tail(self):
        return self.grads[self.n_train:self.n_train + ALIGN]


class ConvLayer:
    """One nn.Conv2d(+ReLU) block of a model: descriptor, matrix-core weight fragments.

    The kernels take input planes in quads (4 planes = one MFMA k-step).  A layer whose Cin is not a
    multiple of 4 (the reference's shipped hyperparams stack 3 frames) runs on a zero-padded copy:
    the input rows are copied into a (B, Cin_pad, H, W) buffer whose extra planes stay 0, the weights
    into a (Cout, Cin_pad, ks, ks) buffer whose extra slices stay 0, and the weight gradient is
    computed padded and sliced back.  Correct, not tuned: the tuned paths are for Cin % 4 == 0."""

    def __init__(self, Cin, H, W, Cout, ks, stride, pad, need_bwd_data, name="conv"):
        self.name = name
        self.cin, self.cin_pad = Cin, (Cin + 3) // 4 * 4
        self.padded = self.cin_pad != Cin
        self.d = ops.conv_desc(self.cin_pad, H, W, Cout, ks, stride, pad)
        self.need_bwd_data = need_bwd_data
        self.out_shape = (Cout, self.d.OH, self.d.OW)
        self.hw = H * W
        self.in_floats = Cin * H * W
        self.wf = self.wb = None
        self._xpad, self._wpad, self._dwpad = {}, None, None

    def _input(self, in_ptr, in_bstride, B, dev, st):
        if not self.padded:
            return in_ptr, in_bstride
        buf = self._xpad.get(B)
        if buf is None:
            buf = self._xpad[B] = torch.zeros(B, self.cin_pad * self.hw, dtype=torch.float32, device=dev)
        ops.copy_rows(in_ptr, in_bstride, buf.data_ptr(), self.cin_pad * self.hw, B, self.cin * self.hw, st)
        return buf.data_ptr(), self.cin_pad * self.hw

    def prep(self, weight, st):
        if self.wf is None:
            self.wf = torch.empty(ops.conv_prep_floats(self.d, 0), dtype=torch.float32, device=weight.device)
            if self.need_bwd_data:
                self.wb = torch.empty(ops.conv_prep_floats(self.d, 1), dtype=torch.float32, device=weight.device)
        if self.padded:
            if self._wpad is None:
                self._wpad = torch.zeros(weight.shape[0], self.cin_pad, *weight.shape[2:], device=weight.device)
            self._wpad[:, :self.cin].copy_(weight.detach())
            weight = self._wpad
        ops.conv_prep(self.d, 0, weight, self.wf, st)
        if self.need_bwd_data:
            ops.conv_prep(self.d, 1, weight, self.wb, st)

    def fwd(self, in_ptr, in_bstride, bias, out, B, st, relu=True, out_bstride=None, signs=None):
        """out: (B, Cout, OH, OW) tensor, or a raw device address + out_bstride (rows of a larger buffer).
        signs = (device address, row stride in words): also leave the sign words of the output there (sign_words > 0)"""
        in_ptr, in_bstride = self._input(in_ptr, in_bstride, B, bias.device, st)
        with ops.span(self.name + ".fwd"):
            optr = out if isinstance(out, int) else out.data_ptr()
            obs = self.d.Cout * self.d.OH * self.d.OW if out_bstride is None else out_bstride
            if signs is not None and in_bstride % 4 == 0 and in_ptr % 16 == 0 and obs % 4 == 0 and optr % 16 == 0:
                ops.conv_fwd_signs(self.d, in_ptr, in_bstride, self.wf, bias, relu, out, signs[0], signs[1], B, st,
                                   out_bstride=out_bstride)
                return True
            ops.conv_fwd(self.d, in_ptr, in_bstride, self.wf, bias, relu, out, B, st, out_bstride=out_bstride)
        return False

    def fwd_frames(self, src, bias, out, B, st, relu=True, out_bstride=None, signs=None):
        """first layer with its input STACKED ON LOAD from the single-frame uint8 store:
        src = (address of sample 0's window, sample stride in bytes, T, address of its valid-plane count, count stride),
        see a2c_conv2d_fwd_frames.  -> True when the sign words were written."""
        fptr, fstride, T, nv_ptr, nv_stride = src
        with ops.span(self.name + ".fwd"):
            ops.conv_fwd_frames(self.d, fptr, fstride, T, nv_ptr, nv_stride, self.wf, bias, relu, out, B, st,
                                out_bstride=out_bstride, signs=signs)
        return signs is not None

    @property
    def frames_ok(self):
        """this layer can read a single-frame uint8 store (forward and weight gradient)"""
        if not hasattr(self, "_fok"):
            self._fok = (not self.padded) and ops.conv_fwd_frames_supported(self.d)
        return self._fok

    @property
    def sign_words(self):
        """uint32 words per sample of the output's sign-word image; 0: this layer's forward cannot write one"""
        if self.padded:
            return 0
        if not hasattr(self, "_sw"):
            self._sw = ops.conv_sign_words(self.d)
        return self._sw

    @property
    def bwd_reads_signs(self):
        if not hasattr(self, "_brs"):
            # stride 2 only: measured (tools/conv3_check.py, N = 4096) the staged sign-word kernels beat the float-mask
            # ones by 12-41 % on the stride-2 layers and lose 15 % on the 84-wide stride-1 layer (16 <- 24)
            self._brs = ((not self.padded) and self.need_bwd_data and self.d.stride == 2
                         and ops.conv_bwd_data_signs_supported(self.d))
        return self._brs

    def bwd_weight(self, in_ptr, in_bstride, dout, dW, db, B, ws, st):
        in_ptr, in_bstride = self._input(in_ptr, in_bstride, B, dout.device, st)
        buf = ws.bytes("conv_wgrad_ws", ops.conv_bwd_weight_ws_bytes(self.d, B))
        dst = dW
        if self.padded:
            if self._dwpad is None:
                self._dwpad = torch.empty(dW.shape[0], self.cin_pad, *dW.shape[2:], device=dW.device)
            dst = self._dwpad
        with ops.span(self.name + ".bwd_weight"):
            ops.conv_bwd_weight(self.d, in_ptr, in_bstride, dout, dst, db, B, buf, st)
        if self.padded:
            dW.copy_(dst[:, :self.cin])

    def bwd_data(self, dout, mask, din, B, st, signs=None, lanemask=None):
        """mask: the float activation below; signs: its sign words (B, words) when its forward left them (preferred);
        lanemask: its lane masks (B, n/64) int64 (A3CModel's conv2: written by the ring kernel beside the a1 stash)"""
        if self.padded:
            raise NotImplementedError("a2c_amd: input gradient of a channel-padded conv layer (only first layers are padded)")
        with ops.span(self.name + ".bwd_data"):
            if lanemask is not None:
                ops.conv_bwd_data_lanemask(self.d, dout, self.wb, lanemask, din, B, st)
            elif signs is not None:
                ops.conv_bwd_data_signs(self.d, dout, self.wb, signs, din, B, st)
            else:
                ops.conv_bwd_data(self.d, dout, self.wb, mask, din, B, st)


def linear_fwd(ws, x_ptr, ldx, W, b, out, M, st, relu=False):
    """out[M,N] = x[M,K] W[N,K]^T + b   (nn.Linear)"""
    N, K = W.shape
    sk = ops.pick_splitk(M, N, K)
    nb = ops.gemm_ws_bytes(M, N, sk, K)
    buf = ws.bytes("gemm_ws", nb) if nb else None
    with ops.span(f"linear.fwd {N}x{K}"):
        ops.gemm(0, 1, M, N, K, x_ptr, ldx, W.data_ptr(), K, out.data_ptr(), out.stride(0), bias=b, relu=relu,
                 splitk=sk, ws=buf, st=st)


def linear_bwd_data(ws, dy, W, dx, M, st, mask=None, n_cols=None):
    """dx[M,K] = dy[M,:n] W[:n,K]  (* (mask > 0)): gradient wrt the layer input, with the ReLU
    derivative of the producer of that input fused in."""
    N, K = W.shape
    n = N if n_cols is None else n_cols
    nb = ops.gemm_ws_bytes(M, K, 1, n)
    buf = ws.bytes("gemm_ws", nb) if nb else None
    with ops.span(f"linear.bwd_data {N}x{K}"):
        ops.gemm(0, 0, M, K, n, dy.data_ptr(), dy.stride(0), W.data_ptr(), K, dx.data_ptr(), dx.stride(0),
                 mask_ptr=0 if mask is None else mask.data_ptr(), ldmask=0 if mask is None else K, ws=buf, st=st)


def linear_bwd_weight(ws, dy, x_ptr, ldx, dW, db, M, st):
    """dW[N,K] = dy[M,N]^T x[M,K];  db[N] = sum_m dy[m,:]"""
    N, K = dW.shape
    sk = ops.pick_splitk(N, K, M)
    nb = ops.gemm_ws_bytes(N, K, sk, M)
    buf = ws.bytes("gemm_ws", nb) if nb else None
    with ops.span(f"linear.bwd_weight {N}x{K}"):
        ops.gemm(1, 0, N, K, M, dy.data_ptr(), dy.stride(0), x_ptr, ldx, dW.data_ptr(), K, splitk=sk, ws=buf, st=st)
    if db is not None:
        cs = ws.bytes("colsum_ws", ops.colsum_ws_bytes(N))
        ops.colsum(dy.data_ptr(), dy.stride(0), M, N, db, cs, st)
