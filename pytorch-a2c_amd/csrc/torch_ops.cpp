// torch.ops.a2c_mi355x.* : PyTorch-ROCm custom-op registration over the C ABI (include/a2c_mi355x.h).
// A thin shim: every op checks device / dtype / layout, takes raw device pointers and torch's current HIP
// stream, and calls the extern "C" launcher of liba2c_mi355x.so.  Only a CUDA(HIP) dispatch key is registered:
// there is no CPU kernel behind any of these ops -- a CPU tensor raises.  (the reference
// call site each op replaces is listed in INTEGRATION.md section 3.)
#include <ATen/hip/HIPContext.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>
#include <torch/types.h>

#include "../../include/a2c_mi355x.h"

namespace {
void* cur_stream() { return (void*)c10::hip::getCurrentHIPStream().stream(); }

void chk(const at::Tensor& t, const char* name, at::ScalarType dt = at::kFloat, bool contig = true) {
  TORCH_CHECK(t.is_cuda(), "a2c_mi355x: `", name, "` must be a HIP tensor (there is no CPU kernel)");
  TORCH_CHECK(t.scalar_type() == dt, "a2c_mi355x: `", name, "` has the wrong dtype");
  TORCH_CHECK(!contig || t.is_contiguous(), "a2c_mi355x: `", name, "` must be contiguous");
}
void ok(int rc, const char* what) { TORCH_CHECK(rc == A2C_OK, what, ": ", a2c_error_string(rc)); }
// zeroed scratch of the deterministic scalar reductions (one per call here: the side door allocates its outputs per call too)
at::Tensor reduce_scratch(const at::Tensor& like) { return at::zeros({A2C_REDUCE_SCRATCH_DOUBLES}, like.options().dtype(at::kDouble)); }

// utils.discount on rows (utils.py:63-79): y[i] = x[i] + g*(dones[i]==1 ? 0 : y[i+1]), bit-exact
at::Tensor discount(const at::Tensor& x, const at::Tensor& dones, double g, int64_t n_seg) {
  chk(x, "x"); chk(dones, "dones");
  TORCH_CHECK(x.numel() == dones.numel() && n_seg >= 1 && x.numel() % n_seg == 0, "a2c_mi355x::discount: bad shapes");
  at::Tensor y = at::empty_like(x);
  ok(a2c_discount_scan(x.data_ptr<float>(), dones.data_ptr<float>(), y.data_ptr<float>(), n_seg, x.numel() / n_seg, (float)g,
                       nullptr, cur_stream()), "a2c_discount_scan");
  return y;
}

// updater.py:70-71 + 86-88: (advs, returns) in one pass
std::tuple<at::Tensor, at::Tensor> gae_returns(const at::Tensor& deltas, const at::Tensor& rewards, const at::Tensor& dones,
                                               double g_adv, double g_ret, int64_t n_seg) {
  chk(deltas, "deltas"); chk(rewards, "rewards"); chk(dones, "dones");
  TORCH_CHECK(n_seg >= 1 && deltas.numel() % n_seg == 0, "a2c_mi355x::gae_returns: bad shapes");
  at::Tensor advs = at::empty_like(deltas), rets = at::empty_like(deltas);
  ok(a2c_gae_returns_fused(deltas.data_ptr<float>(), rewards.data_ptr<float>(), dones.data_ptr<float>(), advs.data_ptr<float>(),
                           rets.data_ptr<float>(), n_seg, deltas.numel() / n_seg, (float)g_adv, (float)g_ret, nullptr,
                           cur_stream()), "a2c_gae_returns_fused");
  return {advs, rets};
}

// runner.py:94-97 + utils.py:45-60: softmax + inverse-CDF sample with explicit uniforms
at::Tensor softmax_sample(const at::Tensor& logits, const at::Tensor& u) {
  chk(logits, "logits"); chk(u, "u");
  TORCH_CHECK(logits.dim() == 2 && u.numel() == logits.size(0), "a2c_mi355x::softmax_sample: logits (B,A), u (B,)");
  at::Tensor acts = at::empty({logits.size(0)}, logits.options().dtype(at::kLong));
  ok(a2c_softmax_sample(logits.data_ptr<float>(), logits.size(1), u.data_ptr<float>(), acts.data_ptr<int64_t>(), 1, nullptr,
                        (int)logits.size(0), (int)logits.size(1), cur_stream()), "a2c_softmax_sample");
  return acts;
}

// utils.next_state batched (utils.py:26-43): frames fp32 (B,HW) or uint8 (B,HW); prev (B,C,HW); reset (B,)
at::Tensor frame_stack_push(const at::Tensor& frame_new, const at::Tensor& reset_mask, const at::Tensor& prev) {
  chk(prev, "prev"); chk(reset_mask, "reset_mask");
  TORCH_CHECK(prev.dim() == 3 && frame_new.dim() == 2 && frame_new.size(0) == prev.size(0) && frame_new.size(1) == prev.size(2),
              "a2c_mi355x::frame_stack_push: prev (B,C,HW), frame_new (B,HW)");
  const int B = (int)prev.size(0), C = (int)prev.size(1), HW = (int)prev.size(2);
  at::Tensor out = at::empty_like(prev);
  if (frame_new.scalar_type() == at::kByte) {
    chk(frame_new, "frame_new", at::kByte);
    ok(a2c_frame_stack_push_u8(frame_new.data_ptr<uint8_t>(), HW, reset_mask.data_ptr<float>(), prev.data_ptr<float>(),
                               (int64_t)C * HW, out.data_ptr<float>(), (int64_t)C * HW, B, C, HW, cur_stream()),
       "a2c_frame_stack_push_u8");
  } else {
    chk(frame_new, "frame_new");
    ok(a2c_frame_stack_push(frame_new.data_ptr<float>(), reset_mask.data_ptr<float>(), prev.data_ptr<float>(), (int64_t)C * HW,
                            out.data_ptr<float>(), (int64_t)C * HW, B, C, HW, cur_stream()), "a2c_frame_stack_push");
  }
  return out;
}

// updater.py:100-106,124-128: loss sums + d/dlogits + d/dvals in one pass; returns (dlogits, dvals, sums[3] double)
std::tuple<at::Tensor, at::Tensor, at::Tensor> loss_fwd_bwd(const at::Tensor& logits, const at::Tensor& vals,
                                                            const at::Tensor& actions, const at::Tensor& advs,
                                                            const at::Tensor& returns, double pi_coef, double val_coef,
                                                            double entr_coef) {
  chk(logits, "logits"); chk(vals, "vals"); chk(actions, "actions", at::kLong); chk(advs, "advs"); chk(returns, "returns");
  const int64_t n = logits.size(0);
  at::Tensor dl = at::empty_like(logits), dv = at::empty_like(vals);
  at::Tensor sums = at::empty({3}, logits.options().dtype(at::kDouble));
  ok(a2c_loss_fwd_bwd(logits.data_ptr<float>(), logits.size(1), vals.data_ptr<float>(), 1, actions.data_ptr<int64_t>(),
                      advs.data_ptr<float>(), returns.data_ptr<float>(), nullptr, n, n, (int)logits.size(1), (float)pi_coef,
                      (float)val_coef, (float)entr_coef, dl.data_ptr<float>(), logits.size(1), dv.data_ptr<float>(), 1,
                      sums.data_ptr<double>(), reduce_scratch(logits).data_ptr<double>(), cur_stream()), "a2c_loss_fwd_bwd");
  return {dl, dv, sums};
}

// nn.Linear forward on the fp32 matrix cores: y = x W^T + b [ReLU]
at::Tensor linear(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, bool relu) {
  chk(x, "x"); chk(w, "weight"); chk(b, "bias");
  TORCH_CHECK(x.dim() == 2 && w.dim() == 2 && x.size(1) == w.size(1) && b.numel() == w.size(0), "a2c_mi355x::linear: shapes");
  at::Tensor y = at::empty({x.size(0), w.size(0)}, x.options());
  ok(a2c_gemm_f32_nt(x.size(0), w.size(0), x.size(1), x.data_ptr<float>(), x.size(1), w.data_ptr<float>(), w.size(1),
                     y.data_ptr<float>(), w.size(0), b.data_ptr<float>(), relu ? 1 : 0, cur_stream()), "a2c_gemm_f32_nt");
  return y;
}

// clip_grad_norm_ + RMSprop step over flat arenas (updater.py:129-132), in place; returns the pre-clip norm
at::Tensor clip_rmsprop_(at::Tensor params, at::Tensor grads, at::Tensor square_avg, double max_norm, double lr, double alpha,
                         double eps) {
  chk(params, "params"); chk(grads, "grads"); chk(square_avg, "square_avg");
  at::Tensor sumsq = at::empty({1}, params.options().dtype(at::kDouble)), norm = at::empty({1}, params.options());
  ok(a2c_gradnorm_sq(grads.data_ptr<float>(), grads.numel(), sumsq.data_ptr<double>(), reduce_scratch(grads).data_ptr<double>(), cur_stream()),
     "a2c_gradnorm_sq");
  ok(a2c_clip_rmsprop(params.data_ptr<float>(), grads.data_ptr<float>(), square_avg.data_ptr<float>(), params.numel(),
                      sumsq.data_ptr<double>(), max_norm, lr, alpha, eps, norm.data_ptr<float>(), cur_stream()), "a2c_clip_rmsprop");
  return norm;
}
// ---- conv layers (models.py:98,119,312): stateless tensor-in / tensor-out forms -- each call prepares the weight fragments
// (a2c_conv2d_prep_weights) itself; the hot path keeps them across calls through the C ABI (engine.ConvLayer)
a2c_conv_desc conv_desc(const at::Tensor& x, const at::Tensor& w, int64_t stride, int64_t pad) {
  TORCH_CHECK(x.dim() == 4 && w.dim() == 4 && x.size(1) == w.size(1) && w.size(2) == w.size(3), "a2c_mi355x conv: x (B,Cin,H,W), weight (Cout,Cin,k,k)");
  a2c_conv_desc d;
  d.Cin = (int)x.size(1); d.H = (int)x.size(2); d.W = (int)x.size(3);
  d.Cout = (int)w.size(0); d.ks = (int)w.size(2); d.stride = (int)stride; d.pad = (int)pad;
  d.OH = (d.H - d.ks + 2 * d.pad) / d.stride + 1; d.OW = (d.W - d.ks + 2 * d.pad) / d.stride + 1;
  TORCH_CHECK(d.Cin % 4 == 0, "a2c_mi355x conv: Cin must be a multiple of 4 (the models pad 3-frame stacks, engine.ConvLayer)");
  return d;
}
at::Tensor prep(const a2c_conv_desc& d, int kind, const at::Tensor& w) {
  at::Tensor f = at::empty({(int64_t)a2c_conv2d_prep_floats(&d, kind)}, w.options());
  ok(a2c_conv2d_prep_weights(&d, kind, w.data_ptr<float>(), f.data_ptr<float>(), cur_stream()), "a2c_conv2d_prep_weights");
  return f;
}
at::Tensor conv2d_fwd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b, int64_t stride, int64_t pad, bool relu) {
  chk(x, "x"); chk(w, "weight"); chk(b, "bias");
  const a2c_conv_desc d = conv_desc(x, w, stride, pad);
  at::Tensor f = prep(d, 0, w), out = at::empty({x.size(0), d.Cout, d.OH, d.OW}, x.options());
  ok(a2c_conv2d_fwd(&d, x.data_ptr<float>(), (int64_t)d.Cin * d.H * d.W, f.data_ptr<float>(), b.data_ptr<float>(), relu ? 1 : 0,
                    out.data_ptr<float>(), (int64_t)d.Cout * d.OH * d.OW, (int)x.size(0), cur_stream()), "a2c_conv2d_fwd");
  return out;
}
// din = conv_transpose(dout, W) * (mask > 0)   (mask = the layer's input activation, or an undefined tensor)
at::Tensor conv2d_bwd_data(const at::Tensor& dout, const at::Tensor& w, const c10::optional<at::Tensor>& mask, int64_t H, int64_t W,
                           int64_t stride, int64_t pad) {
  chk(dout, "dout"); chk(w, "weight");
  at::Tensor din = at::empty({dout.size(0), w.size(1), H, W}, dout.options());
  const a2c_conv_desc d = conv_desc(din, w, stride, pad);
  TORCH_CHECK(dout.size(1) == d.Cout && dout.size(2) == d.OH && dout.size(3) == d.OW, "a2c_mi355x::conv2d_bwd_data: dout shape");
  if (mask.has_value()) { chk(*mask, "mask"); TORCH_CHECK(mask->sizes() == din.sizes(), "mask shape"); }
  at::Tensor f = prep(d, 1, w);
  ok(a2c_conv2d_bwd_data(&d, dout.data_ptr<float>(), f.data_ptr<float>(), mask.has_value() ? mask->data_ptr<float>() : nullptr,
                         din.data_ptr<float>(), (int)dout.size(0), cur_stream()), "a2c_conv2d_bwd_data");
  return din;
}
std::tuple<at::Tensor, at::Tensor> conv2d_bwd_weight(const at::Tensor& x, const at::Tensor& dout, int64_t ks, int64_t stride, int64_t pad) {
  chk(x, "x"); chk(dout, "dout");
  at::Tensor dW = at::empty({dout.size(1), x.size(1), ks, ks}, x.options()), db = at::empty({dout.size(1)}, x.options());
  const a2c_conv_desc d = conv_desc(x, dW, stride, pad);
  TORCH_CHECK(dout.size(0) == x.size(0) && dout.size(2) == d.OH && dout.size(3) == d.OW, "a2c_mi355x::conv2d_bwd_weight: dout shape");
  const size_t nb = a2c_conv2d_bwd_weight_ws_bytes(&d, (int)x.size(0));
  at::Tensor ws = at::empty({(int64_t)(nb + 3) / 4 + 4}, x.options());
  ok(a2c_conv2d_bwd_weight(&d, x.data_ptr<float>(), (int64_t)d.Cin * d.H * d.W, dout.data_ptr<float>(), dW.data_ptr<float>(),
                           db.data_ptr<float>(), (int)x.size(0), ws.data_ptr<float>(), (size_t)ws.numel() * 4, cur_stream()),
     "a2c_conv2d_bwd_weight");
  return {dW, db};
}

// ---- dense fp32 GEMMs on the matrix cores (models.py:40-47, 472-475): C = A B, A B^T (+ bias, ReLU), A^T B
at::Tensor gemm_nn(const at::Tensor& a, const at::Tensor& b) {
  chk(a, "a"); chk(b, "b");
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.size(1) == b.size(0), "a2c_mi355x::gemm_nn: (M,K) x (K,N)");
  at::Tensor c = at::empty({a.size(0), b.size(1)}, a.options());
  ok(a2c_gemm_f32_nn(a.size(0), b.size(1), a.size(1), a.data_ptr<float>(), a.size(1), b.data_ptr<float>(), b.size(1),
                     c.data_ptr<float>(), b.size(1), nullptr, 0, cur_stream()), "a2c_gemm_f32_nn");
  return c;
}
at::Tensor gemm_tn(const at::Tensor& a, const at::Tensor& b) {      // a (K,M), b (K,N) -> (M,N): weight gradients dy^T x
  chk(a, "a"); chk(b, "b");
  TORCH_CHECK(a.dim() == 2 && b.dim() == 2 && a.size(0) == b.size(0), "a2c_mi355x::gemm_tn: (K,M)^T x (K,N)");
  const int64_t M = a.size(1), N = b.size(1), K = a.size(0);
  at::Tensor c = at::empty({M, N}, a.options());
  int splitk = 1;
  const int64_t tiles = ((M + 127) / 128) * ((N + 127) / 128);
  if (tiles < 256) splitk = (int)std::max<int64_t>(1, std::min<int64_t>(512 / tiles, K / 32));
  const size_t nb = a2c_gemm_ws_bytes(M, N, splitk);
  at::Tensor ws = at::empty({(int64_t)(nb + 3) / 4 + 4}, a.options());
  ok(a2c_gemm_f32_tn(M, N, K, a.data_ptr<float>(), M, b.data_ptr<float>(), N, c.data_ptr<float>(), N, splitk, ws.data_ptr<float>(),
                     (size_t)ws.numel() * 4, cur_stream()), "a2c_gemm_f32_tn");
  return c;
}

// ---- GRU cell (models.py:465-476): h_new = z*h + (1-z)*tanh(x Wx2 + (r*h) Wh2 + b2); W_x (3,in,h), W_h (3,h,h), b (3,1,h)
std::vector<at::Tensor> gru_cell_fwd(const at::Tensor& x, const at::Tensor& h, const at::Tensor& Wx, const at::Tensor& Wh, const at::Tensor& b) {
  chk(x, "x"); chk(h, "h"); chk(Wx, "W_x"); chk(Wh, "W_h"); chk(b, "b");
  const int64_t B = x.size(0), in = x.size(1), hd = h.size(1);
  TORCH_CHECK(Wx.sizes() == at::IntArrayRef({3, in, hd}) && Wh.sizes() == at::IntArrayRef({3, hd, hd}) && b.numel() == 3 * hd, "a2c_mi355x::gru_cell_fwd: shapes");
  auto o = x.options();
  at::Tensor gx = at::empty({B, 3 * hd}, o), gh = at::empty({B, 2 * hd}, o);
  for (int g = 0; g < 3; ++g)       // x.mm(W_x[g]) into column block g
    ok(a2c_gemm_f32_nn(B, hd, in, x.data_ptr<float>(), in, Wx.data_ptr<float>() + g * in * hd, hd, gx.data_ptr<float>() + g * hd, 3 * hd,
                       nullptr, 0, cur_stream()), "a2c_gemm_f32_nn");
  for (int g = 0; g < 2; ++g)
    ok(a2c_gemm_f32_nn(B, hd, hd, h.data_ptr<float>(), hd, Wh.data_ptr<float>() + g * hd * hd, hd, gh.data_ptr<float>() + g * hd, 2 * hd,
                       nullptr, 0, cur_stream()), "a2c_gemm_f32_nn");
  at::Tensor z = at::empty({B, hd}, o), r = at::empty({B, hd}, o), rh = at::empty({B, hd}, o), c = at::empty({B, hd}, o), hn = at::empty({B, hd}, o);
  ok(a2c_gru_gates(gx.data_ptr<float>(), gh.data_ptr<float>(), b.data_ptr<float>(), h.data_ptr<float>(), z.data_ptr<float>(),
                   r.data_ptr<float>(), rh.data_ptr<float>(), (int)B, (int)hd, cur_stream()), "a2c_gru_gates");
  at::Tensor rhu = gemm_nn(rh, Wh[2].contiguous());
  ok(a2c_gru_out(gx.data_ptr<float>(), rhu.data_ptr<float>(), b.data_ptr<float>(), h.data_ptr<float>(), z.data_ptr<float>(),
                 c.data_ptr<float>(), hn.data_ptr<float>(), (int)B, (int)hd, cur_stream()), "a2c_gru_out");
  return {hn, z, r, c};
}
// gradients of the cell's pre-activations and of h: given dh_new and the saved (h, z, r, c): (dz_pre, dr_pre, dc_pre, dh)
std::vector<at::Tensor> gru_cell_bwd(const at::Tensor& dh_new, const at::Tensor& h, const at::Tensor& z, const at::Tensor& r,
                                     const at::Tensor& c, const at::Tensor& Wh) {
  chk(dh_new, "dh_new"); chk(h, "h"); chk(z, "z"); chk(r, "r"); chk(c, "c"); chk(Wh, "W_h");
  const int64_t B = h.size(0), hd = h.size(1);
  auto o = h.options();
  at::Tensor dc = at::empty({B, hd}, o), dz = at::empty({B, hd}, o), dh = at::empty({B, hd}, o), dzp = at::empty({B, hd}, o), drp = at::empty({B, hd}, o);
  ok(a2c_gru_out_bwd(dh_new.data_ptr<float>(), h.data_ptr<float>(), z.data_ptr<float>(), c.data_ptr<float>(), dc.data_ptr<float>(),
                     dz.data_ptr<float>(), dh.data_ptr<float>(), (int)B, (int)hd, cur_stream()), "a2c_gru_out_bwd");
  at::Tensor d_rh = at::empty({B, hd}, o);      // dc_pre . W_h[2]^T
  at::Tensor w2 = Wh[2].contiguous();
  ok(a2c_gemm_f32_nt(B, hd, hd, dc.data_ptr<float>(), hd, w2.data_ptr<float>(), hd, d_rh.data_ptr<float>(), hd, nullptr, 0, cur_stream()),
     "a2c_gemm_f32_nt");
  ok(a2c_gru_gates_bwd(d_rh.data_ptr<float>(), dz.data_ptr<float>(), h.data_ptr<float>(), z.data_ptr<float>(), r.data_ptr<float>(),
                       dzp.data_ptr<float>(), drp.data_ptr<float>(), dh.data_ptr<float>(), (int)B, (int)hd, cur_stream()), "a2c_gru_gates_bwd");
  return {dzp, drp, dc, dh};
}

// ---- LayerNorm over the last dim, eps 1e-5 (models.py:392): y, and the backward's dx / per-row dw, db pieces
std::tuple<at::Tensor, at::Tensor, at::Tensor> layernorm_fwd(const at::Tensor& x, const at::Tensor& w, const at::Tensor& b) {
  chk(x, "x"); chk(w, "weight"); chk(b, "bias");
  TORCH_CHECK(x.dim() == 2 && w.numel() == x.size(1) && b.numel() == x.size(1), "a2c_mi355x::layernorm_fwd: x (rows, n)");
  at::Tensor y = at::empty_like(x), mean = at::empty({x.size(0)}, x.options()), rstd = at::empty({x.size(0)}, x.options());
  ok(a2c_layernorm_fwd(x.data_ptr<float>(), w.data_ptr<float>(), b.data_ptr<float>(), y.data_ptr<float>(), mean.data_ptr<float>(),
                       rstd.data_ptr<float>(), x.size(0), (int)x.size(1), cur_stream()), "a2c_layernorm_fwd");
  return {y, mean, rstd};
}
std::tuple<at::Tensor, at::Tensor> layernorm_bwd(const at::Tensor& dy, const at::Tensor& x, const at::Tensor& w, const at::Tensor& mean,
                                                 const at::Tensor& rstd) {
  chk(dy, "dy"); chk(x, "x"); chk(w, "weight"); chk(mean, "mean"); chk(rstd, "rstd");
  at::Tensor dx = at::empty_like(x), dwr = at::empty_like(x);      // dwr[row] = dy * xhat: its column sum is the weight gradient
  ok(a2c_layernorm_bwd(dy.data_ptr<float>(), x.data_ptr<float>(), w.data_ptr<float>(), mean.data_ptr<float>(), rstd.data_ptr<float>(),
                       dx.data_ptr<float>(), dwr.data_ptr<float>(), x.size(0), (int)x.size(1), 0, cur_stream()), "a2c_layernorm_bwd");
  return {dx, dwr};
}

// ---- runner.py:212-232 for B envs: one env step's records into the rollout-major buffers, in place
void rollout_record_(const at::Tensor& rew, const at::Tensor& done, const at::Tensor& val, at::Tensor val_prev, at::Tensor rewards,
                     at::Tensor dones, at::Tensor deltas, int64_t T, int64_t t, int64_t slot0, double gamma, bool pong) {
  chk(rew, "rew"); chk(done, "done"); chk(val, "val"); chk(val_prev, "val_prev"); chk(rewards, "rewards"); chk(dones, "dones"); chk(deltas, "deltas");
  const int B = (int)rew.numel();
  TORCH_CHECK(done.numel() == B && val.numel() == B && val_prev.numel() == B && (slot0 + B) * T <= rewards.numel(), "a2c_mi355x::rollout_record_: shapes");
  ok(a2c_rollout_record(rew.data_ptr<float>(), done.data_ptr<float>(), val.data_ptr<float>(), 1, val_prev.data_ptr<float>(),
                        rewards.data_ptr<float>(), dones.data_ptr<float>(), deltas.data_ptr<float>(), nullptr, nullptr, 0, B, T, t,
                        slot0, (float)gamma, pong ? 1 : 0, cur_stream()), "a2c_rollout_record");
}

// clip_grad_norm_ + Adam step over flat arenas (updater.py:129-132, 226-229), in place; returns the pre-clip norm
at::Tensor clip_adam_(at::Tensor params, at::Tensor grads, at::Tensor exp_avg, at::Tensor exp_avg_sq, int64_t step, double max_norm,
                      double lr, double beta1, double beta2, double eps) {
  chk(params, "params"); chk(grads, "grads"); chk(exp_avg, "exp_avg"); chk(exp_avg_sq, "exp_avg_sq");
  at::Tensor sumsq = at::empty({1}, params.options().dtype(at::kDouble)), norm = at::empty({1}, params.options());
  ok(a2c_gradnorm_sq(grads.data_ptr<float>(), grads.numel(), sumsq.data_ptr<double>(), reduce_scratch(grads).data_ptr<double>(), cur_stream()),
     "a2c_gradnorm_sq");
  ok(a2c_clip_adam(params.data_ptr<float>(), grads.data_ptr<float>(), exp_avg.data_ptr<float>(), exp_avg_sq.data_ptr<float>(),
                   params.numel(), sumsq.data_ptr<double>(), max_norm, lr, beta1, beta2, eps, step, norm.data_ptr<float>(), cur_stream()),
     "a2c_clip_adam");
  return norm;
}
}  // namespace

// In-place ops over the kernel launchers themselves, one per entry point, generated from the header
// (tools/gen_torch_abi_ops.py): what the hot loop calls when A2C_TORCH_OPS=1 (a2c_amd/ops.py: TorchAbi).
#include "torch_ops_abi.inc"

TORCH_LIBRARY(a2c_mi355x, m) {
  A2C_ABI_OPS_DEF(m);
  m.def("discount(Tensor x, Tensor dones, float g, int n_seg=1) -> Tensor");
  m.def("gae_returns(Tensor deltas, Tensor rewards, Tensor dones, float g_adv, float g_ret, int n_seg) -> (Tensor, Tensor)");
  m.def("softmax_sample(Tensor logits, Tensor u) -> Tensor");
  m.def("frame_stack_push(Tensor frame_new, Tensor reset_mask, Tensor prev) -> Tensor");
  m.def("loss_fwd_bwd(Tensor logits, Tensor vals, Tensor actions, Tensor advs, Tensor returns, float pi_coef, float val_coef, "
        "float entr_coef) -> (Tensor, Tensor, Tensor)");
  m.def("linear(Tensor x, Tensor weight, Tensor bias, bool relu=False) -> Tensor");
  m.def("clip_rmsprop_(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) square_avg, float max_norm, float lr, float alpha=0.99, "
        "float eps=1e-8) -> Tensor");
  m.def("conv2d_fwd(Tensor x, Tensor weight, Tensor bias, int stride=1, int pad=0, bool relu=True) -> Tensor");
  m.def("conv2d_bwd_data(Tensor dout, Tensor weight, Tensor? mask, int H, int W, int stride=1, int pad=0) -> Tensor");
  m.def("conv2d_bwd_weight(Tensor x, Tensor dout, int ks, int stride=1, int pad=0) -> (Tensor, Tensor)");
  m.def("gemm_nn(Tensor a, Tensor b) -> Tensor");
  m.def("gemm_tn(Tensor a, Tensor b) -> Tensor");
  m.def("gru_cell_fwd(Tensor x, Tensor h, Tensor W_x, Tensor W_h, Tensor b) -> Tensor[]");
  m.def("gru_cell_bwd(Tensor dh_new, Tensor h, Tensor z, Tensor r, Tensor c, Tensor W_h) -> Tensor[]");
  m.def("layernorm_fwd(Tensor x, Tensor weight, Tensor bias) -> (Tensor, Tensor, Tensor)");
  m.def("layernorm_bwd(Tensor dy, Tensor x, Tensor weight, Tensor mean, Tensor rstd) -> (Tensor, Tensor)");
  m.def("rollout_record_(Tensor rew, Tensor done, Tensor val, Tensor(a!) val_prev, Tensor(b!) rewards, Tensor(c!) dones, "
        "Tensor(d!) deltas, int T, int t, int slot0, float gamma, bool pong=False) -> ()");
  m.def("clip_adam_(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) exp_avg, Tensor(d!) exp_avg_sq, int step, float max_norm, float lr, "
        "float beta1=0.9, float beta2=0.999, float eps=1e-8) -> Tensor");
}

TORCH_LIBRARY_IMPL(a2c_mi355x, CUDA, m) {      // "CUDA" is the HIP dispatch key on PyTorch-ROCm; no CPU implementation exists
  A2C_ABI_OPS_IMPL(m);
  m.impl("discount", &discount);
  m.impl("gae_returns", &gae_returns);
  m.impl("softmax_sample", &softmax_sample);
  m.impl("frame_stack_push", &frame_stack_push);
  m.impl("loss_fwd_bwd", &loss_fwd_bwd);
  m.impl("linear", &linear);
  m.impl("clip_rmsprop_", &clip_rmsprop_);
  m.impl("conv2d_fwd", &conv2d_fwd);
  m.impl("conv2d_bwd_data", &conv2d_bwd_data);
  m.impl("conv2d_bwd_weight", &conv2d_bwd_weight);
  m.impl("gemm_nn", &gemm_nn);
  m.impl("gemm_tn", &gemm_tn);
  m.impl("gru_cell_fwd", &gru_cell_fwd);
  m.impl("gru_cell_bwd", &gru_cell_bwd);
  m.impl("layernorm_fwd", &layernorm_fwd);
  m.impl("layernorm_bwd", &layernorm_bwd);
  m.impl("rollout_record_", &rollout_record_);
  m.impl("clip_adam_", &clip_adam_);
}
