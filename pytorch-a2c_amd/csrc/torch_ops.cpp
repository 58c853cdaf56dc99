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
                      sums.data_ptr<double>(), cur_stream()), "a2c_loss_fwd_bwd");
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
  ok(a2c_gradnorm_sq(grads.data_ptr<float>(), grads.numel(), sumsq.data_ptr<double>(), cur_stream()), "a2c_gradnorm_sq");
  ok(a2c_clip_rmsprop(params.data_ptr<float>(), grads.data_ptr<float>(), square_avg.data_ptr<float>(), params.numel(),
                      sumsq.data_ptr<double>(), max_norm, lr, alpha, eps, norm.data_ptr<float>(), cur_stream()), "a2c_clip_rmsprop");
  return norm;
}
}  // namespace

TORCH_LIBRARY(a2c_mi355x, m) {
  m.def("discount(Tensor x, Tensor dones, float g, int n_seg=1) -> Tensor");
  m.def("gae_returns(Tensor deltas, Tensor rewards, Tensor dones, float g_adv, float g_ret, int n_seg) -> (Tensor, Tensor)");
  m.def("softmax_sample(Tensor logits, Tensor u) -> Tensor");
  m.def("frame_stack_push(Tensor frame_new, Tensor reset_mask, Tensor prev) -> Tensor");
  m.def("loss_fwd_bwd(Tensor logits, Tensor vals, Tensor actions, Tensor advs, Tensor returns, float pi_coef, float val_coef, "
        "float entr_coef) -> (Tensor, Tensor, Tensor)");
  m.def("linear(Tensor x, Tensor weight, Tensor bias, bool relu=False) -> Tensor");
  m.def("clip_rmsprop_(Tensor(a!) params, Tensor(b!) grads, Tensor(c!) square_avg, float max_norm, float lr, float alpha=0.99, "
        "float eps=1e-8) -> Tensor");
}

TORCH_LIBRARY_IMPL(a2c_mi355x, CUDA, m) {      // "CUDA" is the HIP dispatch key on PyTorch-ROCm; no CPU implementation exists
  m.impl("discount", &discount);
  m.impl("gae_returns", &gae_returns);
  m.impl("softmax_sample", &softmax_sample);
  m.impl("frame_stack_push", &frame_stack_push);
  m.impl("loss_fwd_bwd", &loss_fwd_bwd);
  m.impl("linear", &linear);
  m.impl("clip_rmsprop_", &clip_rmsprop_);
}
