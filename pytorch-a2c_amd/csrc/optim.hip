// Gradient clipping + optimiser step over the flat parameter arena (updater.py:129-132,
// 226-229).  All parameters with a gradient live in one contiguous fp32 arena (params, grads,
// optimiser state are parallel arrays), so clip_grad_norm_ + RMSprop/Adam is two memory-bound
// launches: a sum-of-squares reduction and one fused clip+update pass
// (RMSprop: 4 reads... params, grads, square_avg in; params, grads, square_avg out = 24 B/param;
//  Adam: 32 B/param).  The same arena is the RCCL all-reduce buffer on multi-GPU runs.
#include "a2c_common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n, double* out, double* scratch) {
  __shared__ double sm[4];
  double a = 0.0;
  const long n4 = n >> 2;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
    const float4 v = g4[i];
    a += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) a += (double)g[i] * g[i];
  a = block_sum_256(a, sm);
  const double v1[1] = {a};
  grid_sum_ordered<1>(v1, out, scratch, sm);       // fixed-order second stage: no fp64 atomics
}

// torch.nn.utils.clip_grad_norm_: coef = max_norm / (total_norm + 1e-6), clamped to 1.0,
// and the gradients are multiplied by it unconditionally.
__device__ __forceinline__ float clip_coef(const double* sumsq, float max_norm, float* norm_out) {
  const float norm = (float)sqrt(sumsq[0]);
  if (norm_out && blockIdx.x == 0 && threadIdx.x == 0) *norm_out = norm;
  const float c = max_norm / (norm + 1e-6f);
  return c > 1.0f ? 1.0f : c;
}

__device__ __forceinline__ void rmsprop1(float& p, float& g, float& sq, float coef, float lr, float alpha,
                                         float oma, float eps) {
  g = g * coef;
  sq = __fadd_rn(__fmul_rn(sq, alpha), __fmul_rn(__fmul_rn(oma, g), g));  // mul_(alpha).addcmul_(g,g,1-alpha)
  const float avg = __fadd_rn(sqrtf(sq), eps);                            // sqrt().add_(eps)
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-lr, g), avg));                    // addcdiv_(g, avg, value=-lr)
}

__global__ __launch_bounds__(256) void clip_rmsprop_kernel(float* __restrict__ p, float* __restrict__ g,
                                                           float* __restrict__ sq, long n, const double* sumsq,
                                                           float max_norm, float lr, float alpha, float oma,
                                                           float eps, float* norm_out) {
  const float coef = clip_coef(sumsq, max_norm, norm_out);
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* s4 = reinterpret_cast<float4*>(sq);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
    float4 pv = p4[i], gv = g4[i], sv = s4[i];
    rmsprop1(pv.x, gv.x, sv.x, coef, lr, alpha, oma, eps);
    rmsprop1(pv.y, gv.y, sv.y, coef, lr, alpha, oma, eps);
    rmsprop1(pv.z, gv.z, sv.z, coef, lr, alpha, oma, eps);
    rmsprop1(pv.w, gv.w, sv.w, coef, lr, alpha, oma, eps);
    p4[i] = pv; g4[i] = gv; s4[i] = sv;
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
    rmsprop1(p[i], g[i], sq[i], coef, lr, alpha, oma, eps);
}

__device__ __forceinline__ void adam1(float& p, float& g, float& m, float& v, float coef, float omb1, float beta2,
                                      float omb2, float eps, float step_size, float bc2_sqrt) {
  g = g * coef;
  m = __fadd_rn(m, __fmul_rn(omb1, __fsub_rn(g, m)));                       // exp_avg.lerp_(grad, 1-beta1)
  v = __fadd_rn(__fmul_rn(v, beta2), __fmul_rn(__fmul_rn(omb2, g), g));     // mul_(beta2).addcmul_(g,g,1-beta2)
  const float denom = __fadd_rn(__fdiv_rn(sqrtf(v), bc2_sqrt), eps);       // (sqrt/bc2_sqrt).add_(eps)
  p = __fadd_rn(p, __fdiv_rn(__fmul_rn(-step_size, m), denom));            // addcdiv_(m, denom, -step_size)
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                        float* __restrict__ m, float* __restrict__ v, long n,
                                                        const double* sumsq, float max_norm, float omb1, float beta2,
                                                        float omb2, float eps, float step_size, float bc2_sqrt,
                                                        float* norm_out) {
  const float coef = clip_coef(sumsq, max_norm, norm_out);
  const long n4 = n >> 2;
  float4* p4 = reinterpret_cast<float4*>(p);
  float4* g4 = reinterpret_cast<float4*>(g);
  float4* m4 = reinterpret_cast<float4*>(m);
  float4* v4 = reinterpret_cast<float4*>(v);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += gridDim.x * 256L) {
    float4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
    adam1(pv.x, gv.x, mv.x, vv.x, coef, omb1, beta2, omb2, eps, step_size, bc2_sqrt);
    adam1(pv.y, gv.y, mv.y, vv.y, coef, omb1, beta2, omb2, eps, step_size, bc2_sqrt);
    adam1(pv.z, gv.z, mv.z, vv.z, coef, omb1, beta2, omb2, eps, step_size, bc2_sqrt);
    adam1(pv.w, gv.w, mv.w, vv.w, coef, omb1, beta2, omb2, eps, step_size, bc2_sqrt);
    p4[i] = pv; g4[i] = gv; m4[i] = mv; v4[i] = vv;
  }
  for (long i = (n4 << 2) + blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L)
    adam1(p[i], g[i], m[i], v[i], coef, omb1, beta2, omb2, eps, step_size, bc2_sqrt);
}
}  // namespace

namespace {
__global__ void pack_scalars_kernel(const double* __restrict__ loss_sums, const float* __restrict__ norm,
                                    const int* __restrict__ err, double* __restrict__ out) {
  if (threadIdx.x == 0) {
    out[0] = loss_sums[0]; out[1] = loss_sums[1]; out[2] = loss_sums[2];
    out[3] = (double)norm[0];
    out[4] = err ? (double)err[0] : 0.0;
  }
}
}  // namespace

extern "C" {
int a2c_pack_update_scalars(const double* loss_sums, const float* grad_norm, const int* err, double* out5,
                            a2c_stream_t stream) {
  if (!loss_sums || !grad_norm || !out5) return A2C_ERR_ARG;
  hipLaunchKernelGGL(pack_scalars_kernel, dim3(1), dim3(64), 0, a2c_s(stream), loss_sums, grad_norm, err, out5);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gradnorm_sq(const float* grads, int64_t n, double* sumsq, double* scratch, a2c_stream_t stream) {
  if (n < 0 || !sumsq || !scratch || (n > 0 && !grads) || ((uintptr_t)grads % 16)) return A2C_ERR_ARG;
  if (n == 0) {
    a2c_zero_async(sumsq, sizeof(double), a2c_s(stream));
    return A2C_OK;
  }
  hipLaunchKernelGGL(sumsq_kernel, dim3(a2c_grid_1d((n + 3) / 4, 256, A2C_REDUCE_MAX_BLOCKS)), dim3(256), 0, a2c_s(stream), grads,
                     (long)n, sumsq, scratch);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_clip_rmsprop(float* params, float* grads, float* square_avg, int64_t n, const double* sumsq, double max_norm,
                     double lr, double alpha, double eps, float* norm_out, a2c_stream_t stream) {
  if (n < 0 || !sumsq || (n > 0 && (!params || !grads || !square_avg))) return A2C_ERR_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)square_avg) % 16) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  const float oma = (float)(1.0 - alpha);
  hipLaunchKernelGGL(clip_rmsprop_kernel, dim3(a2c_grid_1d((n + 3) / 4, 256)), dim3(256), 0, a2c_s(stream), params,
                     grads, square_avg, (long)n, sumsq, (float)max_norm, (float)lr, (float)alpha, oma, (float)eps,
                     norm_out);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_clip_adam(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, const double* sumsq,
                  double max_norm, double lr, double beta1, double beta2, double eps, int64_t step, float* norm_out,
                  a2c_stream_t stream) {
  if (n < 0 || step < 1 || !sumsq || (n > 0 && (!params || !grads || !exp_avg || !exp_avg_sq))) return A2C_ERR_ARG;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) % 16) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  // torch/optim/adam.py (_single_tensor_adam): python-double scalars, rounded when applied
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1);
  const float bc2_sqrt = (float)sqrt(bc2);
  hipLaunchKernelGGL(clip_adam_kernel, dim3(a2c_grid_1d((n + 3) / 4, 256)), dim3(256), 0, a2c_s(stream), params,
                     grads, exp_avg, exp_avg_sq, (long)n, sumsq, (float)max_norm, (float)(1.0 - beta1), (float)beta2,
                     (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt, norm_out);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
