// A2C loss, forward and backward in one pass (Updater.update_model, updater.py:100-106,
// 124-127).  One lane per rollout sample; per-sample traffic is (2A+5) floats + 8 B action,
// so the kernel is latency/launch bound at the reference's sizes.  The three loss sums are
// reduced per workgroup in fp64; the per-workgroup partials are added in workgroup order by the last one to finish (no atomics).
#include "a2c_common.h"

namespace {
constexpr int MAXA = 32;

__global__ __launch_bounds__(256) void loss_kernel(const float* __restrict__ logits, long ldl,
                                                   const float* __restrict__ vals, long vstride,
                                                   const int64_t* __restrict__ actions,
                                                   const float* __restrict__ advs,
                                                   const float* __restrict__ returns,
                                                   const double* __restrict__ adv_sums, long n_local,
                                                   long n_global, int A, float pi_coef, float val_coef,
                                                   float entr_coef, float* __restrict__ dlogits, long ldd,
                                                   float* __restrict__ dvals, long dvstride, double* loss_sums, double* scratch) {
  __shared__ double sm[4];
  float mean = 0.f, den = 1.f;
  if (adv_sums != nullptr) {
    const double m = adv_sums[0] / (double)n_global;
    double var = (adv_sums[1] - (double)n_global * m * m) / (double)(n_global - 1);
    if (var < 0.0) var = 0.0;
    mean = (float)m;
    den = (float)sqrt(var) + 1e-6f;
  }
  const float invN = 1.0f / (float)n_global;
  double s_pi = 0.0, s_val = 0.0, s_ent = 0.0;
  for (long n = blockIdx.x * 256L + threadIdx.x; n < n_local; n += gridDim.x * 256L) {
    const float* row = logits + n * ldl;
    float x[MAXA];
    float mx = -INFINITY;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) { x[a] = row[a]; mx = fmaxf(mx, x[a]); }
    float se = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) se += expf(x[a] - mx);
    const float lse = logf(se);
    float adv = advs[n];
    if (adv_sums != nullptr) adv = (adv - mean) / den;
    int act = (int)actions[n];
    if (act < 0) act += A;      // log_softs[arange, actions] (updater.py:104): a -1 of sample_action indexes the last action
    float plp = 0.f, lp_act = 0.f;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) {
        const float lsm = x[a] - mx - lse;
        const float p = expf(lsm);
        x[a] = lsm;
        plp += p * lsm;
        if (a == act) lp_act = lsm;
      }
    // d/dlogit_j: -pi_coef*adv/N*(1[j==act]-p_j) + entr_coef/N * p_j*(lsm_j - sum_i p_i lsm_i)
    const float gpi = -pi_coef * adv * invN;
    const float gen = entr_coef * invN;
    float* drow = dlogits + n * ldd;
#pragma unroll
    for (int a = 0; a < MAXA; ++a)
      if (a < A) {
        const float p = expf(x[a]);
        drow[a] = gpi * ((a == act ? 1.f : 0.f) - p) + gen * p * (x[a] - plp);
      }
    const float dv = vals[n * vstride] - returns[n];
    dvals[n * dvstride] = val_coef * 2.f * dv * invN;
    s_pi += (double)(lp_act * adv);
    s_val += (double)(dv * dv);
    s_ent += (double)plp;
  }
  s_pi = block_sum_256(s_pi, sm);
  s_val = block_sum_256(s_val, sm);
  s_ent = block_sum_256(s_ent, sm);
  const double v3[3] = {s_pi, s_val, s_ent};
  grid_sum_ordered<3>(v3, loss_sums, scratch, sm);       // fixed-order second stage: no fp64 atomics
}
}  // namespace

extern "C" int a2c_loss_fwd_bwd(const float* logits, int64_t ld_logits, const float* vals, int64_t val_stride,
                                const int64_t* actions,
                                const float* advs, const float* returns, const double* adv_sums, int64_t n_local,
                                int64_t n_global, int A, float pi_coef, float val_coef, float entr_coef,
                                float* dlogits, int64_t ldd, float* dvals, int64_t dval_stride, double* loss_sums,
                                double* scratch, a2c_stream_t stream) {
  if (n_local < 0 || n_global < n_local || A < 1 || A > MAXA || !loss_sums || !scratch) return A2C_ERR_ARG;
  if (adv_sums && n_global < 2) return A2C_ERR_ARG;
  if (n_local == 0) {
    a2c_zero_async(loss_sums, 3 * sizeof(double), a2c_s(stream));
    return A2C_OK;
  }
  if (!logits || !vals || !actions || !advs || !returns || !dlogits || !dvals) return A2C_ERR_ARG;
  hipLaunchKernelGGL(loss_kernel, dim3(a2c_grid_1d(n_local, 256, A2C_REDUCE_MAX_BLOCKS)), dim3(256), 0, a2c_s(stream), logits,
                     (long)ld_logits, vals, (long)val_stride, actions, advs, returns, adv_sums, (long)n_local,
                     (long)n_global, A, pi_coef, val_coef, entr_coef, dlogits, (long)ldd, dvals, (long)dval_stride,
                     loss_sums, scratch);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
