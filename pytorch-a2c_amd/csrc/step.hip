// One rollout step of the A3CModel-shaped policy in ONE launch (runner.py:190-232 + models.py:60-90):
//   [bookkeeping of the previous env step]  rewards/dones/TD delta           (runner.py:212-232)
//   [frame stack]  state_t = push(state_{t-1}, frame)  -> states row + LDS    (utils.py:26-43, runner.py:199)
//   conv 8x8/s4 (C -> 16) + ReLU -> conv 4x4/s2 (16 -> 32) + ReLU             (models.py:60-66)
//   [logits | value] = flat . Wc^T + bc   (Wc = [pi;value] . proj_matrx: no activation in between, models.py:73)
//   softmax + inverse-CDF sample                                              (runner.py:94-97, utils.py:45-60)
//   [bootstrap of the slot's last step]                                       (runner.py:236-245)
// One WORKGROUP (8 waves) owns one env: the 113 KB state is staged ONCE into LDS (and written to
// the rollout buffer on the way), conv1's 25.6 KB and conv2's 10.4 KB activations never leave the
// CU.  With n_envs = 256 that is one workgroup per CU and T+1 launches per slot instead of 4T+3.
//   LDS: img 4 x PLANE1 (0 mod 64 floats apart: conflict-free ds_read_b128 rows) | a1 16 x PLANE2
//   (32 mod 64: conflict-free ds_read_b64) | a2 flat (c, y, x) | reduction scratch.  conv1's 64
//   A fragments live in registers; conv2's 32 KB of fragments are prefetched into registers at
//   kernel start and dropped over the image once conv1 has consumed it.
#include "a2c_common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512;        // threads per workgroup
constexpr int NST = 14;        // float4 staging registers per thread (covers 4 x 84 x 84)
constexpr int HN = 8;          // max heads (n_actions + 1)

struct StepP {
  a2c_a3c_step_args a;
  int OH1, OW1, OH2, OW2, PLANE1, PLANE2, F, F4;
};

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void a3c_step_kernel(StepP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const a2c_a3c_step_args& a = p.a;
  float* __restrict__ img = lds;
  float* __restrict__ a1 = img + 4 * p.PLANE1;
  float* __restrict__ a2 = a1 + 16 * p.PLANE2;
  float* __restrict__ red = a2 + p.F4;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int HW = a.H * a.W, W = a.W;
  const int per4 = HW >> 2, tot4 = 4 * per4;
  const int N = a.n_actions + 1;

  // ---- issue the state loads: plane c of state_t
  const bool rst = a.reset_mask != nullptr && a.reset_mask[b] != 0.f;
  float4 pf[NST];
#pragma unroll
  for (int u = 0; u < NST; ++u) {
    const int idx = tid + u * NT;
    pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (idx < tot4) {
      const int c = idx / per4, rem = idx - c * per4;
      const float* src;
      if (a.frame_new) src = (c == 3) ? a.frame_new + (long)b * HW : (rst ? nullptr : a.prev + (long)b * a.prev_stride + (long)(c + 1) * HW);
      else src = a.prev + (long)b * a.prev_stride + (long)c * HW;
      if (src) pf[u] = *reinterpret_cast<const float4*>(src + (rem << 2));
    }
  }
  // ---- weights that do not depend on the state
  float af[64];
#pragma unroll
  for (int s = 0; s < 64; ++s) af[s] = a.wfrag1[s * 64 + lane];
  const float4* __restrict__ wf2v = reinterpret_cast<const float4*>(a.wfrag2);
  const float4 w2a = wf2v[tid], w2b = wf2v[tid + NT], w2c = wf2v[tid + 2 * NT], w2d = wf2v[tid + 3 * NT];
  float b1[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) b1[rr] = a.bias1[4 * g + rr];

  // ---- bookkeeping of the previous env step (one thread; its loads overlap the staging)
  float bk_r = 0.f, bk_d = 0.f, bk_v = 0.f;
  long bk_e = 0;
  if (tid == 0 && a.rew) {
    bk_e = (a.slot0 + b) * a.T + a.t_rec;
    bk_r = a.rew[b];
    bk_d = a.done[b] != 0.f ? 1.f : 0.f;
    if (a.pong && bk_r != 0.f) bk_d = 1.f;
    a.rewards[bk_e] = bk_r;
    a.dones[bk_e] = bk_d;
    bk_v = a.heads[(long)b * a.ldh + a.n_actions];          // value of the state the env step left
    if (a.t_rec > 0) {
      const float pr = a.rewards[bk_e - 1], pd = a.dones[bk_e - 1];
      const float gv = a.gamma * bk_v;
      a.deltas[bk_e - 1] = (pr + gv * (1.f - pd)) - a.val_prev[b];
    }
    a.val_prev[b] = bk_v;
  }

  // ---- state -> LDS (+ the rollout buffer row)
#pragma unroll
  for (int u = 0; u < NST; ++u) {
    const int idx = tid + u * NT;
    if (idx < tot4) {
      const int c = idx / per4, rem = idx - c * per4;
      *reinterpret_cast<float4*>(img + c * p.PLANE1 + (rem << 2)) = pf[u];
      if (a.out) *reinterpret_cast<float4*>(a.out + (long)b * a.out_stride + (long)c * HW + (rem << 2)) = pf[u];
    }
  }
  __syncthreads();

  // ---- conv1: 16 x (OH1*OW1) = A[16 x 256] . im2col, one 16-pixel tile per wave pass
  {
    const int NP = p.OH1 * p.OW1, ntile = (NP + 15) >> 4;
    for (int tile = w; tile < ntile; tile += NT / 64) {
      const int idx = tile * 16 + j;
      const bool ok = idx < NP;
      const int i = ok ? idx : 0;
      const int r = i / p.OW1, c = i - r * p.OW1;
      const float* __restrict__ l = img + r * 4 * W + c * 4 + g * p.PLANE1;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ky = 0; ky < 8; ++ky) {
        const float4 t0 = *reinterpret_cast<const float4*>(l + ky * W);
        const float4 t1 = *reinterpret_cast<const float4*>(l + ky * W + 4);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 0], t0.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 1], t0.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 2], t0.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 3], t0.w, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 4], t1.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 5], t1.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 6], t1.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 7], t1.w, acc, 0, 0, 0);
      }
      if (ok) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a1[(4 * g + rr) * p.PLANE2 + i] = fmaxf(acc[rr] + b1[rr], 0.f);
      }
    }
  }
  __syncthreads();                                   // img consumed, a1 complete
  {
    float4* __restrict__ iv = reinterpret_cast<float4*>(img);
    iv[tid] = w2a; iv[tid + NT] = w2b; iv[tid + 2 * NT] = w2c; iv[tid + 3 * NT] = w2d;
  }
  // head weights for this thread's K slices: in flight during conv2
  float4 wc[2][HN];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int k = (tid << 2) + q * (NT * 4);
#pragma unroll
    for (int n = 0; n < HN; ++n) {
      wc[q][n] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && k < p.F) wc[q][n] = *reinterpret_cast<const float4*>(a.Wc + (long)n * p.F + k);
    }
  }
  __syncthreads();

  // ---- conv2: 32 x (OH2*OW2), unit = (16-pixel tile, 16-channel half)
  {
    const int NP = p.OH2 * p.OW2, ntile = (NP + 15) >> 4;
    const float* __restrict__ la = img + lane;
    for (int unit = w; unit < ntile * 2; unit += NT / 64) {
      const int tile = unit >> 1, m = unit & 1;
      const int idx = tile * 16 + j;
      const bool ok = idx < NP;
      const int i = ok ? idx : 0;
      const int r = i / p.OW2, c = i - r * p.OW2;
      const float* __restrict__ l = a1 + r * 2 * p.OW1 + c * 2 + g * p.PLANE2;
      float b2[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) b2[rr] = a.bias2[m * 16 + 4 * g + rr];
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4) {
        float bv[16], av[16];
#pragma unroll
        for (int ky = 0; ky < 4; ++ky) {
          const int off = c4 * 4 * p.PLANE2 + ky * p.OW1;
          const float2 t0 = *reinterpret_cast<const float2*>(l + off);
          const float2 t1 = *reinterpret_cast<const float2*>(l + off + 2);
          bv[ky * 4 + 0] = t0.x; bv[ky * 4 + 1] = t0.y; bv[ky * 4 + 2] = t1.x; bv[ky * 4 + 3] = t1.y;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) av[u] = la[((c4 * 16 + u) * 2 + m) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], acc, 0, 0, 0);
      }
      if (ok) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a2[(m * 16 + 4 * g + rr) * NP + i] = fmaxf(acc[rr] + b2[rr], 0.f);
      }
    }
  }
  __syncthreads();

  // ---- heads: N dot products of length F; fixed-order reduction (deterministic)
  {
    float acc[HN];
#pragma unroll
    for (int n = 0; n < HN; ++n) acc[n] = 0.f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int k = (tid << 2) + q * (NT * 4);
      if (k < p.F) {
        const float4 x = *reinterpret_cast<const float4*>(a2 + k);
#pragma unroll
        for (int n = 0; n < HN; ++n)
          if (n < N) acc[n] += x.x * wc[q][n].x + x.y * wc[q][n].y + x.z * wc[q][n].z + x.w * wc[q][n].w;
      }
    }
#pragma unroll
    for (int n = 0; n < HN; ++n)
      if (n < N) {
        const float v = wave_sum(acc[n]);
        if (lane == 0) red[w * HN + n] = v;
      }
  }
  __syncthreads();
  if (tid == 0) {
    float h[HN], vboot = 0.f;
#pragma unroll
    for (int n = 0; n < HN; ++n) {
      h[n] = 0.f;
      if (n < N) {
        h[n] = (((red[0 * HN + n] + red[1 * HN + n]) + (red[2 * HN + n] + red[3 * HN + n])) +
                ((red[4 * HN + n] + red[5 * HN + n]) + (red[6 * HN + n] + red[7 * HN + n]))) + a.bc[n];
        a.heads[(long)b * a.ldh + n] = h[n];
        if (n == a.n_actions) vboot = h[n];
      }
    }
    if (a.u != nullptr) {        // same maths as sample_kernel<true>: softmax, running fp32 cumsum, first >= u
      float mx = -INFINITY;
#pragma unroll
      for (int n = 0; n < HN; ++n)
        if (n < a.n_actions) mx = fmaxf(mx, h[n]);
      float den = 0.f;
#pragma unroll
      for (int n = 0; n < HN; ++n)
        if (n < a.n_actions) den += expf(h[n] - mx);
      const float ub = a.u[b];
      float cs = 0.f;
      int pick = -1;
#pragma unroll
      for (int n = 0; n < HN; ++n)
        if (n < a.n_actions) {
          cs = cs + expf(h[n] - mx) / den;
          if (pick < 0 && cs >= ub) pick = n;
        }
      a.actions[(long)b * a.act_stride] = (int64_t)pick;
    }
    if (a.bootstrap && a.rew) {  // runner.py:236-245 on the step recorded above (e = slot*T + T-1)
      float r = bk_r;
      if (bk_d == 0.f) {
        r = r + a.gamma * vboot;
        a.rewards[bk_e] = r;
        a.dones[bk_e] = 1.f;
      }
      a.deltas[bk_e] = r - bk_v;
    }
  }
}

static inline int plane_pad(int n, int mod64) {      // smallest p >= n with p % 64 == mod64
  int p = ((n + 63) / 64) * 64 + mod64;
  if (p - 64 >= n) p -= 64;
  return p;
}

static bool step_shapes(int C, int H, int W, int n_actions, StepP& p) {
  if (C != 4 || H < 8 || W < 8 || W % 4 || n_actions < 1 || n_actions + 1 > HN) return false;
  p.OH1 = (H - 8) / 4 + 1; p.OW1 = (W - 8) / 4 + 1;
  if (p.OH1 < 4 || p.OW1 < 4 || p.OW1 % 2) return false;
  p.OH2 = (p.OH1 - 4) / 2 + 1; p.OW2 = (p.OW1 - 4) / 2 + 1;
  p.PLANE1 = plane_pad(H * W, 0);
  p.PLANE2 = plane_pad(p.OH1 * p.OW1, 32);
  p.F = 32 * p.OH2 * p.OW2;
  p.F4 = ((p.F + 3) / 4) * 4;
  if (p.F % 4 || p.F > 2 * NT * 4) return false;
  if (C * H * W > NT * NST * 4) return false;
  if (4 * p.PLANE1 < 8192) return false;             // conv2's fragments reuse the image region
  return true;
}
static size_t step_lds(const StepP& p) { return 4 * ((size_t)4 * p.PLANE1 + (size_t)16 * p.PLANE2 + p.F4 + 8 * HN + 16); }
}  // namespace

extern "C" {
int a2c_a3c_step_supported(int C, int H, int W, int n_actions) {
  StepP p;
  return step_shapes(C, H, W, n_actions, p) && step_lds(p) <= 160 * 1024 ? 1 : 0;
}

int a2c_a3c_step(const a2c_a3c_step_args* args, a2c_stream_t stream) {
  if (!args) return A2C_ERR_ARG;
  StepP p;
  p.a = *args;
  const a2c_a3c_step_args& a = p.a;
  if (a.B < 0) return A2C_ERR_ARG;
  if (a.B == 0) return A2C_OK;
  if (!step_shapes(a.C, a.H, a.W, a.n_actions, p) || step_lds(p) > 160 * 1024) return A2C_ERR_ARG;
  if (!a.prev) return A2C_ERR_ARG;
  if (!a.wfrag1 || !a.bias1 || !a.wfrag2 || !a.bias2 || !a.Wc || !a.bc || !a.heads) return A2C_ERR_ARG;
  if (a.u && !a.actions) return A2C_ERR_ARG;
  if (a.rew && (!a.done || !a.val_prev || !a.rewards || !a.dones || !a.deltas || a.T < 1 || a.t_rec < 0 || a.t_rec >= a.T))
    return A2C_ERR_ARG;
  if (a.bootstrap && (!a.rew || a.t_rec != a.T - 1)) return A2C_ERR_ARG;
  if (a.prev_stride % 4 || (a.out && a.out_stride % 4) || a.ldh < a.n_actions + 1) return A2C_ERR_ARG;
  if ((((uintptr_t)a.prev | (uintptr_t)a.frame_new | (uintptr_t)a.out | (uintptr_t)a.wfrag2 | (uintptr_t)a.Wc) % 16)) return A2C_ERR_ARG;
  const size_t lds = step_lds(p);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)a3c_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return A2C_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(a3c_step_kernel, dim3(a.B), dim3(NT), lds, a2c_s(stream), p);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
