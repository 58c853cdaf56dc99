// Elementwise stages of the reference's custom GRU cell (models.GRU.forward, models.py:465-476)
//   z = sig(x Wx0 + h Wh0 + b0);  r = sig(x Wx1 + h Wh1 + b1)
//   h' = z*h + (1-z)*tanh(x Wx2 + (r*h) Wh2 + b2)
// and torch.nn.LayerNorm of the FCModel/GRUFCModel value head (models.py:392, 510).
// The six matrix products run on a2c_gemm_f32; these kernels are the fused gate math between
// them (memory-bound, B*h elements), forward and backward.
#include <stdlib.h>
#include "a2c_common.h"

namespace {
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void gru_gates_kernel(const float* __restrict__ gx, const float* __restrict__ gh,
                                                        const float* __restrict__ b, const float* __restrict__ h,
                                                        float* __restrict__ z, float* __restrict__ r,
                                                        float* __restrict__ rh, long B, int hd) {
  const long n = B * hd;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const long row = i / hd;
    const int c = (int)(i - row * hd);
    const float zz = sigmoidf_((gx[row * 3 * hd + c] + gh[row * 2 * hd + c]) + b[c]);
    const float rr = sigmoidf_((gx[row * 3 * hd + hd + c] + gh[row * 2 * hd + hd + c]) + b[hd + c]);
    z[i] = zz;
    r[i] = rr;
    rh[i] = rr * h[i];
  }
}

__global__ __launch_bounds__(256) void gru_out_kernel(const float* __restrict__ gx, const float* __restrict__ rhu,
                                                      const float* __restrict__ b, const float* h,
                                                      const float* __restrict__ z, float* __restrict__ cnd,
                                                      float* hn, long B, int hd) {      // hn may alias h (in-place rollout step)
  const long n = B * hd;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const long row = i / hd;
    const int c = (int)(i - row * hd);
    const float cc = tanhf((gx[row * 3 * hd + 2 * hd + c] + rhu[i]) + b[2 * hd + c]);
    const float zz = z[i];
    if (cnd) cnd[i] = cc;
    hn[i] = zz * h[i] + (1.f - zz) * cc;
  }
}

__global__ __launch_bounds__(256) void gru_out_bwd_kernel(const float* __restrict__ dhn, const float* __restrict__ h,
                                                          const float* __restrict__ z, const float* __restrict__ c,
                                                          float* __restrict__ dc_pre, float* __restrict__ dz,
                                                          float* __restrict__ dh, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float g = dhn[i], zz = z[i], cc = c[i];
    dc_pre[i] = g * (1.f - zz) * (1.f - cc * cc);
    dz[i] = g * (h[i] - cc);
    dh[i] = g * zz;
  }
}

// the same with the BPTT carry folded in (updater.py:161-166 unrolled backwards): the total gradient of this step's h_new is
// dh_new + carry * (1 - done), carry = the gradient that reached the NEXT step's h_in = h_new * (1 - done)
__global__ __launch_bounds__(256) void gru_out_bwd_carry_kernel(const float* __restrict__ dhn, const float* carry,
                                                                const float* __restrict__ dones, long dstride,
                                                                const float* __restrict__ h, const float* __restrict__ z,
                                                                const float* __restrict__ c, float* __restrict__ dc_pre,
                                                                float* __restrict__ dz, float* dh, long n, int hd) {
  // (carry may BE dh: element i is read before it is written, by the same thread)
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const long row = i / hd;
    const float g = dhn[i] + carry[i] * (1.f - dones[row * dstride]), zz = z[i], cc = c[i];
    dc_pre[i] = g * (1.f - zz) * (1.f - cc * cc);
    dz[i] = g * (h[i] - cc);
    dh[i] = g * zz;
  }
}

__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float* __restrict__ d_rh,
                                                            const float* __restrict__ dz, const float* __restrict__ h,
                                                            const float* __restrict__ z, const float* __restrict__ r,
                                                            float* __restrict__ dz_pre, float* __restrict__ dr_pre,
                                                            float* __restrict__ dh, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const float zz = z[i], rr = r[i], drh = d_rh[i];
    dz_pre[i] = dz[i] * zz * (1.f - zz);
    dr_pre[i] = drh * h[i] * rr * (1.f - rr);
    dh[i] += drh * rr;
  }
}


// ---------------------------------------------------------------------------------------------------------------
// The whole cell of a rollout step in TWO launches (models.py:465-476 at batch n_envs): the five launches above it
// replaces -- x [Wx0|Wx1|Wx2], h [Wh0|Wh1], gates, (r*h) Wh2, out -- are 5-6 us each at 256 envs, i.e. launch-bound.
//   gru_cell_zr_kernel   column tile of gate g in {z, r, candidate}: waves 0-3 the x-side product (K split four ways),
//                        waves 4-7 the h-side product (z, r only); epilogue z / r / r*h, or the candidate's x-side sum
//   gru_cell_out_kernel  (r*h) Wh2 tile (K split four ways) + tanh + the convex combination -> c, h_new
// Same products, same K split, same MFMA order and the same order of additions as a2c_gemm_f32's small-product kernel
// followed by gru_gates_kernel / gru_out_kernel: bit-identical to the five launches (test).
using f32x16 = __attribute__((ext_vector_type(16))) float;

// one wave's share of a 32 x 32 tile of A (M x K, row-major, lda) . B (K x N, row-major, ldb): K range [kbeg, kend)
__device__ __forceinline__ f32x16 gru_tile_part(const float* __restrict__ arow, const float* __restrict__ bcol, long ldb,
                                                long kbeg, long kend) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  long k = kbeg;
  if ((kend - k) & 15) {                             // head: one 8-row group
    const float4 a1 = *reinterpret_cast<const float4*>(arow + k);
    const float b0 = bcol[k * ldb], b1 = bcol[(k + 1) * ldb], b2 = bcol[(k + 2) * ldb], b3 = bcol[(k + 3) * ldb];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b3, acc, 0, 0, 0);
    k += 8;
  }
  for (; k < kend; k += 16) {                        // two 8-row groups in flight
    float4 av[2];
    float bv[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      av[u] = *reinterpret_cast<const float4*>(arow + k + 8 * u);
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[u][j] = bcol[(k + 8 * u + j) * ldb];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u][3], acc, 0, 0, 0);
    }
  }
  return acc;
}

// NWK = waves per product (K split): 4 reproduces the five launches bit for bit; 8 (default) halves the chain of dependent
// operand loads per wave (the products are latency-bound: 15 -> ~9 us for the gate kernel at 256 envs) and is the same sum
// in another order (A2C_GRU_K4=1 selects 4).
template <int NWK>
__global__ __launch_bounds__(128 * NWK) void gru_cell_zr_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ h,
                                                                const float* __restrict__ WxC, const float* __restrict__ WhC,
                                                                const float* __restrict__ b, float* __restrict__ gx,
                                                                float* __restrict__ z, float* __restrict__ r, float* __restrict__ rh,
                                                                long M, int xs, int hd) {
  __shared__ __attribute__((aligned(16))) float red[2 * NWK][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const int tiles = hd / 32;
  const int gate = blockIdx.x / tiles;
  const long n0 = (long)(blockIdx.x - gate * tiles) * 32, m0 = (long)blockIdx.y * 32;
  const bool xside = w < NWK;
  if (xside || gate < 2) {
    const long K = xside ? xs : hd;
    const int wk = xside ? w : w - NWK;
    const long kq = ((K / 8 + NWK - 1) / NWK) * 8;
    const long kbeg = min(K, wk * kq), kend = min(K, kbeg + kq);
    const float* __restrict__ arow = (xside ? x + min(m0 + li, M - 1) * ldx : h + min(m0 + li, M - 1) * hd) + 4 * lk;
    const long ldb = xside ? 3L * hd : 2L * hd;
    const float* __restrict__ bcol = (xside ? WxC : WhC) + (long)gate * hd + n0 + li + (long)(4 * lk) * ldb;
    const f32x16 acc = gru_tile_part(arow, bcol, ldb, kbeg, kend);
#pragma unroll
    for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  }
  __syncthreads();
  // wave w finishes registers w * 16 / (2 NWK) .. of every lane: col n = lane & 31, row = (q&3) + 8*(q>>2) + 4*(lane>>5)
  constexpr int PER = 16 / (2 * NWK) > 0 ? 16 / (2 * NWK) : 1;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int q = w * PER + e;
    if (q >= 16) continue;
    float gxv = red[0][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) gxv += red[ww][q][lane];
    const long m = m0 + (q & 3) + 8 * (q >> 2) + 4 * lk, n = n0 + li;
    if (m >= M) continue;
    if (gate == 2) {
      gx[m * 3 * hd + 2 * hd + n] = gxv;
      continue;
    }
    float ghv = red[NWK][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) ghv += red[NWK + ww][q][lane];
    const float v = sigmoidf_((gxv + ghv) + b[gate * hd + n]);
    if (gate == 0) z[m * hd + n] = v;
    else {
      r[m * hd + n] = v;
      rh[m * hd + n] = v * h[m * hd + n];
    }
  }
}

template <int NWK>
__global__ __launch_bounds__(64 * NWK) void gru_cell_out_kernel(const float* __restrict__ gx, const float* __restrict__ rh,
                                                                const float* __restrict__ Wh2, const float* __restrict__ b,
                                                                const float* h, const float* __restrict__ z, float* __restrict__ cnd,
                                                                float* hn, long M, int hd) {      // hn may alias h
  __shared__ __attribute__((aligned(16))) float red[NWK][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long n0 = (long)blockIdx.x * 32, m0 = (long)blockIdx.y * 32;
  const long K = hd;
  const long kq = ((K / 8 + NWK - 1) / NWK) * 8;
  const long kbeg = min(K, w * kq), kend = min(K, kbeg + kq);
  const f32x16 acc = gru_tile_part(rh + min(m0 + li, M - 1) * hd + 4 * lk, Wh2 + n0 + li + (long)(4 * lk) * hd, hd, kbeg, kend);
#pragma unroll
  for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16 / NWK; ++e) {
    const int q = w * (16 / NWK) + e;
    float v = red[0][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) v += red[ww][q][lane];
    const long m = m0 + (q & 3) + 8 * (q >> 2) + 4 * lk, n = n0 + li;
    if (m >= M) continue;
    const long i = m * hd + n;
    const float cc = tanhf((gx[m * 3 * hd + 2 * hd + n] + v) + b[2 * hd + n]);
    const float zz = z[i];
    if (cnd) cnd[i] = cc;
    hn[i] = zz * h[i] + (1.f - zz) * cc;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// One step of the BPTT unroll's backward (updater.py:139-169 differentiated; the reference leaves it to autograd) in TWO
// launches instead of five -- gru_out_bwd[_carry], (dc_pre) Wh2^T, gru_gates_bwd, (dz_pre) Wh0^T, (dr_pre) Wh1^T: 128
// strictly serial steps of ~4.5 us per launch are 2.9 ms of a 37 ms update.
//   gru_cell_bwd1_kernel  tile (32 rows x 32 columns) of d_rh = dc_pre Wh2^T with dc_pre = g (1 - z) (1 - c^2),
//                         g = dh_new + carry (1 - done), computed on the fly as the A operand (column block 0 also writes
//                         dc_pre and dz); epilogue: dz_pre, dr_pre and dh = g z + d_rh r of its tile
//   gru_cell_bwd2_kernel  dh += dz_pre Wh0^T, then += dr_pre Wh1^T
// Same products, K split, MFMA order and order of additions as the five launches (a2c_gemm_f32's small-product kernel
// with k-contiguous B): bit-identical (test).  carry must not alias dh (the caller ping-pongs two buffers).
__device__ __forceinline__ void gru_mfma4(f32x16& acc, const float4 a, const float4 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
}
// wave w's share of A (rows, k-contiguous) . B^T (B rows = output columns, k-contiguous), K range [kbeg, kend)
__device__ __forceinline__ f32x16 gru_tile_part_kc(const float* __restrict__ arow, const float* __restrict__ brow, long kbeg, long kend) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (long k = kbeg; k + 8 <= kend; k += 8)
    gru_mfma4(acc, *reinterpret_cast<const float4*>(arow + k), *reinterpret_cast<const float4*>(brow + k));
  return acc;
}

template <int NWK>
__global__ __launch_bounds__(64 * NWK) void gru_cell_bwd1_kernel(const float* __restrict__ dhn, const float* __restrict__ carry,
                                                            const float* __restrict__ dones, long dstride,
                                                            const float* __restrict__ h, const float* __restrict__ z,
                                                            const float* __restrict__ r, const float* __restrict__ c,
                                                            const float* __restrict__ Wh2, float* __restrict__ dcp,
                                                            float* __restrict__ dz, float* __restrict__ dzp, float* __restrict__ drp,
                                                            float* __restrict__ dh, long M, int hd) {
  __shared__ __attribute__((aligned(16))) float red[NWK][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long n0 = (long)blockIdx.x * 32, m0 = (long)blockIdx.y * 32;
  const long K = hd;
  const long kq = ((K / 8 + NWK - 1) / NWK) * 8;
  const long kbeg = min(K, w * kq), kend = min(K, kbeg + kq);
  const long mrow = min(m0 + li, M - 1);
  const float keep = carry ? 1.f - dones[mrow * dstride] : 0.f;
  const float* __restrict__ brow = Wh2 + min(n0 + li, (long)hd - 1) * hd + 4 * lk;
  f32x16 acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  const bool writer = blockIdx.x == 0 && m0 + li < M;
  for (long k = kbeg; k + 8 <= kend; k += 8) {
    const long i = mrow * hd + k + 4 * lk;
    const float4 gd = *reinterpret_cast<const float4*>(dhn + i);
    const float4 zz = *reinterpret_cast<const float4*>(z + i), cc = *reinterpret_cast<const float4*>(c + i);
    float4 g = gd;
    if (carry) {
      const float4 cr = *reinterpret_cast<const float4*>(carry + i);
      g.x = gd.x + cr.x * keep; g.y = gd.y + cr.y * keep; g.z = gd.z + cr.z * keep; g.w = gd.w + cr.w * keep;
    }
    float4 a;
    a.x = g.x * (1.f - zz.x) * (1.f - cc.x * cc.x);
    a.y = g.y * (1.f - zz.y) * (1.f - cc.y * cc.y);
    a.z = g.z * (1.f - zz.z) * (1.f - cc.z * cc.z);
    a.w = g.w * (1.f - zz.w) * (1.f - cc.w * cc.w);
    if (writer) {
      const float4 hh = *reinterpret_cast<const float4*>(h + i);
      *reinterpret_cast<float4*>(dcp + i) = a;
      *reinterpret_cast<float4*>(dz + i) = make_float4(g.x * (hh.x - cc.x), g.y * (hh.y - cc.y), g.z * (hh.z - cc.z), g.w * (hh.w - cc.w));
    }
    gru_mfma4(acc, a, *reinterpret_cast<const float4*>(brow + k));
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  __syncthreads();
#pragma unroll
  for (int e = 0; e < 16 / NWK; ++e) {
    const int q = w * (16 / NWK) + e;
    float drh = red[0][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) drh += red[ww][q][lane];
    const long m = m0 + (q & 3) + 8 * (q >> 2) + 4 * lk, n = n0 + li;
    if (m >= M) continue;
    const long i = m * hd + n;
    const float g = carry ? dhn[i] + carry[i] * (1.f - dones[m * dstride]) : dhn[i];
    const float zz = z[i], rr = r[i], hh = h[i], cc = c[i];
    const float dzv = g * (hh - cc);
    dzp[i] = dzv * zz * (1.f - zz);
    drp[i] = drh * hh * rr * (1.f - rr);
    float d = g * zz;
    d += drh * rr;
    dh[i] = d;
  }
}

template <int NWK>
__global__ __launch_bounds__(128 * NWK) void gru_cell_bwd2_kernel(const float* __restrict__ dzp, const float* __restrict__ drp,
                                                            const float* __restrict__ Wh0, const float* __restrict__ Wh1,
                                                            float* __restrict__ dh, long M, int hd) {
  __shared__ __attribute__((aligned(16))) float red[2 * NWK][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long n0 = (long)blockIdx.x * 32, m0 = (long)blockIdx.y * 32;
  const long K = hd;
  const long kq = ((K / 8 + NWK - 1) / NWK) * 8;
  const int wk = w < NWK ? w : w - NWK;
  const long kbeg = min(K, wk * kq), kend = min(K, kbeg + kq);
  const float* __restrict__ arow = (w < NWK ? dzp : drp) + min(m0 + li, M - 1) * hd + 4 * lk;
  const float* __restrict__ brow = (w < NWK ? Wh0 : Wh1) + min(n0 + li, (long)hd - 1) * hd + 4 * lk;
  const f32x16 acc = gru_tile_part_kc(arow, brow, kbeg, kend);
#pragma unroll
  for (int q = 0; q < 16; ++q) red[w][q][lane] = acc[q];
  __syncthreads();
  constexpr int PER = 16 / (2 * NWK) > 0 ? 16 / (2 * NWK) : 1;
#pragma unroll
  for (int e = 0; e < PER; ++e) {
    const int q = w * PER + e;
    if (q >= 16) continue;
    float v0 = red[0][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) v0 += red[ww][q][lane];
    float v1 = red[NWK][q][lane];
#pragma unroll
    for (int ww = 1; ww < NWK; ++ww) v1 += red[NWK + ww][q][lane];
    const long m = m0 + (q & 3) + 8 * (q >> 2) + 4 * lk, n = n0 + li;
    if (m >= M) continue;
    v0 += dh[m * hd + n];
    v1 += v0;
    dh[m * hd + n] = v1;
  }
}

// one wave per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd,
                                                            long rows, int n) {
  const int lane = threadIdx.x & 63;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4L) {
    const float* xr = x + row * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += xr[i];
    const float m = wave_sum(s) / (float)n;
    float v = 0.f;
    for (int i = lane; i < n; i += 64) { const float d = xr[i] - m; v += d * d; }
    const float rs = 1.0f / sqrtf(wave_sum(v) / (float)n + 1e-5f);
    for (int i = lane; i < n; i += 64) y[row * n + i] = (xr[i] - m) * rs * w[i] + b[i];
    if (lane == 0) { mean[row] = m; rstd[row] = rs; }
  }
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dw_rows, long rows, int n, int accum) {
  const int lane = threadIdx.x & 63;
  for (long row = blockIdx.x * 4L + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4L) {
    const float m = mean[row], rs = rstd[row];
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < n; i += 64) {
      const float xh = (x[row * n + i] - m) * rs;
      const float g = dy[row * n + i] * w[i];
      s1 += g;
      s2 += g * xh;
    }
    s1 = wave_sum(s1) / (float)n;
    s2 = wave_sum(s2) / (float)n;
    for (int i = lane; i < n; i += 64) {
      const float xh = (x[row * n + i] - m) * rs;
      const float d = dy[row * n + i];
      const float v = rs * (d * w[i] - s1 - xh * s2);
      if (accum) dx[row * n + i] += v; else dx[row * n + i] = v;
      dw_rows[row * n + i] = d * xh;
    }
  }
}
}  // namespace

// A2C_GRU_K4=1 (read per call): the cell kernels with a four-way K split, bit-identical to the launch sequences they replace
static bool gru_k4() {
  const char* e = getenv("A2C_GRU_K4");
  return e != nullptr && e[0] == '1';
}

extern "C" {
int a2c_gru_gates(const float* gx, const float* gh, const float* b, const float* h, float* z, float* r, float* rh,
                  int B, int hdim, a2c_stream_t stream) {
  if (B < 0 || hdim < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!gx || !gh || !b || !h || !z || !r || !rh) return A2C_ERR_ARG;
  hipLaunchKernelGGL(gru_gates_kernel, dim3(a2c_grid_1d((long)B * hdim, 256)), dim3(256), 0, a2c_s(stream), gx, gh, b,
                     h, z, r, rh, (long)B, hdim);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_out(const float* gx, const float* rh_u, const float* b, const float* h, const float* z, float* c,
                float* h_new, int B, int hdim, a2c_stream_t stream) {
  if (B < 0 || hdim < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!gx || !rh_u || !b || !h || !z || !h_new) return A2C_ERR_ARG;
  hipLaunchKernelGGL(gru_out_kernel, dim3(a2c_grid_1d((long)B * hdim, 256)), dim3(256), 0, a2c_s(stream), gx, rh_u, b,
                     h, z, c, h_new, (long)B, hdim);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_cell_fwd(const float* x, int64_t ldx, const float* h, const float* WxC, const float* WhC, const float* Wh2,
                     const float* b, float* gx, float* z, float* r, float* rh, float* c, float* h_new, int B, int xs, int hdim,
                     a2c_stream_t stream) {
  if (B < 0 || hdim < 32 || hdim % 32 || xs < 8 || xs % 8 || hdim % 8 || ldx < xs || ldx % 4) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!x || !h || !WxC || !WhC || !Wh2 || !b || !gx || !z || !r || !rh || !h_new) return A2C_ERR_ARG;
  if ((((uintptr_t)x | (uintptr_t)h | (uintptr_t)rh) % 16)) return A2C_ERR_ARG;
  const int tiles = hdim / 32, rows = (B + 31) / 32;
  if (gru_k4()) {
    hipLaunchKernelGGL((gru_cell_zr_kernel<4>), dim3(3 * tiles, rows), dim3(512), 0, a2c_s(stream), x, (long)ldx, h, WxC, WhC, b, gx, z,
                       r, rh, (long)B, xs, hdim);
    A2C_CHECK_LAUNCH();
    hipLaunchKernelGGL((gru_cell_out_kernel<4>), dim3(tiles, rows), dim3(256), 0, a2c_s(stream), gx, rh, Wh2, b, h, z, c, h_new,
                       (long)B, hdim);
  } else {
    hipLaunchKernelGGL((gru_cell_zr_kernel<8>), dim3(3 * tiles, rows), dim3(1024), 0, a2c_s(stream), x, (long)ldx, h, WxC, WhC, b, gx, z,
                       r, rh, (long)B, xs, hdim);
    A2C_CHECK_LAUNCH();
    hipLaunchKernelGGL((gru_cell_out_kernel<8>), dim3(tiles, rows), dim3(512), 0, a2c_s(stream), gx, rh, Wh2, b, h, z, c, h_new,
                       (long)B, hdim);
  }
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_cell_bwd(const float* dh_new, const float* carry, const float* dones, int64_t done_stride, const float* h,
                     const float* z, const float* r, const float* c, const float* Wh, float* dc_pre, float* dz, float* dz_pre,
                     float* dr_pre, float* dh, int B, int hdim, a2c_stream_t stream) {
  if (B < 0 || hdim < 32 || hdim % 32 || (carry && (!dones || done_stride < 1))) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dh_new || !h || !z || !r || !c || !Wh || !dc_pre || !dz || !dz_pre || !dr_pre || !dh || carry == dh) return A2C_ERR_ARG;
  if ((((uintptr_t)dh_new | (uintptr_t)carry | (uintptr_t)h | (uintptr_t)z | (uintptr_t)c | (uintptr_t)Wh | (uintptr_t)dc_pre |
        (uintptr_t)dz | (uintptr_t)dz_pre | (uintptr_t)dr_pre) % 16))
    return A2C_ERR_ARG;
  const int tiles = hdim / 32, rows = (B + 31) / 32;
  const long hh = (long)hdim * hdim;
  if (gru_k4()) {
    hipLaunchKernelGGL((gru_cell_bwd1_kernel<4>), dim3(tiles, rows), dim3(256), 0, a2c_s(stream), dh_new, carry, dones, (long)done_stride,
                       h, z, r, c, Wh + 2 * hh, dc_pre, dz, dz_pre, dr_pre, dh, (long)B, hdim);
    A2C_CHECK_LAUNCH();
    hipLaunchKernelGGL((gru_cell_bwd2_kernel<4>), dim3(tiles, rows), dim3(512), 0, a2c_s(stream), dz_pre, dr_pre, Wh, Wh + hh, dh, (long)B,
                       hdim);
  } else {
    hipLaunchKernelGGL((gru_cell_bwd1_kernel<8>), dim3(tiles, rows), dim3(512), 0, a2c_s(stream), dh_new, carry, dones, (long)done_stride,
                       h, z, r, c, Wh + 2 * hh, dc_pre, dz, dz_pre, dr_pre, dh, (long)B, hdim);
    A2C_CHECK_LAUNCH();
    hipLaunchKernelGGL((gru_cell_bwd2_kernel<8>), dim3(tiles, rows), dim3(1024), 0, a2c_s(stream), dz_pre, dr_pre, Wh, Wh + hh, dh, (long)B,
                       hdim);
  }
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_out_bwd(const float* dh_new, const float* h, const float* z, const float* c, float* dc_pre, float* dz,
                    float* dh, int B, int hdim, a2c_stream_t stream) {
  if (B < 0 || hdim < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dh_new || !h || !z || !c || !dc_pre || !dz || !dh) return A2C_ERR_ARG;
  hipLaunchKernelGGL(gru_out_bwd_kernel, dim3(a2c_grid_1d((long)B * hdim, 256)), dim3(256), 0, a2c_s(stream), dh_new,
                     h, z, c, dc_pre, dz, dh, (long)B * hdim);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_out_bwd_carry(const float* dh_new, const float* carry, const float* dones, int64_t done_stride, const float* h,
                          const float* z, const float* c, float* dc_pre, float* dz, float* dh, int B, int hdim,
                          a2c_stream_t stream) {
  if (B < 0 || hdim < 1 || done_stride < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dh_new || !carry || !dones || !h || !z || !c || !dc_pre || !dz || !dh) return A2C_ERR_ARG;
  hipLaunchKernelGGL(gru_out_bwd_carry_kernel, dim3(a2c_grid_1d((long)B * hdim, 256)), dim3(256), 0, a2c_s(stream), dh_new,
                     carry, dones, (long)done_stride, h, z, c, dc_pre, dz, dh, (long)B * hdim, hdim);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gru_gates_bwd(const float* d_rh, const float* dz, const float* h, const float* z, const float* r,
                      float* dz_pre, float* dr_pre, float* dh, int B, int hdim, a2c_stream_t stream) {
  if (B < 0 || hdim < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!d_rh || !dz || !h || !z || !r || !dz_pre || !dr_pre || !dh) return A2C_ERR_ARG;
  hipLaunchKernelGGL(gru_gates_bwd_kernel, dim3(a2c_grid_1d((long)B * hdim, 256)), dim3(256), 0, a2c_s(stream), d_rh,
                     dz, h, z, r, dz_pre, dr_pre, dh, (long)B * hdim);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                      int64_t rows, int n, a2c_stream_t stream) {
  if (rows < 0 || n < 1) return A2C_ERR_ARG;
  if (rows == 0) return A2C_OK;
  if (!x || !w || !b || !y || !mean || !rstd) return A2C_ERR_ARG;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(a2c_grid_1d(rows, 4)), dim3(256), 0, a2c_s(stream), x, w, b, y, mean,
                     rstd, (long)rows, n);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_layernorm_bwd(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                      float* dx, float* dw_rows, int64_t rows, int n, int accumulate_dx, a2c_stream_t stream) {
  if (rows < 0 || n < 1) return A2C_ERR_ARG;
  if (rows == 0) return A2C_OK;
  if (!dy || !x || !w || !mean || !rstd || !dx || !dw_rows) return A2C_ERR_ARG;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(a2c_grid_1d(rows, 4)), dim3(256), 0, a2c_s(stream), dy, x, w, mean,
                     rstd, dx, dw_rows, (long)rows, n, accumulate_dx);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
