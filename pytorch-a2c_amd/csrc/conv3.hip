// 3x3 / pad 1 convolutions of the conv-stack models (models.py:201-264 ConvModel, 570-636 GRUModel) at 84x84 and its
// stride-2 descendants: forward and stride-1 backward-data as ONE "channel-chunk streaming" implicit GEMM.
//
// Why another kernel family next to conv.hip's generic ones: the PMC profile of those (profiles/r2_pmc_generic_conv.json)
// shows 8.6-10 VALU instructions per MFMA, 34-39 % LDS bank conflicts and 53-61 % of the wave time parked on staging
// waits -- every offset there is computed at run time from a runtime descriptor, staging goes through registers and
// the whole K (all input planes) of a band sits in LDS.  Here:
//   * the layer shape is a TEMPLATE parameter: every tap / channel offset of the MFMA loop is an immediate of the
//     ds_read, the loop is straight-line code (no VALU address arithmetic per step: < 1 VALU per MFMA);
//   * K is streamed in CHUNKS of 8 source channels: a chunk's band image (rows with halo, 8 planes) and its weight
//     fragments are one LDS buffer of 37-56 KB, double buffered; the accumulators of a wave's pixel tiles live in
//     registers across the chunks.  LDS per workgroup no longer grows with the channel count;
//   * a chunk is brought in by LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, no staging registers) by a
//     DEDICATED loader wave (the ninth wave of the workgroup) one chunk ahead of the eight computing waves: its
//     vmcnt(0) waits never see the computing waves' output stores (loads and stores retire through one in-order counter
//     per wave), and one barrier per chunk is the whole synchronisation;
//   * image rows are 4 floats of left pad + W floats (W % 4 == 0): the pad piece comes from a zero page through the
//     DMA's per-lane source address and serves as the left halo of its row AND the right halo of the row above it, rows
//     outside the sample are read from the zero page as well: no border logic in the MFMA loop;
//   * outputs go straight from the accumulators to HBM (D layout: 16 consecutive pixels of one channel per 16 lanes =
//     64-B segments, adjacent tiles of a wave complete the 128-B lines in L2): bias + ReLU (forward) or the ReLU mask of
//     the layer below (backward-data) applied in registers.
// v_mfma_f32_16x16x4_f32: M = 16 destination channels, N = 16 pixels, K-step = 4 source channels at one tap; a wave
// owns TPW pixel tiles x MT channel tiles of accumulators and reads every A fragment once per step for all of them.
// Sums are ordered (chunk, tap, channel quad): fixed, deterministic, a re-association of conv.hip's (tap, quad) order.
#include <stdlib.h>
#include <mutex>
#include "a2c_common.h"

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

struct C3P {
  const float* src; long src_bs;      // source tensor (B, CS, H, W) and its sample stride (floats)
  const float* frag;                  // [chunk][tap][c4][m][64] weight fragments (c3_prep_kernel)
  const float* bias;                  // forward: (CD) or nullptr
  const float* mask;                  // backward-data: (B, CD, H, W) activation whose sign gates the gradient, or nullptr
  float* out; long out_bs;            // (B, CD, OH, OW)
  const float* zero;                  // >= 64 B of zeros in HBM (16-B aligned)
  int B, relu;
};

template <int CS, int CD, int H, int W, int S, int R>
struct C3Geo {
  static constexpr int KC = CS >= 8 ? 8 : 4;          // source channels per chunk
  static constexpr int NCH = CS / KC;
  static constexpr int C4 = KC / 4;
  static constexpr int MT = (CD + 15) / 16;
  static constexpr int OH = (H - 1) / S + 1, OW = (W - 1) / S + 1;
  static constexpr int NBAND = (OH + R - 1) / R;
  static constexpr int NPIX = R * OW;
  static constexpr int NT = (NPIX + 15) / 16;         // 16-pixel tiles per band
  static constexpr int NW = 8;                         // computing waves (two per SIMD); wave NW is the loader
  static constexpr int TPW = (NT + NW - 1) / NW;      // tiles per computing wave
  static constexpr int SR = (R - 1) * S + 3;          // source rows per band (with halo)
  static constexpr int WP = W + 4;                    // 4 floats of pad + the row
  static constexpr int PL0 = SR * WP;
  // planes 16 (mod 32) floats apart: the two k-groups of a 32-lane LDS access fall on disjoint banks (stride 1)
  static constexpr int PLANE = S == 1 ? ((PL0 + 15) / 32) * 32 + 16 : PL0;
  static constexpr int PP = PL0 / 4;                  // 16-byte pieces per plane
  static constexpr int NQ = (PP + 63) / 64;           // DMA instructions per plane
  static constexpr int IMG = KC * PLANE + 8;          // + zeros behind the last plane (right halo of its last row)
  static constexpr int FRAGC = C4 * 9 * MT * 64;      // fragment floats per chunk
  static constexpr int NFQ = (FRAGC / 4 + 63) / 64;
  static constexpr int BUF = ((IMG + FRAGC + 255) / 256) * 256;
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * 4;
  static_assert(W % 4 == 0 && CS % KC == 0 && PL0 % 4 == 0, "shape");
  static_assert(S == 2 || PLANE == PL0, "stride 1 reads the right halo of a plane's last row from the next plane's pad piece");
};

// weights -> fragments.  forward: A[m*16 + i][k] = W[co = m*16+i][ci = ch*KC + c4*4 + g][ty][tx];
// backward-data (stride 1): the correlation over dOut with the flipped kernel,
//   A[m*16 + i][k] = W[co = ch*KC + c4*4 + g][ci = m*16+i][2-ty][2-tx]       (lane = g*16 + i)
__global__ __launch_bounds__(256) void c3_prep_kernel(const float* __restrict__ Wt, float* __restrict__ out, int Cin, int Cout,
                                                      int bwd, int KC, int MT, long total) {
  const int CS = bwd ? Cout : Cin, CD = bwd ? Cin : Cout;
  const int C4 = KC / 4;
  for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += gridDim.x * 256L) {
    const int l = (int)(q & 63);
    long r = q >> 6;
    const int m = (int)(r % MT); r /= MT;
    const int c4 = (int)(r % C4); r /= C4;
    const int tap = (int)(r % 9);
    const int ch = (int)(r / 9);
    const int ty = tap / 3, tx = tap - 3 * ty;
    const int cs = ch * KC + c4 * 4 + (l >> 4), cd = m * 16 + (l & 15);
    float v = 0.f;
    if (cd < CD && cs < CS) {
      if (bwd) v = Wt[(((long)cs * Cin + cd) * 3 + (2 - ty)) * 3 + (2 - tx)];
      else v = Wt[(((long)cd * Cin + cs) * 3 + ty) * 3 + tx];
    }
    out[q] = v;
  }
}

template <int CS, int CD, int H, int W, int S, int R, bool BWD>
__global__ __launch_bounds__(576) void c3_kernel(C3P p) {
  using G = C3Geo<CS, CD, H, W, S, R>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;              // work item k = (tile k / NCH, chunk k % NCH)
  // the zeros that never change: behind each buffer's last plane
  if (tid < 16) {
    lds[G::KC * G::PLANE + (tid & 7)] = 0.f;
    lds[G::BUF + G::KC * G::PLANE + (tid & 7)] = 0.f;
  }
  if (w == G::NW) {
    // ------------------------------------------------------------------ loader wave
    int roff[G::NQ], rrow[G::NQ];                 // this lane's piece of DMA instruction q: source offset / image row
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / (G::WP / 4), i = pi - r * (G::WP / 4);
      rrow[q] = (pi < G::PP) ? (i == 0 ? -100000 : r) : -200000;      // pad piece: zeros; beyond the plane: no lane
      roff[q] = r * W + 4 * (i - 1);
    }
    auto dma = [&](long k) {
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const int ch = (int)(k % G::NCH);
      const long b = tile / G::NBAND;
      const int band = (int)(tile - b * G::NBAND);
      const int y0 = band * R * S - 1;
      float* __restrict__ buf = lds + (k & 1) * G::BUF;
      const float* __restrict__ sb = p.src + b * p.src_bs + ((long)ch * G::KC * H + y0) * W;
#pragma unroll
      for (int c = 0; c < G::KC; ++c) {
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
          if (rrow[q] > -200000) {
            const int y = y0 + rrow[q];
            const float* gsrc = (rrow[q] >= 0 && y >= 0 && y < H) ? sb + (long)c * H * W + roff[q] : p.zero;
            __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 256), 16, 0, 0);
          }
        }
      }
      const float* __restrict__ fg = p.frag + (long)ch * G::FRAGC;
#pragma unroll
      for (int q = 0; q < G::NFQ; ++q) {
        const int pi = q * 64 + lane;
        if (pi < G::FRAGC / 4)
          __builtin_amdgcn_global_load_lds((gptr_t)(fg + pi * 4), (lptr_t)(buf + G::IMG + q * 256), 16, 0, 0);
      }
    };
    if (nwork > 0) dma(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the chunk has landed in LDS
    __syncthreads();
    for (long k = 0; k < nwork; ++k) {
      if (k + 1 < nwork) dma(k + 1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    return;
  }
  // -------------------------------------------------------------------- computing waves
  int base[G::TPW];
#pragma unroll
  for (int u = 0; u < G::TPW; ++u) {
    const int t = w + G::NW * u;
    const int pp = t * 16 + j;
    const bool ok = t < G::NT && pp < G::NPIX;
    const int px = ok ? pp : 0;
    const int r = px / G::OW, x = px - r * G::OW;
    base[u] = g * G::PLANE + r * S * G::WP + x * S + 3;
  }
  f32x4 acc[G::TPW][G::MT];
#pragma unroll
  for (int u = 0; u < G::TPW; ++u)
#pragma unroll
    for (int m = 0; m < G::MT; ++m) acc[u][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  for (long k = 0; k < nwork; ++k) {
    const float* __restrict__ img = lds + (k & 1) * G::BUF;
    const float* __restrict__ fr = img + G::IMG + lane;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
      for (int c4 = 0; c4 < G::C4; ++c4) {
        float av[G::MT];
#pragma unroll
        for (int m = 0; m < G::MT; ++m) av[m] = fr[((tap * G::C4 + c4) * G::MT + m) * 64];
#pragma unroll
        for (int u = 0; u < G::TPW; ++u) {
          const float bv = img[base[u] + c4 * 4 * G::PLANE + (tap / 3) * G::WP + (tap % 3)];
#pragma unroll
          for (int m = 0; m < G::MT; ++m) acc[u][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[m], acc[u][m], 0, 0, 0);
        }
        if (G::TPW * G::MT >= 12) __builtin_amdgcn_sched_barrier(0);      // keep the scheduler from hoisting later steps' reads (registers)
      }
    }
    if ((int)(k % G::NCH) == G::NCH - 1) {
      // ---- the band is complete: accumulators -> HBM.  With the PIXELS as the MFMA's A operand the D tile is
      // [pixel][channel]: lane (j, g) holds pixels 4g .. 4g+3 of channel j -- one 16-byte store per lane and tile
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const long b = tile / G::NBAND;
      const int band = (int)(tile - b * G::NBAND);
      const int npix_ok = min(R, G::OH - band * R) * G::OW;
      const long o0 = b * p.out_bs + (long)band * R * G::OW;
#pragma unroll
      for (int u = 0; u < G::TPW; ++u) {
        const int t = w + G::NW * u;
        const int p0 = t * 16 + 4 * g;                       // first of this lane's 4 pixels
        float4 mk[G::MT];
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          mk[m] = make_float4(1.f, 1.f, 1.f, 1.f);
          const int cd = m * 16 + j;
          if (BWD && p.mask != nullptr && cd < CD && p0 + 3 < npix_ok)
            mk[m] = *reinterpret_cast<const float4*>(p.mask + o0 + (long)cd * G::OH * G::OW + p0);
        }
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          const int cd = m * 16 + j;
          float4 v = make_float4(acc[u][m][0], acc[u][m][1], acc[u][m][2], acc[u][m][3]);
          if (BWD) {
            if (!(mk[m].x > 0.f)) v.x = 0.f;
            if (!(mk[m].y > 0.f)) v.y = 0.f;
            if (!(mk[m].z > 0.f)) v.z = 0.f;
            if (!(mk[m].w > 0.f)) v.w = 0.f;
          } else {
            const float bs = (p.bias && cd < CD) ? p.bias[cd] : 0.f;
            v.x += bs; v.y += bs; v.z += bs; v.w += bs;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          }
          float* __restrict__ o = p.out + o0 + (long)cd * G::OH * G::OW + p0;
          if (cd < CD) {
            if (p0 + 3 < npix_ok) *reinterpret_cast<float4*>(o) = v;
            else {                                             // the band's ragged end (partial last band)
              if (p0 < npix_ok) o[0] = v.x;
              if (p0 + 1 < npix_ok) o[1] = v.y;
              if (p0 + 2 < npix_ok) o[2] = v.z;
            }
          }
          acc[u][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    __syncthreads();
  }
}

const float* zero_page() {
  static float* z = nullptr;
  static std::once_flag once;
  std::call_once(once, [] {
    if (hipMalloc(&z, 256) != hipSuccess || hipMemset(z, 0, 256) != hipSuccess) z = nullptr;
  });
  return z;
}

template <int CS, int CD, int H, int W, int S, int R, bool BWD>
int c3_launch(const C3P& p, hipStream_t st) {
  using G = C3Geo<CS, CD, H, W, S, R>;
  const void* k = (const void*)c3_kernel<CS, CD, H, W, S, R, BWD>;
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) return A2C_ERR_LAUNCH;
    int n = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 576, G::LDS_BYTES) != hipSuccess || n < 1) n = 1;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    per_cu = n;
  }
  const long total = (long)p.B * G::NBAND;
  const long cap = (long)per_cu * cus;
  const int grid = (int)(total < cap ? total : cap);
  hipLaunchKernelGGL((c3_kernel<CS, CD, H, W, S, R, BWD>), dim3(grid), dim3(576), G::LDS_BYTES, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}
}  // namespace

// ---- what conv.hip's entry points call (not part of the C ABI) -------------------------------------------------
// kind 0 = forward, 1 = backward-data.  c3_supported: this family has an instantiation for the layer.
bool c3_supported(const a2c_conv_desc* d, int kind) {
  static const bool off = getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1';
  if (off || d->ks != 3 || d->pad != 1 || d->H != 84 || d->W != 84) return false;
  if (kind == 0) {
    if (d->stride == 1) return (d->Cin == 4 && d->Cout == 16) || (d->Cin == 16 && d->Cout == 24);
    if (d->stride == 2) return (d->Cin == 24 && d->Cout == 32) || (d->Cin == 16 && d->Cout == 24);
    return false;
  }
  return d->stride == 1 && d->Cin == 16 && d->Cout == 24;
}

size_t c3_prep_floats(const a2c_conv_desc* d, int kind) {
  if (!c3_supported(d, kind)) return 0;
  const int CS = kind ? d->Cout : d->Cin, CD = kind ? d->Cin : d->Cout;
  return (size_t)(CS / 4) * 9 * ((CD + 15) / 16) * 64;
}

int c3_prep(const a2c_conv_desc* d, int kind, const float* weight, float* out, hipStream_t st) {
  const long total = (long)c3_prep_floats(d, kind);
  if (!total) return A2C_OK;
  const int CS = kind ? d->Cout : d->Cin, CD = kind ? d->Cin : d->Cout;
  hipLaunchKernelGGL(c3_prep_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, out, d->Cin, d->Cout, kind,
                     CS >= 8 ? 8 : 4, (CD + 15) / 16, total);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

int c3_fwd(const a2c_conv_desc* d, const float* in, long in_bs, const float* frag, const float* bias, int relu, float* out,
           long out_bs, int B, hipStream_t st) {
  C3P p{in, in_bs, frag, bias, nullptr, out, out_bs, zero_page(), B, relu};
  if (!p.zero) return A2C_ERR_LAUNCH;
  if (d->stride == 1 && d->Cin == 4) return c3_launch<4, 16, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 1 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 2 && d->Cin == 24) return c3_launch<24, 32, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 2, 6, false>(p, st);
  return A2C_ERR_ARG;
}

int c3_bwd_data(const a2c_conv_desc* d, const float* dout, const float* frag, const float* mask, float* din, int B,
                hipStream_t st) {
  C3P p{dout, (long)d->Cout * d->OH * d->OW, frag, nullptr, mask, din, (long)d->Cin * d->H * d->W, zero_page(), B, 0};
  if (!p.zero) return A2C_ERR_LAUNCH;
  if (d->stride == 1 && d->Cin == 16 && d->Cout == 24) return c3_launch<24, 16, 84, 84, 1, 12, true>(p, st);
  return A2C_ERR_ARG;
}
