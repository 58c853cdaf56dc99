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
  unsigned long long* dbg;            // debug: phase stamps of workgroup 0 / wave 0 (a2c_debug_c3_timing), or nullptr
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
  static constexpr int NW = 8;                         // computing waves (two per SIMD); waves NW .. NW+NL-1 are the loaders
  static constexpr int NL = 2;
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
  static constexpr int MROW = R * OW;                  // floats of one channel's output band (contiguous in HBM)
  static constexpr int MASKF = CD * MROW;              // backward-data: the ReLU-mask band, staged in LDS by the loaders
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * 4;
  static constexpr size_t LDS_BYTES_BWD = (2 * (size_t)BUF + MASKF) * 4;
  static_assert(W % 4 == 0 && CS % KC == 0 && PL0 % 4 == 0 && MROW % 4 == 0, "shape");
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
__global__ __launch_bounds__(640) void c3_kernel(C3P p) {
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
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves (planes / fragment pieces dealt round robin)
    const int lw = w - G::NW;
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
        if (c % G::NL != lw) continue;
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
        if (q % G::NL == lw && pi < G::FRAGC / 4)
          __builtin_amdgcn_global_load_lds((gptr_t)(fg + pi * 4), (lptr_t)(buf + G::IMG + q * 256), 16, 0, 0);
      }
    };
    // backward-data: the band's ReLU mask, one linear run per channel, issued while the band's FIRST chunk computes
    auto dma_mask = [&](long k) {      // the slice of the band's mask that travels beside chunk k (all but the last chunk carry one)
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const long b = tile / G::NBAND;
      const int band = (int)(tile - b * G::NBAND);
      const int npx = min(R, G::OH - band * R) * G::OW;
      float* __restrict__ mb = lds + 2 * G::BUF;
      constexpr int CPP = G::NCH > 1 ? (CD + G::NCH - 2) / (G::NCH - 1) : CD;
      const int part = (int)(k % G::NCH);
#pragma unroll 1
      for (int c = part * CPP + lw; c < min(CD, (part + 1) * CPP); c += G::NL) {
        const float* __restrict__ ms = p.mask + b * p.out_bs + (long)c * G::OH * G::OW + (long)band * R * G::OW;
        for (int q = 0; q * 256 < npx; ++q) {
          const int pi = q * 64 + lane;
          if (pi * 4 < npx)
            __builtin_amdgcn_global_load_lds((gptr_t)(ms + pi * 4), (lptr_t)(mb + c * G::MROW + q * 256), 16, 0, 0);
        }
      }
    };
    if (nwork > 0) dma(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the chunk has landed in LDS
    __syncthreads();
    for (long k = 0; k < nwork; ++k) {
      if (k + 1 < nwork) dma(k + 1);
      if (BWD && p.mask != nullptr && (int)(k % G::NCH) < G::NCH - 1) dma_mask(k);
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
  float biasv[G::MT];
#pragma unroll
  for (int m = 0; m < G::MT; ++m) biasv[m] = (!BWD && p.bias && m * 16 + j < CD) ? p.bias[m * 16 + j] : 0.f;
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
      const float* __restrict__ mb = lds + 2 * G::BUF;       // backward-data: the band's mask, staged by the loaders
#pragma unroll
      for (int u = 0; u < G::TPW; ++u) {
        const int t = w + G::NW * u;
        const int p0 = t * 16 + 4 * g;                       // first of this lane's 4 pixels
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          const int cd = m * 16 + j;
          float4 v = make_float4(acc[u][m][0], acc[u][m][1], acc[u][m][2], acc[u][m][3]);
          if (BWD) {
            if (p.mask != nullptr && cd < CD && p0 + 3 < npix_ok) {
              const float4 q = *reinterpret_cast<const float4*>(mb + cd * G::MROW + p0);
              if (!(q.x > 0.f)) v.x = 0.f;
              if (!(q.y > 0.f)) v.y = 0.f;
              if (!(q.z > 0.f)) v.z = 0.f;
              if (!(q.w > 0.f)) v.w = 0.f;
            }
          } else {
            const float bs = biasv[m];
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

// ---------------------------------------------------------------------------------------------------------------
// Backward-data of the STRIDE-2 layers (even input size: dX = 2*HO x 2*WO): the same chunk-streaming skeleton over
// the four output-parity classes.  dX[ci][2q+py][2p+px] is a stride-1 correlation of dOut with the taps of that
// parity: ky = 1 for py = 0 (dOut row q), ky in {0, 2} for py = 1 (rows q+1, q); likewise in x.  All four classes of a
// 16-pixel tile (q, p .. p+15) read the SAME four dOut positions (q + dy, p + dx), so a k-step reads four pixel
// operands per tile and feeds nine MFMAs (one per tap) per channel tile -- and a lane ends up with pixels 4g .. 4g+3
// of both x parities of one channel, i.e. eight consecutive floats of a dX row: two 16-byte stores.
// dOut is the small tensor (a quarter of dX): its band image comes in by 4-byte LDS-DMA pieces (any row width; the
// zero column behind each row is the right halo), the fragments by 16-byte pieces.
template <int CO, int CI, int HO, int WO, int RQ, int KC>
struct C3BGeo {
  static constexpr int H = 2 * HO, W = 2 * WO;
  static constexpr int NCH = CO / KC, C4 = KC / 4, MT = (CI + 15) / 16;
  static constexpr int NBAND = (HO + RQ - 1) / RQ;
  static constexpr int NPIX = RQ * WO;                 // class pixels per band
  static constexpr int NT = (NPIX + 15) / 16;
  static constexpr int NW = 8, NL = 2;
  static constexpr int TP = (NT + NW - 1) / NW;        // tile positions per wave
  static constexpr int WP = WO + 1;                    // + the zero column
  static constexpr int PL0 = (RQ + 1) * WP;
  static constexpr int PLANE = ((PL0 + 15) / 32) * 32 + 16;
  static constexpr int NQ = (PL0 + 63) / 64;           // 4-byte DMA instructions per plane
  static constexpr int IMG = ((KC * PLANE + 3) / 4) * 4;
  static constexpr int FRAGC = C4 * 9 * MT * 64;
  static constexpr int NFQ = (FRAGC / 4 + 63) / 64;
  static constexpr int BUF = ((IMG + FRAGC + 255) / 256) * 256;
  static constexpr int MROW = 2 * RQ * W;              // floats of one channel's dX band (rows are contiguous in HBM)
  static constexpr int MASKF = CI * MROW;              // the ReLU-mask band, brought in by the loaders while the band computes
  static constexpr size_t LDS_BYTES = (2 * (size_t)BUF + MASKF) * 4;
  static_assert(CO % KC == 0 && NCH >= 2 && MROW % 4 == 0, "shape");
};

// fragments of the stride-2 backward pass: [chunk][c4][tap = ky*3+kx][m][64]; lane (j, g): W[co = ch*KC+c4*4+g][ci = m*16+j][ky][kx]
__global__ __launch_bounds__(256) void c3b_prep_kernel(const float* __restrict__ Wt, float* __restrict__ out, int Cin, int Cout,
                                                       int KC, int MT, long total) {
  const int C4 = KC / 4;
  for (long q = blockIdx.x * 256L + threadIdx.x; q < total; q += gridDim.x * 256L) {
    const int l = (int)(q & 63);
    long r = q >> 6;
    const int m = (int)(r % MT); r /= MT;
    const int tap = (int)(r % 9); r /= 9;
    const int c4 = (int)(r % C4);
    const int ch = (int)(r / C4);
    const int co = ch * KC + c4 * 4 + (l >> 4), ci = m * 16 + (l & 15);
    out[q] = (ci < Cin && co < Cout) ? Wt[((long)co * Cin + ci) * 9 + tap] : 0.f;
  }
}

template <int CO, int CI, int HO, int WO, int RQ, int KC>
__global__ __launch_bounds__(640) void c3b_kernel(C3P p) {
  using G = C3BGeo<CO, CI, HO, WO, RQ, KC>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - G::NW;
    int roff[G::NQ], rrow[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / G::WP, x = pi - r * G::WP;
      rrow[q] = (pi < G::PL0) ? (x == WO ? -100000 : r) : -200000;     // zero column; beyond the plane: no lane
      roff[q] = r * WO + x;
    }
    auto dma = [&](long k) {
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const int ch = (int)(k % G::NCH);
      const long b = tile / G::NBAND;
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      float* __restrict__ buf = lds + (k & 1) * G::BUF;
      const float* __restrict__ sb = p.src + b * p.src_bs + ((long)ch * KC * HO + q0) * WO;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (c % G::NL != lw) continue;
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
          if (rrow[q] > -200000) {
            const float* gsrc = (rrow[q] >= 0 && q0 + rrow[q] < HO) ? sb + (long)c * HO * WO + roff[q] : p.zero;
            __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 64), 4, 0, 0);
          }
        }
      }
      const float* __restrict__ fg = p.frag + (long)ch * G::FRAGC;
#pragma unroll
      for (int q = 0; q < G::NFQ; ++q) {
        const int pi = q * 64 + lane;
        if (q % G::NL == lw && pi < G::FRAGC / 4)
          __builtin_amdgcn_global_load_lds((gptr_t)(fg + pi * 4), (lptr_t)(buf + G::IMG + q * 256), 16, 0, 0);
      }
    };
    // the band's ReLU mask (the activation below this layer, same geometry as dX): one linear run per channel, issued
    // while the band's FIRST chunk computes -- the epilogue behind its last chunk then reads LDS, not HBM
    auto dma_mask = [&](long k) {      // the slice of the band's mask that travels beside chunk k (all but the last chunk carry one)
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const long b = tile / G::NBAND;
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int rows = 2 * min(RQ, HO - q0);
      float* __restrict__ mb = lds + 2 * G::BUF;
      constexpr int CPP = (CI + G::NCH - 2) / (G::NCH - 1);
      const int part = (int)(k % G::NCH);
#pragma unroll 1
      for (int c = part * CPP + lw; c < min(CI, (part + 1) * CPP); c += G::NL) {
        const float* __restrict__ ms = p.mask + b * p.out_bs + ((long)c * G::H + 2 * q0) * G::W;
        for (int q = 0; q * 256 < rows * G::W; ++q) {
          const int pi = q * 64 + lane;
          if (pi * 4 < rows * G::W)
            __builtin_amdgcn_global_load_lds((gptr_t)(ms + pi * 4), (lptr_t)(mb + c * G::MROW + q * 256), 16, 0, 0);
        }
      }
    };
    if (nwork > 0) dma(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (long k = 0; k < nwork; ++k) {
      if (k + 1 < nwork) dma(k + 1);
      if (p.mask != nullptr && (int)(k % G::NCH) < G::NCH - 1) dma_mask(k);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    return;
  }
  // -------------------------------------------------------------------- computing waves
  int base[G::TP];
#pragma unroll
  for (int u = 0; u < G::TP; ++u) {
    const int t = w + G::NW * u;
    const int pp = t * 16 + j;
    const int px = (t < G::NT && pp < G::NPIX) ? pp : 0;
    const int r = px / WO, x = px - r * WO;
    base[u] = g * G::PLANE + r * G::WP + x;
  }
  f32x4 acc[G::TP][4][G::MT];                    // [tile position][class py*2+px][channel tile]
#pragma unroll
  for (int u = 0; u < G::TP; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < G::MT; ++m) acc[u][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const bool stamp = p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long ts = stamp ? wall_clock64() : 0, tsum[3] = {0, 0, 0};
#define C3_TS(i) do { if (stamp) { const unsigned long long n_ = wall_clock64(); tsum[i] += n_ - ts; ts = n_; } } while (0)
  for (long k = 0; k < nwork; ++k) {
    const float* __restrict__ img = lds + (k & 1) * G::BUF;
    const float* __restrict__ fr = img + G::IMG + lane;
#pragma unroll
    for (int c4 = 0; c4 < G::C4; ++c4) {
      float wv[9][G::MT];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int m = 0; m < G::MT; ++m) wv[tap][m] = fr[((c4 * 9 + tap) * G::MT + m) * 64];
#pragma unroll
      for (int u = 0; u < G::TP; ++u) {
        const float* __restrict__ s = img + base[u] + c4 * 4 * G::PLANE;
        const float s00 = s[0], s01 = s[1], s10 = s[G::WP], s11 = s[G::WP + 1];
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          // class (py, px): taps ky in {1} / {0 (row q+1), 2 (row q)}, kx likewise; tap index = ky*3 + kx
          acc[u][0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s00, wv[4][m], acc[u][0][m], 0, 0, 0);
          acc[u][1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s01, wv[3][m], acc[u][1][m], 0, 0, 0);
          acc[u][2][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s10, wv[1][m], acc[u][2][m], 0, 0, 0);
          acc[u][3][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s11, wv[0][m], acc[u][3][m], 0, 0, 0);
          acc[u][1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s00, wv[5][m], acc[u][1][m], 0, 0, 0);
          acc[u][2][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s00, wv[7][m], acc[u][2][m], 0, 0, 0);
          acc[u][3][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s10, wv[2][m], acc[u][3][m], 0, 0, 0);
          acc[u][3][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s01, wv[6][m], acc[u][3][m], 0, 0, 0);
          acc[u][3][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(s00, wv[8][m], acc[u][3][m], 0, 0, 0);
        }
      }
    }
    C3_TS(0);
    if ((int)(k % G::NCH) == G::NCH - 1) {
      // ---- the band is complete: lane (j, g) holds class pixels 4g .. 4g+3 of channel j, both x parities: rows of dX
      const long tile = blockIdx.x + (k / G::NCH) * gridDim.x;
      const long b = tile / G::NBAND;
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int npix_ok = min(RQ, HO - q0) * WO;
      const long o0 = b * p.out_bs;
#pragma unroll
      for (int u = 0; u < G::TP; ++u) {
        const int t = w + G::NW * u;
        const int c0 = t * 16 + 4 * g;
        // Pixel pair (c, c+1), c = c0 + 2h: 4 consecutive floats of dX row 2(q0+qr)+py -- unless the class grid is odd
        // wide and the pair straddles two of its rows (then two 8-byte halves).  The mask band sits in LDS.
        const float* __restrict__ mb = lds + 2 * G::BUF;
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          const int ci = m * 16 + j;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * h;
            const int qr = c / WO, pc = c - qr * WO;
            const bool okA = ci < CI && c < npix_ok, okB = ci < CI && c + 1 < npix_ok, split = pc + 1 >= WO;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int lrow = (2 * qr + py) * G::W;              // band-relative row offset
              const int la = ci * G::MROW + lrow + 2 * pc, lbo = split ? ci * G::MROW + lrow + 2 * G::W : la + 2;
              const long offA = o0 + ((long)ci * G::H + 2 * q0) * G::W + lrow + 2 * pc;
              const long offB = split ? o0 + ((long)ci * G::H + 2 * q0) * G::W + lrow + 2 * G::W : offA + 2;
              float4 v = make_float4(acc[u][py * 2][m][2 * h], acc[u][py * 2 + 1][m][2 * h], acc[u][py * 2][m][2 * h + 1],
                                     acc[u][py * 2 + 1][m][2 * h + 1]);
              if (p.mask != nullptr) {
                if (okA) {
                  const float2 q2 = *reinterpret_cast<const float2*>(mb + la);
                  if (!(q2.x > 0.f)) v.x = 0.f;
                  if (!(q2.y > 0.f)) v.y = 0.f;
                }
                if (okB) {
                  const float2 q2 = *reinterpret_cast<const float2*>(mb + lbo);
                  if (!(q2.x > 0.f)) v.z = 0.f;
                  if (!(q2.y > 0.f)) v.w = 0.f;
                }
              }
              if (okA) {
                if (okB && !split) *reinterpret_cast<float4*>(p.out + offA) = v;
                else {
                  *reinterpret_cast<float2*>(p.out + offA) = make_float2(v.x, v.y);
                  if (okB) *reinterpret_cast<float2*>(p.out + offB) = make_float2(v.z, v.w);
                }
              }
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[u][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    C3_TS(1);
    __syncthreads();
    C3_TS(2);
  }
  if (stamp) {
    p.dbg[0] = tsum[0]; p.dbg[1] = tsum[1]; p.dbg[2] = tsum[2]; p.dbg[3] = (unsigned long long)nwork;
  }
#undef C3_TS
}

template <int CO, int CI, int HO, int WO, int RQ, int KC>
int c3b_launch(const C3P& p, hipStream_t st) {
  using G = C3BGeo<CO, CI, HO, WO, RQ, KC>;
  const void* k = (const void*)c3b_kernel<CO, CI, HO, WO, RQ, KC>;
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) return A2C_ERR_LAUNCH;
    int n = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 640, G::LDS_BYTES) != hipSuccess || n < 1) n = 1;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    per_cu = n;
  }
  const long total = (long)p.B * G::NBAND;
  const long cap = (long)per_cu * cus;
  const int grid = (int)(total < cap ? total : cap);
  hipLaunchKernelGGL((c3b_kernel<CO, CI, HO, WO, RQ, KC>), dim3(grid), dim3(640), G::LDS_BYTES, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
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
  constexpr size_t LDSB = BWD ? G::LDS_BYTES_BWD : G::LDS_BYTES;
  static_assert(LDSB <= 160 * 1024, "LDS");
  static_assert(!BWD || G::NCH >= 2, "the mask band is staged one chunk ahead of its use");
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB) != hipSuccess) return A2C_ERR_LAUNCH;
    int n = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 640, LDSB) != hipSuccess || n < 1) n = 1;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    per_cu = n;
  }
  const long total = (long)p.B * G::NBAND;
  const long cap = (long)per_cu * cus;
  const int grid = (int)(total < cap ? total : cap);
  hipLaunchKernelGGL((c3_kernel<CS, CD, H, W, S, R, BWD>), dim3(grid), dim3(640), LDSB, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}
}  // namespace

static unsigned long long* g_c3_dbg = nullptr;
extern "C" int a2c_debug_c3_timing(unsigned long long* dev_buf) {      // debug hook (not part of the drop-in boundary)
  g_c3_dbg = dev_buf;
  return A2C_OK;
}

// ---- what conv.hip's entry points call (not part of the C ABI) -------------------------------------------------
// kind 0 = forward, 1 = backward-data.  c3_supported: this family has an instantiation for the layer.
static bool c3b_shape(const a2c_conv_desc* d);
bool c3_supported(const a2c_conv_desc* d, int kind) {
  static const bool off = getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1';
  if (off) return false;
  if (kind == 1 && d->stride == 2) return c3b_shape(d);
  if (d->ks != 3 || d->pad != 1 || d->H != 84 || d->W != 84) return false;
  if (kind == 0) {
    if (d->stride == 1) return (d->Cin == 4 && d->Cout == 16) || (d->Cin == 16 && d->Cout == 24);
    if (d->stride == 2) return (d->Cin == 24 && d->Cout == 32) || (d->Cin == 16 && d->Cout == 24);
    return false;
  }
  if (d->stride == 1) return d->Cin == 16 && d->Cout == 24;
  // stride 2, even input: ConvModel conv3 (24 <- 32 @84), conv4 (32 <- 64 @42); GRUModel conv2 (16 <- 24 @84), conv3 (24 <- 32 @42)
  return d->stride == 2;
}

static bool c3b_shape(const a2c_conv_desc* d) {
  if (d->ks != 3 || d->pad != 1 || d->stride != 2) return false;
  return (d->H == 84 && d->W == 84 && ((d->Cin == 24 && d->Cout == 32) || (d->Cin == 16 && d->Cout == 24))) ||
         (d->H == 42 && d->W == 42 && ((d->Cin == 32 && d->Cout == 64) || (d->Cin == 24 && d->Cout == 32)));
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
  if (kind == 1 && d->stride == 2) {
    hipLaunchKernelGGL(c3b_prep_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, out, d->Cin, d->Cout, 8,
                       (CD + 15) / 16, total);
    if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
    return A2C_OK;
  }
  hipLaunchKernelGGL(c3_prep_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, out, d->Cin, d->Cout, kind,
                     CS >= 8 ? 8 : 4, (CD + 15) / 16, total);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

int c3_fwd(const a2c_conv_desc* d, const float* in, long in_bs, const float* frag, const float* bias, int relu, float* out,
           long out_bs, int B, hipStream_t st) {
  C3P p{in, in_bs, frag, bias, nullptr, out, out_bs, zero_page(), B, relu, g_c3_dbg};
  if (!p.zero) return A2C_ERR_LAUNCH;
  if (d->stride == 1 && d->Cin == 4) return c3_launch<4, 16, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 1 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 2 && d->Cin == 24) return c3_launch<24, 32, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 2, 6, false>(p, st);
  return A2C_ERR_ARG;
}

int c3_bwd_data(const a2c_conv_desc* d, const float* dout, const float* frag, const float* mask, float* din, int B,
                hipStream_t st) {
  C3P p{dout, (long)d->Cout * d->OH * d->OW, frag, nullptr, mask, din, (long)d->Cin * d->H * d->W, zero_page(), B, 0, g_c3_dbg};
  if (!p.zero) return A2C_ERR_LAUNCH;
  if (d->stride == 1 && d->Cin == 16 && d->Cout == 24) return c3_launch<24, 16, 84, 84, 1, 12, true>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3b_launch<32, 24, 42, 42, 6, 8>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3b_launch<24, 16, 42, 42, 6, 8>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3b_launch<64, 32, 21, 21, 11, 8>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3b_launch<32, 24, 21, 21, 11, 8>(p, st);
  return A2C_ERR_ARG;
}
