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
  static constexpr int PB = W % 4 == 0 ? 4 : 1;       // floats per LDS-DMA piece: 16-byte pieces need 16-byte aligned rows
  static constexpr int WP = W + PB;                   // one piece of pad + the row
  static constexpr int PL0 = SR * WP;
  // planes 16 (mod 32) floats apart: the two k-groups of a 32-lane LDS access fall on disjoint banks (stride 1)
  static constexpr int PLANE = S == 1 ? ((PL0 + 15) / 32) * 32 + 16 : PL0;
  static constexpr int PP = PL0 / PB;                 // pieces per plane
  static constexpr int NQ = (PP + 63) / 64;           // DMA instructions per plane
  static constexpr int IMG = ((KC * PLANE + 3) / 4) * 4 + 8;   // + zeros behind the last plane (right halo of its last row)
  static constexpr int FRAGC = C4 * 9 * MT * 64;      // fragment floats per chunk
  static constexpr int NFQ = (FRAGC / 4 + 63) / 64;
  static constexpr int BUF = ((IMG + FRAGC + 255) / 256) * 256;
  static constexpr int MROW = R * OW;                  // floats of one channel's output band (contiguous in HBM)
  static constexpr int MASKF = CD * MROW;              // backward-data: the ReLU-mask band, staged in LDS by the loaders
  static constexpr size_t LDS_BYTES = 2 * (size_t)BUF * 4;
  static constexpr size_t LDS_BYTES_BWD = (2 * (size_t)BUF + MASKF) * 4;
  static constexpr bool VEC = (OH * OW) % 4 == 0 && MROW % 4 == 0;     // 16-byte output stores
  static_assert(CS % KC == 0 && PL0 % PB == 0 && (KC * PLANE) % 4 == 0, "shape");
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
    lds[G::IMG - 8 + (tid & 7)] = 0.f;
    lds[G::BUF + G::IMG - 8 + (tid & 7)] = 0.f;
    if (G::KC * G::PLANE < G::IMG - 8) {                  // (plane sizes that are not a multiple of 4 floats)
      lds[G::KC * G::PLANE + (tid & 3)] = 0.f;
      lds[G::BUF + G::KC * G::PLANE + (tid & 3)] = 0.f;
    }
  }
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves (planes / fragment pieces dealt round robin)
    const int lw = w - G::NW;
    int roff[G::NQ], rrow[G::NQ];                 // this lane's piece of DMA instruction q: source offset / image row
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / (G::WP / G::PB), i = pi - r * (G::WP / G::PB);
      rrow[q] = (pi < G::PP) ? (i == 0 ? -100000 : r) : -200000;      // pad piece: zeros; beyond the plane: no lane
      roff[q] = r * W + G::PB * (i - 1);
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
            if (G::PB == 4) __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 256), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 64), 4, 0, 0);
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
    base[u] = g * G::PLANE + r * S * G::WP + x * S + G::PB - 1;
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
            if (G::VEC && p0 + 3 < npix_ok) *reinterpret_cast<float4*>(o) = v;
            else {                                             // rows that are not 16-byte aligned / the band's ragged end
              if (p0 < npix_ok) o[0] = v.x;
              if (p0 + 1 < npix_ok) o[1] = v.y;
              if (p0 + 2 < npix_ok) o[2] = v.z;
              if (p0 + 3 < npix_ok) o[3] = v.w;
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
        if (G::MT > 1) __builtin_amdgcn_sched_barrier(0);      // (registers: keep the next position's reads behind these MFMAs)
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
          const float* __restrict__ mc = mb + ci * G::MROW;                               // this channel's mask band (LDS)
          float* __restrict__ oc = p.out + o0 + ((long)ci * G::H + 2 * q0) * G::W;          // ... and its dX band (HBM)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * h;
            const int qr = c / WO, pc = c - qr * WO;
            const bool okA = ci < CI && c < npix_ok, okB = ci < CI && c + 1 < npix_ok, split = pc + 1 >= WO;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int la = (2 * qr + py) * G::W + 2 * pc;                               // band-relative offsets of the two pixels
              const int lb = split ? (2 * qr + py + 2) * G::W : la + 2;
              float4 v = make_float4(acc[u][py * 2][m][2 * h], acc[u][py * 2 + 1][m][2 * h], acc[u][py * 2][m][2 * h + 1],
                                     acc[u][py * 2 + 1][m][2 * h + 1]);
              if (p.mask != nullptr) {
                if (okA) {
                  const float2 q2 = *reinterpret_cast<const float2*>(mc + la);
                  if (!(q2.x > 0.f)) v.x = 0.f;
                  if (!(q2.y > 0.f)) v.y = 0.f;
                }
                if (okB) {
                  const float2 q2 = *reinterpret_cast<const float2*>(mc + lb);
                  if (!(q2.x > 0.f)) v.z = 0.f;
                  if (!(q2.y > 0.f)) v.w = 0.f;
                }
              }
              if (okA) {
                if (okB && !split) *reinterpret_cast<float4*>(oc + la) = v;
                else {
                  *reinterpret_cast<float2*>(oc + la) = make_float2(v.x, v.y);
                  if (okB) *reinterpret_cast<float2*>(oc + lb) = make_float2(v.z, v.w);
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

// ---------------------------------------------------------------------------------------------------------------
// Weight gradient: dW[co][ci][ky][kx] = sum over samples and output pixels of dOut[co][oy][ox] * X[ci][S*oy+ky-1][S*ox+kx-1],
// db[co] = sum dOut.  The same streaming skeleton with the PIXELS as the GEMM's K: one MFMA step = 4 consecutive
// output pixels of a row; A operand = the input seen through the (ci, ky, kx) window of row n = ci*9 + ky*3 + kx
// (16 such rows per tile, a chunk of KC input channels = ceil(9*KC/16) tiles), B operand = dOut (16 output channels per
// tile).  D = [n][co]: a lane ends with 4 consecutive n of one co -- a 16-byte run of the (co, ci, ky, kx) gradient.
// All accumulators (every chunk's tiles) stay in registers for the whole launch; the 8 computing waves split into
// NCG groups over the output channels and 8/NCG groups over the pixel steps, each wave writes its partial sums to
// its own slab once, at the end; a fixed-order reduction adds the slabs (deterministic).
template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG>
struct C3WGeo {
  static constexpr int NW = 8, NL = 2, NPG = NW / NCG;
  static constexpr int NCH = CS / KC;
  static constexpr int NTC = (KC * 9 + 15) / 16;       // n-tiles per chunk
  static constexpr int MT = ((CD + 15) / 16 + NCG - 1) / NCG * NCG;
  static constexpr int MTW = MT / NCG;
  static constexpr int OH = (H - 1) / S + 1, OW = (W - 1) / S + 1;
  static constexpr int NBAND = (OH + R - 1) / R;
  static constexpr int OWP = (OW + 3) / 4 * 4;
  static constexpr int KSR = OWP / 4, KS = R * KSR;    // MFMA steps per row / band
  static constexpr int SR = (R - 1) * S + 3;
  static constexpr int PB = W % 4 == 0 ? 4 : 1;
  static constexpr int WP = W + PB;
  static constexpr int PL0 = SR * WP, PLANE = PL0;
  static constexpr int PP = PL0 / PB, NQ = (PP + 63) / 64;
  static constexpr int XB = ((KC * PLANE + 3) / 4) * 4 + 64;      // (+ slack: the last step of a row may read past it)
  static constexpr int DPB = OW % 4 == 0 ? 4 : 1;                 // dOut pieces: whole 16-byte runs when rows are 16-byte multiples
  static constexpr int DPL0 = R * OWP;
  static constexpr int DPLANE = (DPL0 - 4 + 31) / 32 * 32 + 4;    // 4 (mod 32) floats apart: channels spread over the banks
  static constexpr int DPP = DPL0 / DPB, DNQ = (DPP + 63) / 64;
  static constexpr int DB = CD * DPLANE;
  static constexpr int K = CS * 9;
  static constexpr long PER = (long)CD * K + CD;                  // floats per slab: dW | db
  static constexpr size_t LDS_BYTES = (2 * (size_t)XB + 2 * (size_t)DB) * 4;
  static_assert(CS % KC == 0 && NW % NCG == 0 && PL0 % PB == 0 && DPLANE >= DPL0 && DPLANE % 4 == 0, "shape");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct C3WP {
  const float* x; long x_bs;          // layer input (B, CS, H, W)
  const float* dout;                  // (B, CD, OH, OW) dense
  float* slab;                        // [grid * NPG][CD*CS*9 + CD]
  const float* zero;
  int B;
};

__global__ __launch_bounds__(256) void c3w_reduce_kernel(const float* __restrict__ slab, int nslab, long per, long nW,
                                                         float* __restrict__ dW, float* __restrict__ db) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < per; i += gridDim.x * 256L) {
    float s = 0.f;
    int z = 0;
    for (; z + 8 <= nslab; z += 8) {       // 8 independent loads in flight, fixed summation order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = slab[(long)(z + u) * per + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; z < nslab; ++z) s += slab[(long)z * per + i];
    if (i < nW) dW[i] = s;
    else if (db) db[i - nW] = s;
  }
}

template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG>
__global__ __launch_bounds__(640) void c3w_kernel(C3WP p) {
  using G = C3WGeo<CS, CD, H, W, S, R, KC, NCG>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;
  float* __restrict__ xbuf = lds;                      // 2 x XB
  float* __restrict__ dbuf = lds + 2 * G::XB;          // 2 x DB
  // whatever the DMAs never write (pad columns, slack behind the planes) must read as finite numbers: zeros
  for (int i = tid; i < 2 * G::XB + 2 * G::DB; i += 640) lds[i] = 0.f;
  __syncthreads();
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - G::NW;
    int roff[G::NQ], rrow[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / (G::WP / G::PB), i = pi - r * (G::WP / G::PB);
      rrow[q] = (pi < G::PP) ? (i == 0 ? -100000 : r) : -200000;
      roff[q] = r * W + G::PB * (i - 1);
    }
    int doff[G::DNQ], drow[G::DNQ];                   // dOut pieces: band row and source offset of this lane's piece
#pragma unroll
    for (int q = 0; q < G::DNQ; ++q) {
      const int pi = q * 64 + lane;
      const int f = pi * G::DPB;                       // first float of the piece in the (row, OWP) band image
      const int r = f / G::OWP, x = f - r * G::OWP;
      drow[q] = (pi < G::DPP) ? (x < G::OW ? r : -100000) : -200000;      // pad column: zeros
      doff[q] = r * G::OW + x;
    }
    auto dma = [&](long k) {
      const long it = k / G::NCH;
      const long tile = blockIdx.x + it * gridDim.x;
      const int ch = (int)(k % G::NCH);
      const long b = tile / G::NBAND;
      const int band = (int)(tile - b * G::NBAND);
      const int y0 = band * R * S - 1;
      float* __restrict__ xb = xbuf + (k & 1) * G::XB;
      const float* __restrict__ sb = p.x + b * p.x_bs + ((long)ch * KC * H + y0) * W;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (c % G::NL != lw) continue;
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
          if (rrow[q] > -200000) {
            const int y = y0 + rrow[q];
            const float* gsrc = (rrow[q] >= 0 && y >= 0 && y < H) ? sb + (long)c * H * W + roff[q] : p.zero;
            if (G::PB == 4) __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(xb + c * G::PLANE + q * 256), 16, 0, 0);
            else __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(xb + c * G::PLANE + q * 64), 4, 0, 0);
          }
        }
      }
      if (ch == 0) {        // the band's dOut rows, all output channels
        float* __restrict__ db_ = dbuf + (it & 1) * G::DB;
        const float* __restrict__ ds = p.dout + (b * CD * G::OH + (long)band * R) * G::OW;
#pragma unroll 1
        for (int c = lw; c < CD; c += G::NL) {
#pragma unroll
          for (int q = 0; q < G::DNQ; ++q) {
            if (drow[q] > -200000) {
              const float* gsrc = (drow[q] >= 0 && band * R + drow[q] < G::OH) ? ds + (long)c * G::OH * G::OW + doff[q] : p.zero;
              if (G::DPB == 4) __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(db_ + c * G::DPLANE + q * 256), 16, 0, 0);
              else __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(db_ + c * G::DPLANE + q * 64), 4, 0, 0);
            }
          }
        }
      }
    };
    if (nwork > 0) dma(0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    for (long k = 0; k < nwork; ++k) {
      if (k + 1 < nwork) dma(k + 1);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
    return;
  }
  // -------------------------------------------------------------------- computing waves
  const int cg = w % NCG, pg = w / NCG;
  int aoff[G::NTC];                                   // A operand: window offset of row n = nt*16 + j (ci_local, ky, kx) + pixel g
#pragma unroll
  for (int nt = 0; nt < G::NTC; ++nt) {
    const int n = nt * 16 + j;
    const int nn = n < KC * 9 ? n : 0;
    const int cl = nn / 9, tap = nn - cl * 9;
    aoff[nt] = cl * G::PLANE + (tap / 3) * G::WP + (tap % 3) + g * S + G::PB - 1;
  }
  int boff[G::MTW];                                   // B operand: dOut channel (cg*MTW + m)*16 + j, pixel g
#pragma unroll
  for (int m = 0; m < G::MTW; ++m) {
    const int co = (cg * G::MTW + m) * 16 + j;
    boff[m] = (co < CD ? co : 0) * G::DPLANE + g;
  }
  f32x4 acc[G::NCH][G::NTC][G::MTW];
#pragma unroll
  for (int c = 0; c < G::NCH; ++c)
#pragma unroll
    for (int nt = 0; nt < G::NTC; ++nt)
#pragma unroll
      for (int m = 0; m < G::MTW; ++m) acc[c][nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbs[G::MTW];
#pragma unroll
  for (int m = 0; m < G::MTW; ++m) dbs[m] = 0.f;
  __syncthreads();
  for (long it = 0; it < nmine; ++it) {
    const float* __restrict__ dimg = dbuf + (it & 1) * G::DB;
#pragma unroll
    for (int ch = 0; ch < G::NCH; ++ch) {
      const float* __restrict__ ximg = xbuf + ((it * G::NCH + ch) & 1) * G::XB;
      int r = 0, xq = pg;
      while (xq >= G::KSR) { xq -= G::KSR; ++r; }
      for (int s = pg; s < G::KS; s += G::NPG) {
        const int xo = r * S * G::WP + xq * 4 * S, dofs = r * G::OWP + xq * 4;
        float av[G::NTC], bv[G::MTW];
#pragma unroll
        for (int nt = 0; nt < G::NTC; ++nt) av[nt] = ximg[aoff[nt] + xo];
#pragma unroll
        for (int m = 0; m < G::MTW; ++m) bv[m] = dimg[boff[m] + dofs];
#pragma unroll
        for (int nt = 0; nt < G::NTC; ++nt)
#pragma unroll
          for (int m = 0; m < G::MTW; ++m) acc[ch][nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv[m], acc[ch][nt][m], 0, 0, 0);
        if (ch == 0) {
#pragma unroll
          for (int m = 0; m < G::MTW; ++m) dbs[m] += bv[m];
        }
        xq += G::NPG;
        while (xq >= G::KSR) { xq -= G::KSR; ++r; }
      }
      __syncthreads();
    }
  }
  // ---- this wave's partial sums -> its slab (dW as [co][ci*9 + tap], then db)
  float* __restrict__ sl = p.slab + ((long)blockIdx.x * G::NPG + pg) * G::PER;
#pragma unroll
  for (int m = 0; m < G::MTW; ++m) {
    const int co = (cg * G::MTW + m) * 16 + j;
#pragma unroll
    for (int ch = 0; ch < G::NCH; ++ch)
#pragma unroll
      for (int nt = 0; nt < G::NTC; ++nt) {
        const int n = nt * 16 + 4 * g;                 // 4 consecutive window rows of this chunk
        if (co < CD && n < KC * 9)
          *reinterpret_cast<float4*>(sl + (long)co * G::K + ch * KC * 9 + n) =
              make_float4(acc[ch][nt][m][0], acc[ch][nt][m][1], acc[ch][nt][m][2], acc[ch][nt][m][3]);
      }
    float v = dbs[m];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (g == 0 && co < CD) sl[(long)CD * G::K + co] = v;
  }
}

template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG>
int c3w_launch(const C3WP& p0, float* dW, float* db, size_t ws_bytes, hipStream_t st, size_t* need) {
  using G = C3WGeo<CS, CD, H, W, S, R, KC, NCG>;
  const void* k = (const void*)c3w_kernel<CS, CD, H, W, S, R, KC, NCG>;
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) return A2C_ERR_LAUNCH;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p0.B * G::NBAND;
  const int grid = (int)(total < cus ? total : cus);
  const size_t bytes = (size_t)grid * G::NPG * G::PER * 4;
  if (need) { *need = (size_t)cus * G::NPG * G::PER * 4; return A2C_OK; }
  if (ws_bytes < bytes) return A2C_ERR_WORKSPACE;
  hipLaunchKernelGGL((c3w_kernel<CS, CD, H, W, S, R, KC, NCG>), dim3(grid), dim3(640), G::LDS_BYTES, st, p0);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  hipLaunchKernelGGL(c3w_reduce_kernel, dim3(a2c_grid_1d(G::PER, 256)), dim3(256), 0, st, (const float*)p0.slab, grid * G::NPG,
                     G::PER, (long)CD * G::K, dW, db);
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
  if (d->ks != 3 || d->pad != 1) return false;
  if (kind == 0 && d->stride == 2 && d->H == 42 && d->W == 42) return (d->Cin == 32 && d->Cout == 64) || (d->Cin == 24 && d->Cout == 32);
  if (kind == 0 && d->stride == 2 && d->H == 21 && d->W == 21) return d->Cin == 32 && d->Cout == 48;
  if (d->H != 84 || d->W != 84) return false;
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
  if (d->stride == 1 && d->H == 84 && d->Cin == 4) return c3_launch<4, 16, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 1 && d->H == 84 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3_launch<24, 32, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3_launch<32, 64, 42, 42, 2, 11, false>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3_launch<24, 32, 42, 42, 2, 11, false>(p, st);
  if (d->stride == 2 && d->H == 21 && d->Cin == 32) return c3_launch<32, 48, 21, 21, 2, 11, false>(p, st);
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

// weight gradient: instantiations and their workspace
#define C3W_CASES(X)                                                                           \
  X(4, 16, 84, 84, 1, 8, 4, 1)   /* conv1 of both models                                    */ \
  X(16, 24, 84, 84, 1, 4, 16, 1) /* ConvModel conv2                                         */ \
  X(24, 32, 84, 84, 2, 6, 8, 2)  /* ConvModel conv3                                         */ \
  X(32, 64, 42, 42, 2, 7, 8, 4)  /* ConvModel conv4                                         */ \
  X(16, 24, 84, 84, 2, 6, 8, 1)  /* GRUModel conv2                                          */ \
  X(24, 32, 42, 42, 2, 7, 8, 2)  /* GRUModel conv3                                          */

bool c3w_supported(const a2c_conv_desc* d) {
  static const bool off = (getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1') ||
                          (getenv("A2C_NO_C3W") != nullptr && getenv("A2C_NO_C3W")[0] == '1');
  static const bool all = getenv("A2C_C3W_ALL") != nullptr && getenv("A2C_C3W_ALL")[0] == '1';
  if (off || d->ks != 3 || d->pad != 1) return false;
  // Measured against conv.hip's wgrad_kernel at N = 4096 (tools/conv3_check.py): this kernel wins on the first layer
  // (4 -> 16: 0.92 vs 1.01 ms) and on ConvModel's conv4 (32 -> 64 @42: 1.29 vs 1.55 ms), ties on the 84-wide
  // 16 -> 24 layers (2.87-2.95 vs 3.00 ms, 1.15 vs 1.11 ms: matrix-bound in both, 67 TF with a quarter of the 24-channel
  // tiles empty -- and the generic instance for them spilled registers), and loses on the 24 -> 32 layers, which stay.
  if (!all && d->Cin == 24) return false;
#define C3W_MATCH(cs, cd, h, w_, s_, r, kc, ncg) if (d->Cin == cs && d->Cout == cd && d->H == h && d->W == w_ && d->stride == s_) return true;
  C3W_CASES(C3W_MATCH)
#undef C3W_MATCH
  return false;
}

static int c3w_dispatch(const a2c_conv_desc* d, const C3WP& p, float* dW, float* db, size_t ws_bytes, hipStream_t st, size_t* need) {
#define C3W_RUN(cs, cd, h, w_, s_, r, kc, ncg) \
  if (d->Cin == cs && d->Cout == cd && d->H == h && d->W == w_ && d->stride == s_) return c3w_launch<cs, cd, h, w_, s_, r, kc, ncg>(p, dW, db, ws_bytes, st, need);
  C3W_CASES(C3W_RUN)
#undef C3W_RUN
  return A2C_ERR_ARG;
}

size_t c3w_ws_bytes(const a2c_conv_desc* d) {
  if (!c3w_supported(d)) return 0;
  size_t need = 0;
  C3WP p{};
  p.B = 1;
  return c3w_dispatch(d, p, nullptr, nullptr, 0, nullptr, &need) == A2C_OK ? need : 0;
}

int c3w_bwd_weight(const a2c_conv_desc* d, const float* in, long in_bs, const float* dout, float* dW, float* db, int B, void* ws,
                   size_t ws_bytes, hipStream_t st) {
  C3WP p{in, in_bs, dout, (float*)ws, zero_page(), B};
  if (!p.zero) return A2C_ERR_LAUNCH;
  return c3w_dispatch(d, p, dW, db, ws_bytes, st, nullptr);
}
