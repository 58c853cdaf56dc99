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
// v_mfma_f32_16x16x4_f32: M = 16 pixels (A operand), N = 16 destination channels, K-step = 4 source channels at one tap;
// a wave owns TPW pixel tiles x MT channel tiles of accumulators and reads every fragment once per step for all of them.
// Sums are ordered (chunk, tap, channel quad): fixed, deterministic, a re-association of conv.hip's (tap, quad) order.
//
// What is in this file, in order:
//   c3_kernel    forward / stride-1 backward-data, outputs stored straight from the accumulators (inline-asm stores, see
//                c3_gstore128); SG = forward that also leaves the ReLU mask as SIGN WORDS (an eleventh wave writes them out)
//   c3b_kernel   stride-2 backward-data with the float mask band staged in LDS
//   c3s_kernel   / c3bs_kernel: the STAGED variants -- output through an LDS image to two storer waves, the mask read as
//                sign words, (c3bs) a ring of chunk images with counted vmcnt, resident fragments, 16-byte dOut pieces
//   c3w_kernel   weight gradient (pixels as K)
// and the dispatch (c3_fwd, c3_bwd_data, c3w_bwd_weight) that conv.hip's entry points call.  Which kernel runs where, and
// the measurements behind each choice: DESIGN.md section 4.
#include <stdlib.h>
#include <mutex>
#include "a2c_common.h"

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

// Global stores of the kernels below go through inline asm: in a kernel that issues LDS-DMA hipcc's waitcnt insertion puts
// `s_waitcnt vmcnt(0)` in front of an LDS read whenever the wave has a vector-memory operation of its own in flight -- after
// an epilogue that is a wait for the acknowledgement of every output store (2-4 us under load) in front of the next band's
// first MFMA operand.  Stores it cannot see, it does not wait for (nothing here ever reads back what it stored).
// (s_nop 1: the hazard recognizer does not see them either -- a store of more than 8 bytes needs 2 wait states before a
// VALU instruction may overwrite its data registers.)
__device__ __forceinline__ void c3_gstore128(float* p, const float4 v) {
  const f32x4 q = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void c3_gstore64(float* p, const float2 v) {
  using f32x2 = __attribute__((ext_vector_type(2))) float;
  const f32x2 q = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}
__device__ __forceinline__ void c3_gstore32(float* p, const float v) {
  asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

// (sign words: see the staged kernels below)  LDS accesses of a wave that also has global stores in flight go through
// inline asm for the same reason the stores above do.
__device__ __forceinline__ unsigned c3_lds_addr(const void* p) { return (unsigned)(unsigned long)(lptr_t)p; }
__device__ __forceinline__ float c3_lds_read32(const void* p) {
  float v;
  asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(c3_lds_addr(p)));
  return v;
}
__device__ __forceinline__ void c3_lds_zero32(void* p) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(c3_lds_addr(p)), "v"(0u) : "memory");
}
__device__ __forceinline__ void c3_lds_wait(float& a, float& b, float& c, float& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}
// computing waves, forward: the sign bits of four consecutive band pixels p0 .. p0+3 of one channel -> its sign words
template <int OW, int RW>
__device__ __forceinline__ void c3_sign4(unsigned* __restrict__ bbc, int p0, int npix, const float4 q4) {
  if (OW % 4 == 0) {                                   // the four pixels share a row and a word
    const unsigned nib = (q4.x > 0.f ? 1u : 0u) | (q4.y > 0.f ? 2u : 0u) | (q4.z > 0.f ? 4u : 0u) | (q4.w > 0.f ? 8u : 0u);
    const int r = p0 / OW, x = p0 - r * OW;
    if (p0 < npix && nib) atomicOr(&bbc[r * RW + (x >> 5)], nib << (x & 31));
  } else {
    const float e[4] = {q4.x, q4.y, q4.z, q4.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = p0 + i, r = q / OW, x = q - r * OW;
      if (q < npix && e[i] > 0.f) atomicOr(&bbc[r * RW + (x >> 5)], 1u << (x & 31));
    }
  }
}

// Workgroup barrier of the staged kernels: LDS traffic of this wave done (lgkmcnt), then s_barrier -- NOT __syncthreads(),
// whose fence also drains vmcnt: a storer would sit at every barrier until its global stores are acknowledged (microseconds
// under load) and hold up the computing waves with it.  Loaders wait for their LDS-DMA (vmcnt) themselves before they arrive.
__device__ __forceinline__ void c3_bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Stack-on-load source of a FIRST layer: the 4 input planes of sample b are a window of 4
// consecutive uint8 frames of a single-frame store -- f + (b / T) * bs + (b % T) * H * W bytes (rollout step: T = 1 and
// bs = the slot stride; update: sample n = slot * T + t) -- of which the oldest 4 - nv[b * nv_s] are planes from before
// the env's last reset and read as zero (utils.py:37-42).  The loader waves expand the bytes to the fp32 image the
// computing waves read: same values, same summation order as from the fp32 state (runner.py:199 `FloatTensor(state)`).
struct C3U8 {
  const unsigned char* f; long bs; const int* nv; long nv_s; long T;
};

// loader waves, uint8 source: the rows y0 .. y0 + SR - 1 of the planes c % NL == lw of one sample -> the fp32 image
// (rows of WP = W + 4 floats: 4 floats of pad, zeroed once per launch by c3_u8_zero_pads, then the row)
template <int H, int W, int SR, int WP, int PLANE, int NL>
__device__ __forceinline__ void c3_u8_stage(float* __restrict__ buf, const C3U8& u, long b, int y0, int lw, int lane) {
  static_assert(W % 4 == 0, "rows of whole dwords");
  constexpr int RP = W / 4, PPL = SR * RP, NQ8 = (PPL + 63) / 64;
  const unsigned bu_ = (unsigned)b, slot_ = bu_ / (unsigned)u.T;                  // (b < 2^31: 32-bit division)
  const unsigned char* __restrict__ fb = u.f + (long)slot_ * u.bs + (long)(bu_ - slot_ * (unsigned)u.T) * (H * W);
  const int nvl = u.nv ? u.nv[b * u.nv_s] : 4;
  static_assert(4 % NL == 0, "planes dealt evenly to the loader waves");
  constexpr int NPL = 4 / NL;                  // planes of this loader wave: c = lw + NL * s
  unsigned v[NPL][NQ8];                        // ALL loads first (one latency), then expand + write
#pragma unroll
  for (int s = 0; s < NPL; ++s) {
    const int c = lw + NL * s;
    const bool live = c >= 4 - nvl;
#pragma unroll
    for (int q = 0; q < NQ8; ++q) {
      const int pi = q * 64 + lane, r = pi / RP, i = pi - r * RP, y = y0 + r;
      v[s][q] = (pi < PPL && live && y >= 0 && y < H) ? *reinterpret_cast<const unsigned*>(fb + ((long)c * H + y) * W + 4 * i) : 0u;
    }
  }
#pragma unroll
  for (int s = 0; s < NPL; ++s) {
    const int c = lw + NL * s;
#pragma unroll
    for (int q = 0; q < NQ8; ++q) {
      const int pi = q * 64 + lane, r = pi / RP, i = pi - r * RP;
      if (pi < PPL)
        *reinterpret_cast<float4*>(buf + c * PLANE + r * WP + 4 + 4 * i) =
            make_float4((float)(v[s][q] & 0xffu), (float)((v[s][q] >> 8) & 0xffu), (float)((v[s][q] >> 16) & 0xffu),
                        (float)(v[s][q] >> 24));
    }
  }
}
// The same in two halves for a loader that keeps one sample's bytes in flight while the previous one is expanded
// (c3w_kernel): NPL * NQ8 UNCONDITIONAL dword loads (clamped addresses: the instruction count feeds a counted vmcnt),
// the zero planes / rows are selected when the bytes are expanded.
template <int H, int W, int SR, int NL>
struct C3U8Regs {
  static constexpr int RP = W / 4, PPL = SR * RP, NQ8 = (PPL + 63) / 64, NPL = 4 / NL, NLD = NPL * NQ8;
  unsigned v[NPL][NQ8];
  int nvl, y0;
};
template <int H, int W, int SR, int NL>
__device__ __forceinline__ void c3_u8_load(C3U8Regs<H, W, SR, NL>& rg, const C3U8& u, long b, int y0, int lw, int lane) {
  using RG = C3U8Regs<H, W, SR, NL>;
  const unsigned bu_ = (unsigned)b, slot_ = bu_ / (unsigned)u.T;                  // (b < 2^31: 32-bit division)
  const unsigned char* __restrict__ fb = u.f + (long)slot_ * u.bs + (long)(bu_ - slot_ * (unsigned)u.T) * (H * W);
  rg.nvl = u.nv ? u.nv[b * u.nv_s] : 4;
  rg.y0 = y0;
#pragma unroll
  for (int s = 0; s < RG::NPL; ++s) {
    const int c = lw + NL * s;
#pragma unroll
    for (int q = 0; q < RG::NQ8; ++q) {
      const int pi = min(q * 64 + lane, RG::PPL - 1), r = pi / RG::RP, i = pi - r * RG::RP;
      const int y = min(max(y0 + r, 0), H - 1);
      rg.v[s][q] = *reinterpret_cast<const unsigned*>(fb + ((long)c * H + y) * W + 4 * i);
    }
  }
}
template <int H, int W, int SR, int WP, int PLANE, int NL>
__device__ __forceinline__ void c3_u8_expand(float* __restrict__ buf, const C3U8Regs<H, W, SR, NL>& rg, int lw, int lane) {
  using RG = C3U8Regs<H, W, SR, NL>;
#pragma unroll
  for (int s = 0; s < RG::NPL; ++s) {
    const int c = lw + NL * s;
    const bool live = c >= 4 - rg.nvl;
#pragma unroll
    for (int q = 0; q < RG::NQ8; ++q) {
      const int pi = q * 64 + lane, r = pi / RG::RP, i = pi - r * RG::RP, y = rg.y0 + r;
      const unsigned vv = (live && y >= 0 && y < H) ? rg.v[s][q] : 0u;
      if (pi < RG::PPL)
        *reinterpret_cast<float4*>(buf + c * PLANE + r * WP + 4 + 4 * i) =
            make_float4((float)(vv & 0xffu), (float)((vv >> 8) & 0xffu), (float)((vv >> 16) & 0xffu), (float)(vv >> 24));
    }
  }
}
template <int SR, int WP, int PLANE>
__device__ __forceinline__ void c3_u8_zero_pads(float* __restrict__ buf, int lane, int lw, int nl) {
  for (int i = lw * 64 + lane; i < 4 * SR; i += 64 * nl)
    *reinterpret_cast<float4*>(buf + (i / SR) * PLANE + (i % SR) * WP) = make_float4(0.f, 0.f, 0.f, 0.f);
}

template <int N>
__device__ __forceinline__ void c3_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct C3P {
  const float* src; long src_bs;      // source tensor (B, CS, H, W) and its sample stride (floats)
  const float* frag;                  // [chunk][tap][c4][m][64] weight fragments (c3_prep_kernel)
  const float* bias;                  // forward: (CD) or nullptr
  const float* mask;                  // backward-data: (B, CD, H, W) activation whose sign gates the gradient, or nullptr
  float* out; long out_bs;            // (B, CD, OH, OW)
  const float* zero;                  // >= 64 B of zeros in HBM (16-B aligned)
  int B, relu;
  unsigned long long* dbg;            // debug: phase stamps of workgroup 0 / wave 0 (a2c_debug_c3_timing), or nullptr
  // sign words (staged kernels): bit x & 31 of word [c][y][x >> 5] of a sample = (activation[c][y][x] > 0)
  unsigned* sg_out;                   // forward: written for the output, or nullptr
  const unsigned* sg_in;              // backward-data: the ReLU mask of the layer below as sign words, or nullptr
  long sg_bs;                         // sample stride of the sign words (in words)
  C3U8 u8;                            // first layers: uint8 frame-store source instead of src (u8.f != nullptr)
  int prio;                           // loader waves at raised priority
  float* w1_slab;                     // c3bs_kernel<..., W1>: per (workgroup, w1 wave) partial [16 x 36 | 16] of the FIRST layer's gradient
};
// (loader waves issue a few hundred instructions per chunk between the computing waves' MFMA streams: at equal priority they
// were the critical path of the weight-gradient kernels; A2C_C3_PRIO=0 = round 3's schedule, for A/B runs)
static int c3_prio() { static const int v = getenv("A2C_C3_PRIO") ? atoi(getenv("A2C_C3_PRIO")) : 1; return v; }

template <int CS, int CD, int H, int W, int S, int R, int KCO = 0, int NLO = 0>
struct C3Geo {
  static constexpr int KC = KCO ? KCO : (CS >= 8 ? 8 : 4);      // source channels per chunk (c3_kc() tells c3_prep the same)
  static constexpr int NCH = CS / KC;
  static constexpr int C4 = KC / 4;
  static constexpr int MT = (CD + 15) / 16;
  static constexpr int OH = (H - 1) / S + 1, OW = (W - 1) / S + 1;
  static constexpr int NBAND = (OH + R - 1) / R;
  static constexpr int NPIX = R * OW;
  static constexpr int NT = (NPIX + 15) / 16;         // 16-pixel tiles per band
  static constexpr int NW = 8;                         // computing waves (two per SIMD); waves NW .. NW+NL-1 are the loaders
  static constexpr int NL = NLO ? NLO : (W % 4 != 0) ? 4 : 2;     // (dword pieces: four loader waves share the DMA instructions)
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
  static constexpr int SRW = (OW + 31) / 32, SBITC = R * SRW;          // forward with sign words: words per row / channel band
  static constexpr int SBITB = ((CD * SBITC + 3) / 4) * 4;
  static constexpr size_t LDS_BYTES_SG = (2 * (size_t)BUF + SBITB) * 4;
  static constexpr bool VEC = (OH * OW) % 4 == 0 && MROW % 4 == 0;     // 16-byte output stores
  static_assert(CS % KC == 0 && PL0 % PB == 0 && (KC * PLANE) % 4 == 0, "shape");
  // stride 1 reads the right halo of a plane's last row from the next plane's pad piece: PLANE == PL0, or the kernel
  // zeroes the gap between the planes once (GAP floats per plane, never written by the DMA)
  static constexpr int GAP = PLANE - PL0;
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

// SG (forward only): the epilogue also sets the sign bits of what it stores in an LDS band of sign words, and an ELEVENTH
// wave carries the finished band's words to HBM under the next band's first chunk (needs NCH >= 2: a barrier between its
// copy-and-clear and the next epilogue).  The computing waves and the stores are the direct-store kernel's.
// The body of c3_kernel as a device function over the tiles tile0, tile0 + tstride, ... (nmine of them): the stand-alone
// kernel deals the B * NBAND tiles of a launch round robin to its workgroups; the CHAIN kernel below (c3_chain_kernel) hands
// a workgroup the bands of ONE sample, layer after layer.  NLO: loader waves of the block (0: the layer's own count),
// XSG: the block has a sign-word wave (index NW + NL) whether or not this layer writes sign words -- a layer without them
// parks it at the barriers (every wave of a workgroup must arrive at every s_barrier).
template <int CS, int CD, int H, int W, int S, int R, bool BWD, int KCO, bool SG, int NLO, bool XSG, bool CH, int D = 2>
__device__ __forceinline__ void c3_body(const C3P& p, float* __restrict__ lds, const long tile0_, const long nmine_) {
  using G = C3Geo<CS, CD, H, W, S, R, KCO, NLO>;
  // D chunk buffers in a ring, the loaders LOOK = D - 1 chunks ahead of the computing waves.  D = 2 is the plain double
  // buffer (the loader waits for everything it issued: vmcnt(0)); D = 3 keeps a second chunk in flight across the barrier
  // with a COUNTED vmcnt -- every chunk is the same number of DMA instructions per loader wave (NI below) -- so that a
  // chunk's DMA latency (2-3 us under load against ~1.4 us of MFMAs per chunk at rollout batch) overlaps two chunks' work.
  constexpr int LOOK = D - 1;
  static_assert(D == 2 || (D == 3 && !BWD && !(CS == 4 && W % 4 == 0)), "ring of three: forward, fp32 source");
  static_assert(G::KC % G::NL == 0 || D == 2, "counted vmcnt: every loader wave carries the same planes of every chunk");
  static_assert(!SG || (!BWD && G::NCH >= 2), "sign words: forward, two chunks or more");
  static_assert(!SG || XSG, "a layer that writes sign words needs the block's sign-word wave");
  unsigned* const sbits = reinterpret_cast<unsigned*>(lds + D * G::BUF);      // (SG)
  int tid_ = threadIdx.x;
  long nmine = nmine_;
  if (CH) {
    // (opaque copies: in the chain kernel the compiler otherwise (a) hoists every layer's lane-derived tables -- base[],
    // roff[], rrow[], bias -- out of the sample loop and keeps them ALL live and (b) unrolls the work-item loop of the
    // one-band layers, whose trip count it can see: 574 spilled registers, 3x the run time)
    asm volatile("" : "+v"(tid_));
    asm volatile("" : "+s"(nmine));
  }
  const long tile0 = CH ? tile0_ : (long)blockIdx.x;
  const long tstride = CH ? 1L : (long)gridDim.x;
  const int tid = tid_, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long nwork = nmine * G::NCH;              // work item k = (tile k / NCH, chunk k % NCH)
  // the zeros that never change: behind each buffer's last plane
  if (tid < 16) {
#pragma unroll
    for (int bsel = 0; bsel < D; ++bsel) {
      lds[bsel * G::BUF + G::IMG - 8 + (tid & 7)] = 0.f;
      if (G::KC * G::PLANE < G::IMG - 8)                  // (plane sizes that are not a multiple of 4 floats)
        lds[bsel * G::BUF + G::KC * G::PLANE + (tid & 3)] = 0.f;
    }
  }
  if (S == 1 && G::GAP > 0) {                     // (stride 1 reads one float past a plane's last row)
    constexpr int GP = G::GAP > 0 ? G::GAP : 1;
    for (int i = tid; i < D * G::KC * GP; i += 64 * (G::NW + G::NL + (XSG ? 1 : 0))) {
      const int bsel = i / (G::KC * GP), r = i - bsel * (G::KC * GP);
      lds[bsel * G::BUF + (r / GP) * G::PLANE + G::PL0 + (r % GP)] = 0.f;
    }
  }
  if (XSG && !SG && w == G::NW + G::NL) {         // (a chain layer without sign words: the wave only keeps the barrier count)
    __syncthreads();
    for (long k = 0; k < nwork; ++k) __syncthreads();
    return;
  }
  if (SG && w == G::NW + G::NL) {
    // ------------------------------------------------------------------ sign-word writer
    for (int i = lane; i < G::SBITB; i += 64) sbits[i] = 0u;
    c3_bar();
    for (long k = 0; k < nwork; ++k) {
      c3_bar();                                   // (raw: this wave must not wait for its stores' acknowledgements here)
      if ((int)((unsigned)k % (unsigned)G::NCH) == G::NCH - 1 && p.sg_out != nullptr) {      // the band's epilogue is behind that barrier
        const long tile = tile0 + ((unsigned)k / (unsigned)G::NCH) * tstride;
        const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
        const int band = (int)(tile - b * G::NBAND);
        const int nw = min(R, G::OH - band * R) * G::SRW;
        unsigned* __restrict__ gb = p.sg_out + b * p.sg_bs + (long)band * R * G::SRW;
        // all channels' words as one list, four per lane and pass: this wave is at the next barrier late by whatever this takes
        constexpr int TOT = CD * G::SBITC;
#pragma unroll 1
        for (int i0 = lane; i0 < TOT; i0 += 256) {
          int idx[4];
          float wv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            idx[e] = min(i0 + 64 * e, TOT - 1);
            wv[e] = c3_lds_read32(sbits + idx[e]);
          }
          c3_lds_wait(wv[0], wv[1], wv[2], wv[3]);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int c = idx[e] / G::SBITC, i = idx[e] - c * G::SBITC;
            if (i0 + 64 * e < TOT && i < nw) {
              c3_gstore32(reinterpret_cast<float*>(gb + (long)c * G::OH * G::SRW + i), wv[e]);
              c3_lds_zero32(sbits + idx[e]);
            }
          }
        }
      }
    }
    return;
  }
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves (planes / fragment pieces dealt round robin)
    const int lw = w - G::NW;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    int roff[G::NQ], rrow[G::NQ];                 // this lane's piece of DMA instruction q: source offset / image row
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / (G::WP / G::PB), i = pi - r * (G::WP / G::PB);
      rrow[q] = (pi < G::PP) ? (i == 0 ? -100000 : r) : -200000;      // pad piece: zeros; beyond the plane: no lane
      roff[q] = r * W + G::PB * (i - 1);
    }
    constexpr bool U8OK = CS == 4 && W % 4 == 0 && !BWD;
    const bool u8 = U8OK && p.u8.f != nullptr;
    if constexpr (U8OK) {
      if (u8) {
        c3_u8_zero_pads<G::SR, G::WP, G::PLANE>(lds, lane, lw, G::NL);
        c3_u8_zero_pads<G::SR, G::WP, G::PLANE>(lds + G::BUF, lane, lw, G::NL);
      }
    }
    auto dma = [&](long k) {
      const long tile = tile0 + ((unsigned)k / (unsigned)G::NCH) * tstride;
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int y0 = band * R * S - 1;
      float* __restrict__ buf = lds + (D == 2 ? (k & 1) : (long)((unsigned)k % (unsigned)D)) * G::BUF;
      const float* __restrict__ sb = p.src + b * p.src_bs + ((long)ch * G::KC * H + y0) * W;
      const bool staged = U8OK && u8;              // (the uint8 window is expanded below, behind the fragment DMA)
#pragma unroll
      for (int c = 0; c < G::KC; ++c) {
        if (staged || c % G::NL != lw) continue;
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
      if constexpr (U8OK) {
        if (u8) c3_u8_stage<H, W, G::SR, G::WP, G::PLANE, G::NL>(buf, p.u8, b, y0, lw, lane);
      }
    };
    // backward-data: the band's ReLU mask, one linear run per channel, issued while the band's FIRST chunk computes
    auto dma_mask = [&](long k) {      // the slice of the band's mask that travels beside chunk k (all but the last chunk carry one)
      const long tile = tile0 + ((unsigned)k / (unsigned)G::NCH) * tstride;
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int npx = min(R, G::OH - band * R) * G::OW;
      float* __restrict__ mb = lds + D * G::BUF;
      constexpr int CPP = G::NCH > 1 ? (CD + G::NCH - 2) / (G::NCH - 1) : CD;
      const int part = (int)((unsigned)k % (unsigned)G::NCH);
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
    if constexpr (D == 2) {
      if (nwork > 0) dma(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the chunk has landed in LDS
      __syncthreads();
      for (long k = 0; k < nwork; ++k) {
        if (k + 1 < nwork) dma(k + 1);
        if (BWD && p.mask != nullptr && (int)((unsigned)k % (unsigned)G::NCH) < G::NCH - 1) dma_mask(k);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
      }
    } else {
      // DMA instructions of one chunk issued by loader wave lw: its planes' pieces + its share of the fragment pieces (a
      // partly filled instruction still issues: the predicates are per lane)
      constexpr int NIP = (G::KC / G::NL) * G::NQ;
      auto wait_younger = [&](bool one_younger) {      // everything but the youngest chunk (if any) has landed
        if (!one_younger) { c3_wait_vm<0>(); return; }
        static_assert(NIP + (G::NFQ + G::NL - 1) / G::NL <= 63, "vmcnt is 6 bits");
        if (lw == 0) c3_wait_vm<NIP + (G::NFQ + G::NL - 1) / G::NL>();
        else if (lw == 1) c3_wait_vm<NIP + (G::NFQ + G::NL - 2 > 0 ? (G::NFQ + G::NL - 2) / G::NL : 0)>();
        else if (lw == 2) c3_wait_vm<NIP + (G::NFQ + G::NL - 3 > 0 ? (G::NFQ + G::NL - 3) / G::NL : 0)>();
        else c3_wait_vm<NIP + (G::NFQ + G::NL - 4 > 0 ? (G::NFQ + G::NL - 4) / G::NL : 0)>();
      };
      static_assert(G::NL <= 4, "wait_younger");
      for (int k = 0; k < LOOK; ++k)
        if (k < nwork) dma(k);
      // (raw barriers: __syncthreads()' fence would drain vmcnt and with it the chunk that is meant to stay in flight)
      wait_younger(nwork > 1);                       // chunk 0 has landed, chunk 1 may still be in flight
      c3_bar();
      for (long k = 0; k < nwork; ++k) {
        if (k + LOOK < nwork) dma(k + LOOK);         // into the slot chunk k - 1 left
        wait_younger(k + 2 < nwork);                 // chunk k + 1 has landed before the barrier that opens it
        c3_bar();
      }
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
    const float* __restrict__ img = lds + (D == 2 ? (k & 1) : (long)((unsigned)k % (unsigned)D)) * G::BUF;
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
    if ((int)((unsigned)k % (unsigned)G::NCH) == G::NCH - 1) {
      // ---- the band is complete: accumulators -> HBM.  With the PIXELS as the MFMA's A operand the D tile is
      // [pixel][channel]: lane (j, g) holds pixels 4g .. 4g+3 of channel j -- one 16-byte store per lane and tile
      const long tile = tile0 + ((unsigned)k / (unsigned)G::NCH) * tstride;
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int npix_ok = min(R, G::OH - band * R) * G::OW;
      const long o0 = b * p.out_bs + (long)band * R * G::OW;
      const float* __restrict__ mb = lds + D * G::BUF;       // backward-data: the band's mask, staged by the loaders
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
            if (SG && p.sg_out != nullptr && cd < CD) c3_sign4<G::OW, G::SRW>(sbits + cd * G::SBITC, p0, npix_ok, v);
          }
          float* __restrict__ o = p.out + o0 + (long)cd * G::OH * G::OW + p0;
          if (cd < CD) {
            if (G::VEC && p0 + 3 < npix_ok) c3_gstore128(o, v);
            else {                                             // rows that are not 16-byte aligned / the band's ragged end
              if (p0 < npix_ok) c3_gstore32(o, v.x);
              if (p0 + 1 < npix_ok) c3_gstore32(o + 1, v.y);
              if (p0 + 2 < npix_ok) c3_gstore32(o + 2, v.z);
              if (p0 + 3 < npix_ok) c3_gstore32(o + 3, v.w);
            }
          }
          acc[u][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    __syncthreads();
  }
}

template <int CS, int CD, int H, int W, int S, int R, bool BWD, int KCO = 0, bool SG = false>
__global__ __launch_bounds__((64 * (8 + C3Geo<CS, CD, H, W, S, R, KCO>::NL + (SG ? 1 : 0)))) void c3_kernel(C3P p) {
  using G = C3Geo<CS, CD, H, W, S, R, KCO>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  c3_body<CS, CD, H, W, S, R, BWD, KCO, SG, 0, SG, false>(p, lds, 0, nmine);
}

// ---------------------------------------------------------------------------------------------------------------
// Forward CHAIN of a rollout step (GRUModel conv2 .. conv5, models.py:570-636): one workgroup = one SAMPLE, running the
// four stride-2 layers one after the other -- layer l+1 only needs layer l's rows of the SAME sample, so there is no
// grid-wide dependency: the workgroup stores layer l's bands (the update's activation stash needs them in HBM anyway),
// drains its stores, and its loader waves read them back (L2 / infinity cache: the CU that wrote them) as layer l+1's
// source.  Per env step of 256 envs the four launches (16-58 us each, a fill and a tail per launch, 3 of 8 computing
// waves busy on conv5's single 6 x 6 band per workgroup) become one launch in which every CU is busy from the first band
// of conv2 to the last tile of conv5.  Same bodies, same tile shapes, same summation order: bit-identical to the four
// launches (test).  Block = 8 computing waves + 4 loader waves + the sign-word wave.
struct C3Chain4 { C3P l[4]; };
__device__ __forceinline__ void c3_layer_sync() {
  // Every store of the finished layer (inline-asm stores the compiler does not track) acknowledged by L2, then the barrier:
  // WORKGROUP scope is all this needs -- writer and reader are waves of one workgroup, i.e. one CU, one (write-through) L1
  // and one XCD's L2.  (An agent-scope release / acquire here is a `buffer_wbl2 sc1` + `buffer_inv sc1` on this multi-XCD
  // part: a write-back and invalidate of the XCD's whole L2 four times per sample -- measured 700 us per env step instead
  // of 125.)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool SG2, bool SG3, bool SG4>
__global__ __launch_bounds__(64 * 13) void c3_chain_gru_kernel(C3Chain4 cp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  using G2 = C3Geo<16, 24, 84, 84, 2, 6, 0, 4>;
  using G3 = C3Geo<24, 32, 42, 42, 2, 11, 0, 4>;
  using G4 = C3Geo<32, 48, 21, 21, 2, 11, 0, 4>;
  using G5 = C3Geo<48, 64, 11, 11, 2, 6, 24, 4>;
  for (long b = blockIdx.x; b < cp.l[0].B; b += gridDim.x) {
    c3_body<16, 24, 84, 84, 2, 6, false, 0, SG2, 4, true, true>(cp.l[0], lds, b * G2::NBAND, G2::NBAND);
    c3_layer_sync();
    c3_body<24, 32, 42, 42, 2, 11, false, 0, SG3, 4, true, true>(cp.l[1], lds, b * G3::NBAND, G3::NBAND);
    c3_layer_sync();
    c3_body<32, 48, 21, 21, 2, 11, false, 0, SG4, 4, true, true>(cp.l[2], lds, b * G4::NBAND, G4::NBAND);
    c3_layer_sync();
    c3_body<48, 64, 11, 11, 2, 6, false, 24, false, 4, true, true>(cp.l[3], lds, b * G5::NBAND, G5::NBAND);
    c3_layer_sync();
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
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    int roff[G::NQ], rrow[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / G::WP, x = pi - r * G::WP;
      rrow[q] = (pi < G::PL0) ? (x == WO ? -100000 : r) : -200000;     // zero column; beyond the plane: no lane
      roff[q] = r * WO + x;
    }
    auto dma = [&](long k) {
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
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
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int rows = 2 * min(RQ, HO - q0);
      float* __restrict__ mb = lds + 2 * G::BUF;
      constexpr int CPP = (CI + G::NCH - 2) / (G::NCH - 1);
      const int part = (int)((unsigned)k % (unsigned)G::NCH);
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
      if (p.mask != nullptr && (int)((unsigned)k % (unsigned)G::NCH) < G::NCH - 1) dma_mask(k);
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
    if ((int)((unsigned)k % (unsigned)G::NCH) == G::NCH - 1) {
      // ---- the band is complete: lane (j, g) holds class pixels 4g .. 4g+3 of channel j, both x parities: rows of dX
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
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
                if (okB && !split) c3_gstore128(oc + la, v);
                else {
                  c3_gstore64(oc + la, make_float2(v.x, v.y));
                  if (okB) c3_gstore64(oc + lb, make_float2(v.z, v.w));
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
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;      // (the kernels walk chunks with 32-bit arithmetic)
  const long cap = (long)per_cu * cus;
  const int grid = (int)(total < cap ? total : cap);
  hipLaunchKernelGGL((c3b_kernel<CO, CI, HO, WO, RQ, KC>), dim3(grid), dim3(640), G::LDS_BYTES, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

// per-role phase stamps of workgroup 0 in the staged kernels (a2c_debug_c3_timing): compiled in with
// -DA2C_C3_STAMPS only -- their registers push the two instances at the register cap into scratch
#ifdef A2C_C3_STAMPS
constexpr bool C3_STAMPS = true;
#else
constexpr bool C3_STAMPS = false;
#endif

// ---------------------------------------------------------------------------------------------------------------
// STAGED variants of the two kernels above: the finished band does not leave through the computing waves.
//
// Measured on the kernels above (in-kernel stamps, profiles/README.md round 3): a workgroup's band ends with all eight
// computing waves pushing 32-97 KB through the CU's store path (7-10 B/clk) while its matrix cores idle, then the
// matrix phase of the next band runs while the store path idles -- conv1's forward spends 4.6k cycles per band in
// MFMAs and 8k in the epilogue.  Here the computing waves write the band (bias + ReLU, or the ReLU mask, applied) into
// an LDS image [channel][band pixels] (ds_write_b128, < 1k cycles) and go on with the next band; two more waves -- the
// STORERS -- copy that image to HBM (ds_read_b128 -> global_store_dwordx4, full 1 KB wave stores) under the next band's
// MFMAs, a slice of the channels per chunk interval.  Barriers per band: one per chunk as before, plus one (X) between
// the last chunk's MFMAs and the write of the image (the storers have read the previous band out of it by then).
// Store-bound layers now run at the store path's rate, matrix-bound ones lose the epilogue.
//
// The ReLU mask of backward-data arrives as SIGN WORDS (one bit per activation, rows padded to 32-bit words) produced
// by the forward of the layer below -- by its storers, from the image they copy anyway: 1/32 of the mask's HBM reads,
// and the LDS the float mask band took is what the output image lives in.
// DS: TWO output images.  With one, a band is  max(MFMAs, drain of the previous band, loads)  THEN the image write, with the storers
// idle meanwhile and the computing waves waiting (barrier X) for the drain to finish; conv1's forward is store-path-bound (64 KB per
// band at 8-10 B/clk = 3.4 us against 1.9 us of MFMAs + 0.9 of image write: measured by switching the phases off one at a time).  With
// two, the computing waves write band k into image k & 1 straight after its MFMAs while the storers drain image (k - 1) & 1: no X
// barrier, the storers never idle.
template <int CS, int CD, int H, int W, int S, int R, bool DS = false>
struct C3SGeo : C3Geo<CS, CD, H, W, S, R, 0, 2> {          // (two loaders + two storers + eight computing waves: three per SIMD)
  using G0 = C3Geo<CS, CD, H, W, S, R, 0, 2>;
  static constexpr int NS = DS ? 4 : 2;                            // storer waves
  static constexpr int MROWP = ((G0::MROW + 3) / 8) * 8 + 4;        // channel pitch of the image: 4 (mod 8) floats, so the 8
                                                                   // channels of a ds_write_b128 lane group hit 8 distinct slots
  static constexpr int STAGE = CD * MROWP;
  static constexpr int RW = (G0::OW + 31) / 32;                     // sign words per row
  static constexpr int BITC = R * RW;                              // ... per channel band
  static constexpr int BITB = ((CD * BITC + 3) / 4) * 4;
  static constexpr int SIMG = STAGE + BITB;                        // one output image + its sign words
  static constexpr size_t LDS_BYTES_S = (2 * (size_t)G0::BUF + (DS ? 2 : 1) * SIMG) * 4;
  static constexpr int NTHR = 64 * (G0::NW + G0::NL + NS);
  static_assert(MROWP >= ((G0::MROW + 3) / 4) * 4, "pitch");
};

struct C3Drain { const float* s; float* o; unsigned* bb; unsigned* go; bool ok; };

// LDS accesses of the STORER waves go through inline asm.  hipcc's waitcnt insertion treats every LDS access of a kernel
// that also issues LDS-DMA as a possible reader of an in-flight DMA and puts `s_waitcnt vmcnt(0)` in front of it -- in a
// storer that is a wait for the acknowledgement of every global store it has issued so far (2-4 us under load) before
// each batch of reads: 8 KB per round trip.  (The storers never touch what the DMA writes.)
__device__ __forceinline__ f32x4 c3_lds_read128(const float* p) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(c3_lds_addr(p)));
  return v;
}
// all LDS reads issued so far have returned; the values are tied to the wait so that no use can move above it
__device__ __forceinline__ void c3_lds_wait(f32x4& a, f32x4& b, f32x4& c, f32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory");
}

// NH (1 or 2) channel bands: LDS image -> HBM.  Straight-line code with the pieces in named registers: all LDS reads
// first, then the stores -- a storer is ONE wave, it cannot afford a wait per piece (and a private array here ends up
// in scratch memory in the instances at the register cap).
// d.s / d.o: the channel's row of the image / its band in HBM (npx floats, contiguous).  At most 4 pieces per lane.
template <int MROW, bool VEC, int NH = 2>
__device__ __forceinline__ void c3_drain_pair(const C3Drain d0, const C3Drain d1, int npx, int lane) {
  const bool on1 = NH > 1 && d1.ok;
  if constexpr (VEC) {
    constexpr int NIT = (MROW / 4 + 63) / 64;
    static_assert(NIT <= 4, "pieces per lane");
    const int nf = npx >> 2;                                 // (npx % 4 == 0: OW even and whole rows, see C3Geo::VEC)
    const int f0 = lane, f1 = 64 + lane, f2 = 128 + lane, f3 = 192 + lane;
    const f32x4 z = (f32x4){0.f, 0.f, 0.f, 0.f};
#define C3_LD(D, F) c3_lds_read128((D).s + 4 * min((F), nf - 1))
    f32x4 a0 = C3_LD(d0, f0), a1 = NIT > 1 ? C3_LD(d0, f1) : z, a2 = NIT > 2 ? C3_LD(d0, f2) : z, a3 = NIT > 3 ? C3_LD(d0, f3) : z;
    f32x4 b0 = NH > 1 ? C3_LD(d1, f0) : z, b1 = (NH > 1 && NIT > 1) ? C3_LD(d1, f1) : z,
          b2 = (NH > 1 && NIT > 2) ? C3_LD(d1, f2) : z, b3 = (NH > 1 && NIT > 3) ? C3_LD(d1, f3) : z;
#undef C3_LD
    c3_lds_wait(a0, a1, a2, a3);
    if (NH > 1) c3_lds_wait(b0, b1, b2, b3);
#define C3_ST(D, F, V, ON) do { if ((ON) && (F) < nf) *reinterpret_cast<f32x4*>((D).o + 4 * (F)) = (V); } while (0)
    C3_ST(d0, f0, a0, true);
    if (NIT > 1) C3_ST(d0, f1, a1, true);
    if (NIT > 2) C3_ST(d0, f2, a2, true);
    if (NIT > 3) C3_ST(d0, f3, a3, true);
    if (NH > 1) {
      C3_ST(d1, f0, b0, on1);
      if (NIT > 1) C3_ST(d1, f1, b1, on1);
      if (NIT > 2) C3_ST(d1, f2, b2, on1);
      if (NIT > 3) C3_ST(d1, f3, b3, on1);
    }
#undef C3_ST
  } else {                                                   // channel bands that do not start on 16-byte boundaries: dwords
    constexpr int NIT = (MROW + 63) / 64;
    static_assert(NIT <= 4, "pieces per lane");
    const int q0 = lane, q1 = 64 + lane, q2 = 128 + lane, q3 = 192 + lane;
#define C3_LD(D, Q) c3_lds_read32((D).s + min((Q), npx - 1))
    float a0 = C3_LD(d0, q0), a1 = NIT > 1 ? C3_LD(d0, q1) : 0.f, a2 = NIT > 2 ? C3_LD(d0, q2) : 0.f, a3 = NIT > 3 ? C3_LD(d0, q3) : 0.f;
    float b0 = NH > 1 ? C3_LD(d1, q0) : 0.f, b1 = (NH > 1 && NIT > 1) ? C3_LD(d1, q1) : 0.f,
          b2 = (NH > 1 && NIT > 2) ? C3_LD(d1, q2) : 0.f, b3 = (NH > 1 && NIT > 3) ? C3_LD(d1, q3) : 0.f;
#undef C3_LD
    c3_lds_wait(a0, a1, a2, a3);
    if (NH > 1) c3_lds_wait(b0, b1, b2, b3);
#define C3_ST(D, Q, V, ON) do { if ((ON) && (Q) < npx) (D).o[(Q)] = (V); } while (0)
    C3_ST(d0, q0, a0, true);
    if (NIT > 1) C3_ST(d0, q1, a1, true);
    if (NIT > 2) C3_ST(d0, q2, a2, true);
    if (NIT > 3) C3_ST(d0, q3, a3, true);
    if (NH > 1) {
      C3_ST(d1, q0, b0, on1);
      if (NIT > 1) C3_ST(d1, q1, b1, on1);
      if (NIT > 2) C3_ST(d1, q2, b2, on1);
      if (NIT > 3) C3_ST(d1, q3, b3, on1);
    }
#undef C3_ST
  }
}

// a channel band's sign words (set by the computing waves while they wrote the image): LDS -> HBM, and back to zero
__device__ __forceinline__ void c3_drain_signs(const C3Drain d, int nwords, int lane) {
  for (int i = lane; i < nwords; i += 64) {
    float w0 = c3_lds_read32(d.bb + i), z1 = 0.f, z2 = 0.f, z3 = 0.f;
    c3_lds_wait(w0, z1, z2, z3);
    d.go[i] = __float_as_uint(w0);
    c3_lds_zero32(d.bb + i);
  }
}

template <int CS, int CD, int H, int W, int S, int R, bool BWD, bool DS = false>
__global__ __launch_bounds__((C3SGeo<CS, CD, H, W, S, R, DS>::NTHR)) void c3s_kernel(C3P p) {
  using G = C3SGeo<CS, CD, H, W, S, R, DS>;
  static_assert(!BWD || G::OW % 4 == 0, "backward-data reads a lane's four mask bits from one word");
  static_assert(!DS || !BWD, "two output images: forward only");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const stage0 = lds + 2 * G::BUF;
  // the output image (and its sign words) of the workgroup's n-th band
  auto stage_of = [&](long n) { return stage0 + (DS ? (int)(n & 1) * G::SIMG : 0); };
  auto bitb_of = [&](long n) { return reinterpret_cast<unsigned*>(stage_of(n) + G::STAGE); };
  unsigned* const bitb = bitb_of(0);                  // (backward-data: the incoming sign words)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;              // work item k = (tile k / NCH, chunk k % NCH)
  // per-role stamps of workgroup 0 (-DA2C_C3_STAMPS builds, a2c_debug_c3_timing, tools/dbg/c3s_stamps.py): [role][band][event]
  const bool xst = C3_STAMPS && p.dbg != nullptr && blockIdx.x == 0 && lane == 0;
#define C3S_ST(ROLE, K, E) do { if (xst && (K) < 16) p.dbg[((ROLE) * 16 + (K)) * 4 + (E)] = wall_clock64(); } while (0)
  if (tid < 16) {
    lds[G::IMG - 8 + (tid & 7)] = 0.f;
    lds[G::BUF + G::IMG - 8 + (tid & 7)] = 0.f;
    if (G::KC * G::PLANE < G::IMG - 8) {
      lds[G::KC * G::PLANE + (tid & 3)] = 0.f;
      lds[G::BUF + G::KC * G::PLANE + (tid & 3)] = 0.f;
    }
  }
  if (S == 1 && G::GAP > 0) {                     // (stride 1 reads one float past a plane's last row)
    constexpr int GP = G::GAP > 0 ? G::GAP : 1;
    for (int i = tid; i < 2 * G::KC * GP; i += G::NTHR) {
      const int bsel = i / (G::KC * GP), r = i - bsel * (G::KC * GP);
      lds[bsel * G::BUF + (r / GP) * G::PLANE + G::PL0 + (r % GP)] = 0.f;
    }
  }
  if (w >= G::NW + G::NL) {
    // ------------------------------------------------------------------ storer waves
    const int sw = w - G::NW - G::NL;
    if (!BWD && sw == 0)
      for (int i = lane; i < G::BITB; i += 64) {
        bitb[i] = 0u;
        if (DS) bitb_of(1)[i] = 0u;
      }
    c3_bar();
    const bool signs = !BWD && p.sg_out != nullptr;
    auto drain = [&](long tile, int part, long n) {
      float* const stage = stage_of(n);
      unsigned* const bitb = bitb_of(n);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int nrows = min(R, G::OH - band * R), npx = nrows * G::OW;
      constexpr int CPP = (CD + G::NCH - 1) / G::NCH;
      constexpr int CMAX = (CPP + G::NS - 1) / G::NS;         // channels of one storer in one slice (at most)
      const int cend = min(CD, (part + 1) * CPP);
#pragma unroll
      for (int i0 = 0; i0 < CMAX; i0 += 2) {
        auto chan = [&](int i) {
          const int c = part * CPP + sw + G::NS * i;
          C3Drain d;
          d.ok = i < CMAX && c < cend;
          const int cc = d.ok ? c : 0;
          d.s = stage + cc * G::MROWP;
          d.o = p.out + b * p.out_bs + (long)cc * G::OH * G::OW + (long)band * R * G::OW;
          d.bb = bitb + cc * G::BITC;
          d.go = signs ? p.sg_out + b * p.sg_bs + ((long)cc * G::OH + band * R) * G::RW : nullptr;
          return d;
        };
        const C3Drain d0 = chan(i0), d1 = chan(i0 + 1);
        if (d0.ok) c3_drain_pair<G::MROW, G::VEC>(d0, d1, npx, lane);
        if (signs) {
          if (d0.ok) c3_drain_signs(d0, nrows * G::RW, lane);
          if (d1.ok) c3_drain_signs(d1, nrows * G::RW, lane);
        }
      }
    };
    long pend = -1, pend_n = 0;
    for (long k = 0; k < nwork; ++k) {
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      if (sw == 0) C3S_ST(1, k, 0);
      if (pend >= 0) drain(pend, ch, pend_n);     // the previous band leaves under this band's chunks, a slice per chunk
      if (sw == 0) C3S_ST(1, k, 1);
      if (!DS && ch == G::NCH - 1) c3_bar();      // X
      if (sw == 0) C3S_ST(1, k, 2);
      c3_bar();
      if (sw == 0) C3S_ST(1, k, 3);
      if (ch == G::NCH - 1) {
        pend_n = (long)((unsigned)k / (unsigned)G::NCH);
        pend = blockIdx.x + pend_n * gridDim.x;
      }
    }
    if (pend >= 0)
      for (int part = 0; part < G::NCH; ++part) drain(pend, part, pend_n);
    return;
  }
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - G::NW;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    int roff[G::NQ], rrow[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      const int r = pi / (G::WP / G::PB), i = pi - r * (G::WP / G::PB);
      rrow[q] = (pi < G::PP) ? (i == 0 ? -100000 : r) : -200000;
      roff[q] = r * W + G::PB * (i - 1);
    }
    constexpr bool U8OK = CS == 4 && W % 4 == 0 && !BWD;
    const bool u8 = U8OK && p.u8.f != nullptr;
    if constexpr (U8OK) {
      if (u8) {
        c3_u8_zero_pads<G::SR, G::WP, G::PLANE>(lds, lane, lw, G::NL);
        c3_u8_zero_pads<G::SR, G::WP, G::PLANE>(lds + G::BUF, lane, lw, G::NL);
      }
    }
    auto dma = [&](long k) {
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int y0 = band * R * S - 1;
      float* __restrict__ buf = lds + (k & 1) * G::BUF;
      const float* __restrict__ sb = p.src + b * p.src_bs + ((long)ch * G::KC * H + y0) * W;
      const bool staged = U8OK && u8;              // (the uint8 window is expanded below, behind the fragment DMA)
#pragma unroll
      for (int c = 0; c < G::KC; ++c) {
        if (staged || c % G::NL != lw) continue;
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
      if constexpr (U8OK) {
        if (u8) {      // the window's bytes and the env's valid-plane count in ONE round trip (unconditional loads, selected on expand)
          C3U8Regs<H, W, G::SR, G::NL> rg;
          c3_u8_load<H, W, G::SR, G::NL>(rg, p.u8, b, y0, lw, lane);
          c3_u8_expand<H, W, G::SR, G::WP, G::PLANE, G::NL>(buf, rg, lw, lane);
        }
      }
    };
    // backward-data: the band's mask as sign words, R * RW words per channel (contiguous in HBM)
    auto dma_signs = [&](long tile) {
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int nw = min(R, G::OH - band * R) * G::RW;
#pragma unroll 1
      for (int c = lw; c < CD; c += G::NL) {
        const unsigned* __restrict__ src = p.sg_in + b * p.sg_bs + ((long)c * G::OH + band * R) * G::RW;
        for (int i0 = 0; i0 < nw; i0 += 64)
          if (i0 + lane < nw)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + i0 + lane), (lptr_t)(bitb + c * G::BITC + i0), 4, 0, 0);
      }
    };
    const bool msk = BWD && p.sg_in != nullptr;
    if (nwork > 0) {
      dma(0);
      if (msk) dma_signs(blockIdx.x);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);            // vmcnt(0): the chunk has landed in LDS
    c3_bar();
    for (long k = 0; k < nwork; ++k) {
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      if (lw == 0) C3S_ST(2, k, 0);
      if (k + 1 < nwork) dma(k + 1);
      if (lw == 0) C3S_ST(2, k, 1);
      if (msk && ch == 0 && k > 0) dma_signs(blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x);     // (the previous band is done with them)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      if (lw == 0) C3S_ST(2, k, 2);
      if (!DS && ch == G::NCH - 1) c3_bar();       // X
      c3_bar();
      if (lw == 0) C3S_ST(2, k, 3);
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
  const bool msk = BWD && p.sg_in != nullptr;
  const bool sgn = !BWD && p.sg_out != nullptr;
  c3_bar();
  for (long k = 0; k < nwork; ++k) {
    const float* __restrict__ img = lds + (k & 1) * G::BUF;
    const float* __restrict__ fr = img + G::IMG + lane;
    if (w == 0) C3S_ST(0, k, 0);
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
        if (G::TPW * G::MT >= 12) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if ((int)((unsigned)k % (unsigned)G::NCH) == G::NCH - 1) {
      if (w == 0) C3S_ST(0, k, 1);
      if (!DS) c3_bar();                              // X: the image is free, this band's sign words are in
      if (w == 0) C3S_ST(0, k, 2);
      float* const stage = stage_of((long)((unsigned)k / (unsigned)G::NCH));
      unsigned* const bitb = bitb_of((long)((unsigned)k / (unsigned)G::NCH));
      // ---- accumulators -> the image.  D tile [pixel][channel]: lane (j, g) holds pixels 4g .. 4g+3 of channel j
#pragma unroll
      for (int u = 0; u < G::TPW; ++u) {
        const int t = w + G::NW * u;
        const int p0 = t * 16 + 4 * g;
#pragma unroll
        for (int m = 0; m < G::MT; ++m) {
          const int cd = m * 16 + j;
          float4 v = make_float4(acc[u][m][0], acc[u][m][1], acc[u][m][2], acc[u][m][3]);
          if (BWD) {
            if (msk && cd < CD && p0 < G::NPIX) {
              const int r = p0 / G::OW, x = p0 - r * G::OW;
              const unsigned nib = bitb[cd * G::BITC + r * G::RW + (x >> 5)] >> (x & 31);
              if (!(nib & 1u)) v.x = 0.f;
              if (!(nib & 2u)) v.y = 0.f;
              if (!(nib & 4u)) v.z = 0.f;
              if (!(nib & 8u)) v.w = 0.f;
            }
          } else {
            const float bs = biasv[m];
            v.x += bs; v.y += bs; v.z += bs; v.w += bs;
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (sgn && cd < CD) c3_sign4<G::OW, G::RW>(bitb + cd * G::BITC, p0, G::NPIX, v);
          }
          if (cd < CD && p0 + 3 < G::MROWP) *reinterpret_cast<float4*>(stage + cd * G::MROWP + p0) = v;
          acc[u][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
    }
    if (w == 0) C3S_ST(0, k, 3);
    c3_bar();
  }
#undef C3S_ST
}

// ---- stride-2 backward-data, staged.  The chunk images live in a ring of D buffers and the loaders run D - 1 chunks
// ahead (a chunk's MFMAs take ~1 us, an LDS-DMA issued under load comes back after 2-3: one chunk of lookahead leaves
// every interval waiting on its barrier); they wait with COUNTED vmcnt -- every chunk is the same number of DMA
// instructions per loader wave -- so the younger chunks stay in flight across the barrier.  FRES: the weight fragments
// of ALL chunks stay resident in LDS (loaded once per workgroup) instead of travelling with every chunk.
// ODD: the layer's input is (2 HO - 1) x (2 WO - 1) (a 3x3 / stride 2 / pad 1 layer on an odd image: 21 -> 11, GRUModel conv4):
// class pixels (q, p) with 2 q + py = H or 2 p + px = W do not exist.  The whole image is ONE band, the stage image is
// dense [ci][H * W] and leaves as ONE flat float4 run (the sample's dX is contiguous and 16-byte aligned as a whole: its
// rows are not), image writes are scalar.
// MS: the computing waves also split the destination-channel tiles (MS groups of MT / MS tiles each) -- a 6 x 6 class image
// (GRUModel conv5: 11 x 11 <- 6 x 6) is three pixel tiles, which would leave five of eight waves idle; with MS = 3 nine waves
// each own one (pixel tile, channel tile) pair.
template <int CO, int CI, int HO, int WO, int RQ, int KC, int D, bool FRES, bool ODD = false, int MS = 1, bool W1 = false>
struct C3BSGeo {
  static constexpr int H = 2 * HO - (ODD ? 1 : 0), W = 2 * WO - (ODD ? 1 : 0);
  static constexpr int NCH = CO / KC, C4 = KC / 4, MT = (CI + 15) / 16;
  static constexpr int NBAND = (HO + RQ - 1) / RQ;
  static constexpr int NPIX = RQ * WO;                 // class pixels per band
  static constexpr int NT = (NPIX + 15) / 16;
  static constexpr int NW = MS == 3 ? 9 : 8, NL = 2, NS = W1 ? 4 : 2;       // (w1 waves: four -- two took 11 us per band against the band's 6.6)
  static constexpr int NTW = NW / MS, MTW = MT / MS;   // waves over the pixel tiles; channel tiles per wave
  static constexpr int NTHR = 64 * (NW + NL + NS);
  static_assert(NW % MS == 0 && MT % MS == 0, "channel-tile split");
  static constexpr int TP = (NT + NTW - 1) / NTW;      // tile positions per wave
  // dOut band image.  PB16: planes and band starts are 16-byte aligned in HBM -> the band's rows of a plane come in as ONE
  // linear run of 16-byte pieces (row pitch WO, no zero column: the right halo is a lane select in the MFMA loop, the
  // rows below the plane are pieces read from the zero page) -- 2 DMA instructions per plane instead of 5: the loaders
  // of the 4-byte version spent most of their time ISSUING.  Otherwise 4-byte pieces and a zero column per row.
  static constexpr bool PB16 = WO % 2 == 0 && (HO * WO) % 4 == 0 && (RQ * WO) % 4 == 0;
  static constexpr int WP = PB16 ? WO : WO + 1;
  static constexpr int PL0 = PB16 ? (((RQ + 1) * WO + 3) / 4) * 4 : (RQ + 1) * WP;
  static constexpr int PLANE = ((PL0 + 15) / 32) * 32 + 16;
  static constexpr int NQ = PB16 ? (PL0 / 4 + 63) / 64 : (PL0 + 63) / 64;      // DMA instructions per plane
  static constexpr int IMG = ((KC * PLANE + 3) / 4) * 4 + 4;                   // (+ the float a last-row right-halo read may touch)
  static constexpr int FRAGC = C4 * 9 * MT * 64;
  static constexpr int NFQ = (FRAGC / 4 + 63) / 64;
  static constexpr int BUF = ((IMG + FRAGC + 255) / 256) * 256;
  static constexpr int MROW = ODD ? H * W : 2 * RQ * W; // floats of one channel's dX band (rows are contiguous in HBM)
  static constexpr int LOOK = D - 1;                               // chunks in flight ahead of the one being computed
  static constexpr int SLOT = FRES ? ((IMG + 63) / 64) * 64 : BUF;              // floats per ring slot
  static constexpr int FRAG_ALL = FRES ? NCH * FRAGC : 0;
  static constexpr int MROWP = ODD ? MROW : ((MROW + 3) / 8) * 8 + 4;
  static constexpr int STAGE = ((CI * MROWP + 3) / 4) * 4;
  static constexpr int RW = (W + 31) / 32;
  static constexpr int BITC = (ODD ? H : 2 * RQ) * RW;
  static_assert(!ODD || ((HO + RQ - 1) / RQ == 1 && (CI * H * W) % 4 == 0), "odd images: one band per sample, a flat 16-byte drain");
  static constexpr int BITB = ((CI * BITC + 3) / 4) * 4;
  // W1 (round 6): the finished dX band is NOT drained to HBM -- dX is the gradient of the FIRST layer's output, and nothing
  // but that layer's weight gradient reads it (models.py:570-622: the input needs no gradient) -- the two storer waves become
  // "w1 waves" that multiply the band (fp32 in `stage`, split into three exact bf16 pieces in registers) against the band's
  // rows of the uint8 frames (exact in bf16) on the bf16 matrix pipe: dW1[co][ci][ky][kx] += dX[co][y][x] * F[ci][y+ky-1][x+kx-1].
  // Frame band: rows 2 q0 - 1 .. 2 q0 + 2 RQ of the sample's 4 planes as raw bytes, [plane][row][W / 4 dwords], in a ring of
  // THREE buffers (band it is multiplied during band it + 1's chunks while the loaders, one band ahead, fill band it + 2's).
  static constexpr int FR = 2 * RQ + 2, FDW = 4 * FR * (W / 4);                 // rows per plane, data dwords per buffer
  static constexpr int NIF = W1 ? 4 : 0;                                        // frame DMA instructions per loader and chunk
  static constexpr int FBUF = W1 ? 8 + NCH * NL * NIF * 64 + 8 : 0;             // floats: pad | data (+ unused tail of the last DMA) | slack
  static constexpr int NFB = 3;
  static_assert(!W1 || (NCH * NL * NIF * 64 >= FDW && W % 4 == 0 && !ODD && MS == 1 && CI == 16 && MT == 1), "w1 waves: shape");
  static constexpr size_t LDS_BYTES_S = ((size_t)D * SLOT + FRAG_ALL + STAGE + BITB + NFB * FBUF) * 4;
  static constexpr bool VEC = (H * W) % 4 == 0 && MROW % 4 == 0;
  static constexpr long W1_PER = 16 * 36 + 16;                                   // floats per w1 slab: dW1 | db1
  // DMA instructions of one chunk per loader wave (constant by construction: planes and fragment pieces are dealt
  // round robin, a partly filled instruction still issues)
  static constexpr int NI0 = (KC / NL) * NQ + (FRES ? 0 : (NFQ + 1) / 2) + NIF;     // loader 0 / loader 1
  static constexpr int NI1 = (KC / NL) * NQ + (FRES ? 0 : NFQ / 2) + NIF;
  static_assert(CO % KC == 0 && KC % NL == 0 && NL == 2, "every chunk is the same number of DMA instructions per loader");
  static_assert(LOOK >= 1 && LOOK <= NCH && (LOOK - 1) * NI0 <= 63, "lookahead: sign words land before X; vmcnt is 6 bits");
};


template <int CO, int CI, int HO, int WO, int RQ, int KC, int D, bool FRES, bool ODD = false, int MS = 1, bool W1 = false>
__global__ __launch_bounds__((C3BSGeo<CO, CI, HO, WO, RQ, KC, D, FRES, ODD, MS, W1>::NTHR)) void c3bs_kernel(C3P p) {
  using G = C3BSGeo<CO, CI, HO, WO, RQ, KC, D, FRES, ODD, MS, W1>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const fres = lds + D * G::SLOT;                     // resident fragments (FRES)
  float* const stage = fres + G::FRAG_ALL;
  unsigned* const bitb = reinterpret_cast<unsigned*>(stage + G::STAGE);
  unsigned* const fbuf0 = bitb + G::BITB;                    // W1: three frame-band buffers of FBUF dwords (data at + 8)
  if constexpr (W1) {                                        // pads / slack read as zero (column -1 of the first row, past the last)
    for (int i = threadIdx.x; i < G::NFB * G::FBUF; i += G::NTHR) fbuf0[i] = 0u;
    __syncthreads();
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;
  if (w >= G::NW + G::NL) {
    // ------------------------------------------------------------------ storer waves
    const int sw = w - G::NW - G::NL;
    c3_bar();
    auto drain = [&](long tile, int part) {
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      if constexpr (ODD) {
        // the whole sample's dX = CI * H * W floats, one contiguous 16-byte aligned run: part `part` of NCH, this storer's half
        constexpr int TOT4 = CI * G::H * G::W / 4, PER = (TOT4 + G::NCH * G::NS - 1) / (G::NCH * G::NS);
        const int i0 = (part * G::NS + sw) * PER, i1 = min(TOT4, i0 + PER);
        float* __restrict__ o = p.out + b * p.out_bs;
        for (int i = i0 + lane; i < i1; i += 256) {              // four pieces per lane and trip: all LDS reads first, then the stores
          f32x4 v0 = c3_lds_read128(stage + 4 * min(i, TOT4 - 1)), v1 = c3_lds_read128(stage + 4 * min(i + 64, TOT4 - 1));
          f32x4 v2 = c3_lds_read128(stage + 4 * min(i + 128, TOT4 - 1)), v3 = c3_lds_read128(stage + 4 * min(i + 192, TOT4 - 1));
          c3_lds_wait(v0, v1, v2, v3);
          if (i < i1) *reinterpret_cast<f32x4*>(o + 4 * i) = v0;
          if (i + 64 < i1) *reinterpret_cast<f32x4*>(o + 4 * (i + 64)) = v1;
          if (i + 128 < i1) *reinterpret_cast<f32x4*>(o + 4 * (i + 128)) = v2;
          if (i + 192 < i1) *reinterpret_cast<f32x4*>(o + 4 * (i + 192)) = v3;
        }
      } else {
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int nrows = 2 * min(RQ, HO - q0), npx = nrows * G::W;
      constexpr int CPP = (CI + G::NCH - 1) / G::NCH;
      constexpr int CMAX = (CPP + G::NS - 1) / G::NS;
      const int cend = min(CI, (part + 1) * CPP);
      constexpr int NH = (G::MT > 1 && G::TP > 1) ? 1 : 2;      // (the computing waves of those instances sit at the register cap)
#pragma unroll
      for (int i0 = 0; i0 < CMAX; i0 += NH) {
        auto chan = [&](int i) {
          const int c = part * CPP + sw + G::NS * i;
          C3Drain d;
          d.ok = i < CMAX && c < cend;
          const int cc = d.ok ? c : 0;
          d.s = stage + cc * G::MROWP;
          d.o = p.out + b * p.out_bs + ((long)cc * G::H + 2 * q0) * G::W;
          d.bb = nullptr;
          d.go = nullptr;
          return d;
        };
        const C3Drain d0 = chan(i0), d1 = chan(i0 + 1);
        if (d0.ok) c3_drain_pair<G::MROW, G::VEC, NH>(d0, d1, npx, lane);
      }
      }
    };
    // ---- W1: this wave's share of the first layer's weight gradient (see C3BSGeo).  MFMA 16x16x32 bf16: A = dX pieces, lane
    // (g, j) = channel co = j, pixels x0 + 8 g .. + 7 of a band row; B = frame windows, lane (g, j) = window row (ci, ky) = (j / 3,
    // j % 3) for j < 12, the same 8 pixels shifted by kx - 1 -- ONE 10-byte read serves kx = 0, 1, 2 (three accumulator tiles);
    // D[co = 4 g + i][j] per kx.  Band row r of part `part`: 4 part + 2 sw + {0, 1}.
    typedef __bf16 bf16x8s __attribute__((ext_vector_type(8)));
    typedef unsigned int u32x4s __attribute__((ext_vector_type(4)));
    if constexpr (W1) {
      if (p.prio) __builtin_amdgcn_s_setprio(2);              // (the band waits for them at X: 7.01 -> 6.87 ms at N = 32,768)
    }
    f32x4 wacc0 = {0.f, 0.f, 0.f, 0.f}, wacc1 = wacc0, wacc2 = wacc0;
    float wdb = 0.f;
    const int g = lane >> 4, j = lane & 15;
    const int wci = j < 12 ? j / 3 : 0, wky = j < 12 ? j % 3 : 0;
    auto w1 = [&](long tile, int it, int part) {
     if constexpr (W1) {
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int nrows = 2 * min(RQ, HO - q0);
      const int nvl = p.u8.nv ? p.u8.nv[b * p.u8.nv_s] : 4;
      const bool blive = j < 12 && wci >= 4 - nvl;                                   // planes older than the last reset read as zero
      const unsigned char* __restrict__ fb = reinterpret_cast<const unsigned char*>(fbuf0 + (it % G::NFB) * G::FBUF + 8);
      constexpr int RPP = (2 * RQ) / (G::NCH * G::NS);                               // rows per part and wave
      static_assert(RPP * G::NCH * G::NS == 2 * RQ, "band rows dealt evenly to parts and w1 waves");
#pragma unroll 1
      for (int rr = 0; rr < RPP; ++rr) {
        const int r = (part * G::NS + sw) * RPP + rr;
        if (r >= nrows) continue;
        const float* __restrict__ arow = stage + j * G::MROWP + r * G::W;
        const int brow = ((wci * G::FR + r + wky) * G::W);                           // byte offset of column 0 of this lane's frame row
#pragma unroll
        for (int x0 = 0; x0 < G::W; x0 += 32) {                                      // (unrolled: the three steps' LDS reads go out together)
          const int px0 = x0 + 8 * g;                                                // this lane's first pixel
          // A: eight dX values of channel j, three exact bf16 pieces each
          float e[8];
          if (px0 + 8 <= G::W) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(arow + px0), v1 = *reinterpret_cast<const f32x4*>(arow + px0 + 4);
            e[0] = v0[0]; e[1] = v0[1]; e[2] = v0[2]; e[3] = v0[3]; e[4] = v1[0]; e[5] = v1[1]; e[6] = v1[2]; e[7] = v1[3];
          } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) e[i] = px0 + i < G::W ? arow[min(px0 + i, G::W - 1)] : 0.f;
          }
          unsigned int ah[4], am[4], al[4];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            unsigned short pc[3][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const float ev = e[2 * i + h];
              const __bf16 h0 = (__bf16)ev;
              const float r1 = ev - (float)h0;
              const __bf16 h1 = (__bf16)r1;
              const float r2 = r1 - (float)h1;
              pc[0][h] = __builtin_bit_cast(unsigned short, h0);
              pc[1][h] = __builtin_bit_cast(unsigned short, h1);
              pc[2][h] = __builtin_bit_cast(unsigned short, (__bf16)r2);
              wdb += ev;
            }
            ah[i] = (unsigned int)pc[0][0] | ((unsigned int)pc[0][1] << 16);
            am[i] = (unsigned int)pc[1][0] | ((unsigned int)pc[1][1] << 16);
            al[i] = (unsigned int)pc[2][0] | ((unsigned int)pc[2][1] << 16);
          }
          // B: frame bytes of columns px0 - 1 .. px0 + 8 (ten), zero outside the row / the plane
          const int a = brow + px0 - 1;                                             // (>= -1: the pad dword in front of the data)
          const unsigned int* __restrict__ q = reinterpret_cast<const unsigned int*>(fb + (a & ~3));
          const unsigned int d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3];
          const unsigned int sh = (unsigned int)(a & 3);
          unsigned int s0 = __builtin_amdgcn_alignbyte(d1, d0, sh), s1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                       s2 = __builtin_amdgcn_alignbyte(d3, d2, sh);
          if (px0 == 0) s0 &= 0xffffff00u;                                           // column -1
          const int nvh = G::W - (px0 - 1);                                          // valid bytes from the first one
          if (nvh < 10) {                                                            // (row tail: the bytes behind it are the next row's)
            const unsigned int m0 = nvh >= 4 ? 0xffffffffu : nvh <= 0 ? 0u : (1u << (8 * nvh)) - 1u;
            const unsigned int m1 = nvh >= 8 ? 0xffffffffu : nvh <= 4 ? 0u : (1u << (8 * (nvh - 4))) - 1u;
            const unsigned int m2 = nvh >= 12 ? 0xffffffffu : nvh <= 8 ? 0u : (1u << (8 * (nvh - 8))) - 1u;
            s0 &= m0; s1 &= m1; s2 &= m2;
          }
          if (!blive) { s0 = 0u; s1 = 0u; s2 = 0u; }
          // bytes -> bf16 (the upper halves of the exact floats), pairs (e0 e1) (e2 e3) (e4 e5) (e6 e7) (e8 e9)
          const unsigned int P0 = __builtin_amdgcn_perm(__float_as_uint((float)((s0 >> 8) & 0xffu)), __float_as_uint((float)(s0 & 0xffu)), 0x07060302u);
          const unsigned int P1 = __builtin_amdgcn_perm(__float_as_uint((float)(s0 >> 24)), __float_as_uint((float)((s0 >> 16) & 0xffu)), 0x07060302u);
          const unsigned int P2 = __builtin_amdgcn_perm(__float_as_uint((float)((s1 >> 8) & 0xffu)), __float_as_uint((float)(s1 & 0xffu)), 0x07060302u);
          const unsigned int P3 = __builtin_amdgcn_perm(__float_as_uint((float)(s1 >> 24)), __float_as_uint((float)((s1 >> 16) & 0xffu)), 0x07060302u);
          const unsigned int P4 = __builtin_amdgcn_perm(__float_as_uint((float)((s2 >> 8) & 0xffu)), __float_as_uint((float)(s2 & 0xffu)), 0x07060302u);
          const u32x4s b0 = {P0, P1, P2, P3};
          const u32x4s b1 = {__builtin_amdgcn_alignbit(P1, P0, 16u), __builtin_amdgcn_alignbit(P2, P1, 16u),
                             __builtin_amdgcn_alignbit(P3, P2, 16u), __builtin_amdgcn_alignbit(P4, P3, 16u)};
          const u32x4s b2 = {P1, P2, P3, P4};
          const u32x4s ahv = {ah[0], ah[1], ah[2], ah[3]}, amv = {am[0], am[1], am[2], am[3]}, alv = {al[0], al[1], al[2], al[3]};
          const bf16x8s A0 = __builtin_bit_cast(bf16x8s, ahv), A1 = __builtin_bit_cast(bf16x8s, amv), A2 = __builtin_bit_cast(bf16x8s, alv);
          const bf16x8s B0 = __builtin_bit_cast(bf16x8s, b0), B1 = __builtin_bit_cast(bf16x8s, b1), B2 = __builtin_bit_cast(bf16x8s, b2);
          wacc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B0, wacc0, 0, 0, 0);
          wacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B1, wacc1, 0, 0, 0);
          wacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A0, B2, wacc2, 0, 0, 0);
          wacc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B0, wacc0, 0, 0, 0);
          wacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B1, wacc1, 0, 0, 0);
          wacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A1, B2, wacc2, 0, 0, 0);
          wacc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B0, wacc0, 0, 0, 0);
          wacc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B1, wacc1, 0, 0, 0);
          wacc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A2, B2, wacc2, 0, 0, 0);
        }
      }
     }
    };
    long pend = -1;
    int pend_it = 0;
    const bool stamp = C3_STAMPS && p.dbg != nullptr && blockIdx.x == 0 && sw == 0 && lane == 0;
    unsigned long long tdr = 0;
    for (long k = 0; k < nwork; ++k) {
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const unsigned long long t0 = stamp ? wall_clock64() : 0;
      if (pend >= 0) {
        if constexpr (W1) w1(pend, pend_it, ch);
        else drain(pend, ch);
      }
      if (stamp) tdr += wall_clock64() - t0;
      if (ch == G::NCH - 1) c3_bar();      // X
      c3_bar();
      if (ch == G::NCH - 1) {
        pend_it = (int)((unsigned)k / (unsigned)G::NCH);
        pend = blockIdx.x + (long)pend_it * gridDim.x;
      }
    }
    if (pend >= 0)
      for (int part = 0; part < G::NCH; ++part) {
        if constexpr (W1) w1(pend, pend_it, part);
        else drain(pend, part);
      }
    if constexpr (W1) {
      // this wave's partial sums -> its slab: dW1 as [co][ci * 9 + ky * 3 + kx], then db1 (a fixed-order reduction adds the slabs)
      float* __restrict__ sl = p.w1_slab + ((long)blockIdx.x * G::NS + sw) * G::W1_PER;
      if (j < 12) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float* __restrict__ o = sl + (4 * g + i) * 36 + wci * 9 + wky * 3;
          o[0] = wacc0[i]; o[1] = wacc1[i]; o[2] = wacc2[i];
        }
      }
      float v = wdb;
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (g == 0) sl[16 * 36 + j] = v;
    }
    if (stamp) p.dbg[4] = tdr;
    return;
  }
  if (w >= G::NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - G::NW;
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    int roff[G::NQ], rrow[G::NQ];
#pragma unroll
    for (int q = 0; q < G::NQ; ++q) {
      const int pi = q * 64 + lane;
      if (G::PB16) {                                     // piece pi = floats 4 pi .. 4 pi + 3 of the plane's band rows
        rrow[q] = (4 * pi < G::PL0) ? 0 : -200000;
        roff[q] = 4 * pi;
      } else {
        const int r = pi / G::WP, x = pi - r * G::WP;
        rrow[q] = (pi < G::PL0) ? (x == WO ? -100000 : r) : -200000;
        roff[q] = r * WO + x;
      }
    }
    // W1: the band's frame rows travel as dwords, 64 consecutive dwords of the [plane][row][W / 4] buffer per instruction;
    // instruction (ch, lw, u) of the band covers dwords ((ch * NL + lw) * NIF + u) * 64 + lane
    int frow[G::NCH * (W1 ? G::NIF : 0) + 1], foff[G::NCH * (W1 ? G::NIF : 0) + 1];
    if constexpr (W1) {
#pragma unroll
      for (int i = 0; i < G::NCH * G::NIF; ++i) {
        const int ch_ = i / G::NIF, u_ = i % G::NIF;
        const int dw = ((ch_ * G::NL + lw) * G::NIF + u_) * 64 + lane;
        const int rowi = dw / (G::W / 4), x4 = dw - rowi * (G::W / 4);
        const int ci = rowi / G::FR, fr = rowi - ci * G::FR;
        frow[i] = dw < G::FDW ? fr : -100000;                      // frame row 2 q0 - 1 + fr of plane ci
        foff[i] = (ci * G::H + fr - 1) * G::W + 4 * x4;            // (+ 2 q0 * W: byte offset in the sample's 4 planes)
      }
    }
    auto dma = [&](long k) {
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      float* __restrict__ buf = lds + ((unsigned)k % (unsigned)D) * G::SLOT;
      const float* __restrict__ sb = p.src + b * p.src_bs + ((long)ch * KC * HO + q0) * WO;
      if constexpr (W1) {
        const unsigned bu_ = (unsigned)b, slot_ = bu_ / (unsigned)p.u8.T;
        const unsigned char* __restrict__ fbs = p.u8.f + (long)slot_ * p.u8.bs + (long)(bu_ - slot_ * (unsigned)p.u8.T) * (G::H * G::W);
        unsigned* __restrict__ fd = fbuf0 + (((unsigned)k / (unsigned)G::NCH) % (unsigned)G::NFB) * G::FBUF + 8;
#pragma unroll
        for (int u = 0; u < G::NIF; ++u) {
          // (the table index must be a compile-time constant per unrolled trip: select among the NCH chunks' entries)
          int fr_ = 0, fo_ = 0;
#pragma unroll
          for (int c_ = 0; c_ < G::NCH; ++c_)
            if (c_ == ch) { fr_ = frow[c_ * G::NIF + u]; fo_ = foff[c_ * G::NIF + u]; }
          const int y = 2 * q0 - 1 + fr_;
          const void* gsrc = (fr_ >= 0 && y >= 0 && y < G::H) ? (const void*)(fbs + fo_ + 2 * q0 * G::W) : (const void*)p.zero;
          __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(fd + ((ch * G::NL + lw) * G::NIF + u) * 64), 4, 0, 0);
        }
      }
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (c % G::NL != lw) continue;
#pragma unroll
        for (int q = 0; q < G::NQ; ++q) {
          if (rrow[q] > -200000) {
            if (G::PB16) {                                // (the plane ends on a piece boundary: HO * WO % 4 == 0)
              const float* gsrc = (q0 * WO + roff[q] < HO * WO) ? sb + (long)c * HO * WO + roff[q] : p.zero;
              __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 256), 16, 0, 0);
            } else {
              const float* gsrc = (rrow[q] >= 0 && q0 + rrow[q] < HO) ? sb + (long)c * HO * WO + roff[q] : p.zero;
              __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(buf + c * G::PLANE + q * 64), 4, 0, 0);
            }
          }
        }
      }
      if (!FRES) {
        const float* __restrict__ fg = p.frag + (long)ch * G::FRAGC;
#pragma unroll
        for (int q = 0; q < G::NFQ; ++q) {
          const int pi = q * 64 + lane;
          if (q % G::NL == lw && pi < G::FRAGC / 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(fg + pi * 4), (lptr_t)(buf + G::IMG + q * 256), 16, 0, 0);
        }
      }
    };
    auto dma_signs = [&](long tile) {
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int nw = ODD ? G::H * G::RW : 2 * min(RQ, HO - q0) * G::RW;
#pragma unroll 1
      for (int c = lw; c < CI; c += G::NL) {
        const unsigned* __restrict__ src = p.sg_in + b * p.sg_bs + ((long)c * G::H + 2 * q0) * G::RW;
        for (int i0 = 0; i0 < nw; i0 += 64)
          if (i0 + lane < nw)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + i0 + lane), (lptr_t)(bitb + c * G::BITC + i0), 4, 0, 0);
      }
    };
    // before the barrier that opens chunk k1: everything up to chunk k1 has landed, the m younger chunks stay in flight
    auto wait_for = [&](long k1) {
      long m = min(k1 + G::LOOK - 1, nwork - 1) - k1;
      if (m < 0) m = 0;
      if (lw == 0) {
        if (G::LOOK >= 4 && m >= 3) c3_wait_vm<(G::LOOK >= 4 ? 3 : 0) * G::NI0>();
        else if (G::LOOK >= 3 && m == 2) c3_wait_vm<(G::LOOK >= 3 ? 2 : 0) * G::NI0>();
        else if (G::LOOK >= 2 && m == 1) c3_wait_vm<(G::LOOK >= 2 ? 1 : 0) * G::NI0>();
        else c3_wait_vm<0>();
      } else {
        if (G::LOOK >= 4 && m >= 3) c3_wait_vm<(G::LOOK >= 4 ? 3 : 0) * G::NI1>();
        else if (G::LOOK >= 3 && m == 2) c3_wait_vm<(G::LOOK >= 3 ? 2 : 0) * G::NI1>();
        else if (G::LOOK >= 2 && m == 1) c3_wait_vm<(G::LOOK >= 2 ? 1 : 0) * G::NI1>();
        else c3_wait_vm<0>();
      }
    };
    static_assert(G::LOOK <= 4, "wait_for");
    const bool msk = p.sg_in != nullptr;
    if (nwork > 0) {
      if (FRES) {                                            // all chunks' fragments, once
        constexpr int NFA = (G::FRAG_ALL / 4 + 63) / 64;
#pragma unroll 1
        for (int q = lw; q < NFA; q += G::NL) {
          const int pi = q * 64 + lane;
          if (pi < G::FRAG_ALL / 4)
            __builtin_amdgcn_global_load_lds((gptr_t)(p.frag + pi * 4), (lptr_t)(fres + q * 256), 16, 0, 0);
        }
      }
      if (msk) dma_signs(blockIdx.x);
      for (int k = 0; k < G::LOOK; ++k)
        if (k < nwork) dma(k);
    }
    wait_for(0);
    c3_bar();
    const bool stamp = C3_STAMPS && p.dbg != nullptr && blockIdx.x == 0 && lw == 0 && lane == 0;
    unsigned long long tld = 0;
    for (long k = 0; k < nwork; ++k) {
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const unsigned long long t0 = stamp ? wall_clock64() : 0;
      if (msk && ch == 0 && k > 0) dma_signs(blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x);     // (the previous band is done with them)
      if (k + G::LOOK < nwork) dma(k + G::LOOK);            // into the slot chunk k - 1 left
      wait_for(k + 1);
      if (stamp) tld += wall_clock64() - t0;
      if (ch == G::NCH - 1) c3_bar();       // X
      c3_bar();
    }
    if (stamp) p.dbg[5] = tld;
    return;
  }
  // -------------------------------------------------------------------- computing waves
  const int wt = MS == 1 ? w : w % G::NTW, wm = MS == 1 ? 0 : w / G::NTW;      // this wave's pixel-tile lane / channel-tile group
  int base[G::TP];
  bool redge[G::TP];                             // PB16: the pixel is the last of its row (its right neighbour is the halo: 0)
#pragma unroll
  for (int u = 0; u < G::TP; ++u) {
    const int t = wt + G::NTW * u;
    const int pp = t * 16 + j;
    const int px = (t < G::NT && pp < G::NPIX) ? pp : 0;
    const int r = px / WO, x = px - r * WO;
    base[u] = g * G::PLANE + r * G::WP + x;
    redge[u] = G::PB16 && x == WO - 1;
  }
  f32x4 acc[G::TP][4][G::MTW];                   // [tile position][class py*2+px][channel tile of this wave]
#pragma unroll
  for (int u = 0; u < G::TP; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int m = 0; m < G::MTW; ++m) acc[u][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const bool msk = p.sg_in != nullptr;
  c3_bar();
  const bool stamp = C3_STAMPS && p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  unsigned long long ts = stamp ? wall_clock64() : 0, tsum[4] = {0, 0, 0, 0};      // MFMAs | wait at X | image write | wait at the chunk barrier
#define C3_TS(i) do { if (stamp) { const unsigned long long n_ = wall_clock64(); tsum[i] += n_ - ts; ts = n_; } } while (0)
  for (long k = 0; k < nwork; ++k) {
    const float* __restrict__ img = lds + ((unsigned)k % (unsigned)D) * G::SLOT;
    const float* __restrict__ fr = (FRES ? fres + ((unsigned)k % (unsigned)G::NCH) * G::FRAGC : img + G::IMG) + lane;
#pragma unroll
    for (int c4 = 0; c4 < G::C4; ++c4) {
      float wv[9][G::MTW];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int m = 0; m < G::MTW; ++m) wv[tap][m] = fr[((c4 * 9 + tap) * G::MT + wm * G::MTW + m) * 64];
#pragma unroll
      for (int u = 0; u < G::TP; ++u) {
        const float* __restrict__ s = img + base[u] + c4 * 4 * G::PLANE;
        const float s00 = s[0], s10 = s[G::WP];
        float s01 = s[1], s11 = s[G::WP + 1];
        if (G::PB16) {
          s01 = redge[u] ? 0.f : s01;
          s11 = redge[u] ? 0.f : s11;
        }
#pragma unroll
        for (int m = 0; m < G::MTW; ++m) {
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
        if (G::MTW > 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
    C3_TS(0);
    if ((int)((unsigned)k % (unsigned)G::NCH) == G::NCH - 1) {
      c3_bar();                                       // X
      C3_TS(1);
      const long tile = blockIdx.x + ((unsigned)k / (unsigned)G::NCH) * gridDim.x;
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int q0 = (int)(tile - b * G::NBAND) * RQ;
      const int npix_ok = min(RQ, HO - q0) * WO;
#pragma unroll
      for (int u = 0; u < G::TP; ++u) {
        const int t = wt + G::NTW * u;
        const int c0 = t * 16 + 4 * g;
#pragma unroll
        for (int m = 0; m < G::MTW; ++m) {
          const int ci = (wm * G::MTW + m) * 16 + j;
          float* __restrict__ sc = stage + ci * G::MROWP;
          const unsigned* __restrict__ bc = bitb + ci * G::BITC;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int c = c0 + 2 * h;
            const int qr = c / WO, pc = c - qr * WO;
            const bool okA = ci < CI && c < npix_ok, okB = ci < CI && c + 1 < npix_ok, split = pc + 1 >= WO;
#pragma unroll
            for (int py = 0; py < 2; ++py) {
              const int ra = 2 * qr + py, xa = 2 * pc;                                    // band row / column of the first pair
              const int rb = split ? ra + 2 : ra, xb = split ? 0 : xa + 2;
              const int la = ra * G::W + xa, lb = rb * G::W + xb;
              float4 v = make_float4(acc[u][py * 2][m][2 * h], acc[u][py * 2 + 1][m][2 * h], acc[u][py * 2][m][2 * h + 1],
                                     acc[u][py * 2 + 1][m][2 * h + 1]);
              if (msk) {
                if (okA && (!ODD || ra < G::H)) {
                  const unsigned q2 = bc[ra * G::RW + (xa >> 5)] >> (xa & 31);
                  if (!(q2 & 1u)) v.x = 0.f;
                  if (!(q2 & 2u)) v.y = 0.f;
                }
                if (okB && (!ODD || rb < G::H)) {
                  const unsigned q2 = bc[rb * G::RW + (xb >> 5)] >> (xb & 31);
                  if (!(q2 & 1u)) v.z = 0.f;
                  if (!(q2 & 2u)) v.w = 0.f;
                }
              }
              if constexpr (ODD) {
                // rows 2 qr + 1 = H and columns 2 pc + 1 = W do not exist; rows of W floats are not 8-byte aligned
                if (okA && ra < G::H) {
                  sc[la] = v.x;
                  if (xa + 1 < G::W) sc[la + 1] = v.y;
                }
                if (okB && rb < G::H) {
                  sc[lb] = v.z;
                  if (xb + 1 < G::W) sc[lb + 1] = v.w;
                }
              } else if (okA) {
                if (WO % 2 == 0 && okB) *reinterpret_cast<float4*>(sc + la) = v;
                else {
                  *reinterpret_cast<float2*>(sc + la) = make_float2(v.x, v.y);
                  if (okB) *reinterpret_cast<float2*>(sc + lb) = make_float2(v.z, v.w);
                }
              }
            }
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) acc[u][c][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      C3_TS(2);
    }
    c3_bar();
    C3_TS(3);
  }
  if (stamp) {
    p.dbg[0] = tsum[0]; p.dbg[1] = tsum[1]; p.dbg[2] = tsum[2]; p.dbg[3] = tsum[3]; p.dbg[6] = (unsigned long long)nwork;
  }
#undef C3_TS
}

template <int CO, int CI, int HO, int WO, int RQ, int KC, int D, bool FRES, bool ODD = false, int MS = 1>
int c3bs_launch(const C3P& p, hipStream_t st) {
  using G = C3BSGeo<CO, CI, HO, WO, RQ, KC, D, FRES, ODD, MS>;
  static_assert(G::LDS_BYTES_S <= 160 * 1024, "LDS");
  const void* k = (const void*)c3bs_kernel<CO, CI, HO, WO, RQ, KC, D, FRES, ODD, MS>;
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES_S) != hipSuccess) return A2C_ERR_LAUNCH;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p.B * G::NBAND;
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;      // (the kernels walk chunks with 32-bit arithmetic)
  const int grid = (int)(total < cus ? total : cus);
  hipLaunchKernelGGL((c3bs_kernel<CO, CI, HO, WO, RQ, KC, D, FRES, ODD, MS>), dim3(grid), dim3(G::NTHR), G::LDS_BYTES_S, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

__global__ void c3w_reduce_kernel(const float* __restrict__ slab, int nslab, long per, long nW, float* __restrict__ dW,
                                  float* __restrict__ db);
// c3bs_kernel<..., W1 = true>: backward-data of layer 2 with the FIRST layer's weight gradient taken from the band in LDS (the
// input gradient of layer 2 never reaches HBM).  ws = [grid * NS] slabs of 16 x 36 + 16 floats; a fixed-order reduction -> dW1, db1.
template <int CO, int CI, int HO, int WO, int RQ, int KC, int D, bool FRES>
size_t c3bs_w1_ws_bytes(int B) {
  using G = C3BSGeo<CO, CI, HO, WO, RQ, KC, D, FRES, false, 1, true>;
  int dev = 0, cus = 256;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
  const long total = (long)B * G::NBAND;
  const long grid = total < cus ? total : cus;
  return (size_t)(grid * G::NS * G::W1_PER) * sizeof(float);
}
template <int CO, int CI, int HO, int WO, int RQ, int KC, int D, bool FRES>
int c3bs_w1_launch(C3P p, float* dW1, float* db1, void* ws, size_t ws_bytes, hipStream_t st) {
  using G = C3BSGeo<CO, CI, HO, WO, RQ, KC, D, FRES, false, 1, true>;
  static_assert(G::LDS_BYTES_S <= 160 * 1024, "LDS");
  const void* k = (const void*)c3bs_kernel<CO, CI, HO, WO, RQ, KC, D, FRES, false, 1, true>;
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES_S) != hipSuccess) return A2C_ERR_LAUNCH;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p.B * G::NBAND;
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;
  const int grid = (int)(total < cus ? total : cus);
  if (ws_bytes < (size_t)grid * G::NS * G::W1_PER * sizeof(float)) return A2C_ERR_WORKSPACE;
  p.w1_slab = (float*)ws;
  hipLaunchKernelGGL((c3bs_kernel<CO, CI, HO, WO, RQ, KC, D, FRES, false, 1, true>), dim3(grid), dim3(G::NTHR), G::LDS_BYTES_S, st, p);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  hipLaunchKernelGGL(c3w_reduce_kernel, dim3(3), dim3(256), 0, st, (const float*)ws, grid * G::NS, G::W1_PER, (long)16 * 36, dW1, db1);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

template <int CS, int CD, int H, int W, int S, int R, bool BWD, bool DS = false>
int c3s_launch(const C3P& p, hipStream_t st) {
  using G = C3SGeo<CS, CD, H, W, S, R, DS>;
  static_assert(G::LDS_BYTES_S <= 160 * 1024, "LDS");
  const void* k = (const void*)c3s_kernel<CS, CD, H, W, S, R, BWD, DS>;
  static int cus = 0;
  if (!cus) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES_S) != hipSuccess) return A2C_ERR_LAUNCH;
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p.B * G::NBAND;
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;      // (the kernels walk chunks with 32-bit arithmetic)
  const int grid = (int)(total < cus ? total : cus);
  hipLaunchKernelGGL((c3s_kernel<CS, CD, H, W, S, R, BWD, DS>), dim3(grid), dim3(G::NTHR), G::LDS_BYTES_S, st, p);
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
template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG, int D = 2>
struct C3WGeo {
  // (rows that are not 16-byte multiples travel as dword pieces: ~270 DMA instructions per band and loader wave on the
  // 42-wide 32 -> 64 layer; four loader waves share them)
  static constexpr int NW = 8, NL = (W % 4 != 0) ? 4 : 2, NPG = NW / NCG, NTHR = 64 * (NW + NL);
  static constexpr int NCH = CS / KC;
  // D input-chunk images in a ring, the loaders LOOK = D - 1 chunks ahead of the computing waves (counted vmcnt, as in
  // c3bs_kernel): with the compute loop pipelined a band's MFMAs take 2-3 us, an LDS-DMA issued under load comes back after
  // 2-3 -- one chunk of lookahead (D = 2) left every band waiting on its barrier.  DD dOut band images cover the bands
  // the chunks in flight belong to.
  static constexpr int LOOK = D - 1, DD = (LOOK + NCH - 1) / NCH + 1;
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
  // dOut band image.  Rows that are 16-byte multiples: 16-byte pieces, row pitch OWP.  Odd-width rows whose BAND is a
  // 16-byte multiple at a 16-byte aligned address (42-wide rows, 6 per band): the band of a channel is DMA'd FLAT (pitch
  // OW, 16-byte pieces: 12 instructions per loader wave and band instead of 60 dword ones) and the computing waves zero
  // the lanes of a row's last step that would read the next row's first pixels.  Otherwise dword pieces.
  static constexpr bool DFLAT = OW % 4 != 0 && (R * OW) % 4 == 0 && (OH * OW) % 4 == 0 && OH % R == 0;
  static constexpr int OWD = DFLAT ? OW : OWP;                    // row pitch of the dOut image
  static constexpr int DPB = (OW % 4 == 0 || DFLAT) ? 4 : 1;
  static constexpr int DPL0 = R * OWD;
  static constexpr int DPLANE = (DPL0 - 4 + 31) / 32 * 32 + 4;    // 4 (mod 32) floats apart: channels spread over the banks
  static constexpr int DPP = DPL0 / DPB, DNQ = (DPP + 63) / 64;
  static constexpr int DB = CD * DPLANE;
  static constexpr int K = CS * 9;
  static constexpr long PER = (long)CD * K + CD;                  // floats per slab: dW | db
  static constexpr size_t LDS_BYTES = (D * (size_t)XB + DD * (size_t)DB) * 4;
  // LDS-DMA instructions per loader wave for a chunk's input planes / a band's dOut channels (vmcnt bookkeeping)
  static constexpr int NIX = (KC / NL) * NQ, NID = (CD / NL) * DNQ;
  static_assert(CS % KC == 0 && NW % NCG == 0 && PL0 % PB == 0 && DPLANE >= DPL0 && DPLANE % 4 == 0, "shape");
  static_assert(KC % NL == 0 && CD % NL == 0 && D >= 2 && D <= 3, "loader bookkeeping");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS");
};

struct C3WP {
  const float* x; long x_bs;          // layer input (B, CS, H, W)
  const float* dout;                  // (B, CD, OH, OW) dense
  float* slab;                        // [grid * NPG][CD*CS*9 + CD]
  const float* zero;
  int B;
  C3U8 u8;                            // first layer: uint8 frame-store source instead of x (u8.f != nullptr)
  int nodma;                          // timing experiment (A2C_C3W_NODMA=1: wrong sums): only the first band is loaded
  int prio;                           // loader waves at raised priority
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

template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG, int D>
__global__ __launch_bounds__((C3WGeo<CS, CD, H, W, S, R, KC, NCG, D>::NTHR)) void c3w_kernel(C3WP p) {
  using G = C3WGeo<CS, CD, H, W, S, R, KC, NCG, D>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long ntile = (long)p.B * G::NBAND;
  long nmine = 0;
  if ((long)blockIdx.x < ntile) nmine = (ntile - 1 - blockIdx.x) / gridDim.x + 1;
  const long nwork = nmine * G::NCH;
  float* __restrict__ xbuf = lds;                      // D x XB
  float* __restrict__ dbuf = lds + D * G::XB;          // DD x DB
  // whatever the DMAs never write (pad columns, slack behind the planes) must read as finite numbers: zeros
  for (int i = tid; i < D * G::XB + G::DD * G::DB; i += G::NTHR) lds[i] = 0.f;
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
    if (p.prio) __builtin_amdgcn_s_setprio(3);
    int doff[G::DNQ], drow[G::DNQ];                   // dOut pieces: band row and source offset of this lane's piece
#pragma unroll
    for (int q = 0; q < G::DNQ; ++q) {
      const int pi = q * 64 + lane;
      const int f = pi * G::DPB;                       // first float of the piece in the (row, OWD) band image
      const int r = f / G::OWD, x = f - r * G::OWD;
      drow[q] = (pi < G::DPP) ? (x < G::OW ? (G::DFLAT ? 0 : r) : -100000) : -200000;      // pad column: zeros
      doff[q] = r * G::OW + x;
    }
    auto dma = [&](long k) {
      const long it = (unsigned)k / (unsigned)G::NCH;
      const long tile = blockIdx.x + it * gridDim.x;
      const int ch = (int)((unsigned)k % (unsigned)G::NCH);
      const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
      const int band = (int)(tile - b * G::NBAND);
      const int y0 = band * R * S - 1;
      float* __restrict__ xb = xbuf + ((unsigned)k % (unsigned)D) * G::XB;
      const float* __restrict__ sb = p.x + b * p.x_bs + ((long)ch * KC * H + y0) * W;
      constexpr bool U8OK = CS == 4 && KC == 4 && W % 4 == 0;      // first layer from the single-frame uint8 store (stack-on-load):
      const bool staged = U8OK && p.u8.f != nullptr;                // the image is written by the loader loop below, not by DMA
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (staged || c % G::NL != lw) continue;
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
        float* __restrict__ db_ = dbuf + ((unsigned)it % (unsigned)G::DD) * G::DB;
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
    if constexpr (CS == 4 && KC == 4 && W % 4 == 0) {
      if (p.u8.f != nullptr) {
        // ---- first layer from the single-frame uint8 store (stack-on-load; one chunk per band: k = tile).  The bytes of
        // chunk k + LOOK + 1 travel to REGISTERS while chunk k + LOOK's are expanded into its ring image, so the loaders
        // never sit out an HBM latency (round 4's first version staged synchronously and ran on the D = 2 instance:
        // 6.2 ms against the fp32 source's 5.0 at N = 32,768).  vmcnt order per iteration: dOut DMA (NID), then the NLD
        // byte loads of the chunk after it.
        using RG = C3U8Regs<H, W, G::SR, G::NL>;
        static_assert(G::NID + RG::NLD <= 63, "counted vmcnt");
        RG rg;
        auto u8l = [&](long k) {
          const long tile = blockIdx.x + k * gridDim.x;
          const long b = (long)((unsigned)tile / (unsigned)G::NBAND);
          const int band = (int)(tile - b * G::NBAND);
          c3_u8_load<H, W, G::SR, G::NL>(rg, p.u8, b, band * R * S - 1, lw, lane);
        };
        auto u8e = [&](long k) { c3_u8_expand<H, W, G::SR, G::WP, G::PLANE, G::NL>(xbuf + ((unsigned)k % (unsigned)D) * G::XB, rg, lw, lane); };
        for (int kk = 0; kk < G::LOOK; ++kk)
          if (kk < nwork) {
            dma(kk);
            u8l(kk);
            c3_wait_vm<0>();
            u8e(kk);
          }
        if (G::LOOK < nwork) u8l(G::LOOK);
        c3_bar();
        for (long k = 0; k < nwork; ++k) {
          const bool more = k + G::LOOK < nwork, ahead = k + G::LOOK + 1 < nwork;
          if (more) {
            dma(k + G::LOOK);                     // dOut band of chunk k + LOOK (its bytes are already on the way)
            c3_wait_vm<G::NID>();                 // ... the bytes have landed
            u8e(k + G::LOOK);                     // into the image chunk k - 1 left
            if (ahead) {
              u8l(k + G::LOOK + 1);
            }
          }
          // opening chunk k + 1: its dOut band and its image are complete; what may stay in flight is younger
          if (G::LOOK >= 2 && more) {
            if (ahead) c3_wait_vm<G::NID + RG::NLD>();
            else c3_wait_vm<G::NID>();
          } else {
            if (ahead && more) c3_wait_vm<RG::NLD>();
            else c3_wait_vm<0>();
          }
          c3_bar();
        }
        return;
      }
    }
    // before the barrier that opens chunk k1: everything up to chunk k1 has landed; with LOOK = 2 the one younger chunk
    // (k1 + 1) stays in flight: its instruction count is NIX (+ NID when it opens a band).  (vmcnt holds 63 at most: a
    // smaller immediate only waits longer.)
    auto wait_for = [&](long k1) {
      constexpr int CA = G::NIX > 63 ? 63 : G::NIX, CB = G::NIX + G::NID > 63 ? 63 : G::NIX + G::NID;
      if (G::LOOK >= 2 && k1 + 1 < nwork && !p.nodma) {
        if ((int)((unsigned)(k1 + 1) % (unsigned)G::NCH) == 0) c3_wait_vm<CB>();
        else c3_wait_vm<CA>();
      } else {
        c3_wait_vm<0>();
      }
    };
    for (int kk = 0; kk < G::LOOK; ++kk)
      if (kk < nwork) dma(kk);
    wait_for(0);
    c3_bar();                                 // (raw s_barrier: __syncthreads() would drain vmcnt to 0)
    for (long k = 0; k < nwork; ++k) {
      if (k + G::LOOK < nwork && !p.nodma) dma(k + G::LOOK);          // into the image chunk k - 1 left
      wait_for(k + 1);
      c3_bar();
    }
    return;
  }
  // -------------------------------------------------------------------- computing waves
  const int wu = __builtin_amdgcn_readfirstlane(w);   // (scalar: the step walk below is SALU work, its branches uniform)
  const int cg = wu % NCG, pg = wu / NCG;
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
    const float* __restrict__ dimg = dbuf + ((unsigned)it % (unsigned)G::DD) * G::DB;
#pragma unroll
    for (int ch = 0; ch < G::NCH; ++ch) {
      const float* __restrict__ ximg = xbuf + ((it * G::NCH + ch) % D) * G::XB;
      // This wave's pixel steps s = pg, pg + NPG, ... (a step = 4 consecutive output pixels of a row).  The operands of
      // step s + NPG are read BEFORE the MFMAs of step s (two register sets, the loop unrolled twice): with one set the
      // loop was LDS latency + address arithmetic + 3-6 MFMAs per trip, the matrix pipe idle two thirds of the time.
      auto ld = [&](int st, float (&av)[G::NTC], float (&bv)[G::MTW]) {
        const int r = st / G::KSR, xq = st - r * G::KSR;            // (scalar)
        const int xo = r * S * G::WP + xq * 4 * S, dofs = r * G::OWD + xq * 4;
#pragma unroll
        for (int nt = 0; nt < G::NTC; ++nt) av[nt] = ximg[aoff[nt] + xo];
#pragma unroll
        for (int m = 0; m < G::MTW; ++m) bv[m] = dimg[boff[m] + dofs];
        if (G::DFLAT && xq == G::KSR - 1) {            // (uniform branch) flat image: pixels past the row's end are the next row's
#pragma unroll
          for (int m = 0; m < G::MTW; ++m) bv[m] = g < G::OW % 4 ? bv[m] : 0.f;
        }
      };
      auto mm = [&](const float (&av)[G::NTC], const float (&bv)[G::MTW]) {
#pragma unroll
        for (int nt = 0; nt < G::NTC; ++nt)
#pragma unroll
          for (int m = 0; m < G::MTW; ++m) acc[ch][nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nt], bv[m], acc[ch][nt][m], 0, 0, 0);
        if (ch == 0) {
#pragma unroll
          for (int m = 0; m < G::MTW; ++m) dbs[m] += bv[m];
        }
      };
      float a0[G::NTC], b0[G::MTW], a1[G::NTC], b1[G::MTW];
      int st = pg;
      if (st < G::KS) ld(st, a0, b0);
      while (st < G::KS) {
        if (st + G::NPG < G::KS) ld(st + G::NPG, a1, b1);
        mm(a0, b0);
        st += G::NPG;
        if (st >= G::KS) break;
        if (st + G::NPG < G::KS) ld(st + G::NPG, a0, b0);
        mm(a1, b1);
        st += G::NPG;
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

template <int CS, int CD, int H, int W, int S, int R, int KC, int NCG, int D>
int c3w_launch(const C3WP& p0, float* dW, float* db, size_t ws_bytes, hipStream_t st, size_t* need) {
  using G = C3WGeo<CS, CD, H, W, S, R, KC, NCG, D>;
  const void* k = (const void*)c3w_kernel<CS, CD, H, W, S, R, KC, NCG, D>;
  static int cus = 0, per_cu = 1;
  if (!cus) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS_BYTES) != hipSuccess) return A2C_ERR_LAUNCH;
    int dev = 0, n = 0;
    hipDeviceProp_t prop;
    // short bands (small LDS images): several workgroups per CU -- each has ONE band of lookahead, so the bands in flight
    // per CU (what hides the 2-3 us an LDS-DMA takes under load) scale with the residency
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, G::NTHR, G::LDS_BYTES) == hipSuccess && n >= 1) per_cu = n > 3 ? 3 : n;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  const long total = (long)p0.B * G::NBAND;
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;      // (the kernels walk chunks with 32-bit arithmetic)
  const long cap = (long)cus * per_cu;
  const int grid = (int)(total < cap ? total : cap);
  const size_t bytes = (size_t)grid * G::NPG * G::PER * 4;
  if (need) { *need = (size_t)cap * G::NPG * G::PER * 4; return A2C_OK; }
  if (ws_bytes < bytes) return A2C_ERR_WORKSPACE;
  hipLaunchKernelGGL((c3w_kernel<CS, CD, H, W, S, R, KC, NCG, D>), dim3(grid), dim3(G::NTHR), G::LDS_BYTES, st, p0);
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

template <int CS, int CD, int H, int W, int S, int R, bool BWD, int KCO = 0, bool SG = false>
int c3_launch(const C3P& p, hipStream_t st) {
  using G = C3Geo<CS, CD, H, W, S, R, KCO>;
  const void* k = (const void*)c3_kernel<CS, CD, H, W, S, R, BWD, KCO, SG>;
  constexpr size_t LDSB = BWD ? G::LDS_BYTES_BWD : (SG ? G::LDS_BYTES_SG : G::LDS_BYTES);
  constexpr int NTHR = 64 * (G::NW + G::NL + (SG ? 1 : 0));
  static_assert(LDSB <= 160 * 1024, "LDS");
  static_assert(!BWD || G::NCH >= 2, "the mask band is staged one chunk ahead of its use");
  static int per_cu = 0, cus = 0;
  if (!per_cu) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB) != hipSuccess) return A2C_ERR_LAUNCH;
    int n = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, NTHR, LDSB) != hipSuccess || n < 1) n = 1;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    per_cu = n;
  }
  const long total = (long)p.B * G::NBAND;
  if (total * G::NCH >= (1L << 31)) return A2C_ERR_ARG;      // (the kernels walk chunks with 32-bit arithmetic)
  const long cap = (long)per_cu * cus;
  const int grid = (int)(total < cap ? total : cap);
  hipLaunchKernelGGL((c3_kernel<CS, CD, H, W, S, R, BWD, KCO, SG>), dim3(grid), dim3(NTHR), LDSB, st, p);
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
bool c3_supported(const a2c_conv_desc* d, int kind);
// stride-2 backward-data onto an ODD image (GRUModel conv4: 32 <- 48, 21 x 21 <- 11 x 11; conv5: 48 <- 64, 11 x 11 <- 6 x 6): the
// staged sign-word kernel only (c3bs_kernel<..., ODD>); A2C_NO_ODD_BS=1 leaves the layers on conv.hip's generic band kernel
static bool c3bs_odd_shape(const a2c_conv_desc* d) {
  const char* e = getenv("A2C_NO_ODD_BS");             // (read per call: tests compare the two kernels)
  const bool off = e != nullptr && e[0] == '1';
  if (off || d->ks != 3 || d->pad != 1 || d->stride != 2 || d->H != d->W) return false;
  return (d->H == 21 && d->Cin == 32 && d->Cout == 48) ||          // GRUModel conv4: 21 x 21 <- 11 x 11
         (d->H == 11 && d->Cin == 48 && d->Cout == 64);            // GRUModel conv5: 11 x 11 <- 6 x 6
}
// backward-data with the FLOAT activation as the mask has a kernel of this family (the odd-image instance reads sign words only)
bool c3_bwd_mask_supported(const a2c_conv_desc* d) { return c3_supported(d, 1) && !c3bs_odd_shape(d); }
bool c3_supported(const a2c_conv_desc* d, int kind) {
  static const bool off = getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1';
  if (off) return false;
  if (kind == 1 && d->stride == 2) return c3b_shape(d) || c3bs_odd_shape(d);
  if (d->ks != 3 || d->pad != 1) return false;
  if (kind == 0 && d->stride == 2 && d->H == 42 && d->W == 42) return (d->Cin == 32 && d->Cout == 64) || (d->Cin == 24 && d->Cout == 32);
  if (kind == 0 && d->stride == 2 && d->H == 21 && d->W == 21) return d->Cin == 32 && d->Cout == 48;
  if (kind == 0 && d->stride == 2 && d->H == 11 && d->W == 11) return d->Cin == 48 && d->Cout == 64;
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

// source channels per chunk of the layer's kernel (the fragment layout is [chunk][tap][c4][m][64])
static int c3_kc(const a2c_conv_desc* d, int kind) {
  const int CS = kind ? d->Cout : d->Cin;
  if (kind == 0 && d->H == 11 && d->Cin == 48) return 24;      // GRUModel conv5: 6 x 6 outputs, one band per sample, two big chunks
  return CS >= 8 ? 8 : 4;
}

size_t c3_prep_floats(const a2c_conv_desc* d, int kind) {
  if (!c3_supported(d, kind)) return 0;
  const int CS = kind ? d->Cout : d->Cin, CD = kind ? d->Cin : d->Cout;
  return (size_t)(CS / 4) * 9 * ((CD + 15) / 16) * 64;
}

int c3_prep(const a2c_conv_desc* d, int kind, const float* weight, float* out, hipStream_t st) {
  const long total = (long)c3_prep_floats(d, kind);
  if (!total) return A2C_OK;
  const int CD = kind ? d->Cin : d->Cout;
  if (kind == 1 && d->stride == 2) {
    hipLaunchKernelGGL(c3b_prep_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, out, d->Cin, d->Cout, 8,
                       (CD + 15) / 16, total);
    if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
    return A2C_OK;
  }
  hipLaunchKernelGGL(c3_prep_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, out, d->Cin, d->Cout, kind,
                     c3_kc(d, kind), (CD + 15) / 16, total);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

// sign words per sample of the layer's OUTPUT ([Cout][OH][ceil(OW/32)]), 0 when its forward cannot write them
long c3_sign_words(const a2c_conv_desc* d) {
  if (!c3_supported(d, 0) || d->H == 11) return 0;
  return (long)d->Cout * d->OH * ((d->OW + 31) / 32);
}
// backward-data with the mask given as the sign words of the layer's INPUT ([Cin][H][ceil(W/32)])
bool c3_bwd_signs_supported(const a2c_conv_desc* d) { return c3_supported(d, 1); }

int c3_fwd(const a2c_conv_desc* d, const float* in, long in_bs, const float* frag, const float* bias, int relu, float* out,
           long out_bs, unsigned* signs, long signs_bs, int B, hipStream_t st) {
  C3P p{in, in_bs, frag, bias, nullptr, out, out_bs, zero_page(), B, relu, g_c3_dbg, signs, nullptr, signs_bs};
  if (!p.zero) return A2C_ERR_LAUNCH;
  p.prio = c3_prio();
  if (signs != nullptr && B > 64) {   // the direct-store kernels with the sign-word writer wave, where they have one (two chunks or more)
    if (d->stride == 1 && d->H == 84 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 1, 12, false, 0, true>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3_launch<24, 32, 84, 84, 2, 6, false, 0, true>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3_launch<16, 24, 84, 84, 2, 6, false, 0, true>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3_launch<32, 64, 42, 42, 2, 11, false, 0, true>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3_launch<24, 32, 42, 42, 2, 11, false, 0, true>(p, st);
  }
  if (signs != nullptr) {        // the staged kernels: their storers also carry the sign words
    if (d->stride == 1 && d->H == 84 && d->Cin == 4) return c3s_launch<4, 16, 84, 84, 1, 12, false>(p, st);
    if (d->stride == 1 && d->H == 84 && d->Cin == 16) return c3s_launch<16, 24, 84, 84, 1, 6, false>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3s_launch<24, 32, 84, 84, 2, 6, false>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3s_launch<16, 24, 84, 84, 2, 6, false>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3s_launch<32, 64, 42, 42, 2, 11, false>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3s_launch<24, 32, 42, 42, 2, 11, false>(p, st);
    if (d->stride == 2 && d->H == 21 && d->Cin == 32) return c3s_launch<32, 48, 21, 21, 2, 11, false>(p, st);
    return A2C_ERR_ARG;
  }
  // Without sign words the direct-store kernels: measured (tools/conv3_check.py, bench.py's rollout site timers) the staged
  // forward gains 10-20 % only on the store-bound first layer at B >= 256 and loses where a workgroup sees one band (B = 32).
  // Small batches (one band or less per CU with the tall bands): shorter bands, more workgroups, two per CU.
  const bool small = B <= 64;
  if (d->stride == 1 && d->H == 84 && d->Cin == 4)
    return small ? c3_launch<4, 16, 84, 84, 1, 6, false>(p, st) : c3s_launch<4, 16, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 1 && d->H == 84 && d->Cin == 16)
    return small ? c3_launch<16, 24, 84, 84, 1, 6, false>(p, st) : c3_launch<16, 24, 84, 84, 1, 12, false>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 24)
    return small ? c3_launch<24, 32, 84, 84, 2, 3, false>(p, st) : c3_launch<24, 32, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 16)
    return small ? c3_launch<16, 24, 84, 84, 2, 3, false>(p, st) : c3_launch<16, 24, 84, 84, 2, 6, false>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 32)
    return small ? c3_launch<32, 64, 42, 42, 2, 3, false>(p, st) : c3_launch<32, 64, 42, 42, 2, 11, false>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3_launch<24, 32, 42, 42, 2, 11, false>(p, st);
  if (d->stride == 2 && d->H == 21 && d->Cin == 32) return c3_launch<32, 48, 21, 21, 2, 11, false>(p, st);
  if (d->stride == 2 && d->H == 11 && d->Cin == 48) return c3_launch<48, 64, 11, 11, 2, 6, false, 24>(p, st);
  return A2C_ERR_ARG;
}

// rollout-step chain: GRUModel conv2 .. conv5 (16 -> 24 @84, 24 -> 32 @42, 32 -> 48 @21, 48 -> 64 @11, all 3x3 / stride 2 / pad 1)
bool c3_chain_supported(const a2c_conv_desc* d, int n) {
  static const bool off = getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1';
  const char* nc = getenv("A2C_NO_CHAIN");        // (read per call: tests compare the chain with the four launches)
  if (off || (nc != nullptr && nc[0] == '1') || n != 4) return false;
  static const int cin[4] = {16, 24, 32, 48}, cout[4] = {24, 32, 48, 64}, hw[4] = {84, 42, 21, 11};
  for (int i = 0; i < 4; ++i)
    if (d[i].ks != 3 || d[i].pad != 1 || d[i].stride != 2 || d[i].Cin != cin[i] || d[i].Cout != cout[i] || d[i].H != hw[i] || d[i].W != hw[i])
      return false;
  return true;
}

int c3_chain_fwd(const a2c_conv_desc* d, int n, const float* in, long in_bs, const float* const* frag, const float* const* bias,
                 int relu, float* const* out, const long* out_bs, unsigned* const* signs, const long* signs_bs, int B, hipStream_t st) {
  if (!c3_chain_supported(d, n)) return A2C_ERR_ARG;
  // sign words: of layer 0's output (conv2, for conv3's backward-data), of layer 1's and layer 2's (conv3 / conv4, for the
  // odd-image backward-data kernels of conv4 / conv5) -- a later layer's only together with the earlier ones'
  if ((signs[1] && !signs[0]) || (signs[2] && !signs[1]) || signs[3]) return A2C_ERR_ARG;
  C3Chain4 cp;
  const float* zp = zero_page();
  if (!zp) return A2C_ERR_LAUNCH;
  for (int i = 0; i < 4; ++i) {
    cp.l[i] = C3P{i == 0 ? in : out[i - 1], i == 0 ? in_bs : out_bs[i - 1], frag[i], bias[i], nullptr, out[i], out_bs[i], zp, B, relu,
                  nullptr, signs[i], nullptr, signs[i] ? signs_bs[i] : 0};
    cp.l[i].prio = c3_prio();
  }
  using G2 = C3Geo<16, 24, 84, 84, 2, 6, 0, 4>;
  using G3 = C3Geo<24, 32, 42, 42, 2, 11, 0, 4>;
  using G4 = C3Geo<32, 48, 21, 21, 2, 11, 0, 4>;
  using G5 = C3Geo<48, 64, 11, 11, 2, 6, 24, 4>;
  constexpr size_t m1 = G2::LDS_BYTES_SG > G3::LDS_BYTES_SG ? G2::LDS_BYTES_SG : G3::LDS_BYTES_SG;
  constexpr size_t m2 = G4::LDS_BYTES_SG > G5::LDS_BYTES ? G4::LDS_BYTES_SG : G5::LDS_BYTES;
  constexpr size_t LDSB = m1 > m2 ? m1 : m2;
  static_assert(LDSB <= 160 * 1024, "LDS");
  static int per_cu[4] = {0, 0, 0, 0}, cus = 0;
  const int v = signs[2] ? 3 : signs[1] ? 2 : (signs[0] ? 1 : 0);
  const void* k = v == 0 ? (const void*)c3_chain_gru_kernel<false, false, false> : v == 1 ? (const void*)c3_chain_gru_kernel<true, false, false>
                : v == 2 ? (const void*)c3_chain_gru_kernel<true, true, false> : (const void*)c3_chain_gru_kernel<true, true, true>;
  if (!per_cu[v]) {
    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDSB) != hipSuccess) return A2C_ERR_LAUNCH;
    int nb = 0, dev = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 64 * 13, LDSB) != hipSuccess || nb < 1) nb = 1;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    per_cu[v] = nb;
  }
  const long cap = (long)per_cu[v] * cus;
  const int grid = (int)(B < cap ? B : cap);
  if (v == 0) hipLaunchKernelGGL((c3_chain_gru_kernel<false, false, false>), dim3(grid), dim3(64 * 13), LDSB, st, cp);
  else if (v == 1) hipLaunchKernelGGL((c3_chain_gru_kernel<true, false, false>), dim3(grid), dim3(64 * 13), LDSB, st, cp);
  else if (v == 2) hipLaunchKernelGGL((c3_chain_gru_kernel<true, true, false>), dim3(grid), dim3(64 * 13), LDSB, st, cp);
  else hipLaunchKernelGGL((c3_chain_gru_kernel<true, true, true>), dim3(grid), dim3(64 * 13), LDSB, st, cp);
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

// first layer (4 -> 16 @ 84 x 84) with its input stacked on load from a single-frame uint8 store (C3U8)
bool c3_fwd_frames_supported(const a2c_conv_desc* d) {
  return c3_supported(d, 0) && d->Cin == 4 && d->Cout == 16 && d->H == 84 && d->W == 84 && d->stride == 1;
}
int c3_fwd_frames(const a2c_conv_desc* d, const unsigned char* f, long bs, long T, const int* nv, long nv_s, const float* frag,
                  const float* bias, int relu, float* out, long out_bs, unsigned* signs, long signs_bs, int B, hipStream_t st) {
  if (!c3_fwd_frames_supported(d)) return A2C_ERR_ARG;
  C3P p{nullptr, 0, frag, bias, nullptr, out, out_bs, zero_page(), B, relu, g_c3_dbg, signs, nullptr, signs_bs, C3U8{f, bs, nv, nv_s, T}};
  if (!p.zero) return A2C_ERR_LAUNCH;
  p.prio = c3_prio();
  // 256 envs: 7-row bands, two output images, four storers (49.0 -> 43.7 us; 6-row bands 45.5, 8-row 44.8); A2C_C3S_DS=0: one image
  static const bool ds = !(getenv("A2C_C3S_DS") && getenv("A2C_C3S_DS")[0] == '0');
  if (B > 64 && ds) return c3s_launch<4, 16, 84, 84, 1, 7, false, true>(p, st);      // (32 envs: 224 twelve-row bands are one per CU)
  if (signs != nullptr || B > 64) return c3s_launch<4, 16, 84, 84, 1, 12, false>(p, st);
  return c3_launch<4, 16, 84, 84, 1, 6, false>(p, st);
}

// mask: the float activation below (the kernels that keep its band in LDS) -- or signs: its sign words (the staged kernels)
int c3_bwd_data(const a2c_conv_desc* d, const float* dout, const float* frag, const float* mask, const unsigned* signs,
                long signs_bs, float* din, int B, hipStream_t st) {
  C3P p{dout, (long)d->Cout * d->OH * d->OW, frag, nullptr, mask, din, (long)d->Cin * d->H * d->W, zero_page(), B, 0, g_c3_dbg,
        nullptr, signs, signs_bs};
  if (!p.zero) return A2C_ERR_LAUNCH;
  p.prio = c3_prio();
  if (mask != nullptr) {
    if (d->stride == 1 && d->Cin == 16 && d->Cout == 24) return c3_launch<24, 16, 84, 84, 1, 12, true>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3b_launch<32, 24, 42, 42, 6, 8>(p, st);
    if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3b_launch<24, 16, 42, 42, 6, 8>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3b_launch<64, 32, 21, 21, 11, 8>(p, st);
    if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3b_launch<32, 24, 21, 21, 11, 8>(p, st);
    return A2C_ERR_ARG;
  }
  if (d->stride == 1 && d->Cin == 16 && d->Cout == 24) return c3s_launch<24, 16, 84, 84, 1, 6, true>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 24) return c3bs_launch<32, 24, 42, 42, 6, 8, 3, false>(p, st);
  if (d->stride == 2 && d->H == 84 && d->Cin == 16) return c3bs_launch<24, 16, 42, 42, 6, 8, 4, true>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 32) return c3bs_launch<64, 32, 21, 21, 11, 8, 2, false>(p, st);
  if (d->stride == 2 && d->H == 42 && d->Cin == 24) return c3bs_launch<32, 24, 21, 21, 11, 8, 3, true>(p, st);
  // the odd images, one band per sample, 16-channel chunks (8: 1.64 / 1.33 ms at N = 32,768 against 1.56 / 1.21)
  if (c3bs_odd_shape(d) && d->H == 21) return c3bs_launch<48, 32, 11, 11, 11, 16, 3, false, true>(p, st);
  // conv5: all four chunks' fragments resident (110 KB), nine computing waves = 3 pixel tiles x 3 channel tiles
  if (c3bs_odd_shape(d) && d->H == 11) return c3bs_launch<64, 48, 6, 6, 6, 16, 3, true, true, 3>(p, st);
  return A2C_ERR_ARG;
}

// layer 2's backward-data (sign-word mask) + the first layer's weight gradient from the uint8 frame store, in one pass:
// GRUModel's conv2 (16 <- 24, 84 x 84 <- 42 x 42, stride 2) over conv1 (4 -> 16, 3 x 3, stride 1, pad 1)
bool c3_bwd_data_w1_frames_supported(const a2c_conv_desc* d2, const a2c_conv_desc* d1) {
  static const bool off = getenv("A2C_NO_W1_FRAMES") && getenv("A2C_NO_W1_FRAMES")[0] == '1';
  return !off && c3_supported(d2, 1) && d2->stride == 2 && d2->H == 84 && d2->W == 84 && d2->Cin == 16 && d2->Cout == 24 &&
         d1->Cin == 4 && d1->Cout == 16 && d1->H == 84 && d1->W == 84 && d1->ks == 3 && d1->stride == 1 && d1->pad == 1;
}
size_t c3_bwd_data_w1_frames_ws_bytes(const a2c_conv_desc* d2, const a2c_conv_desc* d1, int B) {
  if (B < 1 || !c3_bwd_data_w1_frames_supported(d2, d1)) return 0;
  return c3bs_w1_ws_bytes<24, 16, 42, 42, 6, 8, 4, true>(B);
}
int c3_bwd_data_w1_frames(const a2c_conv_desc* d2, const a2c_conv_desc* d1, const float* dout, const float* frag, const unsigned* signs,
                          long signs_bs, const unsigned char* f, long bs, long T, const int* nv, float* dW1, float* db1, int B, void* ws,
                          size_t ws_bytes, hipStream_t st) {
  if (!c3_bwd_data_w1_frames_supported(d2, d1)) return A2C_ERR_ARG;
  C3P p{dout, (long)d2->Cout * d2->OH * d2->OW, frag, nullptr, nullptr, nullptr, 0, zero_page(), B, 0, g_c3_dbg,
        nullptr, signs, signs_bs, C3U8{f, bs, nv, 1, T}};
  if (!p.zero) return A2C_ERR_LAUNCH;
  p.prio = c3_prio();
  return c3bs_w1_launch<24, 16, 42, 42, 6, 8, 4, true>(p, dW1, db1, ws, ws_bytes, st);
}

// weight gradient: instantiations and their workspace// weight gradient: instantiations and their workspace
// (CS, CD, H, W, S, R rows per band, KC channels per chunk, NCG, D chunk images in the ring)
#define C3W_CASES(X)                                                                              \
  X(4, 16, 84, 84, 1, 6, 4, 1, 3)   /* conv1 of both models                                    */ \
  X(16, 24, 84, 84, 1, 3, 16, 1, 3) /* ConvModel conv2 (two channel groups: 22.3 ms against 19.3)   */ \
  X(24, 32, 84, 84, 2, 6, 8, 2, 2)  /* ConvModel conv3                                         */ \
  X(32, 64, 42, 42, 2, 7, 8, 4, 3)  /* ConvModel conv4 (R = 5 / 6 with 16-channel chunks, D = 2: 7.8 / 7.5 ms against 7.36) */ \
  X(16, 24, 84, 84, 2, 6, 8, 1, 2)  /* GRUModel conv2 (R = 5, D = 3: 3 % slower; two channel groups: 7.4 ms against 6.9)  */ \
  X(24, 32, 42, 42, 2, 7, 12, 2, 3) /* GRUModel conv3 (12-channel chunks, ring of 3: 2.88 ms; 8 / ring of 2: 3.30) */ \
  X(32, 48, 21, 21, 2, 11, 16, 4, 2) /* GRUModel conv4: the whole 11 x 11 output as one band, 2 chunks of 16 channels (9 full
                                        n-tiles each), 4 channel groups x 2 pixel groups: 1.73 ms at N = 32,768 against the generic
                                        kernel's 2.72 (KC = 8: 2.02; NCG = 2: 4.5, its 160 accumulator registers spilled) */
// (GRUModel conv5, 48 -> 64 @11, whole 6 x 6 output as one band: 1.15-1.30 ms against the generic kernel's 1.09 -- 36 pixels
// per band are 12 MFMA steps per chunk, the per-band overhead decides)
// round 3's instances (A2C_C3W_D2=1): taller bands, one chunk of lookahead
#define C3W_OLD_CASES(X)             \
  X(4, 16, 84, 84, 1, 8, 4, 1, 2)    \
  X(16, 24, 84, 84, 1, 4, 16, 1, 2)  \
  X(32, 64, 42, 42, 2, 7, 8, 4, 2)

bool c3w_supported(const a2c_conv_desc* d) {
  static const bool off = (getenv("A2C_NO_C3") != nullptr && getenv("A2C_NO_C3")[0] == '1') ||
                          (getenv("A2C_NO_C3W") != nullptr && getenv("A2C_NO_C3W")[0] == '1');
  static const bool all = !(getenv("A2C_C3W_ALL") != nullptr && getenv("A2C_C3W_ALL")[0] == '0');
  if (off || d->ks != 3 || d->pad != 1) return false;
  // Measured against conv.hip's wgrad_kernel at N = 4096 (tools/conv3_check.py): this kernel wins on the first layer
  // (4 -> 16: 0.92 vs 1.01 ms) and on ConvModel's conv4 (32 -> 64 @42: 1.29 vs 1.55 ms), ties on the 84-wide
  // 16 -> 24 layers (2.87-2.95 vs 3.00 ms, 1.15 vs 1.11 ms: matrix-bound in both, 67 TF with a quarter of the 24-channel
  // tiles empty -- and the generic instance for them spilled registers), and lost on the 24 -> 32 layers until the loader
  // waves got their priority, the flat dOut bands and four waves for dword rows (round 4; N = 32,768: 11.2 vs 12.0 ms
  // at 84, 3.30 vs 3.56 at 42).  A2C_C3W_ALL=0 keeps those two on the generic kernel.
  if (!all && d->Cin == 24) return false;
#define C3W_MATCH(cs, cd, h, w_, s_, r, kc, ncg, dd) if (d->Cin == cs && d->Cout == cd && d->H == h && d->W == w_ && d->stride == s_) return true;
  C3W_CASES(C3W_MATCH)
#undef C3W_MATCH
  return false;
}

static int c3w_dispatch(const a2c_conv_desc* d, const C3WP& p, float* dW, float* db, size_t ws_bytes, hipStream_t st, size_t* need) {
  const char* e_old = getenv("A2C_C3W_D2");       // (read per call: tests switch it)
  const bool old = e_old != nullptr && e_old[0] == '1';
#define C3W_RUN(cs, cd, h, w_, s_, r, kc, ncg, dd) \
  if (d->Cin == cs && d->Cout == cd && d->H == h && d->W == w_ && d->stride == s_) return c3w_launch<cs, cd, h, w_, s_, r, kc, ncg, dd>(p, dW, db, ws_bytes, st, need);
  if (old) { C3W_OLD_CASES(C3W_RUN) }
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
  static const int nodma = getenv("A2C_C3W_NODMA") != nullptr && getenv("A2C_C3W_NODMA")[0] == '1';
  p.nodma = nodma;
  p.prio = c3_prio();
  if (!p.zero) return A2C_ERR_LAUNCH;
  return c3w_dispatch(d, p, dW, db, ws_bytes, st, nullptr);
}

int c3w_bwd_weight_frames(const a2c_conv_desc* d, const unsigned char* f, long bs, long T, const int* nv, const float* dout,
                          float* dW, float* db, int B, void* ws, size_t ws_bytes, hipStream_t st) {
  if (!(c3w_supported(d) && d->Cin == 4)) return A2C_ERR_ARG;
  C3WP p{nullptr, 0, dout, (float*)ws, zero_page(), B, C3U8{f, bs, nv, 1, T}};
  if (!p.zero) return A2C_ERR_LAUNCH;
  p.prio = c3_prio();
  return c3w_dispatch(d, p, dW, db, ws_bytes, st, nullptr);
}
