// Convolutions of the model classes (nn.Conv2d + ReLU: models.py:98, 312, 685), forward,
// backward-data and backward-weight, NCHW fp32.
//
// Structure (all three): a 256-thread workgroup stages an input tile (all input planes x the
// rows one band of output rows needs) into LDS -- the whole tile is one flattened index space
// swept by all 256 threads with 8 independent 16-byte loads in flight per thread (a plane's rows
// are contiguous in HBM when there is no padding, so the sweep is a straight coalesced copy) --
// then its 4 waves run an implicit GEMM over that tile on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain).  The four k-groups of a wave (4 input planes)
// sit PLANE = 16 (mod 32) floats apart so a B-operand read spreads over the LDS banks
// (stride-1/2 layers conflict-free, the stride-4 layer 2-way).
//
//  forward:  D[co, pix] = sum_k W[co,k] X[k,pix];  k = (tap, ci) with ci fastest, one MFMA step =
//            4 input planes at one tap; A fragments (weights) pre-laid by a2c_conv2d_prep_weights,
//            read coalesced from L1/L2 one 8-step chunk AHEAD of the MFMAs that use them;
//            B fragments gathered from the LDS tile.
//  bwd-data: the S*S output-parity classes of dX are S*S stride-1 correlations of dOut with a
//            sub-kernel of W (taps ky = ry + S*a), i.e. the SAME kernel with other tables.
//            The ReLU derivative of the layer below is fused into the store (mask > 0).
//  bwd-weight: dW[co,k] = sum_pix dOut[co,pix] X[k,pix]: A = dOut tile (LDS), B = input tile
//            (LDS), accumulators persist in registers across all samples of a workgroup,
//            per-workgroup partial slabs + fixed-order reduction (deterministic), bias gradient
//            from the same dOut tile.
#include <stdlib.h>
#include <algorithm>
#include <map>
#include <mutex>
#include "a2c_common.h"

// conv3.hip: shape-specialised channel-chunk streaming kernels for the 3x3 layers at 84x84 (fragments appended to the
// prepared-weight buffers of this file's layouts)
bool c3_supported(const a2c_conv_desc* d, int kind);
size_t c3_prep_floats(const a2c_conv_desc* d, int kind);
int c3_prep(const a2c_conv_desc* d, int kind, const float* weight, float* out, hipStream_t st);
int c3_fwd(const a2c_conv_desc* d, const float* in, long in_bs, const float* frag, const float* bias, int relu, float* out,
           long out_bs, unsigned* signs, long signs_bs, int B, hipStream_t st);
int c3_bwd_data(const a2c_conv_desc* d, const float* dout, const float* frag, const float* mask, const unsigned* signs,
                long signs_bs, float* din, int B, hipStream_t st);
long c3_sign_words(const a2c_conv_desc* d);
bool c3_bwd_signs_supported(const a2c_conv_desc* d);
bool c3_bwd_mask_supported(const a2c_conv_desc* d);
bool c3w_supported(const a2c_conv_desc* d);
size_t c3w_ws_bytes(const a2c_conv_desc* d);
int c3w_bwd_weight(const a2c_conv_desc* d, const float* in, long in_bs, const float* dout, float* dW, float* db, int B, void* ws,
                   size_t ws_bytes, hipStream_t st);
bool c3_fwd_frames_supported(const a2c_conv_desc* d);
bool c3_chain_supported(const a2c_conv_desc* d, int n);
int c3_chain_fwd(const a2c_conv_desc* d, int n, const float* in, long in_bs, const float* const* frag, const float* const* bias,
                 int relu, float* const* out, const long* out_bs, unsigned* const* signs, const long* signs_bs, int B, hipStream_t st);
int c3_fwd_frames(const a2c_conv_desc* d, const unsigned char* f, long bs, long T, const int* nv, long nv_s, const float* frag,
                  const float* bias, int relu, float* out, long out_bs, unsigned* signs, long signs_bs, int B, hipStream_t st);
int c3w_bwd_weight_frames(const a2c_conv_desc* d, const unsigned char* f, long bs, long T, const int* nv, const float* dout,
                          float* dW, float* db, int B, void* ws, size_t ws_bytes, hipStream_t st);
bool c3_bwd_data_w1_frames_supported(const a2c_conv_desc* d2, const a2c_conv_desc* d1);
size_t c3_bwd_data_w1_frames_ws_bytes(const a2c_conv_desc* d2, const a2c_conv_desc* d1, int B);
int c3_bwd_data_w1_frames(const a2c_conv_desc* d2, const a2c_conv_desc* d1, const float* dout, const float* frag, const unsigned* signs,
                          long signs_bs, const unsigned char* f, long bs, long T, const int* nv, float* dW1, float* db1, int B, void* ws,
                          size_t ws_bytes, hipStream_t st);

namespace {
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int MAX_TAPS = 64;
struct BandKey {         // (layer, batch, flags) key of the tile-size tuners' caches
  int v[9];
  bool operator<(const BandKey& o) const { return std::lexicographical_compare(v, v + 9, o.v, o.v + 9); }
};
static int env_kb(const char* name, int dflt_kb) {
  const char* v = getenv(name);
  const int kb = v ? atoi(v) : 0;
  return (kb > 0 ? kb : dflt_kb) * 1024;
}
// LDS per workgroup decides how many workgroups share a CU (160 KB): tunable for experiments
#define IGEMM_LDS_BUDGET env_kb("A2C_IGEMM_LDS_KB", 64)
#define WGRAD_LDS_BUDGET env_kb("A2C_WGRAD_LDS_KB", 48)   // 3 workgroups per CU: the tiled kernels stage synchronously and rely on neighbours for overlap (76 KB: -10..25 % on the 3x3 layers)
constexpr int LDS_HARD_MAX = 160 * 1024;

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// persistent grids: exactly the number of workgroups the chip keeps resident (a static tile
// stride over a grid that is not a multiple of it leaves the last round mostly idle)
static int resident_grid(const void* kernel, size_t lds_bytes, long total_tiles, int block = 256) {
  int per_cu = 0, dev = 0, cus = 256;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds_bytes) != hipSuccess || per_cu < 1) per_cu = 1;
  hipDeviceProp_t prop;
  static int cached_cus = 0;
  if (!cached_cus) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached_cus = prop.multiProcessorCount;
    else cached_cus = 256;
  }
  cus = cached_cus;
  const long g = (long)per_cu * cus;
  return (int)(total_tiles < g ? (total_tiles > 0 ? total_tiles : 1) : g);
}
static inline int ilog2(int v) { int s = 0; while ((1 << s) < v) ++s; return s; }

// ------------------------------------------------------------------ tile geometry
struct SrcTile {         // how a band of TPH pixel rows maps onto an LDS image of the source
  int Cp, IH, IW;        // source planes / size
  int SY, SX;            // source rows / cols per pixel row / col
  int sy0, sx0;          // source row/col of (pixel 0, tap 0)
  int span_y, span_x;    // tap extent
  int PH, PW;            // pixel grid
  int TPH, TIH, WP, PLANE, tiles;
};

static void plan_src(SrcTile& t, int extra_floats_per_row_of_pixels, int extra_fixed, int budget) {
  const int ncols = (t.PW - 1) * t.SX + t.span_x;
  t.WP = (t.sx0 == 0 && ncols <= t.IW) ? t.IW : ncols;    // unpadded layers keep whole rows (contiguous copy)
  if (t.WP < t.IW - t.sx0) t.WP = t.IW - t.sx0;            // whole source rows are staged (vector loads)
  int best = 1;
  for (int tph = 1; tph <= t.PH; ++tph) {
    const int tih = (tph - 1) * t.SY + t.span_y;
    const int plane = ((tih * t.WP + 31) / 32) * 32 + 16;
    const long bytes = 4L * ((long)t.Cp * plane + (long)extra_floats_per_row_of_pixels * tph + extra_fixed + 64);
    if (bytes <= budget) best = tph; else break;
  }
  t.TPH = best;
  t.TIH = (best - 1) * t.SY + t.span_y;
  t.PLANE = ((t.TIH * t.WP + 31) / 32) * 32 + 16;
  t.tiles = ceil_div(t.PH, t.TPH);
}

// ------------------------------------------------------------------ staging (device)
struct StageP {
  const float* src; long bstride;   // sample stride (floats)
  int Cp, IH, IW, TIH, WP, PLANE, sx0, fast;
  int vec;                          // widest aligned load for a source row: 4, 2 or 1 floats
  int flat;                         // the tile is the WHOLE sample (one tile per sample): its Cp*IH*IW floats are one
                                    // contiguous, 16-B aligned run whatever the row width -- float4 loads, scattered to
                                    // the padded image per element (pad rows keep the zeros of stage_zero)
};

constexpr int STAGE_U = 8;          // independent loads in flight per thread

// zero the whole image once per kernel: halo columns / plane slack are never written again
__device__ __forceinline__ void stage_zero(const StageP& s, float* __restrict__ lds) {
  for (int i = threadIdx.x; i < s.Cp * s.PLANE + 64; i += 256) lds[i] = 0.f;
}

template <int VEC>
__device__ __forceinline__ void stage_rows(const StageP& s, float* __restrict__ lds, const float* __restrict__ base,
                                           int y_lo) {
  // flattened (plane, row, vector) space; every data position of the image is rewritten each tile
  // (rows outside the source as zeros), halo columns keep the zeros of stage_zero()
  const int nv = s.IW / VEC;
  const int total = s.Cp * s.TIH * nv;
  for (int i0 = threadIdx.x; i0 < total; i0 += 256 * STAGE_U) {
    float v[STAGE_U][VEC];
    int dst[STAGE_U];
#pragma unroll
    for (int u = 0; u < STAGE_U; ++u) {
      const int idx = i0 + u * 256;
      dst[u] = -1;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[u][e] = 0.f;
      if (idx < total) {
        const int rt = idx / nv, x = (idx - rt * nv) * VEC;
        const int c = rt / s.TIH, r = rt - c * s.TIH;
        const int ys = y_lo + r;
        dst[u] = c * s.PLANE + r * s.WP + x - s.sx0;
        if (ys >= 0 && ys < s.IH) {
          const float* p = base + ((long)c * s.IH + ys) * s.IW + x;
          if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
          else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[u][0] = t.x; v[u][1] = t.y; }
          else v[u][0] = p[0];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < STAGE_U; ++u)
      if (dst[u] >= 0) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) lds[dst[u] + e] = v[u][e];
      }
  }
}

// All 256 threads sweep the image; `fast` = unpadded full-width rows (a plane is one contiguous
// run: float4 loads AND float4 LDS stores); otherwise vector loads per row + scalar LDS stores.
__device__ __forceinline__ void stage_tile(const StageP& s, float* __restrict__ lds, long b, int y_lo) {
  const int tid = threadIdx.x;
  const float* __restrict__ base = s.src + b * s.bstride;
  if (s.fast) {
    // no padding, WP == IW: plane c of the image is the contiguous run of TIH rows from y_lo
    const int per4 = (s.TIH * s.IW) >> 2;                       // float4 per plane image
    const int ok4 = (min(s.TIH, s.IH - y_lo) * s.IW) >> 2;      // float4 that exist in the source
    const int total = s.Cp * per4;
    for (int i0 = tid; i0 < total; i0 += 256 * STAGE_U) {
      float4 v[STAGE_U];
      int dst[STAGE_U];
#pragma unroll
      for (int u = 0; u < STAGE_U; ++u) {
        const int idx = i0 + u * 256;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        dst[u] = -1;
        if (idx < total) {
          const int c = idx / per4, rem = idx - c * per4;
          dst[u] = c * s.PLANE + (rem << 2);
          if (rem < ok4) v[u] = *reinterpret_cast<const float4*>(base + ((long)c * s.IH + y_lo) * s.IW + (rem << 2));
        }
      }
#pragma unroll
      for (int u = 0; u < STAGE_U; ++u)
        if (dst[u] >= 0) *reinterpret_cast<float4*>(lds + dst[u]) = v[u];
    }
  } else if (s.flat && s.vec != 4) {
    const int hw = s.IH * s.IW, total4 = (s.Cp * hw) >> 2;
    for (int i0 = tid; i0 < total4; i0 += 256 * STAGE_U) {
      float4 v[STAGE_U];
#pragma unroll
      for (int u = 0; u < STAGE_U; ++u) {
        const int idx = i0 + u * 256;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < total4) v[u] = reinterpret_cast<const float4*>(base)[idx];
      }
#pragma unroll
      for (int u = 0; u < STAGE_U; ++u) {
        const int idx = i0 + u * 256;
        if (idx < total4) {
          const int e = idx << 2;
          int c = e / hw, rem = e - c * hw;
          int r = rem / s.IW, x = rem - r * s.IW;
          const float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            lds[c * s.PLANE + (r - y_lo) * s.WP + x - s.sx0] = f[k];
            if (++x == s.IW) { x = 0; if (++r == s.IH) { r = 0; ++c; } }
          }
        }
      }
    }
  } else if (s.vec == 4) stage_rows<4>(s, lds, base, y_lo);
  else if (s.vec == 2) stage_rows<2>(s, lds, base, y_lo);
  else stage_rows<1>(s, lds, base, y_lo);
}

// ------------------------------------------------------------------ forward / backward-data
constexpr int CH = 8;                // MFMA steps per A-fragment chunk (prefetched one chunk ahead)
constexpr int MAX_STEPS = 128;

struct IgemmP {
  StageP st;
  float* out; long out_bs; int Mch, OHf, OWf;      // output tensor (B, Mch, OHf, OWf)
  const float* wfrag; const float* bias; const float* mask; int relu;
  int nchunks;                                      // ceil(nsteps / CH); fragments are zero-padded to it
  int PH, PW, oy_mul, oy_add, ox_mul, ox_add;       // pixel (q,p) -> out (q*oy_mul+oy_add, p*ox_mul+ox_add)
  int SY, SX, sy0, TPH, tiles, B;
  // LDS offset of MFMA step s = (a*nb + b)*c4n + c4 (tap row a, tap col b, channel quad c4):
  //   off0 + a*step_a + b*step_b + c4*step_c     -- scalar arithmetic, no table
  int nsteps, nb, c4n, off0, step_a, step_b, step_c;
  // forward only: > 0 = assemble the tile's output [Mch][TPH*PW] in LDS at this float offset and
  // flush it with coalesced stores (a tile's rows of one channel are one contiguous HBM run)
  int out_stage, out_vec;
};

struct StepIter {      // wave-uniform walker over the step offsets
  int off, bi, ci;
  __device__ __forceinline__ void init(const IgemmP& p) { off = p.off0; bi = 0; ci = 0; }
  __device__ __forceinline__ void next(const IgemmP& p) {   // branch-free: selects only, so a chunk of
    ++ci;                                                     // steps stays one basic block
    const bool w1 = (ci == p.c4n);
    ci = w1 ? 0 : ci;
    bi += w1 ? 1 : 0;
    const bool w2 = (bi == p.nb);
    bi = w2 ? 0 : bi;
    off += p.step_c + (w1 ? p.step_b - p.c4n * p.step_c : 0) + (w2 ? p.step_a - p.nb * p.step_b : 0);
  }
};

template <int MT>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long total = (long)p.B * p.tiles;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  stage_zero(p.st, lds);
  for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int qq0 = ti * p.TPH;
    const int rows = min(p.TPH, p.PH - qq0);
    const int NP = rows * p.PW;
    __syncthreads();                       // readers of the previous tile (and the zero fill) are done
    stage_tile(p.st, lds, b, qq0 * p.SY + p.sy0);
    __syncthreads();
    const int npairs = (NP + 31) >> 5;
    for (int pr = w; pr < npairs; pr += 4) {
      const int idx0 = pr * 32 + j, idx1 = idx0 + 16;
      const bool ok0 = idx0 < NP, ok1 = idx1 < NP;
      const int i0 = ok0 ? idx0 : 0, i1 = ok1 ? idx1 : 0;
      const int r0 = i0 / p.PW, c0 = i0 - r0 * p.PW;
      const int r1 = i1 / p.PW, c1 = i1 - r1 * p.PW;
      const float* __restrict__ l0 = lds + r0 * p.SY * WP + c0 * p.SX + g * PLANE;
      const float* __restrict__ l1 = lds + r1 * p.SY * WP + c1 * p.SX + g * PLANE;
      f32x4 acc[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      const float* __restrict__ wf = p.wfrag + lane;
      float a_cur[CH * MT], a_nxt[CH * MT];
#pragma unroll
      for (int i = 0; i < CH * MT; ++i) a_cur[i] = wf[i * 64];
      StepIter it;
      it.init(p);
      for (int ck = 0; ck < p.nchunks; ++ck) {
        if (ck + 1 < p.nchunks) {
#pragma unroll
          for (int i = 0; i < CH * MT; ++i) a_nxt[i] = wf[((ck + 1) * CH * MT + i) * 64];
        }
        float bv0[CH], bv1[CH];
#pragma unroll
        for (int u = 0; u < CH; ++u) {       // all B gathers of the chunk first ...
          const int off = (ck * CH + u < p.nsteps) ? it.off : 0;   // padded steps (A = 0) read a valid word
          it.next(p);
          bv0[u] = l0[off];
          bv1[u] = l1[off];
        }
#pragma unroll
        for (int u = 0; u < CH; ++u)         // ... then its MFMAs back to back
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u * MT + m], bv0[u], acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u * MT + m], bv1[u], acc[m][1], 0, 0, 0);
          }
#pragma unroll
        for (int i = 0; i < CH * MT; ++i) a_cur[i] = a_nxt[i];
      }
      // D map of the 16x16 tile: col (pixel) = lane & 15, row (channel) = 4*(lane>>4) + reg
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (!(nt ? ok1 : ok0)) continue;
        const int r = nt ? r1 : r0, c = nt ? c1 : c0;
        const long yo = (long)(qq0 + r) * p.oy_mul + p.oy_add;
        const long xo = (long)c * p.ox_mul + p.ox_add;
        const long pix = yo * p.OWf + xo;
        const int ip = nt ? i1 : i0;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int co = m * 16 + 4 * g + rr;
            if (co < p.Mch) {
              float v = acc[m][nt][rr];
              if (p.bias) v += p.bias[co];
              if (p.relu) v = fmaxf(v, 0.f);
              if (p.out_stage > 0) {
                lds[p.out_stage + co * p.TPH * p.PW + ip] = v;
              } else {
                const long o = b * p.out_bs + (long)co * p.OHf * p.OWf + pix;
                if (p.mask && !(p.mask[o] > 0.f)) v = 0.f;
                p.out[o] = v;
              }
            }
          }
      }
    }
    if (p.out_stage > 0) {       // coalesced flush: NP contiguous floats per channel
      __syncthreads();
      float* __restrict__ ob = p.out + b * p.out_bs + (long)qq0 * p.OWf;
      const float* __restrict__ sb = lds + p.out_stage;
      const long chs = (long)p.OHf * p.OWf;
      // a wave takes one channel at a time (no per-element division); 16 B vectors when the
      // launcher found every run aligned, else 4 B
      if (p.out_vec == 2) {      // the tile is the whole sample: [Mch][OH*OW] is one contiguous 16-B aligned run
        const int n4 = (p.Mch * NP) >> 2;
        for (int i = threadIdx.x; i < n4; i += 256) reinterpret_cast<float4*>(ob)[i] = reinterpret_cast<const float4*>(sb)[i];
      } else
      for (int co = w; co < p.Mch; co += 4) {
        float* __restrict__ oc = ob + co * chs;
        const float* __restrict__ sc = sb + co * p.TPH * p.PW;
        if (p.out_vec) {
          for (int e = lane << 2; e < NP; e += 256) *reinterpret_cast<float4*>(oc + e) = *reinterpret_cast<const float4*>(sc + e);
        } else {
          for (int e = lane; e < NP; e += 64) oc[e] = sc[e];
        }
      }
    }
  }
}

// "Row-run" forward kernel for the unpadded layers (A3CModel: 8x8/s4 and 4x4/s2), software
// pipelined:
//  * K is ordered (c4, ky, kx) with kx fastest, so the KS taps of one kernel row are KS
//    CONTIGUOUS floats of the image per lane (lane = output pixel, byte offset 4*S*ox, i.e.
//    16 B aligned for S=4, 8 B for S=2): two ds_read_b128 (or b64) feed KS MFMA steps instead of
//    KS bank-conflicting ds_read_b32 gathers.  PLANE = 0 (b128) / 32 (b64) mod 64 floats makes
//    the 4 planes of a wave's k-groups fall on distinct 16-lane bank groups: conflict-free.
//  * the NEXT tile's global loads are issued into registers before the current tile's MFMA
//    phase and written to LDS after it, so HBM latency hides under the matrix work of the same
//    workgroup.  The compute phase touches only LDS (B rows AND the A fragments, copied to LDS
//    once per workgroup) plus the output stores, so no wait on a global load drains the prefetch.
constexpr int PF_N = 10;             // float4 prefetch registers per thread  (256*10*16 B = 40 KB image)

template <int MT, int KS, int S>
__global__ __launch_bounds__(256) void igemm_run_kernel(IgemmP p, int nfrag) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ ldsA = lds;                 // [nsteps*MT][64] weight fragments
  float* __restrict__ img = lds + nfrag;          // tile image
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long total = (long)p.B * p.tiles;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  for (int i = tid; i < nfrag; i += 256) ldsA[i] = p.wfrag[i];
  float bias_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int co = m * 16 + 4 * g + rr;
      bias_r[m][rr] = (p.bias && co < p.Mch) ? p.bias[co] : 0.f;
    }
  // flattened float4 index space of the image (same as stage_tile's fast path)
  const int per4 = (p.st.TIH * p.st.IW) >> 2;
  const int tot4 = p.st.Cp * per4;
  int dst[PF_N], srcoff[PF_N], rem4[PF_N];
#pragma unroll
  for (int u = 0; u < PF_N; ++u) {
    const int idx = tid + u * 256;
    dst[u] = -1; srcoff[u] = 0; rem4[u] = 0;
    if (idx < tot4) {
      const int c = idx / per4, rem = idx - c * per4;
      dst[u] = c * PLANE + (rem << 2);
      srcoff[u] = c * p.st.IH * p.st.IW + (rem << 2);
      rem4[u] = rem;
    }
  }
  float4 pf[PF_N];
  auto issue = [&](long tile) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int y_lo = ti * p.TPH * p.SY + p.sy0;
    const int ok4 = (min(p.st.TIH, p.st.IH - y_lo) * p.st.IW) >> 2;
    const float* __restrict__ base = p.st.src + b * p.st.bstride + (long)y_lo * p.st.IW;
#pragma unroll
    for (int u = 0; u < PF_N; ++u) {
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (dst[u] >= 0 && rem4[u] < ok4) pf[u] = *reinterpret_cast<const float4*>(base + srcoff[u]);
    }
  };
  long tile = blockIdx.x;
  if (tile < total) issue(tile);
  for (; tile < total; tile += gridDim.x) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int qq0 = ti * p.TPH;
    const int rows = min(p.TPH, p.PH - qq0);
    const int NP = rows * p.PW;
    __syncthreads();                       // readers of the previous tile are done
#pragma unroll
    for (int u = 0; u < PF_N; ++u)
      if (dst[u] >= 0) *reinterpret_cast<float4*>(img + dst[u]) = pf[u];
    __syncthreads();
    if (tile + gridDim.x < total) issue(tile + gridDim.x);     // in flight during the MFMA phase below
    const int npairs = (NP + 31) >> 5;
    for (int pr = w; pr < npairs; pr += 4) {
      const int idx0 = pr * 32 + j, idx1 = idx0 + 16;
      const bool ok0 = idx0 < NP, ok1 = idx1 < NP;
      const int i0 = ok0 ? idx0 : 0, i1 = ok1 ? idx1 : 0;
      const int r0 = i0 / p.PW, c0 = i0 - r0 * p.PW;
      const int r1 = i1 / p.PW, c1 = i1 - r1 * p.PW;
      const float* __restrict__ l0 = img + r0 * S * WP + c0 * S + g * PLANE;
      const float* __restrict__ l1 = img + r1 * S * WP + c1 * S + g * PLANE;
      const float* __restrict__ la = ldsA + lane;
      f32x4 acc[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      int s = 0;
      for (int c4 = 0; c4 < p.c4n; ++c4) {
        const int poff = c4 * 4 * PLANE;
#pragma unroll 2
        for (int ky = 0; ky < KS; ++ky, s += KS) {
          const int off = poff + ky * WP;
          float bv0[KS], bv1[KS], av[KS * MT];
          if (S == 4) {                    // 16 B aligned: ds_read_b128
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
              const float4 t0 = *reinterpret_cast<const float4*>(l0 + off + 4 * q);
              const float4 t1 = *reinterpret_cast<const float4*>(l1 + off + 4 * q);
              bv0[4 * q] = t0.x; bv0[4 * q + 1] = t0.y; bv0[4 * q + 2] = t0.z; bv0[4 * q + 3] = t0.w;
              bv1[4 * q] = t1.x; bv1[4 * q + 1] = t1.y; bv1[4 * q + 2] = t1.z; bv1[4 * q + 3] = t1.w;
            }
          } else {                         // 8 B aligned: ds_read_b64
#pragma unroll
            for (int q = 0; q < KS / 2; ++q) {
              const float2 t0 = *reinterpret_cast<const float2*>(l0 + off + 2 * q);
              const float2 t1 = *reinterpret_cast<const float2*>(l1 + off + 2 * q);
              bv0[2 * q] = t0.x; bv0[2 * q + 1] = t0.y;
              bv1[2 * q] = t1.x; bv1[2 * q + 1] = t1.y;
            }
          }
#pragma unroll
          for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) av[u * MT + m] = la[((s + u) * MT + m) * 64];
#pragma unroll
          for (int u = 0; u < KS; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv0[u], acc[m][0], 0, 0, 0);
              acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv1[u], acc[m][1], 0, 0, 0);
            }
        }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (!(nt ? ok1 : ok0)) continue;
        const int r = nt ? r1 : r0, c = nt ? c1 : c0;
        const long pix = ((long)(qq0 + r) * p.oy_mul + p.oy_add) * p.OWf + (long)c * p.ox_mul + p.ox_add;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int co = m * 16 + 4 * g + rr;
            if (co < p.Mch) {
              float v = acc[m][nt][rr] + bias_r[m][rr];
              if (p.relu) v = fmaxf(v, 0.f);
              p.out[b * p.out_bs + (long)co * p.OHf * p.OWf + pix] = v;
            }
          }
      }
    }
  }
}

// "Streaming" forward kernel for the 8x8 / stride 4 layer on 4 input planes (A3CModel conv1) at
// large batch: ONE persistent 8-wave workgroup per CU, a whole input sample in LDS, the NEXT
// sample in flight in registers (14 float4 per thread, issued right before the matrix phase, i.e. a
// full sample of lead time), outputs staged in LDS and flushed with coalesced float4 stores.  Two
// barriers per sample; between them the 25 16-pixel tiles run back to back (A fragments in
// registers, B rows by ds_read_b128).  Every thread executes the same number of global loads and
// stores (out-of-range slots are clamped onto a valid element), so the s_waitcnt vmcnt(n) before
// the LDS commit is exact and the flush stores never stall the prefetch.
constexpr int ST_NT = 512;
constexpr int ST_NS = 14;            // float4 prefetch slots per thread: 4 planes x H*W <= 14*512*4 floats
constexpr int ST_FL = 4;             // float4 flush slots per thread:   Cout x OH*OW <= 4*512*4 floats

struct StreamP {
  const float* in; long in_bs;
  const float* wfrag; const float* bias;
  float* out; long out_bs;
  int B, H, W, OH, OW, Mch, relu, PLANE1, PLANEo;
  int dbg;      // experiments (A2C_STREAM_DBG): 1 = no matrix phase, 2 = no input loads, 4 = no output stores
};

// one float4 slot of the sample: global -> register / register -> LDS.  Named scalars, not arrays:
// the prefetch registers live across the persistent loop's back edge and must not end up in scratch.
#define ST_LD(var, u, src) \
  if (!(p.dbg & 2)) var = *reinterpret_cast<const float4*>((src) + (min(tid + (u) * ST_NT, tot4 - 1) << 2));
#define ST_ST(var, u)                                                                          \
  {                                                                                            \
    const int idx_ = min(tid + (u) * ST_NT, tot4 - 1);                                          \
    const int c_ = (idx_ >= per4) + (idx_ >= 2 * per4) + (idx_ >= 3 * per4);                   \
    *reinterpret_cast<float4*>(img + c_ * p.PLANE1 + ((idx_ - c_ * per4) << 2)) = var;          \
  }
#define ST_LDALL(src) ST_LD(v0, 0, src) ST_LD(v1, 1, src) ST_LD(v2, 2, src) ST_LD(v3, 3, src) ST_LD(v4, 4, src) ST_LD(v5, 5, src) ST_LD(v6, 6, src) \
  ST_LD(v7, 7, src) ST_LD(v8, 8, src) ST_LD(v9, 9, src) ST_LD(v10, 10, src) ST_LD(v11, 11, src) ST_LD(v12, 12, src) ST_LD(v13, 13, src)
#define ST_STALL ST_ST(v0, 0) ST_ST(v1, 1) ST_ST(v2, 2) ST_ST(v3, 3) ST_ST(v4, 4) ST_ST(v5, 5) ST_ST(v6, 6) ST_ST(v7, 7) ST_ST(v8, 8) \
  ST_ST(v9, 9) ST_ST(v10, 10) ST_ST(v11, 11) ST_ST(v12, 12) ST_ST(v13, 13)

__global__ __launch_bounds__(ST_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_stream_kernel(StreamP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ img = lds;
  float* __restrict__ ob = img + 4 * p.PLANE1;        // [16][PLANEo] output staging (+ bias in the 4 spare words per row)
  float* __restrict__ part = ob + 16 * p.PLANEo;      // [8][256] K-split partials of the leftover tile
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int HW = p.H * p.W, W = p.W, NP = p.OH * p.OW;
  const int per4 = HW >> 2, tot4 = 4 * per4;
  // A fragments: all 64 steps in registers (every tile uses them), plus kernel row `w` once more
  // for the K-split leftover tile (register arrays cannot be indexed by the wave id)
  float af[64], aw[8];
#pragma unroll
  for (int s = 0; s < 64; ++s) af[s] = p.wfrag[s * 64 + lane];
#pragma unroll
  for (int kx = 0; kx < 8; ++kx) aw[kx] = p.wfrag[(w * 8 + kx) * 64 + lane];
  const float b10 = (p.bias && 4 * g + 0 < p.Mch) ? p.bias[4 * g + 0] : 0.f, b11 = (p.bias && 4 * g + 1 < p.Mch) ? p.bias[4 * g + 1] : 0.f;
  const float b12 = (p.bias && 4 * g + 2 < p.Mch) ? p.bias[4 * g + 2] : 0.f, b13 = (p.bias && 4 * g + 3 < p.Mch) ? p.bias[4 * g + 3] : 0.f;
  if (tid < 16) ob[tid * p.PLANEo + NP] = (p.bias && tid < p.Mch) ? p.bias[tid] : 0.f;
  const int ntile = (NP + 15) >> 4;
  const int nfull = ntile - (ntile & 7);             // tiles done whole by one wave; the (at most one) leftover is K-split
  const int split0 = nfull * 16;                     // first pixel of the leftover tile (== NP rounded down if none)
  const int out4 = split0 >> 2, otot = p.Mch * out4;  // float4 per channel flushed from the staging rows
  float4 v0 = {}, v1 = {}, v2 = {}, v3 = {}, v4 = {}, v5 = {}, v6 = {}, v7 = {}, v8 = {}, v9 = {}, v10 = {}, v11 = {}, v12 = {}, v13 = {};
  long n = blockIdx.x;
  if (n >= p.B) return;
  {
    const float* __restrict__ src = p.in + n * p.in_bs;
    ST_LDALL(src)
  }
  for (; n < p.B; n += gridDim.x) {
    const long nn = (n + gridDim.x < p.B) ? n + gridDim.x : n;        // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.in + nn * p.in_bs;
    float* __restrict__ dst = p.out + n * p.out_bs;
    ST_STALL                                         // sample n: registers -> LDS
    __syncthreads();
    // matrix phase: 3 whole tiles per wave + one kernel row of the leftover tile, unrolled so that the
    // loads of sample n + grid are issued in instalments between the tiles.  The instalments are
    // straight-line code on purpose: any branch around a global load (even a uniform one) makes the
    // compiler fall back to conservative vmcnt waits (measured 1.36 ms vs 1.23 ms); one burst before
    // the first tile costs 1.45 ms, one load per kernel row inside the tiles 1.50 ms.
#define ST_TILE(T)                                                                                          \
    if (!(p.dbg & 1) && (T) * (ST_NT / 64) + w < nfull) {                                                   \
      const int idx = ((T) * (ST_NT / 64) + w) * 16 + j;                                                    \
      const bool ok = idx < NP;                                                                             \
      const int i = ok ? idx : 0;                                                                           \
      const int r = i / p.OW, c = i - r * p.OW;                                                             \
      const float* __restrict__ l = img + r * 4 * W + c * 4 + g * p.PLANE1;                                 \
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                              \
      _Pragma("unroll") for (int ky = 0; ky < 8; ++ky) {                                                    \
        const float4 t0v = *reinterpret_cast<const float4*>(l + ky * W);                                    \
        const float4 t1v = *reinterpret_cast<const float4*>(l + ky * W + 4);                                \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 0], t0v.x, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 1], t0v.y, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 2], t0v.z, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 3], t0v.w, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 4], t1v.x, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 5], t1v.y, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 6], t1v.z, acc, 0, 0, 0);                    \
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[ky * 8 + 7], t1v.w, acc, 0, 0, 0);                    \
      }                                                                                                     \
      if (ok) {                                                                                             \
        float o0 = acc[0] + b10, o1 = acc[1] + b11, o2 = acc[2] + b12, o3 = acc[3] + b13;                   \
        if (p.relu) { o0 = fmaxf(o0, 0.f); o1 = fmaxf(o1, 0.f); o2 = fmaxf(o2, 0.f); o3 = fmaxf(o3, 0.f); } \
        ob[(4 * g + 0) * p.PLANEo + i] = o0; ob[(4 * g + 1) * p.PLANEo + i] = o1;                           \
        ob[(4 * g + 2) * p.PLANEo + i] = o2; ob[(4 * g + 3) * p.PLANEo + i] = o3;                           \
      }                                                                                                     \
    }
    ST_LD(v0, 0, nsrc) ST_LD(v1, 1, nsrc) ST_LD(v2, 2, nsrc) ST_LD(v3, 3, nsrc)
    ST_TILE(0)
    ST_LD(v4, 4, nsrc) ST_LD(v5, 5, nsrc) ST_LD(v6, 6, nsrc) ST_LD(v7, 7, nsrc)
    ST_TILE(1)
    ST_LD(v8, 8, nsrc) ST_LD(v9, 9, nsrc) ST_LD(v10, 10, nsrc) ST_LD(v11, 11, nsrc)
    ST_TILE(2)
    ST_LD(v12, 12, nsrc) ST_LD(v13, 13, nsrc)
    if (!(p.dbg & 1) && nfull < ntile) {             // leftover tile: this wave's kernel row ky = w (8 of its 64 steps)
      const int idx = nfull * 16 + j;
      const int i = idx < NP ? idx : 0;
      const int r = i / p.OW, c = i - r * p.OW;
      const float* __restrict__ l = img + r * 4 * W + c * 4 + g * p.PLANE1 + w * W;
      const float4 t0v = *reinterpret_cast<const float4*>(l);
      const float4 t1v = *reinterpret_cast<const float4*>(l + 4);
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[0], t0v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[1], t0v.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[2], t0v.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[3], t0v.w, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[4], t1v.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[5], t1v.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[6], t1v.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[7], t1v.w, acc, 0, 0, 0);
      *reinterpret_cast<float4*>(part + w * 256 + lane * 4) = (float4){acc[0], acc[1], acc[2], acc[3]};
    }
    __syncthreads();
    if (nfull < ntile) {                             // leftover tile: one output per thread, kernel rows summed in a
      const int k = tid & 255, co = min(k >> 4, p.Mch - 1), jx = k & 15;      // fixed order, + bias, ReLU (threads >= 256 duplicate)
      const int slot = (jx + 16 * (co >> 2)) * 4 + (co & 3);
      float q[ST_NT / 64];
#pragma unroll
      for (int x = 0; x < ST_NT / 64; ++x) q[x] = part[x * 256 + slot];
      float sum = q[0];
#pragma unroll
      for (int x = 1; x < ST_NT / 64; ++x) sum += q[x];
      sum += ob[co * p.PLANEo + NP];
      if (p.relu) sum = fmaxf(sum, 0.f);
      if (!(p.dbg & 4)) dst[(long)co * NP + split0 + jx] = sum;
    }
#pragma unroll
    for (int u = 0; u < ST_FL; ++u) {                // outputs of the whole tiles of sample n: LDS -> HBM
      const int idx = min(tid + u * ST_NT, otot - 1);
      const int co = idx / out4, e = (idx - co * out4) << 2;
      const float4 t = *reinterpret_cast<const float4*>(ob + co * p.PLANEo + e);
      if (!(p.dbg & 4)) *reinterpret_cast<float4*>(dst + (long)co * NP + e) = t;
    }
  }
}

static size_t stream_lds(const StreamP& p) { return 4 * (size_t)(4 * p.PLANE1 + 16 * p.PLANEo + 8 * 256); }
static bool plan_stream(const a2c_conv_desc* d, StreamP& p) {
  if (!(d->ks == 8 && d->stride == 4 && d->pad == 0 && d->Cin == 4 && d->Cout <= 16 && d->W % 4 == 0)) return false;
  const int NP = d->OH * d->OW;
  if (NP % 4) return false;
  p.H = d->H; p.W = d->W; p.OH = d->OH; p.OW = d->OW; p.Mch = d->Cout;
  p.PLANE1 = ((d->H * d->W + 63) / 64) * 64;
  p.PLANEo = ((NP + 7) / 8) * 8 + 4;
  if (d->H * d->W > ST_NS * ST_NT || d->Cout * NP > ST_FL * ST_NT * 4) return false;
  const int ntile = (NP + 15) / 16;
  if (ntile / 8 > 3 || ntile % 8 > 1 || NP % 16) return false;   // 3 whole tiles per wave + at most one K-split leftover
  return stream_lds(p) <= 160 * 1024;
}

// "Streaming" forward kernel for the 4x4 / stride 2 layer 16 -> 32 channels (A3CModel conv2) at
// large batch: persistent 8-wave workgroups, TWO samples per iteration so that the 2 x 6 tiles x 2
// channel halves = 24 units spread evenly (3 per wave, no K split); wave w keeps the 64 A
// fragments of its channel half m = w & 1 in registers; the next pair of samples is in flight in
// registers; outputs staged in LDS and flushed with coalesced float4 stores.
constexpr int S2_NT = 512, S2_NS = 7, S2_FL = 3;
struct Stream2P {
  const float* in; long in_bs;
  const float* wfrag; const float* bias;
  float* out; long out_bs;
  int B, H, W, OH, OW, Mch, relu, PLANE;
};
#define S2_LD(var, u)                                                                          \
  {                                                                                            \
    const int idx_ = min(tid + (u) * S2_NT, 2 * tot4 - 1);                                      \
    const int hs_ = idx_ >= tot4;                                                              \
    var = *reinterpret_cast<const float4*>((hs_ ? nsrc1 : nsrc0) + ((idx_ - hs_ * tot4) << 2)); \
  }
#define S2_ST(var, u)                                                                          \
  {                                                                                            \
    const int idx_ = min(tid + (u) * S2_NT, 2 * tot4 - 1);                                      \
    const int pl_ = idx_ / per4;                                                               \
    *reinterpret_cast<float4*>(img + pl_ * p.PLANE + ((idx_ - pl_ * per4) << 2)) = var;         \
  }

__global__ __launch_bounds__(S2_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv2_stream_kernel(Stream2P p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ img = lds;                       // [2 samples][16 planes][PLANE]
  float* __restrict__ ob = lds + 32 * p.PLANE;         // [2 samples][Mch*NP] flat (c, y, x)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int HW = p.H * p.W, NP = p.OH * p.OW;
  const int per4 = HW >> 2, tot4 = 16 * per4;          // float4 per plane / per sample
  const int on = p.Mch * NP, on4 = on >> 2;            // output floats / float4 per sample
  const int m = w & 1, wq = w >> 1;                    // channel half; 4 waves share a half
  float af[64];
#pragma unroll
  for (int s = 0; s < 64; ++s) af[s] = p.wfrag[(s * 2 + m) * 64 + lane];
  float bz[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) bz[rr] = (p.bias && m * 16 + 4 * g + rr < p.Mch) ? p.bias[m * 16 + 4 * g + rr] : 0.f;
  const int ntile = (NP + 15) >> 4;                    // per sample
  float4 v0 = {}, v1 = {}, v2 = {}, v3 = {}, v4 = {}, v5 = {}, v6 = {};
  const long npair = (p.B + 1) >> 1;
  long n = blockIdx.x;
  if (n >= npair) return;
  {
    const float* __restrict__ nsrc0 = p.in + (2 * n) * p.in_bs;
    const float* __restrict__ nsrc1 = p.in + min(2 * n + 1, (long)p.B - 1) * p.in_bs;
    S2_LD(v0, 0) S2_LD(v1, 1) S2_LD(v2, 2) S2_LD(v3, 3) S2_LD(v4, 4) S2_LD(v5, 5) S2_LD(v6, 6)
  }
  for (; n < npair; n += gridDim.x) {
    const long nn = (n + gridDim.x < npair) ? n + gridDim.x : n;     // past the end: re-read this pair (discarded)
    const float* __restrict__ nsrc0 = p.in + (2 * nn) * p.in_bs;
    const float* __restrict__ nsrc1 = p.in + min(2 * nn + 1, (long)p.B - 1) * p.in_bs;
    float* __restrict__ dst0 = p.out + (2 * n) * p.out_bs;
    float* __restrict__ dst1 = p.out + min(2 * n + 1, (long)p.B - 1) * p.out_bs;
    S2_ST(v0, 0) S2_ST(v1, 1) S2_ST(v2, 2) S2_ST(v3, 3) S2_ST(v4, 4) S2_ST(v5, 5) S2_ST(v6, 6)
    __syncthreads();
#define S2_UNITS(U0, U1)                                                                                    \
    for (int un = (U0) * 4 + wq; un < (U1) * 4 && un < 2 * ntile; un += 4) {                                \
      const int sm = un >= ntile, tile = un - sm * ntile;                                                   \
      const int idx = tile * 16 + j;                                                                        \
      const bool ok = idx < NP;                                                                             \
      const int i = ok ? idx : 0;                                                                           \
      const int r = i / p.OW, c = i - r * p.OW;                                                             \
      const float* __restrict__ l = img + (sm * 16 + g) * p.PLANE + r * 2 * p.W + c * 2;                    \
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                              \
      _Pragma("unroll") for (int c4 = 0; c4 < 4; ++c4) {                                                    \
        float bv[16];                                                                                       \
        _Pragma("unroll") for (int ky = 0; ky < 4; ++ky) {                                                  \
          const int off = c4 * 4 * p.PLANE + ky * p.W;                                                      \
          const float2 t0 = *reinterpret_cast<const float2*>(l + off);                                      \
          const float2 t1 = *reinterpret_cast<const float2*>(l + off + 2);                                  \
          bv[ky * 4 + 0] = t0.x; bv[ky * 4 + 1] = t0.y; bv[ky * 4 + 2] = t1.x; bv[ky * 4 + 3] = t1.y;       \
        }                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 16; ++u)                                                      \
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c4 * 16 + u], bv[u], acc, 0, 0, 0);                 \
      }                                                                                                     \
      if (ok) {                                                                                             \
        _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                  \
          const int co = m * 16 + 4 * g + rr;                                                               \
          float o = acc[rr] + bz[rr];                                                                       \
          if (p.relu) o = fmaxf(o, 0.f);                                                                    \
          if (co < p.Mch) ob[sm * on + co * NP + i] = o;                                                    \
        }                                                                                                   \
      }                                                                                                     \
    }
    S2_LD(v0, 0) S2_LD(v1, 1) S2_LD(v2, 2) S2_LD(v3, 3)
    S2_UNITS(0, 1)
    S2_LD(v4, 4) S2_LD(v5, 5) S2_LD(v6, 6)
    S2_UNITS(1, 64)
    __syncthreads();
#pragma unroll
    for (int u = 0; u < S2_FL; ++u) {                  // both samples' outputs: LDS -> HBM
      const int idx = min(tid + u * S2_NT, 2 * on4 - 1);
      const int hs = idx >= on4, e = (idx - hs * on4) << 2;
      *reinterpret_cast<float4*>((hs ? dst1 : dst0) + e) = *reinterpret_cast<const float4*>(ob + hs * on + e);
    }
  }
}

static bool plan_stream2(const a2c_conv_desc* d, Stream2P& p) {
  if (!(d->ks == 4 && d->stride == 2 && d->pad == 0 && d->Cin == 16 && d->Cout > 16 && d->Cout <= 32 && d->W % 2 == 0)) return false;
  const int HW = d->H * d->W, NP = d->OH * d->OW;
  if (HW % 4 || (d->Cout * NP) % 4) return false;
  p.H = d->H; p.W = d->W; p.OH = d->OH; p.OW = d->OW; p.Mch = d->Cout;
  p.PLANE = ((HW + 63) / 64) * 64 + 32;                // planes 32 mod 64 floats apart: conflict-free ds_read_b64
  if (2 * 16 * HW > S2_NS * S2_NT * 4 || 2 * d->Cout * NP > S2_FL * S2_NT * 4) return false;
  return 4 * (size_t)(32 * p.PLANE + 2 * d->Cout * NP) <= 160 * 1024;
}

// Same pipeline for the 3x3 / pad 1 layers (ConvModel, GRUModel) whose rows are 16 B aligned
// (W % 4 == 0): the image keeps a one-column zero halo (LDS column = x + 1, zeroed once), source
// rows above/below the picture are prefetched as zeros, the three taps of a kernel row are three
// consecutive LDS words per lane, and the output band is assembled in LDS and flushed with
// coalesced stores (these layers' outputs are up to 14.8 GB per tensor at 32 768 samples).
constexpr int PF3 = 16;              // float4 prefetch registers per thread (64 KB image)
template <int MT, int S>
__global__ __launch_bounds__(256) void igemm_run3_kernel(IgemmP p, int nfrag) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ ldsA = lds;
  float* __restrict__ img = lds + nfrag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const long total = (long)p.B * p.tiles;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  float* __restrict__ outb = img + p.st.Cp * PLANE + 64;          // [Mch][TPH*PW]
  for (int i = tid; i < nfrag; i += 256) ldsA[i] = p.wfrag[i];
  for (int i = tid; i < p.st.Cp * PLANE + 64; i += 256) img[i] = 0.f;    // halo columns stay 0
  float bias_r[MT][4];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int co = m * 16 + 4 * g + rr;
      bias_r[m][rr] = (p.bias && co < p.Mch) ? p.bias[co] : 0.f;
    }
  const int nv = p.st.IW >> 2;
  const int per4 = p.st.TIH * nv;
  const int tot4 = p.st.Cp * per4;
  int dst[PF3], srcoff[PF3], rrow[PF3];
#pragma unroll
  for (int u = 0; u < PF3; ++u) {
    const int idx = tid + u * 256;
    dst[u] = -1; srcoff[u] = 0; rrow[u] = 0;
    if (idx < tot4) {
      const int c = idx / per4, rem = idx - c * per4;
      const int r = rem / nv, x = (rem - r * nv) << 2;
      dst[u] = c * PLANE + r * WP + x + 1;
      srcoff[u] = (c * p.st.IH + r) * p.st.IW + x;
      rrow[u] = r;
    }
  }
  float4 pf[PF3];
  auto issue = [&](long tile) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int y_lo = ti * p.TPH * S - 1;
    const float* __restrict__ base = p.st.src + b * p.st.bstride + (long)y_lo * p.st.IW;
#pragma unroll
    for (int u = 0; u < PF3; ++u) {
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ys = y_lo + rrow[u];
      if (dst[u] >= 0 && ys >= 0 && ys < p.st.IH) pf[u] = *reinterpret_cast<const float4*>(base + srcoff[u]);
    }
  };
  long tile = blockIdx.x;
  if (tile < total) issue(tile);
  for (; tile < total; tile += gridDim.x) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int qq0 = ti * p.TPH;
    const int rows = min(p.TPH, p.PH - qq0);
    const int NP = rows * p.PW;
    __syncthreads();                       // readers of the previous tile (image and out band) are done
#pragma unroll
    for (int u = 0; u < PF3; ++u)
      if (dst[u] >= 0) {
        img[dst[u]] = pf[u].x; img[dst[u] + 1] = pf[u].y; img[dst[u] + 2] = pf[u].z; img[dst[u] + 3] = pf[u].w;
      }
    __syncthreads();
    if (tile + gridDim.x < total) issue(tile + gridDim.x);
    const int npairs = (NP + 31) >> 5;
    for (int pr = w; pr < npairs; pr += 4) {
      const int idx0 = pr * 32 + j, idx1 = idx0 + 16;
      const bool ok0 = idx0 < NP, ok1 = idx1 < NP;
      const int i0 = ok0 ? idx0 : 0, i1 = ok1 ? idx1 : 0;
      const int r0 = i0 / p.PW, c0 = i0 - r0 * p.PW;
      const int r1 = i1 / p.PW, c1 = i1 - r1 * p.PW;
      const float* __restrict__ l0 = img + r0 * S * WP + c0 * S + g * PLANE;
      const float* __restrict__ l1 = img + r1 * S * WP + c1 * S + g * PLANE;
      const float* __restrict__ la = ldsA + lane;
      f32x4 acc[MT][2];
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      for (int c4 = 0; c4 < p.c4n; ++c4) {       // 9 MFMA steps (3 kernel rows x 3 taps) per channel quad
        const int poff = c4 * 4 * PLANE;
        const int s0 = c4 * 9;
        float bv0[9], bv1[9], av[9 * MT];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            bv0[ky * 3 + kx] = l0[poff + ky * WP + kx];
            bv1[ky * 3 + kx] = l1[poff + ky * WP + kx];
          }
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m) av[u * MT + m] = la[((s0 + u) * MT + m) * 64];
#pragma unroll
        for (int u = 0; u < 9; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv0[u], acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv1[u], acc[m][1], 0, 0, 0);
          }
      }
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        if (!(nt ? ok1 : ok0)) continue;
        const int ip = nt ? i1 : i0;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int co = m * 16 + 4 * g + rr;
            if (co < p.Mch) {
              float v = acc[m][nt][rr] + bias_r[m][rr];
              if (p.relu) v = fmaxf(v, 0.f);
              outb[co * p.TPH * p.PW + ip] = v;
            }
          }
      }
    }
    __syncthreads();
    {  // coalesced flush: NP contiguous floats per channel
      float* __restrict__ ob = p.out + b * p.out_bs + (long)qq0 * p.OWf;
      const long chs = (long)p.OHf * p.OWf;
      if (p.out_vec) {
        const int p4 = NP >> 2, n4 = p.Mch * p4;
        for (int i = tid; i < n4; i += 256) {
          const int co = i / p4, e = (i - co * p4) << 2;
          *reinterpret_cast<float4*>(ob + co * chs + e) = *reinterpret_cast<const float4*>(outb + co * p.TPH * p.PW + e);
        }
      } else {
        const int n = p.Mch * NP;
        for (int i = tid; i < n; i += 256) {
          const int co = i / NP, e = i - co * NP;
          ob[co * chs + e] = outb[co * p.TPH * p.PW + e];
        }
      }
    }
  }
}

// layers that take the row-run kernel (and therefore the (c4, ky, kx) fragment order)
static bool run_layout(const a2c_conv_desc* d) {
  return d->pad == 0 && d->W % 4 == 0 && d->Cout <= 32 &&
         ((d->ks == 8 && d->stride == 4) || (d->ks == 4 && d->stride == 2));
}
// 3x3 / pad 1 layers with 16 B aligned rows whose fragments fit LDS: igemm_run3_kernel
static bool run3_layout(const a2c_conv_desc* d) {
  return d->ks == 3 && d->pad == 1 && d->W % 4 == 0 && d->Cout <= 32 && (d->stride == 1 || d->stride == 2) &&
         (size_t)9 * (d->Cin / 4) * ((d->Cout + 15) / 16) * 64 * 4 <= 48 * 1024;
}

// weights -> A fragments.  kind 0 (forward): step s = tap*(Cin/4) + c4, lane l:
//   W[co = mt*16 + (l&15)][ci = 4*c4 + (l>>4)][ky][kx],  tap = ky*ks + kx.
// kind 1 (backward-data), class (ry,rx), taps (a,b): ky = ry + S*a, kx = rx + S*b,
//   step s = tap*(Cout/4) + c4, lane l: W[co = 4*c4 + (l>>4)][ci = mt*16 + (l&15)][ky][kx].
__global__ __launch_bounds__(256) void prep_fwd_kernel(const float* __restrict__ W, float* __restrict__ out, int Cin,
                                                       int Cout, int ks, int MT, int run_order, long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int l = (int)(i & 63);
    long q = i >> 6;
    const int mt = (int)(q % MT);
    q /= MT;
    const int c4n = Cin >> 2;
    int c4 = (int)(q % c4n), tap = (int)(q / c4n);          // default: step = tap*c4n + c4
    if (run_order) { tap = (int)(q % (ks * ks)); c4 = (int)(q / (ks * ks)); if (c4 >= c4n) { c4 = 0; tap = ks * ks; } }
    const int ky = tap / ks, kx = tap - ky * ks;            // run order: step = (c4*ks + ky)*ks + kx
    const int co = mt * 16 + (l & 15), ci = 4 * c4 + (l >> 4);
    out[i] = (co < Cout && tap < ks * ks) ? W[(((long)co * Cin + ci) * ks + ky) * ks + kx] : 0.f;   // pad steps = 0
  }
}

__global__ __launch_bounds__(256) void prep_bwd_kernel(const float* __restrict__ W, float* __restrict__ out, int Cin,
                                                       int Cout, int ks, int S, int ry, int rx, int na, int nb, int MT,
                                                       long total) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int l = (int)(i & 63);
    long q = i >> 6;
    const int mt = (int)(q % MT);
    q /= MT;
    const int c4n = Cout >> 2;
    const int c4 = (int)(q % c4n), tap = (int)(q / c4n);
    const int a = tap / nb, bb = tap - a * nb;
    const int ky = ry + S * a, kx = rx + S * bb;
    const int co = 4 * c4 + (l >> 4), ci = mt * 16 + (l & 15);
    out[i] = (ci < Cin && a < na) ? W[(((long)co * Cin + ci) * ks + ky) * ks + kx] : 0.f;          // pad steps = 0
  }
}

static inline int ntaps_1d(int ks, int S, int r) { return r < ks ? (ks - r + S - 1) / S : 0; }
static inline int pad_steps(int nsteps) { return ceil_div(nsteps > 0 ? nsteps : 1, CH) * CH; }

static bool desc_ok(const a2c_conv_desc* d) {
  if (!d) return false;
  if (d->Cin < 4 || d->Cin % 4 || d->Cout < 4 || d->Cout % 4 || d->Cout > 64 || d->Cin > 64) return false;
  if (d->ks < 1 || d->stride < 1 || d->stride > 4 || (d->stride & (d->stride - 1)) || d->pad < 0 || d->pad >= d->ks) return false;
  if (d->ks * d->ks > MAX_TAPS) return false;
  if (pad_steps(d->ks * d->ks * (d->Cin / 4)) > MAX_STEPS) return false;
  if (pad_steps(ntaps_1d(d->ks, d->stride, 0) * ntaps_1d(d->ks, d->stride, 0) * (d->Cout / 4)) > MAX_STEPS) return false;
  if (d->OH != (d->H - d->ks + 2 * d->pad) / d->stride + 1 || d->OW != (d->W - d->ks + 2 * d->pad) / d->stride + 1) return false;
  return d->OH >= 1 && d->OW >= 1;
}

static size_t bwd_class_offset(const a2c_conv_desc* d, int cls) {   // floats before class `cls`
  const int S = d->stride, MTb = ceil_div(d->Cin, 16);
  size_t off = 0;
  for (int c = 0; c < cls; ++c) {
    const int ry = c / S, rx = c % S;
    off += (size_t)pad_steps(ntaps_1d(d->ks, S, ry) * ntaps_1d(d->ks, S, rx) * (d->Cout / 4)) * MTb * 64;
  }
  return off;
}

template <int MT>
static void launch_igemm_t(const IgemmP& p, int grid, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)igemm_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  grid = resident_grid((const void*)igemm_kernel<MT>, lds, (long)p.B * p.tiles);
  hipLaunchKernelGGL(igemm_kernel<MT>, dim3(grid), dim3(256), lds, st, p);
}

static int launch_igemm(const IgemmP& p, int MT, hipStream_t st) {
  const size_t lds = 4 * ((size_t)p.st.Cp * p.st.PLANE + 64 + (p.out_stage > 0 ? (size_t)p.Mch * p.TPH * p.PW + 16 : 0));
  if (lds > LDS_HARD_MAX) return A2C_ERR_ARG;
  const long total = (long)p.B * p.tiles;
  const int grid = (int)(total < 2048 ? total : 2048);
  switch (MT) {
    case 1: launch_igemm_t<1>(p, grid, lds, st); break;
    case 2: launch_igemm_t<2>(p, grid, lds, st); break;
    case 3: launch_igemm_t<3>(p, grid, lds, st); break;
    case 4: launch_igemm_t<4>(p, grid, lds, st); break;
    default: return A2C_ERR_ARG;
  }
  if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH;
  return A2C_OK;
}

static int stage_vec(const float* src, long bstride, int IH, int IW) {
  if (IW % 4 == 0 && bstride % 4 == 0 && ((long)IH * IW) % 4 == 0 && (uintptr_t)src % 16 == 0) return 4;
  if (IW % 2 == 0 && bstride % 2 == 0 && ((long)IH * IW) % 2 == 0 && (uintptr_t)src % 8 == 0) return 2;
  return 1;
}

static void fill_stage(StageP& s, const SrcTile& t, const float* src, long bstride) {
  s.src = src; s.bstride = bstride;
  s.Cp = t.Cp; s.IH = t.IH; s.IW = t.IW; s.TIH = t.TIH; s.WP = t.WP; s.PLANE = t.PLANE; s.sx0 = t.sx0;
  s.fast = (t.sx0 == 0) && (t.sy0 >= 0) && (t.WP == t.IW) && (t.IW % 4 == 0) && (bstride % 4 == 0) &&
           ((uintptr_t)src % 16 == 0);
  s.vec = stage_vec(src, bstride, t.IH, t.IW);
  s.flat = (t.tiles == 1 && t.sy0 <= 0 && t.TIH + t.sy0 >= t.IH && bstride % 4 == 0 && ((uintptr_t)src % 16 == 0) &&
            ((long)t.Cp * t.IH * t.IW) % 4 == 0 && !getenv("A2C_NO_FLAT_STAGE")) ? 1 : 0;
}

// ------------------------------------------------------------------ backward-weight
struct WgradP {
  StageP st;                      // input tile
  const float* dout;              // (B, Cout, OH, OW) contiguous
  float* slab;                    // [grid][Cout*K + Cout]
  int Cout, K, ks, OH, OW, OWp;   // OWp = OW rounded up to 4
  int S, sy0, TPH, tiles, B;
  int PLANEo;                     // dOut LDS plane stride (= 2 mod 32)
  int nkt;                        // ceil(K/16)
  int dvec;                       // dOut rows are 16 B aligned float4 streams
  int dflat;                      // the tile is the whole sample: dOut (Cout*OH*OW floats) is ONE 16-B aligned run
};

// RS ("row split"): every wave owns ALL KTW = nkt k-tiles and the tile's pixel rows are dealt round-robin
// to the four waves (one A read feeds KTW MFMAs per 16 channels, no idle wave when nkt % 4 != 0); the
// per-wave partial sums are added in wave order through LDS once, at the end of the launch.
// NI > 0 ("prefetch"): the NEXT tile's input rows (NI float4 per thread) and dOut rows (ND float2 per
// thread) are loaded into registers while the MFMA phase of the current tile runs, so a workgroup hides
// its own HBM latency; each thread's (plane, row, vector) slots are tile-invariant and decoded once.
// VI / VD: floats per input / dOut prefetch slot (the widest load the row width allows: 84-wide rows 4 / 2,
// 42-wide inputs 2, 21-wide dOut 1).
template <int MT, int KTW, bool RS = false, int NI = 0, int ND = 0, bool EX = false, int VI = 4, int VD = 2>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgradP p) {
  constexpr bool PF = NI > 0;
  constexpr bool BR = PF && MT * KTW < 18;      // bias gradient summed from the prefetch registers (when they are to spare)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  const int in_floats = p.st.Cp * PLANE;
  float* ldo = lds + in_floats;                       // dOut tile: [MT*16][PLANEo]
  const int lds_total = in_floats + MT * 16 * p.PLANEo + 64;
  for (int i = threadIdx.x; i < lds_total; i += 256) lds[i] = 0.f;   // padding stays 0 (finite) forever

  // this lane's k offsets (natural weight order k = (ci*ks + ky)*ks + kx), k-tiles w, w+4, ...
  int koff[KTW];
#pragma unroll
  for (int q = 0; q < KTW; ++q) {
    const int k = (RS ? q : w + 4 * q) * 16 + j;
    int o = 0;
    if (k < p.K) {
      const int ci = k / (p.ks * p.ks), rem = k - ci * p.ks * p.ks;
      const int ky = rem / p.ks, kx = rem - ky * p.ks;
      o = ci * PLANE + ky * WP + kx;
    }
    koff[q] = o + kk * p.S;                           // + pixel (4*c4 + kk) of the step, S columns apart
  }
  f32x4 acc[MT][KTW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < KTW; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // bias gradient: thread -> (channel, part)
  const int Cm = MT * 16;
  const int nparts = 256 / Cm;
  const int bco = threadIdx.x % Cm, bpart = threadIdx.x / Cm;
  float dbacc = 0.f;

  const long total = (long)p.B * p.tiles;
  const int c4n = p.OWp >> 2;
  // prefetch slots: (plane << 16) | (row << 8) | vector, -1 = none
  int di[PF ? NI : 1], dd[PF ? ND : 1];
  float vi[PF ? NI : 1][VI];
  float vd[PF ? ND : 1][VD];
  float dbs[BR ? ND : 1];         // bias gradient: this thread's dOut slots, summed over its tiles
#pragma unroll
  for (int u = 0; u < (BR ? ND : 1); ++u) dbs[u] = 0.f;
  if (PF) {
    const int nv = p.st.IW / VI, nvd = p.OW / VD;
    const int toti = p.st.Cp * p.st.TIH * nv, totd = p.Cout * p.TPH * nvd;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int rt = idx / nv, c = rt / p.st.TIH;
      di[u] = idx < toti ? (c << 16) | ((rt - c * p.st.TIH) << 8) | (idx - rt * nv) : -1;
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int idx = threadIdx.x + u * 256;
      const int rt = idx / nvd, co = rt / p.TPH;
      dd[u] = idx < totd ? (co << 16) | ((rt - co * p.TPH) << 8) | (idx - rt * nvd) : -1;
    }
  }
  auto issue = [&](long tile) {
    const long b = tile / p.tiles;
    const int q0 = (int)(tile - b * p.tiles) * p.TPH;
    const int rows = min(p.TPH, p.OH - q0), y_lo = q0 * p.S + p.sy0;
    const float* __restrict__ base = p.st.src + b * p.st.bstride;
    const float* __restrict__ dsrc = p.dout + (b * p.Cout * p.OH + q0) * (long)p.OW;
#pragma unroll
    for (int u = 0; u < NI; ++u) {
#pragma unroll
      for (int e = 0; e < VI; ++e) vi[u][e] = 0.f;
      const int ys = y_lo + ((di[u] >> 8) & 255);
      if (di[u] >= 0 && ys >= 0 && ys < p.st.IH) {
        const float* q = base + ((long)(di[u] >> 16) * p.st.IH + ys) * p.st.IW + (di[u] & 255) * VI;
        if (VI == 4) { const float4 t = *reinterpret_cast<const float4*>(q); vi[u][0] = t.x; vi[u][1] = t.y; vi[u][2] = t.z; vi[u][3] = t.w; }
        else if (VI == 2) { const float2 t = *reinterpret_cast<const float2*>(q); vi[u][0] = t.x; vi[u][1] = t.y; }
        else vi[u][0] = q[0];
      }
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
#pragma unroll
      for (int e = 0; e < VD; ++e) vd[u][e] = 0.f;
      const int r = (dd[u] >> 8) & 255;
      if (dd[u] >= 0 && r < rows) {
        const float* q = dsrc + ((long)(dd[u] >> 16) * p.OH + r) * p.OW + (dd[u] & 255) * VD;
        if (VD == 2) { const float2 t = *reinterpret_cast<const float2*>(q); vd[u][0] = t.x; vd[u][1] = t.y; }
        else vd[u][0] = q[0];
      }
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < NI; ++u)
      if (di[u] >= 0) {
        float* d = lds + (di[u] >> 16) * PLANE + ((di[u] >> 8) & 255) * WP + (di[u] & 255) * VI - p.st.sx0;
#pragma unroll
        for (int e = 0; e < VI; ++e) d[e] = vi[u][e];
      }
#pragma unroll
    for (int u = 0; u < ND; ++u)
      if (dd[u] >= 0) {
        float* d = ldo + (dd[u] >> 16) * p.PLANEo + ((dd[u] >> 8) & 255) * p.OWp + (dd[u] & 255) * VD;
#pragma unroll
        for (int e = 0; e < VD; ++e) {
          d[e] = vd[u][e];
          if (BR) dbs[u] += vd[u][e];
        }
      }
  };
  if (PF && blockIdx.x < total) issue(blockIdx.x);
  for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int q0 = ti * p.TPH;
    const int rows = min(p.TPH, p.OH - q0);
    __syncthreads();
    if (PF) {
      commit();
      __syncthreads();
      if (tile + gridDim.x < total) issue(tile + gridDim.x);
    } else {
    stage_tile(p.st, lds, b, q0 * p.S + p.sy0);
    {  // dOut rows q0..q0+rows of every channel (a contiguous run per channel) -> ldo[co][r*OWp + c];
       // pad columns stay 0.  Flattened over all threads, 8 independent loads in flight each.
      const int per = rows * p.OW;
      const float* __restrict__ dsrc = p.dout + (b * p.Cout * p.OH + q0) * (long)p.OW;
      if (p.dflat && p.dvec != 4) {      // whole sample, any row width: float4 loads, scattered to the padded planes
        const int hw = p.OH * p.OW, tot4 = (p.Cout * hw) >> 2;
        const float4* __restrict__ d4 = reinterpret_cast<const float4*>(p.dout + b * (long)p.Cout * hw);
        for (int i0 = threadIdx.x; i0 < tot4; i0 += 256 * STAGE_U) {
          float4 v[STAGE_U];
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u) {
            const int idx = i0 + u * 256;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < tot4) v[u] = d4[idx];
          }
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u) {
            const int idx = i0 + u * 256;
            if (idx < tot4) {
              const int e = idx << 2;
              int co = e / hw, rem = e - co * hw;
              int r = rem / p.OW, x = rem - r * p.OW;
              const float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
              for (int k = 0; k < 4; ++k) {
                ldo[co * p.PLANEo + r * p.OWp + x] = f[k];
                if (++x == p.OW) { x = 0; if (++r == p.OH) { r = 0; ++co; } }
              }
            }
          }
        }
      } else if (p.dvec == 4) {       // OW % 4 == 0, 16 B aligned: a channel's rows are one float4 stream
        const int per4 = per >> 2, tot4 = p.Cout * per4;
        for (int i0 = threadIdx.x; i0 < tot4; i0 += 256 * STAGE_U) {
          float4 v[STAGE_U];
          int dst[STAGE_U];
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u) {
            const int idx = i0 + u * 256;
            dst[u] = -1;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < tot4) {
              const int co = idx / per4, e = (idx - co * per4) << 2;
              const int r = e / p.OW, x = e - r * p.OW;
              dst[u] = co * p.PLANEo + r * p.OWp + x;
              v[u] = *reinterpret_cast<const float4*>(dsrc + (long)co * p.OH * p.OW + e);
            }
          }
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u)
            if (dst[u] >= 0) { ldo[dst[u]] = v[u].x; ldo[dst[u] + 1] = v[u].y; ldo[dst[u] + 2] = v[u].z; ldo[dst[u] + 3] = v[u].w; }
        }
      } else if (p.dvec == 2) {   // OW % 2 == 0, 8 B aligned (42-wide layers): float2 stream, never across a row
        const int per2 = per >> 1, tot2 = p.Cout * per2;
        for (int i0 = threadIdx.x; i0 < tot2; i0 += 256 * STAGE_U) {
          float2 v[STAGE_U];
          int dst[STAGE_U];
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u) {
            const int idx = i0 + u * 256;
            dst[u] = -1;
            v[u] = make_float2(0.f, 0.f);
            if (idx < tot2) {
              const int co = idx / per2, e = (idx - co * per2) << 1;
              const int r = e / p.OW, x = e - r * p.OW;
              dst[u] = co * p.PLANEo + r * p.OWp + x;
              v[u] = *reinterpret_cast<const float2*>(dsrc + (long)co * p.OH * p.OW + e);
            }
          }
#pragma unroll
          for (int u = 0; u < STAGE_U; ++u)
            if (dst[u] >= 0) { ldo[dst[u]] = v[u].x; ldo[dst[u] + 1] = v[u].y; }
        }
      } else {
      const int tot = p.Cout * per;
      for (int i0 = threadIdx.x; i0 < tot; i0 += 256 * STAGE_U) {
        float v[STAGE_U];
        int dst[STAGE_U];
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
          const int idx = i0 + u * 256;
          dst[u] = -1;
          v[u] = 0.f;
          if (idx < tot) {
            const int co = idx / per, e = idx - co * per;
            const int r = e / p.OW, x = e - r * p.OW;
            dst[u] = co * p.PLANEo + r * p.OWp + x;
            v[u] = dsrc[(long)co * p.OH * p.OW + e];
          }
        }
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u)
          if (dst[u] >= 0) ldo[dst[u]] = v[u];
      }
      }
    }
    __syncthreads();
    }
    // bias partial sums over this tile
    if (!BR && bpart < nparts && bco < p.Cout) {
      const float* pl = ldo + bco * p.PLANEo;
      const int n = rows * p.OWp;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four reads in flight
      int i = bpart;
      for (; i + 3 * nparts < n; i += 4 * nparts) { s0 += pl[i]; s1 += pl[i + nparts]; s2 += pl[i + 2 * nparts]; s3 += pl[i + 3 * nparts]; }
      for (; i < n; i += nparts) s0 += pl[i];
      dbacc += (s0 + s1) + (s2 + s3);
    }
    // implicit GEMM over the tile's pixels, 4 pixels per MFMA step
    // RS: the tile's rows*c4n steps in four contiguous runs, one per wave.  EX: every instantiated
    // k-tile is real (or its wave would only idle), so the step is one branch-free block and the next
    // step's operands are read from LDS while this step's MFMAs run.
    {
      const int nsteps = rows * c4n;
      const int s0 = RS ? (nsteps * w) >> 2 : 0, s1 = RS ? (nsteps * (w + 1)) >> 2 : nsteps;
      int r = s0 / c4n, c4 = s0 - r * c4n;
      float a[MT], bv[KTW];
      if (s0 < s1) {
        const int ao = r * p.OWp + kk + 4 * c4, bo = r * p.S * WP + 4 * c4 * p.S;
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = ldo[(m * 16 + j) * p.PLANEo + ao];
#pragma unroll
        for (int q = 0; q < KTW; ++q) bv[q] = lds[koff[q] + bo];
      }
      for (int s = s0; s < s1; ++s) {
        if (++c4 == c4n) { c4 = 0; ++r; }
        const bool more = s + 1 < s1;                       // last step: re-read valid operands
        const int rn = more ? r : 0, cn = more ? c4 : 0;
        const int ao = rn * p.OWp + kk + 4 * cn, bo = rn * p.S * WP + 4 * cn * p.S;
        float an[MT], bn[KTW];
#pragma unroll
        for (int m = 0; m < MT; ++m) an[m] = ldo[(m * 16 + j) * p.PLANEo + ao];
#pragma unroll
        for (int q = 0; q < KTW; ++q) bn[q] = lds[koff[q] + bo];
        if (EX) __builtin_amdgcn_sched_barrier(0);          // keep the reads ahead of the MFMAs
#pragma unroll
        for (int q = 0; q < KTW; ++q)
          if (EX || (RS ? q : w + 4 * q) < p.nkt) {        // wave-uniform
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bv[q], acc[m][q], 0, 0, 0);
          }
#pragma unroll
        for (int m = 0; m < MT; ++m) a[m] = an[m];
#pragma unroll
        for (int q = 0; q < KTW; ++q) bv[q] = bn[q];
      }
    }
  }
  // write this workgroup's partials: D row (co) = 4*(lane>>4)+reg, col (k) = lane&15
  float* sl = p.slab + (long)blockIdx.x * ((long)p.Cout * p.K + p.Cout);
  if (RS) {
    float4* red4 = reinterpret_cast<float4*>(lds);    // [4 waves][KTW][64 lanes], one 16-channel block at a time
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      __syncthreads();
#pragma unroll
      for (int q = 0; q < KTW; ++q)
        red4[(w * KTW + q) * 64 + lane] = make_float4(acc[m][q][0], acc[m][q][1], acc[m][q][2], acc[m][q][3]);
      __syncthreads();
      for (int e = threadIdx.x; e < KTW * 64; e += 256) {
        float4 s = red4[e];
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) {
          const float4 t = red4[ww * KTW * 64 + e];
          s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int k = (e >> 6) * 16 + (e & 15), co = m * 16 + 4 * ((e & 63) >> 4);
        if (k < p.K) {
          if (co < p.Cout) sl[(long)co * p.K + k] = s.x;
          if (co + 1 < p.Cout) sl[(long)(co + 1) * p.K + k] = s.y;
          if (co + 2 < p.Cout) sl[(long)(co + 2) * p.K + k] = s.z;
          if (co + 3 < p.Cout) sl[(long)(co + 3) * p.K + k] = s.w;
        }
      }
    }
  } else
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < KTW; ++q) {
      const int k = (w + 4 * q) * 16 + j;
      if (w + 4 * q < p.nkt && k < p.K) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int co = m * 16 + 4 * kk + rr;
          if (co < p.Cout) sl[(long)co * p.K + k] = acc[m][q][rr];
        }
      }
    }
  __syncthreads();
  float* red = lds;   // reuse: [nparts][Cm]
  if (BR) {           // slot index = channel-major: a channel's partial sums are one contiguous run
#pragma unroll
    for (int u = 0; u < ND; ++u) red[threadIdx.x + u * 256] = dbs[u];
    __syncthreads();
    if (threadIdx.x < p.Cout) {
      const int per = p.TPH * (p.OW / VD);
      float s = 0.f;
      for (int i = 0; i < per; ++i) s += red[threadIdx.x * per + i];
      sl[(long)p.Cout * p.K + threadIdx.x] = s;
    }
    return;
  }
  if (bpart < nparts) red[bpart * Cm + bco] = dbacc;
  __syncthreads();
  if (threadIdx.x < p.Cout) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += red[q * Cm + threadIdx.x];
    sl[(long)p.Cout * p.K + threadIdx.x] = s;
  }
}

// Backward-data for the unpadded ks = 2*S layers with ALL S*S output-parity classes fused into one
// launch: a workgroup stages one sample's dOut (Cout x OH x OW, a contiguous run in HBM) ONCE into
// a zero-haloed LDS image, then walks the classes; every class is a (ks/S)^2-tap stride-1
// correlation (class fragments + tables in `cls`).  Software pipelined like igemm_run_kernel:
// the next sample's dOut is in flight during the MFMA phase; class weight fragments live in LDS.
constexpr int PF_B = 3;              // float4 prefetch registers per thread (256*3*4 = 3072 dOut floats)
constexpr int MAX_CLS = 4;

struct BwdClass { int frag_off, nsteps, PH, PW, oy_add, ox_add; };
struct BwdFusedP {
  const float* dout; float* din; const float* mask; const float* wfrag;
  int Cout, OH, OW, Cin, H, W, S, B;
  int WP, PLANE, nfrag, ncls, nb, c4n, off0, step_a, step_b, step_c;
  BwdClass cls[MAX_CLS];
};

template <int MT>
__global__ __launch_bounds__(256) void bwd_fused_kernel(BwdFusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ ldsA = lds;
  float* __restrict__ img = lds + p.nfrag;
  float* __restrict__ outb = img + p.Cout * p.PLANE + 64;      // dX of the sample, [Cin][H*W], flushed coalesced
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.WP, PLANE = p.PLANE;
  const int HW = p.H * p.W;
  for (int i = tid; i < p.nfrag; i += 256) ldsA[i] = p.wfrag[i];
  for (int i = tid; i < p.Cout * PLANE + 64; i += 256) img[i] = 0.f;       // halo stays 0 forever
  // prefetch map: the sample is Cout*OH*OW contiguous floats; element e -> haloed image position
  const int nel = p.Cout * p.OH * p.OW;
  int dst[PF_B][4];
#pragma unroll
  for (int u = 0; u < PF_B; ++u)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int e = ((tid + u * 256) << 2) + c;
      dst[u][c] = -1;
      if (e < nel) {
        const int co = e / (p.OH * p.OW), rem = e - co * p.OH * p.OW;
        const int y = rem / p.OW, x = rem - y * p.OW;
        dst[u][c] = co * PLANE + (y + 1) * WP + x + 1;
      }
    }
  const bool vec_ok = (nel % 4 == 0);
  float4 pf[PF_B];
  auto issue = [&](long b) {
    const float* __restrict__ src = p.dout + b * nel;
#pragma unroll
    for (int u = 0; u < PF_B; ++u) {
      const int e = (tid + u * 256) << 2;
      pf[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (vec_ok && e + 3 < nel) pf[u] = *reinterpret_cast<const float4*>(src + e);
      else {
        if (e + 0 < nel) pf[u].x = src[e + 0];
        if (e + 1 < nel) pf[u].y = src[e + 1];
        if (e + 2 < nel) pf[u].z = src[e + 2];
        if (e + 3 < nel) pf[u].w = src[e + 3];
      }
    }
  };
  long b = blockIdx.x;
  if (b < p.B) issue(b);
  for (; b < p.B; b += gridDim.x) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < PF_B; ++u) {
      if (dst[u][0] >= 0) img[dst[u][0]] = pf[u].x;
      if (dst[u][1] >= 0) img[dst[u][1]] = pf[u].y;
      if (dst[u][2] >= 0) img[dst[u][2]] = pf[u].z;
      if (dst[u][3] >= 0) img[dst[u][3]] = pf[u].w;
    }
    __syncthreads();
    if (b + gridDim.x < p.B) issue(b + gridDim.x);
    for (int c = 0; c < p.ncls; ++c) {
      const BwdClass& k = p.cls[c];
      const int NP = k.PH * k.PW;
      const int npairs = (NP + 31) >> 5;
      const float* __restrict__ la = ldsA + k.frag_off + lane;
      for (int pr = w; pr < npairs; pr += 4) {
        const int idx0 = pr * 32 + j, idx1 = idx0 + 16;
        const bool ok0 = idx0 < NP, ok1 = idx1 < NP;
        const int i0 = ok0 ? idx0 : 0, i1 = ok1 ? idx1 : 0;
        const int r0 = i0 / k.PW, c0 = i0 - r0 * k.PW;
        const int r1 = i1 / k.PW, c1 = i1 - r1 * k.PW;
        const float* __restrict__ l0 = img + r0 * WP + c0 + g * PLANE;
        const float* __restrict__ l1 = img + r1 * WP + c1 + g * PLANE;
        f32x4 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        int off = p.off0, bi = 0, ci = 0;
        for (int s0 = 0; s0 < k.nsteps; s0 += CH) {       // nsteps % CH == 0 on this path
          float bv0[CH], bv1[CH], av[CH * MT];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            bv0[u] = l0[off];
            bv1[u] = l1[off];
#pragma unroll
            for (int m = 0; m < MT; ++m) av[u * MT + m] = la[((s0 + u) * MT + m) * 64];
            ++ci;
            const bool w1 = (ci == p.c4n);
            ci = w1 ? 0 : ci;
            bi += w1 ? 1 : 0;
            const bool w2 = (bi == p.nb);
            bi = w2 ? 0 : bi;
            off += p.step_c + (w1 ? p.step_b - p.c4n * p.step_c : 0) + (w2 ? p.step_a - p.nb * p.step_b : 0);
          }
#pragma unroll
          for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
              acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv0[u], acc[m][0], 0, 0, 0);
              acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u * MT + m], bv1[u], acc[m][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (!(nt ? ok1 : ok0)) continue;
          const int r = nt ? r1 : r0, cc = nt ? c1 : c0;
          const int pix = (r * p.S + k.oy_add) * p.W + cc * p.S + k.ox_add;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int ch = m * 16 + 4 * g + rr;
              if (ch < p.Cin) outb[ch * HW + pix] = acc[m][nt][rr];
            }
        }
      }
    }
    __syncthreads();                                            // every class has landed in outb
    {  // coalesced flush: 16 B per lane, ReLU-derivative mask of the layer below fused in
      const int n4 = (p.Cin * HW) >> 2;
      float4* __restrict__ o4 = reinterpret_cast<float4*>(p.din + b * (long)p.Cin * HW);
      const float4* __restrict__ m4 = p.mask ? reinterpret_cast<const float4*>(p.mask + b * (long)p.Cin * HW) : nullptr;
      for (int i = tid; i < n4; i += 256) {
        float4 v = *reinterpret_cast<const float4*>(outb + (i << 2));
        if (m4) {
          const float4 mk = m4[i];
          if (!(mk.x > 0.f)) v.x = 0.f;
          if (!(mk.y > 0.f)) v.y = 0.f;
          if (!(mk.z > 0.f)) v.z = 0.f;
          if (!(mk.w > 0.f)) v.w = 0.f;
        }
        o4[i] = v;
      }
    }
  }
}

// "Streaming" backward-data kernel for the 4x4 / stride 2 layer with Cout = 32, Cin <= 16 (A3CModel
// conv2) at large batch.  Same skeleton as conv_stream_kernel: persistent 8-wave workgroups, the
// sample's dOut (haloed) in LDS, the next sample's dOut and this sample's ReLU mask in flight in
// registers, dX assembled in LDS and flushed with coalesced float4 stores.  Wave (class, half): one
// of the 4 output-parity classes (its 32 A fragments live in registers for the whole kernel) and
// every other 16-pixel tile of that class's 10 x 10 pixel grid -- 7 tiles per SIMD and sample.
constexpr int BS_NT = 512, BS_PD = 6, BS_PM = 4;
struct BstreamP {
  const float* dout; float* din; const float* mask; const float* wfrag;
  int Cout, OH, OW, Cin, H, W, B;
  int WP, PLANE, off0, step_a, step_b, step_c;
  BwdClass cls[4];
};
#define BS_LDD(var, u, src) var = *reinterpret_cast<const float4*>((src) + (min(tid + (u) * BS_NT, nel4 - 1) << 2));
#define BS_LDM(var, u, src) var = *reinterpret_cast<const float4*>((src) + (min(tid + (u) * BS_NT, n4 - 1) << 2));
// where the four floats of this thread's u-th float4 of dOut go in the haloed LDS image: the same for every sample, so
// the (channel, row, column) arithmetic -- five integer divisions per float4 -- is done ONCE per launch (BS_OFF), two
// 16-bit float offsets per register; the per-sample staging is 24 plain ds_write_b32
#define BS_OFF(u)                                                                               \
  {                                                                                             \
    const int e_ = min(tid + (u) * BS_NT, nel4 - 1) << 2;                                        \
    const int co_ = e_ / ohw, rem_ = e_ - co_ * ohw;                                            \
    unsigned int o_[4];                                                                         \
    _Pragma("unroll") for (int c_ = 0; c_ < 4; ++c_) {                                          \
      int cc_ = co_, r_ = rem_ + c_;                                                            \
      if (r_ >= ohw) { r_ -= ohw; ++cc_; }                                                      \
      const int y_ = r_ / p.OW, x_ = r_ - y_ * p.OW;                                            \
      o_[c_] = (unsigned int)(cc_ * PLANE + (y_ + 1) * WP + x_ + 1);                            \
    }                                                                                           \
    soff[2 * (u)] = o_[0] | (o_[1] << 16);                                                      \
    soff[2 * (u) + 1] = o_[2] | (o_[3] << 16);                                                  \
  }
#define BS_STD(var, u)                                                                          \
  {                                                                                             \
    img[soff[2 * (u)] & 0xffffu] = var.x; img[soff[2 * (u)] >> 16] = var.y;                     \
    img[soff[2 * (u) + 1] & 0xffffu] = var.z; img[soff[2 * (u) + 1] >> 16] = var.w;             \
  }

template <bool MASK>
__global__ __launch_bounds__(BS_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_stream_kernel(BstreamP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ img = lds;                              // dOut of the sample with a one-pixel zero halo
  float* __restrict__ outb = lds + p.Cout * p.PLANE + 64;     // dX of the sample, [Cin][H*W]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.WP, PLANE = p.PLANE, HW = p.H * p.W;
  const int ohw = p.OH * p.OW, nel4 = (p.Cout * ohw) >> 2, n4 = (p.Cin * HW) >> 2;
  for (int i = tid; i < p.Cout * PLANE + 64; i += BS_NT) img[i] = 0.f;       // halo stays 0 forever
  const BwdClass k = p.cls[w & 3];
  const int half = w >> 2;
  float af[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) af[s] = p.wfrag[k.frag_off + s * 64 + lane];
  const int NP = k.PH * k.PW, ntile = (NP + 15) >> 4;
  float4 d0 = {}, d1 = {}, d2 = {}, d3 = {}, d4 = {}, d5 = {}, m0 = {}, m1 = {}, m2 = {}, m3 = {};
  unsigned int soff[12];
  BS_OFF(0) BS_OFF(1) BS_OFF(2) BS_OFF(3) BS_OFF(4) BS_OFF(5)
  long b = blockIdx.x;
  if (b >= p.B) return;
  {
    const float* __restrict__ src = p.dout + b * (long)p.Cout * ohw;
    BS_LDD(d0, 0, src) BS_LDD(d1, 1, src) BS_LDD(d2, 2, src) BS_LDD(d3, 3, src) BS_LDD(d4, 4, src) BS_LDD(d5, 5, src)
  }
  for (; b < p.B; b += gridDim.x) {
    const long nb = (b + gridDim.x < p.B) ? b + gridDim.x : b;          // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.dout + nb * (long)p.Cout * ohw;
    const float* __restrict__ msrc = p.mask + b * (long)p.Cin * HW;
    float* __restrict__ dst = p.din + b * (long)p.Cin * HW;
    BS_STD(d0, 0) BS_STD(d1, 1) BS_STD(d2, 2) BS_STD(d3, 3) BS_STD(d4, 4) BS_STD(d5, 5)
    __syncthreads();
    /* two tiles of this wave at a time: two INDEPENDENT accumulator chains sharing the A fragments (a single chain issues */ \
    /* one MFMA per 40-cycle dependent latency instead of one per 32); the second tile may not exist (wave-uniform)        */
#define BS_PAIR(TA)                                                                                         \
    {                                                                                                       \
      const int tA = (TA) * 2 + half, tB = tA + 2;                                                          \
      if (tA < ntile) {                                                                                     \
        const bool two = tB < ntile;                                                                        \
        const int idxA = tA * 16 + j, idxB = (two ? tB : tA) * 16 + j;                                      \
        const bool okA = idxA < NP, okB = two && idxB < NP;                                                 \
        const int iA = okA ? idxA : 0, iB = okB ? idxB : 0;                                                 \
        const int rA = iA / k.PW, cA = iA - rA * k.PW, rB = iB / k.PW, cB = iB - rB * k.PW;                 \
        const float* __restrict__ lA = img + rA * WP + cA + g * PLANE + p.off0;                             \
        const float* __restrict__ lB = img + rB * WP + cB + g * PLANE + p.off0;                             \
        f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accB = accA;                                              \
        _Pragma("unroll") for (int a = 0; a < 2; ++a) {                                                     \
          float bA[16], bB[16];                                                                             \
          _Pragma("unroll") for (int bb = 0; bb < 2; ++bb)                                                  \
            _Pragma("unroll") for (int c4 = 0; c4 < 8; ++c4) {                                              \
              bA[bb * 8 + c4] = lA[a * p.step_a + bb * p.step_b + c4 * p.step_c];                           \
              bB[bb * 8 + c4] = lB[a * p.step_a + bb * p.step_b + c4 * p.step_c];                           \
            }                                                                                               \
          if (two) {                                                                                        \
            _Pragma("unroll") for (int s = 0; s < 16; ++s) {                                                \
              accA = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a * 16 + s], bA[s], accA, 0, 0, 0);            \
              accB = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a * 16 + s], bB[s], accB, 0, 0, 0);            \
            }                                                                                               \
          } else {                                                                                          \
            _Pragma("unroll") for (int s = 0; s < 16; ++s)                                                  \
              accA = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a * 16 + s], bA[s], accA, 0, 0, 0);            \
          }                                                                                                 \
        }                                                                                                   \
        if (okA) {                                                                                          \
          const int pix = (rA * 2 + k.oy_add) * p.W + cA * 2 + k.ox_add;                                    \
          _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                                  \
            if (4 * g + rr < p.Cin) outb[(4 * g + rr) * HW + pix] = accA[rr];                               \
        }                                                                                                   \
        if (okB) {                                                                                          \
          const int pix = (rB * 2 + k.oy_add) * p.W + cB * 2 + k.ox_add;                                    \
          _Pragma("unroll") for (int rr = 0; rr < 4; ++rr)                                                  \
            if (4 * g + rr < p.Cin) outb[(4 * g + rr) * HW + pix] = accB[rr];                               \
        }                                                                                                   \
      }                                                                                                     \
    }
    if (MASK) { BS_LDM(m0, 0, msrc) BS_LDM(m1, 1, msrc) BS_LDM(m2, 2, msrc) BS_LDM(m3, 3, msrc) }
    BS_LDD(d0, 0, nsrc) BS_LDD(d1, 1, nsrc) BS_LDD(d2, 2, nsrc)
    BS_PAIR(0)                                                  // tiles half, half + 2
    BS_LDD(d3, 3, nsrc) BS_LDD(d4, 4, nsrc) BS_LDD(d5, 5, nsrc)
    for (int tp = 2; tp * 2 + half < ntile; tp += 2) BS_PAIR(tp) // tiles half + 4, half + 6, ...
    __syncthreads();                                            // every class has landed in outb
#define BS_FLUSH(mv, u)                                                                         \
    {                                                                                           \
      const int i_ = min(tid + (u) * BS_NT, n4 - 1) << 2;                                        \
      float4 v_ = *reinterpret_cast<const float4*>(outb + i_);                                  \
      if (MASK) {                                                                               \
        if (!(mv.x > 0.f)) v_.x = 0.f;                                                          \
        if (!(mv.y > 0.f)) v_.y = 0.f;                                                          \
        if (!(mv.z > 0.f)) v_.z = 0.f;                                                          \
        if (!(mv.w > 0.f)) v_.w = 0.f;                                                          \
      }                                                                                         \
      *reinterpret_cast<float4*>(dst + i_) = v_;                                                \
    }
    BS_FLUSH(m0, 0) BS_FLUSH(m1, 1) BS_FLUSH(m2, 2) BS_FLUSH(m3, 3)
  }
}

// bwd_stream_kernel, second form (round 6).  Two things the first form pays for per sample, measured: (1) the flush of the
// sample's dX (25.6 KB: LDS read, mask, global store) sits between two matrix phases with the matrix pipe idle, and the loop
// top's wait for the next sample's dOut also waits for those fresh stores (loads and stores retire through one in-order
// counter); (2) the ReLU mask is the fp32 activation itself, 25.6 KB per sample read for one bit per element (42 % of the
// kernel's HBM bytes).  Here the dX image is DOUBLE BUFFERED in LDS: sample b's classes land in image b & 1 while image
// (b - 1) & 1 is flushed at the HEAD of the iteration -- stores first, then the matrix phase, then the next sample's dOut
// loads, so that no wait of the iteration sits behind a fresh store -- and the mask can come as ONE BIT per activation
// (MODE 2, "lane masks"): bit e & 7 of byte e >> 3 of the sample's row = (act[e] > 0) in flat (c, y, x) order, so the four
// bits of a lane's float4 are one nibble and a flush unit needs ONE byte load per lane (the ring kernel writes the bits
// beside its a1 stash: 800 B per sample instead of 25.6 KB).  The bytes of sample b are fetched during b's own matrix
// phase (first version: at the flush, four loads whose latency nothing covered -- 0.73 -> 0.89 ms).  Same sums in the same
// order as the first form: bit-identical dX.
// MODE 0: no mask, 1: float mask, 2: lane masks.
static unsigned long long* g_bs2_dbg = nullptr;      // a2c_debug_bwd_stream_timing
struct Bstream2P {
  BstreamP s;
  const unsigned long long* lmask;      // MODE 2: (B, lmw) 64-bit words
  int lmw;                              // words per sample = Cin * H * W / 64
  int order;                            // third form: 0 = the two waves of a SIMD out of step, 1 = side work first, 2 = first tile pair first
  unsigned long long* dbg;              // second form: phase stamps of workgroup 0 (a2c_debug_bwd_stream_timing), [wave][8] shader clocks
};
template <int MODE, bool STAMP = false>
__global__ __launch_bounds__(BS_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_stream2_kernel(Bstream2P pp) {
  const BstreamP& p = pp.s;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ img = lds;                              // dOut of the sample with a one-pixel zero halo
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.WP, PLANE = p.PLANE, HW = p.H * p.W;
  const int obs = p.Cin * HW;                                 // floats per dX image
  float* __restrict__ outb0 = lds + p.Cout * p.PLANE + 64;    // two dX images, [Cin][H*W] each
  const int ohw = p.OH * p.OW, nel4 = (p.Cout * ohw) >> 2, n4 = obs >> 2;
  for (int i = tid; i < p.Cout * PLANE + 64; i += BS_NT) img[i] = 0.f;       // halo stays 0 forever
  const BwdClass k = p.cls[w & 3];
  const int half = w >> 2;
  float af[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) af[s] = p.wfrag[k.frag_off + s * 64 + lane];
  const int NP = k.PH * k.PW, ntile = (NP + 15) >> 4;
  float4 d0 = {}, d1 = {}, d2 = {}, d3 = {}, d4 = {}, d5 = {}, m0 = {}, m1 = {}, m2 = {}, m3 = {};
  unsigned int soff[12];
  BS_OFF(0) BS_OFF(1) BS_OFF(2) BS_OFF(3) BS_OFF(4) BS_OFF(5)
  long b = blockIdx.x;
  if (b >= p.B) return;
  {
    const float* __restrict__ src = p.dout + b * (long)p.Cout * ohw;
    BS_LDD(d0, 0, src) BS_LDD(d1, 1, src) BS_LDD(d2, 2, src) BS_LDD(d3, 3, src) BS_LDD(d4, 4, src) BS_LDD(d5, 5, src)
  }
  // flush unit u of sample pb_ out of image ob_: float4 number tid + u * BS_NT
#define BS2_FLUSH(mv, u, pb_, ob_)                                                                          \
  {                                                                                                         \
    float* __restrict__ dst_ = p.din + (pb_) * (long)obs;                                                   \
    if (MODE == 2) {                                                                                        \
      const int q_ = tid + (u) * BS_NT;                  /* this lane's float4; its four bits were fetched  */   \
      if (q_ < n4) {                                     /* one iteration ago (BS2_LDB): byte u of lmb      */   \
        const unsigned int nb_ = (lmb >> (8 * (u) + 4 * (q_ & 1))) & 0xfu;                                  \
        const int i_ = q_ << 2;                                                                             \
        float4 v_ = *reinterpret_cast<const float4*>((ob_) + i_);                                           \
        if (!(nb_ & 1u)) v_.x = 0.f;                                                                        \
        if (!(nb_ & 2u)) v_.y = 0.f;                                                                        \
        if (!(nb_ & 4u)) v_.z = 0.f;                                                                        \
        if (!(nb_ & 8u)) v_.w = 0.f;                                                                        \
        *reinterpret_cast<float4*>(dst_ + i_) = v_;                                                         \
      }                                                                                                     \
    } else {                                                                                                \
      const int i_ = min(tid + (u) * BS_NT, n4 - 1) << 2;                                                    \
      float4 v_ = *reinterpret_cast<const float4*>((ob_) + i_);                                             \
      if (MODE == 1) {                                                                                      \
        if (!(mv.x > 0.f)) v_.x = 0.f;                                                                      \
        if (!(mv.y > 0.f)) v_.y = 0.f;                                                                      \
        if (!(mv.z > 0.f)) v_.z = 0.f;                                                                      \
        if (!(mv.w > 0.f)) v_.w = 0.f;                                                                      \
      }                                                                                                     \
      *reinterpret_cast<float4*>(dst_ + i_) = v_;                                                           \
    }                                                                                                       \
  }
  long pb = -1;
  int cur = 0;
  unsigned int lmb = 0;                                         // MODE 2: the mask bits of sample pb for this thread's four flush units
  // phase stamps (tools/bwd_stream_timing.py): per wave of workgroup 0, shader clocks summed over its samples
  const bool stamp = STAMP && pp.dbg != nullptr && blockIdx.x == 0;       // (a template flag: the sums cost 18 VGPRs the kernel does not have)
  unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = stamp ? (unsigned long long)clock64() : 0ull;
#define BS2_TS(i) do { if (stamp) { const unsigned long long n_ = (unsigned long long)clock64(); tsum[i] += n_ - tprev; tprev = n_; } } while (0)
  for (; b < p.B; b += gridDim.x) {
    const long nb = (b + gridDim.x < p.B) ? b + gridDim.x : b;          // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.dout + nb * (long)p.Cout * ohw;
    const float* __restrict__ msrc = p.mask + b * (long)obs;
    float* __restrict__ outb = outb0 + cur * obs;
    BS_STD(d0, 0) BS_STD(d1, 1) BS_STD(d2, 2) BS_STD(d3, 3) BS_STD(d4, 4) BS_STD(d5, 5)
    BS2_TS(0);                                                  // wait for the sample's dOut + staging
    __syncthreads();
    BS2_TS(1);                                                  // barrier 1
    if (pb >= 0) {                                              // the previous sample's dX: stores first
      const float* __restrict__ ob = outb0 + (cur ^ 1) * obs;
      BS2_FLUSH(m0, 0, pb, ob) BS2_FLUSH(m1, 1, pb, ob) BS2_FLUSH(m2, 2, pb, ob) BS2_FLUSH(m3, 3, pb, ob)
    }
    BS2_TS(2);                                                  // flush of the previous sample
    BS_PAIR(0)                                                  // tiles half, half + 2
    BS2_TS(3);                                                  // first tile pair
    if (MODE == 1) { BS_LDM(m0, 0, msrc) BS_LDM(m1, 1, msrc) BS_LDM(m2, 2, msrc) BS_LDM(m3, 3, msrc) }
    unsigned int lb0 = 0, lb1 = 0, lb2 = 0, lb3 = 0;
    if (MODE == 2) {           // THIS sample's mask bits (its flush is the next iteration's first act): one byte per flush unit
      const unsigned char* __restrict__ lmp = reinterpret_cast<const unsigned char*>(pp.lmask) + b * (long)pp.lmw * 8;
      const int nby = n4 >> 1;
      lb0 = lmp[min((tid + 0 * BS_NT) >> 1, nby - 1)]; lb1 = lmp[min((tid + 1 * BS_NT) >> 1, nby - 1)];
      lb2 = lmp[min((tid + 2 * BS_NT) >> 1, nby - 1)]; lb3 = lmp[min((tid + 3 * BS_NT) >> 1, nby - 1)];
    }
    BS_LDD(d0, 0, nsrc) BS_LDD(d1, 1, nsrc) BS_LDD(d2, 2, nsrc) BS_LDD(d3, 3, nsrc) BS_LDD(d4, 4, nsrc) BS_LDD(d5, 5, nsrc)
    BS2_TS(4);                                                  // issue of the next sample's loads
    for (int tp = 2; tp * 2 + half < ntile; tp += 2) BS_PAIR(tp) // tiles half + 4, half + 6, ...
    if (MODE == 2) lmb = lb0 | (lb1 << 8) | (lb2 << 16) | (lb3 << 24);
    BS2_TS(5);                                                  // remaining tile pairs
    __syncthreads();                                            // every class has landed in this sample's image
    BS2_TS(6);                                                  // barrier 2
    pb = b;
    cur ^= 1;
    if (stamp) tsum[7] += 1;
  }
  {
    const float* __restrict__ ob = outb0 + (cur ^ 1) * obs;
    BS2_FLUSH(m0, 0, pb, ob) BS2_FLUSH(m1, 1, pb, ob) BS2_FLUSH(m2, 2, pb, ob) BS2_FLUSH(m3, 3, pb, ob)
  }
  if (stamp && lane == 0)
    for (int i = 0; i < 8; ++i) pp.dbg[w * 8 + i] = tsum[i];
#undef BS2_TS
}

// Third form, OPT-IN (A2C_BWD_STREAM_FORM=3): built after the counters, measured, and slower -- 0.855 ms with the waves out of
// step, 0.869 with the side work first on all waves, 0.913 with the first tile pair first, against 0.74 for the second form
// (same box, alternating runs, tools/ab_bs3.sh).  Kept for the record and for the next look with a phase-stamp build.
// (round 6, after the counters: MFMA pipe 54 % busy, the rest is time in which BOTH waves of a SIMD sit in the same
// non-matrix phase -- staging, flush, the LDS reads at the head of a k block -- because two barriers per sample keep them in
// step).  ONE barrier per sample: the dOut image is double buffered too (sample b + 1 is staged into the other image during
// sample b's interval), and the two waves of a SIMD (class k, halves 0 and 1) run the interval's work in DIFFERENT orders --
// half 0: first tile pair, flush, staging, loads, rest; half 1: flush, staging, loads, then its tiles -- so that one wave's
// stores / LDS writes / address arithmetic run under the other's MFMAs.  Same sums in the same order: bit-identical dX.
template <int MODE>
__global__ __launch_bounds__(BS_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void bwd_stream3_kernel(Bstream2P pp) {
  const BstreamP& p = pp.s;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.WP, PLANE = p.PLANE, HW = p.H * p.W;
  const int obs = p.Cin * HW;                                 // floats per dX image
  const int IMG = p.Cout * p.PLANE + 64;                      // floats per dOut image (one-pixel zero halo)
  float* __restrict__ img0 = lds;                             // two dOut images
  float* __restrict__ outb0 = lds + 2 * IMG;                  // two dX images, [Cin][H*W] each
  const int ohw = p.OH * p.OW, nel4 = (p.Cout * ohw) >> 2, n4 = obs >> 2;
  for (int i = tid; i < 2 * IMG; i += BS_NT) img0[i] = 0.f;   // halos stay 0 forever
  const BwdClass k = p.cls[w & 3];
  const int half = __builtin_amdgcn_readfirstlane(w >> 2);
  float af[32];
#pragma unroll
  for (int s = 0; s < 32; ++s) af[s] = p.wfrag[k.frag_off + s * 64 + lane];
  const int NP = k.PH * k.PW, ntile = (NP + 15) >> 4;
  float4 d0 = {}, d1 = {}, d2 = {}, d3 = {}, d4 = {}, d5 = {}, m0 = {}, m1 = {}, m2 = {}, m3 = {};
  unsigned int soff[12];
  BS_OFF(0) BS_OFF(1) BS_OFF(2) BS_OFF(3) BS_OFF(4) BS_OFF(5)
  long b = blockIdx.x;
  if (b >= p.B) return;
#define BS3_STD(var, u, base)                                                                   \
  {                                                                                             \
    (base)[soff[2 * (u)] & 0xffffu] = var.x; (base)[soff[2 * (u)] >> 16] = var.y;               \
    (base)[soff[2 * (u) + 1] & 0xffffu] = var.z; (base)[soff[2 * (u) + 1] >> 16] = var.w;       \
  }
  __syncthreads();                                            // the zero fill, before anything is staged
  {
    const float* __restrict__ src = p.dout + b * (long)p.Cout * ohw;
    BS_LDD(d0, 0, src) BS_LDD(d1, 1, src) BS_LDD(d2, 2, src) BS_LDD(d3, 3, src) BS_LDD(d4, 4, src) BS_LDD(d5, 5, src)
    BS3_STD(d0, 0, img0) BS3_STD(d1, 1, img0) BS3_STD(d2, 2, img0) BS3_STD(d3, 3, img0) BS3_STD(d4, 4, img0) BS3_STD(d5, 5, img0)
    const long b1 = (b + gridDim.x < p.B) ? b + gridDim.x : b;
    const float* __restrict__ s1 = p.dout + b1 * (long)p.Cout * ohw;
    BS_LDD(d0, 0, s1) BS_LDD(d1, 1, s1) BS_LDD(d2, 2, s1) BS_LDD(d3, 3, s1) BS_LDD(d4, 4, s1) BS_LDD(d5, 5, s1)
  }
  __syncthreads();
  long pb = -1;
  int cur = 0;
  unsigned int lmb = 0;                                       // MODE 2: the mask bits of sample pb for this thread's four flush units
  for (; b < p.B; b += gridDim.x) {
    const long nb2 = (b + 2L * gridDim.x < p.B) ? b + 2L * gridDim.x : b;      // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.dout + nb2 * (long)p.Cout * ohw;
    const float* __restrict__ msrc = p.mask + b * (long)obs;
    const float* __restrict__ img = img0 + cur * IMG;         // this sample's dOut (staged during the previous interval)
    float* __restrict__ imgn = img0 + (cur ^ 1) * IMG;        // the next sample's
    float* __restrict__ outb = outb0 + cur * obs;             // this sample's dX
    const float* __restrict__ ob = outb0 + (cur ^ 1) * obs;   // the previous sample's, flushed in this interval
    unsigned int lb0 = 0, lb1 = 0, lb2 = 0, lb3 = 0;
    // everything of the interval that is not this sample's matrix work
#define BS3_SIDE                                                                                                   \
    {                                                                                                              \
      if (pb >= 0) { BS2_FLUSH(m0, 0, pb, ob) BS2_FLUSH(m1, 1, pb, ob) BS2_FLUSH(m2, 2, pb, ob) BS2_FLUSH(m3, 3, pb, ob) } \
      BS3_STD(d0, 0, imgn) BS3_STD(d1, 1, imgn) BS3_STD(d2, 2, imgn) BS3_STD(d3, 3, imgn) BS3_STD(d4, 4, imgn) BS3_STD(d5, 5, imgn) \
      if (MODE == 1) { BS_LDM(m0, 0, msrc) BS_LDM(m1, 1, msrc) BS_LDM(m2, 2, msrc) BS_LDM(m3, 3, msrc) }           \
      if (MODE == 2) {                                                                                             \
        const unsigned char* __restrict__ lmp = reinterpret_cast<const unsigned char*>(pp.lmask) + b * (long)pp.lmw * 8; \
        const int nby = n4 >> 1;                                                                                   \
        lb0 = lmp[min((tid + 0 * BS_NT) >> 1, nby - 1)]; lb1 = lmp[min((tid + 1 * BS_NT) >> 1, nby - 1)];          \
        lb2 = lmp[min((tid + 2 * BS_NT) >> 1, nby - 1)]; lb3 = lmp[min((tid + 3 * BS_NT) >> 1, nby - 1)];          \
      }                                                                                                            \
      BS_LDD(d0, 0, nsrc) BS_LDD(d1, 1, nsrc) BS_LDD(d2, 2, nsrc) BS_LDD(d3, 3, nsrc) BS_LDD(d4, 4, nsrc) BS_LDD(d5, 5, nsrc) \
    }
    if (pp.order == 2 || (pp.order == 0 && half == 0)) {
      BS_PAIR(0)
      BS3_SIDE
    } else {
      BS3_SIDE
      BS_PAIR(0)
    }
    for (int tp = 2; tp * 2 + half < ntile; tp += 2) BS_PAIR(tp) // tiles half + 4, half + 6, ...
    if (MODE == 2) lmb = lb0 | (lb1 << 8) | (lb2 << 16) | (lb3 << 24);
    __syncthreads();                                            // this sample's dX complete, the next sample's dOut staged
    pb = b;
    cur ^= 1;
  }
#undef BS3_SIDE
  {
    const float* __restrict__ ob = outb0 + (cur ^ 1) * obs;
    BS2_FLUSH(m0, 0, pb, ob) BS2_FLUSH(m1, 1, pb, ob) BS2_FLUSH(m2, 2, pb, ob) BS2_FLUSH(m3, 3, pb, ob)
  }
#undef BS3_STD
}

// ---------------------------------------------------------------------------------------------------------------
// bwd_x6_kernel (round 6): the same layer -- A3CModel conv2's backward-data, 32 -> 16 channels, 4 x 4, stride 2, 9 x 9 ->
// 20 x 20 -- on the BF16 matrix pipe with fp32 results.  bwd_stream2_kernel is bound by the fp32 MFMAs it issues (82 TF, the
// pipe 54 % busy in step-locked phases); here both operands are split into three bf16 pieces (exact) and
// the six piece products with qa + qb <= 2 are issued (the dropped ones are below 2^-24 of |a b|: gemm_x6_kernel's argument),
// 6 v_mfma_f32_16x16x32_bf16 (16 cycles) per 32-deep k step instead of 8 v_mfma_f32_16x16x4_f32 (32 cycles): 0.375 of the
// matrix time.  What makes the layer fit the bf16 instruction:
//   * all four output-parity classes gather the SAME dOut pixels: dX[ci][2cy+ry][2cx+rx] = sum over taps (a, b) and co of
//     dOut[co][cy-a][cx-b] W[co][ci][ry+2a][rx+2b] -- per sample ONE product D[(class, ci)][(cy, cx)] = Wt[(class, ci)][(tap, co)]
//     G[(tap, co)][(cy, cx)] with M = 64, N = 100 (7 tiles of 16), K = 128;
//   * k runs over co inside a tap, so a lane's 8 consecutive k are 8 consecutive channels of ONE dOut pixel: the sample is
//     staged as [piece][co / 8][pixel][8] bf16 (a thread loads the 8 channels of its pixel, splits, three ds_write_b128) and an
//     MFMA operand is ONE ds_read_b128 at a per-lane pixel slot (taps outside the 9 x 9 grid read a zero slot) -- no halo
//     image, no gather instructions;
//   * a wave owns two classes (their weight fragments, 2 x 4 taps x 3 pieces, live in 96 registers for the whole launch) and
//     two 16-pixel tiles: a dOut fragment read feeds up to 12 MFMAs; (tile, class) units are dealt 4 / 4 / 3 / 3 to the four
//     waves of a class pair so that the two waves of a SIMD carry 7 of the 28 units each.
// The rest is bwd_stream2_kernel's skeleton: next sample's dOut in flight in registers during the matrix phase, two dX images
// in LDS, the previous sample flushed (ReLU mask bits, coalesced float4 stores) at the head of the iteration.
constexpr int BX_NT = 512, BX_SLOT0 = 81, BX_GST = 82 * 8, BX_PST = 4 * BX_GST, BX_IMG = 3 * BX_PST;    // bf16 elements
typedef __bf16 bf16x8x __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4x __attribute__((ext_vector_type(4)));
struct BwdX6P {
  const float* dout; float* din; const float* wfrag; const unsigned long long* lmask;
  int lmw, B;
  int frag_off[4];
  // RANK: dOut is not read but formed while the sample is staged -- dOut[e] = (a2[e] > 0) ? sum_n dl[n] Wc[n][e] : 0, the sums of
  // small_n_bwd_data_bits_kernel (n ascending, separate multiply and add): A3CModel's da2 never exists in HBM
  const float* dl; long ldl; int nlog; const float* Wc; const unsigned char* a2b; long a2b_row;
};
__device__ __forceinline__ void bx_split8(const float e[8], u32x4x o[3]) {
  unsigned short pc[3][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 h0 = (__bf16)e[i];
    const float r1 = e[i] - (float)h0;                 // exact
    const __bf16 h1 = (__bf16)r1;
    const float r2 = r1 - (float)h1;                   // exact, at most 8 significant bits
    pc[0][i] = __builtin_bit_cast(unsigned short, h0);
    pc[1][i] = __builtin_bit_cast(unsigned short, h1);
    pc[2][i] = __builtin_bit_cast(unsigned short, (__bf16)r2);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q)
    o[q] = (u32x4x){(unsigned int)pc[q][0] | ((unsigned int)pc[q][1] << 16), (unsigned int)pc[q][2] | ((unsigned int)pc[q][3] << 16),
                    (unsigned int)pc[q][4] | ((unsigned int)pc[q][5] << 16), (unsigned int)pc[q][6] | ((unsigned int)pc[q][7] << 16)};
}
template <int MODE, bool RANK = false>               // MODE 0: no mask, 2: lane masks
__global__ __launch_bounds__(BX_NT) void bwd_x6_kernel(BwdX6P p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsx[];
  unsigned short* __restrict__ img = reinterpret_cast<unsigned short*>(ldsx);          // dOut of the sample, three piece images
  float* __restrict__ outb0 = reinterpret_cast<float*>(ldsx + 2 * BX_IMG);            // two dX images, [16][400] each
  constexpr int HW = 400, OBS = 6400, N4 = 1600, OHW = 81;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int cp = w & 1, mg = w >> 1;                  // class pair (ry = cp; rx = 0, 1) and tile group
  if (tid < 12) *reinterpret_cast<u32x4x*>(img + tid * BX_GST + BX_SLOT0 * 8) = (u32x4x){0u, 0u, 0u, 0u};    // the zero pixel of every plane
  // weight fragments: lane (ci = j, k group g) of class c, tap t: W[co = 8 g + i][ci][ry + 2 a][rx + 2 b], i = 0..7, three pieces
  bf16x8x wf[2][4][3];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int co = 8 * g + i;
        e[i] = p.wfrag[p.frag_off[2 * cp + c] + (t * 8 + (co >> 2)) * 64 + (((co & 3) << 4) | j)];
      }
      u32x4x o[3];
      bx_split8(e, o);
#pragma unroll
      for (int q = 0; q < 3; ++q) wf[c][t][q] = __builtin_bit_cast(bf16x8x, o[q]);
    }
  // this wave's two tiles, which of its classes each carries, and where lane j's pixel of each tile reads / writes
  const int tl0 = mg == 3 ? 5 : 2 * mg, tl1 = tl0 + 1;
  const bool do00 = mg != 3, do01 = true, do10 = true, do11 = mg != 2;          // do<job><class>
  int boff[2][4], pix[2][2];
  bool okp[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int m = (jb ? tl1 : tl0) * 16 + j;
    okp[jb] = m < 100;
    const int cy = m / 10, cx = m - cy * 10;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int oy = cy - (t >> 1), ox = cx - (t & 1);
      const int slot = (okp[jb] && oy >= 0 && oy <= 8 && ox >= 0 && ox <= 8) ? oy * 9 + ox : BX_SLOT0;
      boff[jb][t] = g * BX_GST + slot * 8;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) pix[jb][c] = (2 * cy + cp) * 20 + 2 * cx + c;
  }
  // staging role: thread (pixel s, channel group gg) of the sample's 81 x 4 cells -- the LAST 324 threads: waves 4-7 carry three
  // (tile, class) units against the four of waves 0-3, and the next sample is formed and split in that slack (before the
  // iteration's closing barrier); only the three ds_write_b128 stay at the head of the next iteration
  const bool stg = tid >= BX_NT - 4 * OHW;
  const int st_ = stg ? tid - (BX_NT - 4 * OHW) : 0;
  const int ss = st_ % OHW, gg = st_ / OHW;
  float pre[8];
  u32x4x so[3] = {(u32x4x){0u, 0u, 0u, 0u}, (u32x4x){0u, 0u, 0u, 0u}, (u32x4x){0u, 0u, 0u, 0u}};
  // RANK: this thread's eight elements e = (8 gg + i) * 81 + ss of the composed matrix (all samples), the next sample's dl and
  // the eight mask bytes of its elements
  float gl[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned int mbv[RANK ? 8 : 1];
  const int e0 = 8 * gg * OHW + ss;
  // (the composed matrix transposed in LDS, a float4 of up to four logit rows per element: in registers the 32 values per
  // thread spill, and a spill reload's vmcnt(0) waits for every prefetch load and flush store in flight)
  float4* __restrict__ wcl = reinterpret_cast<float4*>(ldsx + 2 * BX_IMG + 2 * 4 * OBS);
  if (RANK) {
    for (int e = tid; e < 32 * OHW; e += BX_NT)
      wcl[e] = make_float4(p.Wc[e], p.nlog > 1 ? p.Wc[32 * OHW + e] : 0.f, p.nlog > 2 ? p.Wc[2 * 32 * OHW + e] : 0.f,
                           p.nlog > 3 ? p.Wc[3 * 32 * OHW + e] : 0.f);
  }
  long b = blockIdx.x;
  if (b >= p.B) return;
#define BX_LDD(src)                                                                         \
  if (stg) { _Pragma("unroll") for (int i = 0; i < 8; ++i) pre[i] = (src)[(8 * gg + i) * OHW + ss]; }
#define BX_LDR(bn)                                                                          \
  if (stg) {                                                                                \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) gl[n] = n < p.nlog ? p.dl[(bn) * p.ldl + n] : 0.f;      \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) mbv[RANK ? i : 0] = p.a2b[(bn) * p.a2b_row + ((e0 + i * OHW) >> 3)];   \
  }
  if (RANK) { BX_LDR(b) }
  else {
    const float* __restrict__ src = p.dout + b * (long)(32 * OHW);
    BX_LDD(src)
  }
  // the fetched sample -> its three piece vectors (RANK: formed first)
#define BX_FORM()                                                                                           \
  if (stg) {                                                                                                \
    if (RANK) {                                                                                             \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                       \
        const float4 w4 = wcl[e0 + i * OHW];                                                                \
        float a = 0.f;                                                                                      \
        a += gl[0] * w4.x; a += gl[1] * w4.y; a += gl[2] * w4.z; a += gl[3] * w4.w;   /* (columns past nlog add 0 * 0) */ \
        if (!((mbv[RANK ? i : 0] >> ((e0 + i * OHW) & 7)) & 1u)) a = 0.f;                                   \
        pre[i] = a;                                                                                         \
      }                                                                                                     \
    }                                                                                                       \
    bx_split8(pre, so);                                                                                     \
  }
#define BX_FLUSH(u, pb_, ob_)                                                                               \
  {                                                                                                         \
    const int q_ = tid + (u) * BX_NT;                                                                       \
    if (q_ < N4) {                                                                                          \
      float* __restrict__ dst_ = p.din + (pb_) * (long)OBS;                                                 \
      const int i_ = q_ << 2;                                                                               \
      float4 v_ = *reinterpret_cast<const float4*>((ob_) + i_);                                             \
      if (MODE == 2) {                                                                                      \
        const unsigned int nb_ = (lmb >> (8 * (u) + 4 * (q_ & 1))) & 0xfu;                                  \
        if (!(nb_ & 1u)) v_.x = 0.f;                                                                        \
        if (!(nb_ & 2u)) v_.y = 0.f;                                                                        \
        if (!(nb_ & 4u)) v_.z = 0.f;                                                                        \
        if (!(nb_ & 8u)) v_.w = 0.f;                                                                        \
      }                                                                                                     \
      *reinterpret_cast<float4*>(dst_ + i_) = v_;                                                           \
    }                                                                                                       \
  }
  long pb = -1;
  int cur = 0;
  unsigned int lmb = 0;
  __syncthreads();                                            // the zero pixels (RANK: the composed matrix)
  BX_FORM()
  for (; b < p.B; b += gridDim.x) {
    const long nb = (b + gridDim.x < p.B) ? b + gridDim.x : b;          // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.dout + nb * (long)(32 * OHW);
    float* __restrict__ outb = outb0 + cur * OBS;
    if (stg) {
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4x*>(img + q * BX_PST + gg * BX_GST + ss * 8) = so[q];
    }
    __syncthreads();
    if (pb >= 0) {                                              // the previous sample's dX: stores first
      const float* __restrict__ ob = outb0 + (cur ^ 1) * OBS;
      BX_FLUSH(0, pb, ob) BX_FLUSH(1, pb, ob) BX_FLUSH(2, pb, ob) BX_FLUSH(3, pb, ob)
    }
    unsigned int lb0 = 0, lb1 = 0, lb2 = 0, lb3 = 0;
    if (MODE == 2) {           // THIS sample's mask bits (its flush is the next iteration's first act): one byte per flush unit
      const unsigned char* __restrict__ lmp = reinterpret_cast<const unsigned char*>(p.lmask) + b * (long)p.lmw * 8;
      lb0 = lmp[min((tid + 0 * BX_NT) >> 1, N4 / 2 - 1)]; lb1 = lmp[min((tid + 1 * BX_NT) >> 1, N4 / 2 - 1)];
      lb2 = lmp[min((tid + 2 * BX_NT) >> 1, N4 / 2 - 1)]; lb3 = lmp[min((tid + 3 * BX_NT) >> 1, N4 / 2 - 1)];
    }
    if (RANK) { BX_LDR(nb) } else { BX_LDD(nsrc) }
    {
      // a<job><class>: the (0,0) products; s<job><class>: the five small ones (their roundings are relative to a sum 2^-8 of the
      // first; the main chain takes one addition per tap: against fp64 no less accurate than the fp32 kernel)
      f32x4 a00 = (f32x4){0.f, 0.f, 0.f, 0.f}, a01 = a00, a10 = a00, a11 = a00, s00 = a00, s01 = a00, s10 = a00, s11 = a00;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        bf16x8x d0[3], d1[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          d0[q] = *reinterpret_cast<const bf16x8x*>(img + q * BX_PST + boff[0][t]);
          d1[q] = *reinterpret_cast<const bf16x8x*>(img + q * BX_PST + boff[1][t]);
        }
        // rising magnitude: (2,0) (1,1) (0,2) | (1,0) (0,1) | (0,0); the four chains interleaved
#define BX_MM(P, QA, QB)                                                                                               \
        if (do00) P##00 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][t][QA], d0[QB], P##00, 0, 0, 0);                 \
        if (do01) P##01 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][t][QA], d0[QB], P##01, 0, 0, 0);                 \
        if (do10) P##10 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][t][QA], d1[QB], P##10, 0, 0, 0);                 \
        if (do11) P##11 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][t][QA], d1[QB], P##11, 0, 0, 0);
        BX_MM(s, 2, 0) BX_MM(s, 1, 1) BX_MM(s, 0, 2) BX_MM(s, 1, 0) BX_MM(s, 0, 1) BX_MM(a, 0, 0)
#undef BX_MM
      }
      // D: lane holds pixel j (column), channels 4 g + r (rows)
      if (okp[0]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (do00) outb[(4 * g + r) * HW + pix[0][0]] = a00[r] + s00[r];
          if (do01) outb[(4 * g + r) * HW + pix[0][1]] = a01[r] + s01[r];
        }
      }
      if (okp[1]) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (do10) outb[(4 * g + r) * HW + pix[1][0]] = a10[r] + s10[r];
          if (do11) outb[(4 * g + r) * HW + pix[1][1]] = a11[r] + s11[r];
        }
      }
    }
    if (MODE == 2) lmb = lb0 | (lb1 << 8) | (lb2 << 16) | (lb3 << 24);
    BX_FORM()                                                   // the next sample (fetched during the matrix phase above)
    __syncthreads();                                            // every class has landed in this sample's image
    pb = b;
    cur ^= 1;
  }
  {
    const float* __restrict__ ob = outb0 + (cur ^ 1) * OBS;
    BX_FLUSH(0, pb, ob) BX_FLUSH(1, pb, ob) BX_FLUSH(2, pb, ob) BX_FLUSH(3, pb, ob)
  }
#undef BX_FLUSH
#undef BX_LDD
#undef BX_LDR
#undef BX_FORM
}

// the mask bits of an activation tensor (see bwd_stream2_kernel): a lane per float4, a byte per pair of lanes
__global__ __launch_bounds__(256) void lanemask_kernel(const float* __restrict__ act, unsigned char* __restrict__ lm, long n4) {
  for (long q = blockIdx.x * 256L + threadIdx.x; q < n4; q += gridDim.x * 256L) {          // n4 is even: whole lane pairs
    const float4 v = reinterpret_cast<const float4*>(act)[q];
    const unsigned int nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
    const unsigned int hi = __shfl_xor(nib, 1);
    if (!(q & 1)) lm[q >> 1] = (unsigned char)(nib | (hi << 4));
  }
}

// Generic backward-data, any (ks, S <= 2, pad): a workgroup owns a BAND of TY rows of dX (all
// columns, all input channels).  It stages the band's dOut rows (with zero halo) once, runs every
// output-parity class as a stride-1 correlation into an LDS copy of the dX band, then flushes
// the band with coalesced 16 B stores (ReLU-derivative mask fused, 16 B mask loads).  Replaces S*S
// launches that each re-staged dOut and scattered 4-byte writes at stride S.
struct BandClass { int ry, rx, na, nb, frag_off, nsteps, nchunks, p0, PWc; };
struct BwdBandP {
  StageP st;                       // dOut image staging (general path, halo)
  float* din; const float* mask; const float* wfrag;
  int Cin, H, W, S, P, ks, TY, bands, B, ncls, c4n, ox_lo, out_floats;
  int mgroups, mt_total;           // bwd_band_kernel<MT>: the input channels in mgroups groups of 16 * MT, one workgroup each (1 = all)
  BandClass cls[MAX_CLS];
};

__device__ __forceinline__ int fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

template <int MT, int NTU>
__global__ __launch_bounds__(256) void bwd_band_kernel(BwdBandP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ outb = lds;                         // [Cin][TY*W]
  float* __restrict__ img = lds + p.out_floats;           // [Cout][PLANE]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  // channel groups (whole-sample bands only): a 32-channel 21 x 21 sample is 56 KB of dX + 40 KB of dOut = ONE workgroup per
  // CU running stage -> matrix -> flush in sequence; as two 16-channel workgroups of 68 KB each two are resident and overlap
  const long total = (long)p.B * p.bands * p.mgroups;
  stage_zero(p.st, img);
  for (long tile = blockIdx.x; tile < total; tile += gridDim.x) {
    const int mg = (int)(tile % p.mgroups);
    const long tb = tile / p.mgroups;
    const long b = tb / p.bands;
    const int Y0 = (int)(tb - b * p.bands) * p.TY;
    const int rowsY = min(p.TY, p.H - Y0);
    const int oy_lo = fdiv(Y0 + p.P - (p.ks - 1), p.S);
    const int ch0 = mg * 16 * MT;                          // first input channel of this workgroup
    __syncthreads();
    stage_tile(p.st, img, b, oy_lo);
    __syncthreads();
    // work units = (class, pair of 16-pixel tiles), dealt to the four waves ACROSS the classes: with a loop per class a small
    // image (11 x 11: 2 + 1 + 1 + 1 pairs) kept wave 0 busy four times in a row and the others idle
    int np_c[MAX_CLS], units = 0;
#pragma unroll
    for (int c = 0; c < MAX_CLS; ++c) {
      np_c[c] = 0;
      if (c < p.ncls) {
        const BandClass& k = p.cls[c];
        int q_lo = (Y0 + p.P - k.ry + p.S - 1) / p.S;
        if (q_lo < 0) q_lo = 0;
        const int y_first = p.S * q_lo + k.ry - p.P;
        const int rows_c = y_first < Y0 + rowsY ? (Y0 + rowsY - 1 - y_first) / p.S + 1 : 0;
        np_c[c] = (rows_c * k.PWc + 16 * NTU - 1) / (16 * NTU);
      }
      units += np_c[c];
    }
    for (int un = __builtin_amdgcn_readfirstlane(w); un < units; un += 4) {      // (scalar: the class parameters are s_loads)
      int c = 0, pr = un;
#pragma unroll
      for (int cc = 0; cc < MAX_CLS - 1; ++cc)
        if (c == cc && pr >= np_c[cc]) { pr -= np_c[cc]; c = cc + 1; }
      const BandClass& k = p.cls[c];
      // class rows in this band: y = S*q + ry - P in [Y0, Y0 + rowsY)
      int q_lo = (Y0 + p.P - k.ry + p.S - 1) / p.S;
      if (q_lo < 0) q_lo = 0;
      const int y_first = p.S * q_lo + k.ry - p.P;
      const int rows_c = y_first < Y0 + rowsY ? (Y0 + rowsY - 1 - y_first) / p.S + 1 : 0;
      const int NP = rows_c * k.PWc;
      {
        // NTU 16-pixel tiles per unit share every A fragment (one global load per NTU MFMAs); tiles past the class's last
        // pixel are skipped (wave-uniform)
        const int ntl = min(NTU, (NP - pr * 16 * NTU + 15) >> 4);
        bool ok[NTU];
        int rT[NTU], cT[NTU];
        const float* __restrict__ lT[NTU];
#pragma unroll
        for (int t = 0; t < NTU; ++t) {
          const int idx = (pr * NTU + t) * 16 + j;
          ok[t] = idx < NP;
          const int i_ = ok[t] ? idx : 0;
          rT[t] = i_ / k.PWc;
          cT[t] = i_ - rT[t] * k.PWc;
          // class pixel (q, pc) -> dOut position (q, pc) relative to the image origin (oy_lo, ox_lo)
          lT[t] = img + (q_lo + rT[t] - oy_lo) * WP + (k.p0 + cT[t] - p.ox_lo) + g * PLANE;
        }
        f32x4 acc[MT][NTU];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int t = 0; t < NTU; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // fragment of (step s, channel tile m) at [(s * mt_total + m)][64]; this workgroup's tiles are mg * MT .. + MT - 1
        const float* __restrict__ wf = p.wfrag + k.frag_off + lane + (long)mg * MT * 64;
        const int fstep = p.mt_total * 64;
        float a_cur[CH * MT], a_nxt[CH * MT];
#pragma unroll
        for (int u = 0; u < CH; ++u)
#pragma unroll
          for (int m = 0; m < MT; ++m) a_cur[u * MT + m] = wf[u * fstep + m * 64];
        int off = 0, bi = 0, ci = 0;
        for (int ck = 0; ck < k.nchunks; ++ck) {
          if (ck + 1 < k.nchunks) {
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
              for (int m = 0; m < MT; ++m) a_nxt[u * MT + m] = wf[((ck + 1) * CH + u) * fstep + m * 64];
          }
          float bv[NTU][CH];
#pragma unroll
          for (int u = 0; u < CH; ++u) {
            const int o = (ck * CH + u < k.nsteps) ? off : 0;     // padded steps (A = 0) read a valid word
#pragma unroll
            for (int t = 0; t < NTU; ++t) bv[t][u] = lT[t][o];
            ++ci;
            const bool w1 = (ci == p.c4n);
            ci = w1 ? 0 : ci;
            bi += w1 ? 1 : 0;
            const bool w2 = (bi == k.nb);
            bi = w2 ? 0 : bi;
            off += 4 * PLANE + (w1 ? -1 - p.c4n * 4 * PLANE : 0) + (w2 ? -WP + k.nb : 0);
          }
#pragma unroll
          for (int u = 0; u < CH; ++u)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
              for (int t = 0; t < NTU; ++t)
                if (t < 2 || t < ntl)                             // (tiles 0 and 1 as before; uniform branch for the rest)
                  acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[u * MT + m], bv[t][u], acc[m][t], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < CH * MT; ++i) a_cur[i] = a_nxt[i];
        }
#pragma unroll
        for (int t = 0; t < NTU; ++t) {
          if (!ok[t]) continue;
          const int yy = y_first + rT[t] * p.S - Y0;              // row inside the band
          const int xx = (k.p0 + cT[t]) * p.S + k.rx - p.P;
          const int pix = yy * p.W + xx;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int ch = m * 16 + 4 * g + rr;
              if (ch0 + ch < p.Cin) outb[ch * p.TY * p.W + pix] = acc[m][t][rr];
            }
        }
      }
    }
    __syncthreads();
    // coalesced flush of the band: rows Y0 .. Y0+rowsY of every channel are ONE contiguous run of
    // rowsY*W floats per channel; a wave takes a channel at a time (no per-element division), with
    // the widest vector the row pitch allows (W % 4 == 0: 16 B, W % 2 == 0: 8 B, else 4 B)
    if (p.st.flat) {
      // the band is the whole sample and TY == H: the dX block [Cin][H*W] is one contiguous 16-B aligned run in LDS,
      // in HBM and in the mask, whatever the row width -- float4 everywhere, the mask words of a batch loaded first
      const int cg = p.mgroups > 1 ? min(16 * MT, p.Cin - ch0) : p.Cin;      // channels of this workgroup
      const int n4 = (cg * p.H * p.W) >> 2;
      const long o0 = (b * (long)p.Cin + ch0) * p.H * p.W;
      for (int i0 = tid; i0 < n4; i0 += 256 * STAGE_U) {
        float4 mk4[STAGE_U];
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
          const int i = i0 + u * 256;
          mk4[u] = make_float4(1.f, 1.f, 1.f, 1.f);
          if (p.mask && i < n4) mk4[u] = reinterpret_cast<const float4*>(p.mask + o0)[i];
        }
#pragma unroll
        for (int u = 0; u < STAGE_U; ++u) {
          const int i = i0 + u * 256;
          if (i < n4) {
            float4 v = reinterpret_cast<const float4*>(outb)[i];
            if (!(mk4[u].x > 0.f)) v.x = 0.f;
            if (!(mk4[u].y > 0.f)) v.y = 0.f;
            if (!(mk4[u].z > 0.f)) v.z = 0.f;
            if (!(mk4[u].w > 0.f)) v.w = 0.f;
            reinterpret_cast<float4*>(p.din + o0)[i] = v;
          }
        }
      }
    } else {
      const int per = rowsY * p.W;
      // a channel's band is one contiguous run: 16-B vectors when the run (not necessarily each row) is 16-B aligned
      const bool run4 = (per & 3) == 0 && ((Y0 * p.W) & 3) == 0 && ((p.H * p.W) & 3) == 0 && ((p.TY * p.W) & 3) == 0;
      const int vec = ((p.W & 3) == 0 || run4) ? 4 : (p.W & 1) == 0 ? 2 : 1;
      for (int ch = w; ch < p.Cin; ch += 4) {
        const long o0 = ((b * p.Cin + ch) * (long)p.H + Y0) * p.W;
        const float* __restrict__ sb = outb + ch * p.TY * p.W;
        if (vec == 4) {
          for (int e = lane << 2; e < per; e += 256) {
            float4 v = *reinterpret_cast<const float4*>(sb + e);
            if (p.mask) {
              const float4 mk = *reinterpret_cast<const float4*>(p.mask + o0 + e);
              if (!(mk.x > 0.f)) v.x = 0.f;
              if (!(mk.y > 0.f)) v.y = 0.f;
              if (!(mk.z > 0.f)) v.z = 0.f;
              if (!(mk.w > 0.f)) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(p.din + o0 + e) = v;
          }
        } else if (vec == 2) {
          for (int e = lane << 1; e < per; e += 128) {
            float2 v = *reinterpret_cast<const float2*>(sb + e);
            if (p.mask) {
              const float2 mk = *reinterpret_cast<const float2*>(p.mask + o0 + e);
              if (!(mk.x > 0.f)) v.x = 0.f;
              if (!(mk.y > 0.f)) v.y = 0.f;
            }
            *reinterpret_cast<float2*>(p.din + o0 + e) = v;
          }
        } else {
          for (int e = lane; e < per; e += 64) {
            float v = sb[e];
            if (p.mask && !(p.mask[o0 + e] > 0.f)) v = 0.f;
            p.din[o0 + e] = v;
          }
        }
      }
    }
  }
}

// Pipelined twin of bwd_band_kernel for the fragment sets that fit LDS (every 3x3 layer of the reference models up
// to 48 KB of fragments).  Same band / class / step structure and the same (class, tap a, tap b, channel quad)
// summation order, so the results are bit-identical; what changes is where the time went in the PMC profile of the
// synchronous kernel (12 VALU instructions per MFMA, 46 % of the wave time parked on staging):
//   * the A fragments of all classes live in LDS for the whole kernel (were: global loads per chunk and pixel pair),
//   * the tap walk is three nested loops over (a, b, channel quad) with running LDS addresses (was: a per-lane
//     counter machine of selects per MFMA step),
//   * the NEXT band's dOut rows are loaded into registers before the matrix phase of the current band, and the
//     ReLU-derivative mask of the CURRENT band at its start, so neither the staging nor the flush waits on HBM.
constexpr int PFB2_MAX = 16;         // dOut prefetch slots per thread (VEC floats each): template parameter PFB2 = 8, 12 or 16
constexpr int PFM2 = 12;             // mask prefetch slots per thread (float4 each)
constexpr int PFM2_W1 = 8;           // ... of the fused variant (registers: two workgroups per CU)
struct BwdBand2P {
  BwdBandP b;
  int nfrag;                         // floats of the prepared backward fragments (all classes)
  int mask_pf;                       // 1: the band's mask fits PFM2 float4 per thread and rows are 16-B multiples
};

// W1 = true fuses the weight gradient of the layer BELOW behind this backward-data pass: when that layer is the first
// of the stack (3x3 / stride 1 / pad 1, <= 48 weight columns: 4 frames x 9 taps), the dX band this kernel assembles
// in LDS is exactly the dOut tile its weight gradient needs, and nothing else reads dX.  The band is masked in place,
// multiplied against the band's input rows (prefetched like dOut: fw.x) on the matrix cores, and never leaves the
// chip: 2 x Cin*H*W*4 bytes per sample (GRUModel/ConvModel: 903 KB of the 2.1 MB the two separate passes moved)
// and one launch saved.  Per-workgroup partials of dW / db go to fw.slab for wgrad_reduce_kernel.
struct FuseW1P {
  const float* x; long x_bs;                 // the lower layer's input (B, Cin1, H, W) and its sample stride
  float* slab;                               // [grid][Cout1*K1 + Cout1], Cout1 = this layer's Cin
  int Cin1, K1, WPx, PLANEx, TIHx, xoff;     // LDS image of the band's input rows (TIHx = TY + 2 rows of W + 2 columns) at lds + xoff
};
constexpr int NIX = 4;                       // float4 prefetch slots of the input rows per thread

template <int MT, int VEC, int PFB2, bool W1 = false>
__global__ __launch_bounds__(256) void bwd_band2_kernel(BwdBand2P pp, FuseW1P fw) {
  constexpr int PFM = W1 ? PFM2_W1 : PFM2;       // mask prefetch slots per thread
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const BwdBandP& p = pp.b;
  float* __restrict__ frag = lds;                          // [nfrag]
  float* __restrict__ outb = lds + pp.nfrag;                // [Cin][TY*W]
  float* __restrict__ img = outb + p.out_floats;            // [Cout][PLANE]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  const long total = (long)p.B * p.bands;
  for (int i = tid; i < pp.nfrag; i += 256) frag[i] = p.wfrag[i];
  stage_zero(p.st, img);
  // ---- fused lower-layer weight gradient: input-row image, prefetch slots, this lane's k offsets, accumulators
  float* __restrict__ xt = lds + (W1 ? fw.xoff : 0);
  int dxs[W1 ? NIX : 1];
  float4 vx[W1 ? NIX : 1];
  int koff1[3];
  f32x4 acc1[3];
  float dbacc1 = 0.f;
  if (W1) {
    for (int i = tid; i < fw.Cin1 * fw.PLANEx + 64; i += 256) xt[i] = 0.f;
    const int nvx = p.W >> 2, totx = fw.Cin1 * fw.TIHx * nvx;
#pragma unroll
    for (int u = 0; u < NIX; ++u) {
      const int idx = tid + u * 256;
      const int rt = idx / nvx, c = rt / fw.TIHx;
      dxs[u] = idx < totx ? (c << 16) | ((rt - c * fw.TIHx) << 8) | (idx - rt * nvx) : -1;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int k = q * 16 + j;
      int o = 0;
      if (k < fw.K1) {
        const int ci = k / 9, rem = k - ci * 9, ky = rem / 3;
        o = ci * fw.PLANEx + ky * fw.WPx + (rem - ky * 3);
      }
      koff1[q] = o + g;                                   // + pixel (4*c4 + g) of the step
      acc1[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }
  // this thread's slots of the dOut band image: (plane c, image row r, VEC floats at column x)
  const int nv = p.st.IW / VEC;
  const int tot_v = p.st.Cp * p.st.TIH * nv;
  int dsc[PFB2];                  // (plane << 20) | (row << 10) | vector index, -1 = none
#pragma unroll
  for (int u = 0; u < PFB2; ++u) {
    const int idx = tid + u * 256;
    const int rt = idx / nv, c = rt / p.st.TIH;
    dsc[u] = idx < tot_v ? (c << 20) | ((rt - c * p.st.TIH) << 10) | (idx - rt * nv) : -1;
  }
  float pf[PFB2][VEC];
  auto issue = [&](long tile) {
    const long b = tile / p.bands;
    const int Y0 = (int)(tile - b * p.bands) * p.TY;
    const int oy_lo = fdiv(Y0 + p.P - (p.ks - 1), p.S);
    const float* __restrict__ base = p.st.src + b * p.st.bstride + (long)oy_lo * p.st.IW;
#pragma unroll
    for (int u = 0; u < PFB2; ++u) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) pf[u][e] = 0.f;
      const int r = (dsc[u] >> 10) & 1023, ys = oy_lo + r;
      if (dsc[u] >= 0 && ys >= 0 && ys < p.st.IH) {
        const float* q = base + ((long)(dsc[u] >> 20) * p.st.IH + r) * p.st.IW + (dsc[u] & 1023) * VEC;
        if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(q); pf[u][0] = t.x; pf[u][1] = t.y; pf[u][2] = t.z; pf[u][3] = t.w; }
        else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(q); pf[u][0] = t.x; pf[u][1] = t.y; }
        else pf[u][0] = q[0];
      }
    }
    if (W1) {      // the lower layer's input rows Y0-1 .. Y0+TY of every plane
      const float* __restrict__ xb = fw.x + b * fw.x_bs;
#pragma unroll
      for (int u = 0; u < NIX; ++u) {
        vx[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int ys = Y0 - 1 + ((dxs[u] >> 8) & 255);
        if (dxs[u] >= 0 && ys >= 0 && ys < p.H)
          vx[u] = *reinterpret_cast<const float4*>(xb + ((long)(dxs[u] >> 16) * p.H + ys) * p.W + ((dxs[u] & 255) << 2));
      }
    }
  };
  long tile = blockIdx.x;
  if (tile < total) issue(tile);
  for (; tile < total; tile += gridDim.x) {
    const long b = tile / p.bands;
    const int Y0 = (int)(tile - b * p.bands) * p.TY;
    const int rowsY = min(p.TY, p.H - Y0);
    const int oy_lo = fdiv(Y0 + p.P - (p.ks - 1), p.S);
    __syncthreads();                       // readers of the previous band (image and dX band) are done
#pragma unroll
    for (int u = 0; u < PFB2; ++u)
      if (dsc[u] >= 0) {
        float* d = img + (dsc[u] >> 20) * PLANE + ((dsc[u] >> 10) & 1023) * WP + (dsc[u] & 1023) * VEC - p.st.sx0;
#pragma unroll
        for (int e = 0; e < VEC; ++e) d[e] = pf[u][e];
      }
    if (W1) {
#pragma unroll
      for (int u = 0; u < NIX; ++u)
        if (dxs[u] >= 0) {
          float* d = xt + (dxs[u] >> 16) * fw.PLANEx + ((dxs[u] >> 8) & 255) * fw.WPx + ((dxs[u] & 255) << 2) + 1;
          d[0] = vx[u].x; d[1] = vx[u].y; d[2] = vx[u].z; d[3] = vx[u].w;
        }
    }
    __syncthreads();
    if (tile + gridDim.x < total) issue(tile + gridDim.x);
    // the ReLU-derivative mask of THIS band: in flight during the matrix phase, consumed by the flush
    const int per = rowsY * p.W;
    float4 mk[PFM];
    if (pp.mask_pf && p.mask) {
      const int per4 = per >> 2, n4 = p.Cin * per4;
#pragma unroll
      for (int u = 0; u < PFM; ++u) {
        const int i = tid + u * 256;
        mk[u] = make_float4(1.f, 1.f, 1.f, 1.f);
        if (i < n4) {
          const int ch = i / per4, e = (i - ch * per4) << 2;
          mk[u] = *reinterpret_cast<const float4*>(p.mask + ((b * p.Cin + ch) * (long)p.H + Y0) * p.W + e);
        }
      }
    }
    for (int c = 0; c < p.ncls; ++c) {
      const BandClass& k = p.cls[c];
      int q_lo = (Y0 + p.P - k.ry + p.S - 1) / p.S;
      if (q_lo < 0) q_lo = 0;
      const int y_first = p.S * q_lo + k.ry - p.P;
      const int rows_c = y_first < Y0 + rowsY ? (Y0 + rowsY - 1 - y_first) / p.S + 1 : 0;
      const int NP = rows_c * k.PWc;
      const int npairs = (NP + 31) >> 5;
      const float* __restrict__ fc = frag + k.frag_off + lane;
      for (int pr = w; pr < npairs; pr += 4) {
        const int idx0 = pr * 32 + j, idx1 = idx0 + 16;
        const bool ok0 = idx0 < NP, ok1 = idx1 < NP;
        const int i0 = ok0 ? idx0 : 0, i1 = ok1 ? idx1 : 0;
        const int r0 = i0 / k.PWc, c0 = i0 - r0 * k.PWc;
        const int r1 = i1 / k.PWc, c1 = i1 - r1 * k.PWc;
        const float* __restrict__ l0 = img + (q_lo + r0 - oy_lo) * WP + (k.p0 + c0 - p.ox_lo) + g * PLANE;
        const float* __restrict__ l1 = img + (q_lo + r1 - oy_lo) * WP + (k.p0 + c1 - p.ox_lo) + g * PLANE;
        f32x4 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
          acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const float* __restrict__ fa = fc;                 // fragments of step s = (a*nb + b)*c4n + c4: MT*64 floats each
        for (int ta = 0; ta < k.na; ++ta)
          for (int tb = 0; tb < k.nb; ++tb) {
            const float* __restrict__ t0 = l0 - ta * WP - tb;
            const float* __restrict__ t1 = l1 - ta * WP - tb;
            for (int c4 = 0; c4 < p.c4n; c4 += 2) {        // two channel quads per trip (c4n is even): 2*(MT+2) LDS reads in flight
              float av[2][MT], bv0[2], bv1[2];
#pragma unroll
              for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int m = 0; m < MT; ++m) av[u][m] = fa[((c4 + u) * MT + m) * 64];
                bv0[u] = t0[(c4 + u) * 4 * PLANE];
                bv1[u] = t1[(c4 + u) * 4 * PLANE];
              }
#pragma unroll
              for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                  acc[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bv0[u], acc[m][0], 0, 0, 0);
                  acc[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u][m], bv1[u], acc[m][1], 0, 0, 0);
                }
            }
            fa += p.c4n * MT * 64;
          }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
          if (!(nt ? ok1 : ok0)) continue;
          const int r = nt ? r1 : r0, cc = nt ? c1 : c0;
          const int yy = y_first + r * p.S - Y0;
          const int xx = (k.p0 + cc) * p.S + k.rx - p.P;
          const int pix = yy * p.W + xx;
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const int ch = m * 16 + 4 * g + rr;
              if (ch < p.Cin) outb[ch * p.TY * p.W + pix] = acc[m][nt][rr];
            }
        }
      }
    }
    __syncthreads();
    if (pp.mask_pf) {        // rows are 16-B multiples: one float4 per slot, mask already in registers
      const int per4 = per >> 2, n4 = p.Cin * per4;
#pragma unroll
      for (int u = 0; u < PFM; ++u) {
        const int i = tid + u * 256;
        if (i < n4) {
          const int ch = i / per4, e = (i - ch * per4) << 2;
          float4 v = *reinterpret_cast<const float4*>(outb + ch * p.TY * p.W + e);
          if (p.mask) {
            if (!(mk[u].x > 0.f)) v.x = 0.f;
            if (!(mk[u].y > 0.f)) v.y = 0.f;
            if (!(mk[u].z > 0.f)) v.z = 0.f;
            if (!(mk[u].w > 0.f)) v.w = 0.f;
          }
          if (W1) *reinterpret_cast<float4*>(outb + ch * p.TY * p.W + e) = v;         // masked in place: it stays on chip
          else *reinterpret_cast<float4*>(p.din + ((b * p.Cin + ch) * (long)p.H + Y0) * p.W + e) = v;
        }
      }
      if (W1) {
        __syncthreads();
        {  // bias gradient of the lower layer: channel = tid & 15, 16 interleaved parts, four reads in flight
          const float* __restrict__ pl = outb + (tid & 15) * p.TY * p.W;
          const int part = tid >> 4;
          float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
          int i = part;
          for (; i + 48 < per; i += 64) { s0 += pl[i]; s1 += pl[i + 16]; s2 += pl[i + 32]; s3 += pl[i + 48]; }
          for (; i < per; i += 16) s0 += pl[i];
          if ((tid & 15) < p.Cin) dbacc1 += (s0 + s1) + (s2 + s3);
        }
        // dW1[co][k] += sum over the band's pixels of dX[co][y][x] * x[ci][y+ky-1][x+kx-1]: 4 pixels per MFMA step,
        // the band's rowsY * W/4 steps in four contiguous runs (one per wave), operands read one step ahead
        const int c4n = p.W >> 2, nsteps = rowsY * c4n;
        const int s0 = (nsteps * w) >> 2, s1 = (nsteps * (w + 1)) >> 2;
        int r = s0 / c4n, c4 = s0 - r * c4n;
        float av, bv[3];
        if (s0 < s1) {
          av = outb[j * p.TY * p.W + r * p.W + 4 * c4 + g];
#pragma unroll
          for (int q = 0; q < 3; ++q) bv[q] = xt[koff1[q] + r * fw.WPx + 4 * c4];
        }
        for (int s = s0; s < s1; ++s) {
          if (++c4 == c4n) { c4 = 0; ++r; }
          const bool more = s + 1 < s1;
          const int rn = more ? r : 0, cn = more ? c4 : 0;
          const float an = outb[j * p.TY * p.W + rn * p.W + 4 * cn + g];
          float bn[3];
#pragma unroll
          for (int q = 0; q < 3; ++q) bn[q] = xt[koff1[q] + rn * fw.WPx + 4 * cn];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 3; ++q) acc1[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[q], acc1[q], 0, 0, 0);
          av = an;
#pragma unroll
          for (int q = 0; q < 3; ++q) bv[q] = bn[q];
        }
      }
    } else {
      const int vec = (p.W & 3) == 0 ? 4 : (p.W & 1) == 0 ? 2 : 1;
      for (int ch = w; ch < p.Cin; ch += 4) {
        const long o0 = ((b * p.Cin + ch) * (long)p.H + Y0) * p.W;
        const float* __restrict__ sb = outb + ch * p.TY * p.W;
        if (vec == 4) {
          for (int e = lane << 2; e < per; e += 256) {
            float4 v = *reinterpret_cast<const float4*>(sb + e);
            if (p.mask) {
              const float4 m4 = *reinterpret_cast<const float4*>(p.mask + o0 + e);
              if (!(m4.x > 0.f)) v.x = 0.f;
              if (!(m4.y > 0.f)) v.y = 0.f;
              if (!(m4.z > 0.f)) v.z = 0.f;
              if (!(m4.w > 0.f)) v.w = 0.f;
            }
            *reinterpret_cast<float4*>(p.din + o0 + e) = v;
          }
        } else if (vec == 2) {
          for (int e = lane << 1; e < per; e += 128) {
            float2 v = *reinterpret_cast<const float2*>(sb + e);
            if (p.mask) {
              const float2 m2 = *reinterpret_cast<const float2*>(p.mask + o0 + e);
              if (!(m2.x > 0.f)) v.x = 0.f;
              if (!(m2.y > 0.f)) v.y = 0.f;
            }
            *reinterpret_cast<float2*>(p.din + o0 + e) = v;
          }
        } else {
          for (int e = lane; e < per; e += 64) {
            float v = sb[e];
            if (p.mask && !(p.mask[o0 + e] > 0.f)) v = 0.f;
            p.din[o0 + e] = v;
          }
        }
      }
    }
  }
  if (W1) {      // per-workgroup partials: the four waves' tiles added in wave order, then the bias parts
    __syncthreads();
    float4* red4 = reinterpret_cast<float4*>(lds);              // [4 waves][3 k-tiles][64 lanes]
#pragma unroll
    for (int q = 0; q < 3; ++q) red4[(w * 3 + q) * 64 + lane] = make_float4(acc1[q][0], acc1[q][1], acc1[q][2], acc1[q][3]);
    float* redb = lds + 4 * 3 * 64 * 4;                         // [16 parts][16 channels]
    redb[(tid >> 4) * 16 + (tid & 15)] = dbacc1;
    __syncthreads();
    float* sl = fw.slab + (long)blockIdx.x * ((long)p.Cin * fw.K1 + p.Cin);
    if (tid < 3 * 64) {
      float4 sv = red4[tid];
#pragma unroll
      for (int ww = 1; ww < 4; ++ww) {
        const float4 t = red4[ww * 3 * 64 + tid];
        sv.x += t.x; sv.y += t.y; sv.z += t.z; sv.w += t.w;
      }
      const int k = (tid >> 6) * 16 + (tid & 15), co = 4 * ((tid & 63) >> 4);
      if (k < fw.K1) {
        if (co < p.Cin) sl[(long)co * fw.K1 + k] = sv.x;
        if (co + 1 < p.Cin) sl[(long)(co + 1) * fw.K1 + k] = sv.y;
        if (co + 2 < p.Cin) sl[(long)(co + 2) * fw.K1 + k] = sv.z;
        if (co + 3 < p.Cin) sl[(long)(co + 3) * fw.K1 + k] = sv.w;
      }
    }
    if (tid < p.Cin) {
      float sb = 0.f;
      for (int q = 0; q < 16; ++q) sb += redb[q * 16 + tid];
      sl[(long)p.Cin * fw.K1 + tid] = sb;
    }
  }
}
// "Row-run" weight-gradient kernel for the unpadded layers (ks = 2*S: 8x8/s4, 4x4/s2), software
// pipelined like igemm_run_kernel.  kx = kxh*S + kxl: for a fixed (ci, ky, kxh) the S taps kxl are
// S contiguous floats at column S*(ox + kxh), so ONE ds_read_b128 (b64) per lane yields the B
// operands of S different k-tiles for the same pixel.  Lanes j of a "row group" enumerate the
// 16 (ci, ky, kxh) combinations cidx = rg*16 + j = (ci*ks + ky)*2 + kxh; wave w owns row groups
// w, w+4, ...  (conv1: one input plane per wave).  Per MFMA step (4 pixels): MT b32 reads of
// dOut + RGW vector reads feed MT*RGW*S MFMAs.
constexpr int PF_D = 12;             // dOut prefetch registers per thread (256*12 floats per tile)

struct WrunP {
  StageP st;                      // input image (fast layout, WP == IW)
  const float* dout; float* slab;
  int Cout, K, ks, OH, OW, OWp, TPH, tiles, B, PLANEo;
};

template <int MT, int S, int RGW, int C4N>
__global__ __launch_bounds__(256) void wgrad_run_kernel(WrunP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int WP = p.st.WP, PLANE = p.st.PLANE;
  const int in_floats = p.st.Cp * PLANE;
  float* __restrict__ img = lds;
  float* __restrict__ ldo = lds + in_floats;          // dOut tile: [MT*16][PLANEo]
  const int lds_total = in_floats + MT * 16 * p.PLANEo + 64;
  for (int i = tid; i < lds_total; i += 256) lds[i] = 0.f;          // padding stays 0 (finite) forever

  // this lane's (ci, ky, kxh) per row group -> image offset and natural weight index of kxl = 0
  int boff[RGW], knat[RGW];
#pragma unroll
  for (int q = 0; q < RGW; ++q) {
    const int cidx = (w + 4 * q) * 16 + j;
    const int kxh = cidx & 1, cy = cidx >> 1;
    const int ci = cy / p.ks, ky = cy - ci * p.ks;
    boff[q] = ci * PLANE + ky * WP + kxh * S + kk * S;                 // + pixel (4*c4 + kk) of the step
    knat[q] = (ci * p.ks + ky) * p.ks + kxh * S;
  }
  f32x4 acc[MT][RGW][S];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < RGW; ++q)
#pragma unroll
      for (int x = 0; x < S; ++x) acc[m][q][x] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int Cm = MT * 16, nparts = 256 / Cm;
  const int bco = tid % Cm, bpart = tid / Cm;
  float dbacc = 0.f;

  // prefetch maps (tile independent): image float4s and dOut floats of this thread
  const int per4 = (p.st.TIH * p.st.IW) >> 2;
  const int tot4 = p.st.Cp * per4;
  int idst[PF_N], isrc[PF_N], irem[PF_N];
#pragma unroll
  for (int u = 0; u < PF_N; ++u) {
    const int idx = tid + u * 256;
    idst[u] = -1; isrc[u] = 0; irem[u] = 0;
    if (idx < tot4) {
      const int c = idx / per4, rem = idx - c * per4;
      idst[u] = c * PLANE + (rem << 2);
      isrc[u] = c * p.st.IH * p.st.IW + (rem << 2);
      irem[u] = rem;
    }
  }
  const int dper = p.TPH * p.OW;                      // dOut floats per channel per full tile
  int ddst[PF_D], dsrc[PF_D], drow[PF_D];
#pragma unroll
  for (int u = 0; u < PF_D; ++u) {
    const int idx = tid + u * 256;
    ddst[u] = -1; dsrc[u] = 0; drow[u] = 0;
    if (idx < p.Cout * dper) {
      const int co = idx / dper, e = idx - co * dper;
      const int r = e / p.OW, x = e - r * p.OW;
      ddst[u] = co * p.PLANEo + r * p.OWp + x;
      dsrc[u] = co * p.OH * p.OW + e;
      drow[u] = r;
    }
  }
  float4 pfi[PF_N];
  float pfd[PF_D];
  const long total = (long)p.B * p.tiles;
  auto issue = [&](long tile) {
    const long b = tile / p.tiles;
    const int ti = (int)(tile - b * p.tiles);
    const int q0 = ti * p.TPH;
    const int y_lo = q0 * S;
    const int ok4 = (min(p.st.TIH, p.st.IH - y_lo) * p.st.IW) >> 2;
    const int rows = min(p.TPH, p.OH - q0);
    const float* __restrict__ ib = p.st.src + b * p.st.bstride + (long)y_lo * p.st.IW;
    const float* __restrict__ db = p.dout + (b * p.Cout * p.OH + q0) * (long)p.OW;
#pragma unroll
    for (int u = 0; u < PF_N; ++u) {
      pfi[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idst[u] >= 0 && irem[u] < ok4) pfi[u] = *reinterpret_cast<const float4*>(ib + isrc[u]);
    }
#pragma unroll
    for (int u = 0; u < PF_D; ++u) {
      pfd[u] = 0.f;
      if (ddst[u] >= 0 && drow[u] < rows) pfd[u] = db[dsrc[u]];
    }
  };
  long tile = blockIdx.x;
  if (tile < total) issue(tile);
  for (; tile < total; tile += gridDim.x) {
    const int ti = (int)(tile % p.tiles);
    const int rows = min(p.TPH, p.OH - ti * p.TPH);
    __syncthreads();                       // readers of the previous tile (and the zero fill) are done
#pragma unroll
    for (int u = 0; u < PF_N; ++u)
      if (idst[u] >= 0) *reinterpret_cast<float4*>(img + idst[u]) = pfi[u];
#pragma unroll
    for (int u = 0; u < PF_D; ++u)
      if (ddst[u] >= 0) ldo[ddst[u]] = pfd[u];     // rows past the bottom were loaded as 0
    __syncthreads();
    if (tile + gridDim.x < total) issue(tile + gridDim.x);     // in flight during the MFMA phase below
    if (bpart < nparts && bco < p.Cout) {          // bias partial sums over this tile
      const float* pl = ldo + bco * p.PLANEo;
      const int n = rows * p.OWp;
      float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // four reads in flight
      int i = bpart;
      for (; i + 3 * nparts < n; i += 4 * nparts) { s0 += pl[i]; s1 += pl[i + nparts]; s2 += pl[i + 2 * nparts]; s3 += pl[i + 3 * nparts]; }
      for (; i < n; i += nparts) s0 += pl[i];
      dbacc += (s0 + s1) + (s2 + s3);
    }
    for (int r = 0; r < rows; ++r) {
      const float* __restrict__ arow = ldo + j * p.PLANEo + r * p.OWp + kk;
      const float* __restrict__ brow = img + r * S * WP;
      float av[C4N][MT];
      float bv[C4N][RGW][S];
#pragma unroll
      for (int c4 = 0; c4 < C4N; ++c4) {           // all LDS reads of the pixel row first ...
#pragma unroll
        for (int m = 0; m < MT; ++m) av[c4][m] = arow[m * 16 * p.PLANEo + 4 * c4];
#pragma unroll
        for (int q = 0; q < RGW; ++q) {
          const float* src = brow + boff[q] + 4 * c4 * S;
          if (S == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            bv[c4][q][0] = t.x; bv[c4][q][1] = t.y; bv[c4][q][2] = t.z; bv[c4][q][3] = t.w;
          } else {
            const float2 t = *reinterpret_cast<const float2*>(src);
            bv[c4][q][0] = t.x; bv[c4][q][1] = t.y;
          }
        }
      }
#pragma unroll
      for (int c4 = 0; c4 < C4N; ++c4)             // ... then its MFMAs
#pragma unroll
        for (int q = 0; q < RGW; ++q)
#pragma unroll
          for (int x = 0; x < S; ++x)
#pragma unroll
            for (int m = 0; m < MT; ++m)
              acc[m][q][x] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c4][m], bv[c4][q][x], acc[m][q][x], 0, 0, 0);
    }
  }
  // partials: D row (co) = 4*(lane>>4)+reg, col = lane&15 -> k = knat[q] + kxl
  float* sl = p.slab + (long)blockIdx.x * ((long)p.Cout * p.K + p.Cout);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < RGW; ++q)
#pragma unroll
      for (int x = 0; x < S; ++x)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int co = m * 16 + 4 * kk + rr;
          if (co < p.Cout) sl[(long)co * p.K + knat[q] + x] = acc[m][q][x][rr];
        }
  __syncthreads();
  float* red = lds;   // reuse: [nparts][Cm]
  if (bpart < nparts) red[bpart * Cm + bco] = dbacc;
  __syncthreads();
  if (tid < p.Cout) {
    float s = 0.f;
    for (int q = 0; q < nparts; ++q) s += red[q * Cm + tid];
    sl[(long)p.Cout * p.K + tid] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad_x6_kernel (round 6): A3CModel conv2's weight gradient -- dW[co][ci][ky][kx] = sum over samples and the 9 x 9 output
// pixels of dOut[co][oy][ox] a1[ci][2 oy + ky][2 ox + kx] -- on the BF16 matrix pipe with fp32 results, the twin of
// bwd_x6_kernel: both operands split into three bf16 pieces (exact), the six piece products with qa + qb <= 2 issued
// (gemm_x6_kernel's argument), v_mfma_f32_16x16x32_bf16 with k = PIXELS.  Per sample one product D[32 co][256 n] +=
// A[co][k] B[n][k] over 96 pixel slots in 12 groups of 8: groups 0..8 = the eight pixels ox 0..7 of output row oy, group 9 =
// the column ox = 8 of rows 0..7, group 10 = the corner pixel (8, 8) + zeros, group 11 = zeros (81 of 96 slots carry data).
//   * the sample's a1 (25.6 KB) and dOut (10.4 KB) arrive by LDS-DMA into an fp32 scratch, issued for the NEXT sample at the
//     head of the matrix phase (no prefetch registers);
//   * a conversion phase splits them into the piece images: a1 as phase runs P[ci][y][x parity][m] = a1[ci][y][2 m + parity]
//     (a tap column kx = 2 kxh + kxl of output pixels ox 0..7 is elements m = ox + kxh of parity kxl: eight consecutive
//     elements; kxh = 1 shifts the fragment by one element with v_alignbit, wave-uniform per 16-column tile), the column
//     group and the corner as 16-byte cells per weight column, dOut as [co][slot];
//   * wave w owns the weight columns of input channels 4 w .. 4 w + 3 (four 16-column tiles: a channel pair x kxh) and both
//     16-row co tiles: 18 fragment reads feed 48 MFMAs per 32-slot step;
//   * per-workgroup slabs [32 x 256 + 32] (db = sum of dOut in fp32) go to wgrad_reduce_kernel as before.
namespace wx {
constexpr int NT = 512;
constexpr int SCR_A1 = 0, SCR_DO = 25600, SCR_BYTES = SCR_DO + 16384;      // the fp32 scratch: a STATIC array of its own -- as part of
// the dynamic block the compiler cannot tell the LDS-DMA's writes from the fragment reads and waits vmcnt(0) in the matrix phase
constexpr int P0 = 0, PQ = 15360, PBLK = 48;                               // the piece images (dynamic block), bytes
// P[q][(ci, y)]: a 48-byte block = [parity 0: m 0..7][parity 1: m 0..7][m 8, 9 of parity 0 | of parity 1 | 8 B pad]: a fragment is
// ONE ds_read_b128 (+ the tail dword when the run is shifted); at a 24-byte run pitch the two 8-byte halves compiled to
// ds_read2_b64: 8 LDS cycles per wave-instruction at 128 B/clk instead of 4 at 256
constexpr int C2_0 = P0 + 3 * PQ, CQ = 4096, X2_0 = C2_0 + 3 * CQ;
constexpr int A_0 = X2_0 + 3 * CQ, AQ = 6656, AROW = 208;
constexpr int ZERO = A_0 + 3 * AQ, LDS_BYTES = ZERO + 32;
}  // namespace wx
struct WgradX6P {
  const float* in; long in_bs; const float* dout; float* slab; int B; int dbg;
  // RANK (see BwdX6P): dOut formed in the kernel, written to the fp32 scratch in place of the DMA
  const float* dl; long ldl; int nlog; const float* Wc; const unsigned char* a2b; long a2b_row;
};
__device__ __forceinline__ void wx_split2(float a, float b, unsigned int o[3]) {
  const __bf16 a0 = (__bf16)a, b0 = (__bf16)b;
  const float ra = a - (float)a0, rb = b - (float)b0;                 // exact
  const __bf16 a1 = (__bf16)ra, b1 = (__bf16)rb;
  const float sa = ra - (float)a1, sb = rb - (float)b1;               // exact, at most 8 significant bits
  const __bf16 a2 = (__bf16)sa, b2 = (__bf16)sb;
  o[0] = (unsigned int)__builtin_bit_cast(unsigned short, a0) | ((unsigned int)__builtin_bit_cast(unsigned short, b0) << 16);
  o[1] = (unsigned int)__builtin_bit_cast(unsigned short, a1) | ((unsigned int)__builtin_bit_cast(unsigned short, b1) << 16);
  o[2] = (unsigned int)__builtin_bit_cast(unsigned short, a2) | ((unsigned int)__builtin_bit_cast(unsigned short, b2) << 16);
}
template <bool RANK>
__global__ __launch_bounds__(wx::NT) void wgrad_x6_kernel(WgradX6P p) {
  using namespace wx;
  typedef const void __attribute__((address_space(1)))* gptr_t;
  typedef void __attribute__((address_space(3)))* lptr_t;
  __shared__ __attribute__((aligned(16))) unsigned char scrw[SCR_BYTES];
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsw[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wq = w & 3, th = w >> 2;                   // channel group (input channels 4 wq .. 4 wq + 3) and tile half
  const float* __restrict__ sa1 = reinterpret_cast<const float*>(scrw + SCR_A1);
  const float* __restrict__ sdo = reinterpret_cast<const float*>(scrw + SCR_DO);
  for (int i = tid; i < LDS_BYTES / 16; i += NT) *reinterpret_cast<u32x4x*>(ldsw + i * 16) = (u32x4x){0u, 0u, 0u, 0u};
  // ---- matrix-phase addresses of this lane: tiles tt = 0 (kxh = 0) and 1 (kxh = 1) of channel pair 2 wq + th
  int b0[2], b2[2], t0[2], t2[2], ncol[2];
  const int qs2 = g == 0 ? PQ : (g == 3 ? 0 : CQ);
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) {
    const int ci = 2 * (2 * wq + th) + (l16 >> 3), ky = (l16 >> 1) & 3, kxl = l16 & 1;
    const int n = (ci * 4 + ky) * 4 + 2 * tt + kxl;
    ncol[tt] = n;
    b0[tt] = P0 + (ci * 20 + 2 * g + ky) * PBLK + 16 * kxl;                  // step 0: output row oy = g (step 1: + 8 rows)
    b2[tt] = g == 0 ? P0 + (ci * 20 + 16 + ky) * PBLK + 16 * kxl : g == 1 ? C2_0 + n * 16 : g == 2 ? X2_0 + n * 16 : ZERO;
    t0[tt] = P0 + (ci * 20 + 2 * g + ky) * PBLK + 32 + 4 * kxl;              // the tail dword (m 8, 9) of that run
    t2[tt] = g == 0 ? P0 + (ci * 20 + 16 + ky) * PBLK + 32 + 4 * kxl : ZERO;
  }
  const int aaddr = A_0 + l16 * AROW + g * 16;
  // two accumulators per tile: the (0,0) products (one addition per 32 pixel slots: fewer roundings than the fp32 MFMA chain's one
  // per 4) and the five small ones (their roundings are relative to a sum 2^-8 of the first): against fp64 the result is no less
  // accurate than the fp32 kernel's
  f32x4 acc[2][2], acs[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc[mt][tt] = acs[mt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;
  const int bco = tid & 31, bpart = tid >> 5;
  auto dma = [&](long b) {
    const float* __restrict__ ga = p.in + b * p.in_bs;
    const float* __restrict__ gd = p.dout + b * 2592L;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i < 3 || w == 0)
        __builtin_amdgcn_global_load_lds((gptr_t)(ga + (long)(i * NT + tid) * 4), (lptr_t)(scrw + SCR_A1 + (i * NT + w * 64) * 16), 16, 0, 0);
    }
    if (!RANK) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {           // 648 chunks of 16 B; the lanes past them re-read chunk 647 into the scratch's slack
        const int c = min(i * NT + tid, 647);
        __builtin_amdgcn_global_load_lds((gptr_t)(gd + (long)c * 4), (lptr_t)(scrw + SCR_DO + (i * NT + w * 64) * 16), 16, 0, 0);
      }
    }
  };
  // RANK: this thread's elements e = tid + 512 j of the sample's dOut: their rows of the composed matrix (all samples), the raw
  // inputs of the NEXT sample to be formed (its dl and the mask bytes of the elements)
  float wc[RANK ? 6 : 1][4], gl[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned int mbv[RANK ? 6 : 1];
  if (RANK) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) wc[RANK ? j : 0][n] = (tid + 512 * j < 2592 && n < p.nlog) ? p.Wc[(long)n * 2592 + tid + 512 * j] : 0.f;
  }
  auto raw = [&](long bn) {
#pragma unroll
    for (int n = 0; n < 4; ++n) gl[n] = n < p.nlog ? p.dl[bn * p.ldl + n] : 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) mbv[RANK ? j : 0] = p.a2b[bn * p.a2b_row + (min(tid + 512 * j, 2591) >> 3)];
  };
  auto form = [&]() {        // dOut[e] = (a2[e] > 0) ? sum_n dl[n] Wc[n][e] : 0 -- small_n_bwd_data_bits_kernel's sums -- into the scratch
    float* __restrict__ dst = reinterpret_cast<float*>(scrw + SCR_DO);
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = tid + 512 * j;
      float a = 0.f;
#pragma unroll
      for (int n = 0; n < 4; ++n) a += gl[n] * wc[RANK ? j : 0][n];
      if (!((mbv[RANK ? j : 0] >> (e & 7)) & 1u)) a = 0.f;
      if (e < 2592) dst[e] = a;
    }
  };
  // one a1 half row (ci, y, parity): elements m = 0..9 of the phase run
  auto half_row = [&](int h) {
    const int ci = h / 40, rem = h - ci * 40, y = rem >> 1, par = rem & 1;
    const float* __restrict__ row = sa1 + ci * 400 + y * 20 + par;
    unsigned int o[5][3];
#pragma unroll
    for (int i = 0; i < 5; ++i) wx_split2(row[4 * i], row[4 * i + 2], o[i]);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      unsigned char* dst = ldsw + P0 + q * PQ + (h >> 1) * PBLK;
      *reinterpret_cast<u32x4x*>(dst + 16 * par) = (u32x4x){o[0][q], o[1][q], o[2][q], o[3][q]};
      *reinterpret_cast<unsigned int*>(dst + 32 + 4 * par) = o[4][q];
    }
  };
  long b = blockIdx.x;
  if (RANK && b < p.B) {
    raw(b);
    form();
    raw(b + gridDim.x < p.B ? b + gridDim.x : b);
  }
  if (b < p.B) dma(b);
  for (; b < p.B; b += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // the sample's scratch has landed; the previous matrix phase is over
    // ---- conversion (waves 0-1: two half rows; 2-3: half row + dOut group; 4-7: half row, weight column, dOut group)
    if (!(p.dbg & 1)) {
      half_row(tid);
      if (tid < 128) half_row(512 + tid);
      if (tid >= 256) {  // the column ox = 8 (rows oy 0..7) and the corner of weight column n
        const int n = tid - 256, ci = n >> 4, ky = (n >> 2) & 3, kx = n & 3;
        const float* __restrict__ col = sa1 + ci * 400 + ky * 20 + 16 + kx;
        unsigned int o[4][3], oc[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) wx_split2(col[80 * i], col[80 * i + 40], o[i]);
        wx_split2(col[320], 0.f, oc);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          *reinterpret_cast<u32x4x*>(ldsw + C2_0 + q * CQ + n * 16) = (u32x4x){o[0][q], o[1][q], o[2][q], o[3][q]};
          *reinterpret_cast<unsigned int*>(ldsw + X2_0 + q * CQ + n * 16) = oc[q];
        }
      }
    }
    if (!(p.dbg & 2) && tid >= 128 && tid < 480) {     // dOut group (co, gi): 0..8 rows, 9 the column, 10 the corner
      const int id = tid - 128, co = id / 11, gi = id - co * 11;
      const float* __restrict__ src = sdo + co * 81;
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = gi < 9 ? gi * 9 + i : gi == 9 ? i * 9 + 8 : 80;
        e[i] = (gi == 10 && i > 0) ? 0.f : src[idx];
      }
      u32x4x o[3];
      bx_split8(e, o);
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4x*>(ldsw + A_0 + q * AQ + co * AROW + gi * 16) = o[q];
    }
    {  // bias partial sums (fp32, fixed order)
      const float* __restrict__ src = sdo + bco * 81;
      float s0 = 0.f;
      for (int i = bpart; i < 81; i += 16) s0 += src[i];
      dbacc += s0;
    }
    __syncthreads();                                   // pieces complete; the scratch is free
    if (RANK && b + gridDim.x < p.B) {                 // the next sample's dOut from the raw inputs fetched one iteration ago
      form();                                          // (before the DMA: the compiler orders plain LDS writes behind LDS-DMA with a vmcnt(0))
      const long b2 = b + 2L * gridDim.x;
      raw(b2 < p.B ? b2 : b);                          // (and the small loads first: the loop top waits for the youngest)
    }
    if (b + gridDim.x < p.B && !(p.dbg & 8)) dma(b + gridDim.x);
    // ---- matrix phase: three steps of 32 pixel slots; all twelve fragments of a step first, then its 24 MFMAs
    if (!(p.dbg & 4))
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      // The fragment reads go through inline asm: hipcc's waitcnt insertion treats every LDS access of a kernel that issues
      // LDS-DMA as a possible reader of an in-flight DMA and puts `s_waitcnt vmcnt(0)` in front of it -- here a wait for the next
      // sample's scratch (issued a few instructions earlier) at the head of the matrix phase that was meant to cover it.
      bf16x8x a[2][3], bf[2][3];
      u32x4x ra[2][3], rb[2][3];
      unsigned int rt[3] = {0u, 0u, 0u};
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          asm volatile("ds_read_b128 %0, %1" : "=v"(ra[mt][q]) : "v"((unsigned)(unsigned long)(lptr_t)(ldsw + aaddr + mt * 16 * AROW + s * 64 + q * AQ)));
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int base = s == 0 ? b0[tt] : s == 1 ? b0[tt] + 8 * PBLK : b2[tt];
        const int tail = s == 0 ? t0[tt] : s == 1 ? t0[tt] + 8 * PBLK : t2[tt];
        const int qs = s == 2 ? qs2 : PQ;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          asm volatile("ds_read_b128 %0, %1" : "=v"(rb[tt][q]) : "v"((unsigned)(unsigned long)(lptr_t)(ldsw + base + q * qs)));
          if (tt == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(rt[q]) : "v"((unsigned)(unsigned long)(lptr_t)(ldsw + tail + q * qs)));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[0][2]), "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(ra[1][2]), "+v"(rb[0][0]), "+v"(rb[0][1]),
                     "+v"(rb[0][2]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[1][2]), "+v"(rt[0]), "+v"(rt[1]), "+v"(rt[2])
                   :: "memory");
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 3; ++q) a[mt][q] = __builtin_bit_cast(bf16x8x, ra[mt][q]);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        bf[0][q] = __builtin_bit_cast(bf16x8x, rb[0][q]);
        u32x4x v = rb[1][q];                            // kxh = 1: the run one element further
        const u32x4x sh = (u32x4x){__builtin_amdgcn_alignbit(v[1], v[0], 16), __builtin_amdgcn_alignbit(v[2], v[1], 16),
                                   __builtin_amdgcn_alignbit(v[3], v[2], 16), __builtin_amdgcn_alignbit(rt[q], v[3], 16)};
        if (s < 2 || g == 0) v = sh;                    // (step 2: only the lanes reading a row run)
        bf[1][q] = __builtin_bit_cast(bf16x8x, v);
      }
      // rising magnitude: (2,0) (1,1) (0,2) | (1,0) (0,1) | (0,0)   (a: dOut pieces, b: a1 pieces); the four chains interleaved
#define WX_MM(AC, QA, QB)                                                                                               \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                  \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                                \
          AC[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][QA], bf[tt][QB], AC[mt][tt], 0, 0, 0);
      WX_MM(acs, 2, 0) WX_MM(acs, 1, 1) WX_MM(acs, 0, 2) WX_MM(acs, 1, 0) WX_MM(acs, 0, 1) WX_MM(acc, 0, 0)
#undef WX_MM
    }
  }
  // partials: D row (co) = mt * 16 + 4 g + r, column = this lane's weight column of tile tt
  float* sl = p.slab + (long)blockIdx.x * (32 * 256 + 32);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(mt * 16 + 4 * g + r) * 256 + ncol[tt]] = acc[mt][tt][r] + acs[mt][tt][r];
  __syncthreads();
  float* red = reinterpret_cast<float*>(scrw);          // [16 parts][32]
  red[bpart * 32 + bco] = dbacc;
  __syncthreads();
  if (tid < 32) {
    float sdb = 0.f;
    for (int q = 0; q < 16; ++q) sdb += red[q * 32 + tid];
    sl[32 * 256 + tid] = sdb;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// wgrad_x6p_kernel, OPT-IN (A2C_WGRAD_X6=2): built to run the conversion of sample n + 1 UNDER the matrix phase of sample n,
// measured, and no faster -- 0.48-0.50 ms against 0.46 for wgrad_x6_kernel (same box, tools/wgrad_x6_timing.py): matrix phase
// alone 0.31, conversion alone 0.21-0.25, together 0.48 whether the two waves of a SIMD run them in opposite order or in the
// same order.  The phases do not compete for the matrix pipe and the vector ALU but for the LDS: ~390 ds_write_b32 / b16 per
// sample at 4 cycles each on the store path beside ~650 fragment reads, 40 % of whose LDS cycles are bank conflicts.  Kept for
// the record and for the next step there (wider stores, a conflict-free run pitch); dW is bit-identical to wgrad_x6_kernel.
// wgrad_x6_kernel runs its two phases in sequence (conversion 0.22 ms + matrix phase 0.31 of its 0.47-0.51 ms): both waves of a
// SIMD convert, then both multiply.  Here the piece images exist twice (two sets of 72 KB: phase runs at a 20-byte pitch --
// fragments are four or five ds_read_b32 --, the corner cells as single dwords) and there is no fp32 scratch: a thread converts
// straight from the float4s it prefetched (the next sample's, in registers during the whole interval).  Per sample ONE barrier;
// between two barriers every wave multiplies sample n out of set n & 1 and converts sample n + 1 into the other set -- waves
// 0-3 multiply first, waves 4-7 convert first, and waves w and w + 4 share a SIMD: its matrix pipe and its vector ALU work at
// the same time.  Same MFMA order per accumulator as wgrad_x6_kernel: dW is bit-identical to it (db: sums in another order).
namespace wxp {
constexpr int NT = 512;
constexpr int PQ = 12800, PRUN = 20, P0 = 0;                       // [q][run 640][10 el]
constexpr int C2_0 = P0 + 3 * PQ, CQ = 4096;                       // [q][n 256][8 el]
constexpr int X2_0 = C2_0 + 3 * CQ, XQ = 1024;                     // [q][n 256] one dword: the corner element, 0
constexpr int A_0 = X2_0 + 3 * XQ, AQ = 6656, AROW = 208;          // [q][co 32][104 el]
constexpr int SET = A_0 + 3 * AQ;                                  // 73,728
constexpr int LDS_BYTES = 2 * SET;
static_assert(SET == 73728 && SET % 16 == 0, "set");
}  // namespace wxp
template <bool RANK>
__global__ __launch_bounds__(wxp::NT) void wgrad_x6p_kernel(WgradX6P p) {
  using namespace wxp;
  extern __shared__ __attribute__((aligned(16))) unsigned char ldsp[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const int wq = w & 3, th = w >> 2;                   // channel pair 2 wq + th; th also = "converts first"
  for (int i = tid; i < LDS_BYTES / 16; i += NT) *reinterpret_cast<u32x4x*>(ldsp + i * 16) = (u32x4x){0u, 0u, 0u, 0u};
  // ---- matrix-phase addresses of this lane (within a set)
  int ncol[2];
  const int ci_l = 2 * (2 * wq + th) + (l16 >> 3), ky_l = (l16 >> 1) & 3, kxl_l = l16 & 1;
#pragma unroll
  for (int tt = 0; tt < 2; ++tt) ncol[tt] = (ci_l * 4 + ky_l) * 4 + 2 * tt + kxl_l;
  const int prow0 = P0 + ((ci_l * 20 + 2 * g + ky_l) * 2 + kxl_l) * PRUN;      // step 0: output row oy = g (step 1: + 16 runs)
  const int prow8 = P0 + ((ci_l * 20 + 16 + ky_l) * 2 + kxl_l) * PRUN;         // step 2, g = 0: output row 8
  const int aaddr = A_0 + l16 * AROW + g * 16;
  f32x4 acc[2][2], acs[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) acc[mt][tt] = acs[mt][tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // ---- conversion role: regular float4s r = tid + 512 j of the sample's a1 (columns 0..15 of every row: j < 3, the third only
  // for tid < 256), edge float4s d = tid - 192 (columns 16..19 of row (ci, y) = (d / 20, d % 20): tid >= 192), dOut elements
  // e = tid + 512 j (j < 6)
  int rsrc[3], rdst[3];                                // float offset in the sample; byte offset of the par-0 dword in a set
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int r = min(tid + 512 * j, 1279), ci = r / 80, rem = r - ci * 80, y = rem >> 2, f = rem & 3;
    rsrc[j] = (ci * 20 + y) * 20 + 4 * f;
    rdst[j] = P0 + ((ci * 20 + y) * 2) * PRUN + 4 * f;
  }
  const bool edge = tid >= 192;
  const int ed = edge ? tid - 192 : 0, eci = ed / 20, ey = ed - eci * 20;
  const int esrc = (eci * 20 + ey) * 20 + 16, edst = P0 + ((eci * 20 + ey) * 2) * PRUN + 16;
  f32x4 pa[3], pe;
  float pd[6];
  float wc[RANK ? 6 : 1][4], gl[4] = {0.f, 0.f, 0.f, 0.f};
  unsigned int mbv[RANK ? 6 : 1];
  float dbs[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (RANK) {
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
      for (int n = 0; n < 4; ++n) wc[RANK ? j : 0][n] = (tid + 512 * j < 2592 && n < p.nlog) ? p.Wc[(long)n * 2592 + tid + 512 * j] : 0.f;
  }
  auto load_raw = [&](long bn) {
    const float* __restrict__ ga = p.in + bn * p.in_bs;
#pragma unroll
    for (int j = 0; j < 3; ++j)
      if (j < 2 || tid < 256) pa[j] = *reinterpret_cast<const f32x4*>(ga + rsrc[j]);
    if (edge) pe = *reinterpret_cast<const f32x4*>(ga + esrc);
    if (RANK) {
#pragma unroll
      for (int n = 0; n < 4; ++n) gl[n] = n < p.nlog ? p.dl[bn * p.ldl + n] : 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) mbv[RANK ? j : 0] = p.a2b[bn * p.a2b_row + (min(tid + 512 * j, 2591) >> 3)];
    } else {
      const float* __restrict__ gd = p.dout + bn * 2592L;
#pragma unroll
      for (int j = 0; j < 6; ++j) pd[j] = gd[min(tid + 512 * j, 2591)];
    }
  };
  auto convert = [&](unsigned char* __restrict__ set) {
    // a1: a float4 (x0 .. x0 + 3) is one dword of the parity-0 run (x0, x0 + 2) and one of the parity-1 run (x0 + 1, x0 + 3)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      if (j < 2 || tid < 256) {
        unsigned int o0[3], o1[3];
        wx_split2(pa[j][0], pa[j][2], o0);
        wx_split2(pa[j][1], pa[j][3], o1);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          *reinterpret_cast<unsigned int*>(set + q * PQ + rdst[j]) = o0[q];
          *reinterpret_cast<unsigned int*>(set + q * PQ + rdst[j] + PRUN) = o1[q];
        }
      }
    }
    if (edge) {      // columns 16..19 = taps kx 0..3 of the column ox = 8: also the column cells (oy <= 7) and the corner (oy = 8)
      unsigned int o0[3], o1[3];
      wx_split2(pe[0], pe[2], o0);
      wx_split2(pe[1], pe[3], o1);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        *reinterpret_cast<unsigned int*>(set + q * PQ + edst) = o0[q];
        *reinterpret_cast<unsigned int*>(set + q * PQ + edst + PRUN) = o1[q];
      }
#pragma unroll
      for (int kh = 0; kh < 2; ++kh) {
        const int ky = (ey & 1) + 2 * kh, dd = ey - ky, oy = dd >> 1;
        if (dd >= 0 && oy <= 8) {
          const int n0 = (eci * 4 + ky) * 4;
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const unsigned short v0 = (unsigned short)(o0[q] & 0xffffu), v2 = (unsigned short)(o0[q] >> 16);
            const unsigned short v1 = (unsigned short)(o1[q] & 0xffffu), v3 = (unsigned short)(o1[q] >> 16);
            if (oy <= 7) {
              unsigned short* c = reinterpret_cast<unsigned short*>(set + C2_0 + q * CQ + n0 * 16) + oy;
              c[0] = v0; c[8] = v1; c[16] = v2; c[24] = v3;
            } else {
              unsigned short* x = reinterpret_cast<unsigned short*>(set + X2_0 + q * XQ + n0 * 4);
              x[0] = v0; x[2] = v1; x[4] = v2; x[6] = v3;
            }
          }
        }
      }
    }
    // dOut: element e = (co, oy, ox) -> slot 8 oy + ox (ox < 8), 72 + oy (ox = 8, oy < 8), 80 (the corner)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int e = tid + 512 * j;
      float a;
      if (RANK) {
        a = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) a += gl[n] * wc[RANK ? j : 0][n];
        if (!((mbv[RANK ? j : 0] >> (e & 7)) & 1u)) a = 0.f;
      } else {
        a = pd[j];
      }
      if (e < 2592) {
        dbs[j] += a;
        const int co = e / 81, px = e - co * 81, oy = px / 9, ox = px - oy * 9;
        const int slot = ox < 8 ? 8 * oy + ox : (oy < 8 ? 72 + oy : 80);
        const __bf16 h0 = (__bf16)a;
        const float r1 = a - (float)h0;
        const __bf16 h1 = (__bf16)r1;
        const float r2 = r1 - (float)h1;
        unsigned short* dst = reinterpret_cast<unsigned short*>(set + A_0 + co * AROW) + slot;
        dst[0] = __builtin_bit_cast(unsigned short, h0);
        dst[AQ / 2] = __builtin_bit_cast(unsigned short, h1);
        dst[AQ] = __builtin_bit_cast(unsigned short, (__bf16)r2);
      }
    }
  };
  auto matrix = [&](const unsigned char* __restrict__ set) {
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      bf16x8x a[2][3], bf[2][3];
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int q = 0; q < 3; ++q)
          a[mt][q] = *reinterpret_cast<const bf16x8x*>(set + aaddr + mt * 16 * AROW + s * 64 + q * AQ);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const unsigned int* pr = reinterpret_cast<const unsigned int*>(set + q * PQ + (s == 0 ? prow0 : s == 1 ? prow0 + 16 * PRUN : (g == 0 ? prow8 : prow0)));
          const unsigned int d0 = pr[0], d1 = pr[1], d2 = pr[2], d3 = pr[3];
          u32x4x v = (u32x4x){d0, d1, d2, d3};
          if (tt == 1) {                                // kxh = 1: the run one element further
            const unsigned int d4 = pr[4];
            v = (u32x4x){__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                         __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)};
          }
          if (s == 2) {                                 // lanes g = 1: the column cells, g = 2: the corner, g = 3: nothing
            const u32x4x vc = *reinterpret_cast<const u32x4x*>(set + C2_0 + q * CQ + ncol[tt] * 16);
            const unsigned int vx = *reinterpret_cast<const unsigned int*>(set + X2_0 + q * XQ + ncol[tt] * 4);
            if (g == 1) v = vc;
            else if (g == 2) v = (u32x4x){vx, 0u, 0u, 0u};
            else if (g == 3) v = (u32x4x){0u, 0u, 0u, 0u};
          }
          bf[tt][q] = __builtin_bit_cast(bf16x8x, v);
        }
      }
#define WXP_MM(AC, QA, QB)                                                                                              \
      _Pragma("unroll") for (int mt = 0; mt < 2; ++mt)                                                                  \
        _Pragma("unroll") for (int tt = 0; tt < 2; ++tt)                                                                \
          AC[mt][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][QA], bf[tt][QB], AC[mt][tt], 0, 0, 0);
      WXP_MM(acs, 2, 0) WXP_MM(acs, 1, 1) WXP_MM(acs, 0, 2) WXP_MM(acs, 1, 0) WXP_MM(acs, 0, 1) WXP_MM(acc, 0, 0)
#undef WXP_MM
    }
  };
  long b = blockIdx.x;
  if (b < p.B) {
    __syncthreads();                                   // the zero fill
    load_raw(b);
    convert(ldsp);
    if (b + gridDim.x < p.B) load_raw(b + gridDim.x);
    __syncthreads();
  }
  int cur = 0;
  for (; b < p.B; b += gridDim.x) {
    const bool nxt = b + gridDim.x < p.B;
    const long b2 = b + 2L * gridDim.x;
    unsigned char* __restrict__ sc = ldsp + cur * SET;
    unsigned char* __restrict__ sn = ldsp + (cur ^ 1) * SET;
    const bool do_c = nxt && !(p.dbg & 1), do_l = b2 < p.B && !(p.dbg & 8), do_m = !(p.dbg & 4);
    if (th == 0 || (p.dbg & 16)) {                     // (dbg 16: every wave multiplies first -- the in-phase order, for timing)
      if (do_m) matrix(sc);
      if (do_c) convert(sn);
      if (nxt && do_l) load_raw(b2);
    } else {
      if (do_c) convert(sn);
      if (nxt && do_l) load_raw(b2);
      if (do_m) matrix(sc);
    }
    __syncthreads();
    cur ^= 1;
  }
  // partials: D row (co) = mt * 16 + 4 g + r, column = this lane's weight column of tile tt
  float* sl = p.slab + (long)blockIdx.x * (32 * 256 + 32);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sl[(mt * 16 + 4 * g + r) * 256 + ncol[tt]] = acc[mt][tt][r] + acs[mt][tt][r];
  float* red = reinterpret_cast<float*>(ldsp);          // [e 3072]: this workgroup's sum over its samples of dOut element e
#pragma unroll
  for (int j = 0; j < 6; ++j) red[tid + 512 * j] = dbs[j];
  __syncthreads();
  if (tid < 32) {
    float sdb = 0.f;
    for (int i = 0; i < 81; ++i) sdb += red[tid * 81 + i];
    sl[32 * 256 + tid] = sdb;
  }
}

// "Streaming" backward-weight kernel for the 8x8 / stride 4 layer on 4 input planes (A3CModel
// conv1), the twin of conv_stream_kernel: one persistent 8-wave workgroup per CU, the input sample
// and its dOut in LDS, the NEXT sample's 18 float4 per thread in flight in registers (issued in
// three straight-line instalments between the row segments of the matrix phase).  Wave (q, h):
// k-group q = 64 of the 256 weight columns (4 accumulators), output rows of parity h; the two
// parities are added in a fixed order at the end, per-workgroup slabs go to wgrad_reduce_kernel.
// dW[co][k] += sum_px dOut[co][px] * x[k][px]:  A = dOut (16 co x 4 px), B = image (4 px x 16 k).
struct WstreamP {
  const float* in; long in_bs;
  const float* dout;              // (B, Cout, OH, OW) contiguous
  float* slab;                    // [grid][Cout*K + Cout]
  int B, H, W, OH, OW, Cout, K, PLANE1, PLANEo, WP;
  // U8: the input comes from the single-frame uint8 store of the rollout instead of the stacked fp32 states:
  // sample n = (slot r, step t) is the CONTIGUOUS window of 4 frames fstore[r][t .. t+3] (T+4 frames per slot),
  // planes older than the env's last reset (c < 4 - nvalid[n]) are zero (utils.py:37-42)
  const unsigned char* fstore; long fs_slot_stride; int T; const int* nvalid;
};
typedef unsigned int u32x4w __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ws_u8x4(unsigned int w) {
  return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
}
// one DWORD (4 pixels) per lane and load: consecutive lanes then write consecutive float4 of the LDS image (the first
// version loaded 16 pixels per lane: its four float4 writes per lane were 64 B apart between lanes -- 4-way bank conflicts
// in the commit phase every wave waits on: 1.41 vs 1.27 ms for the fp32 source)
#define WS_LDU(var, u, src) var = reinterpret_cast<const unsigned int*>(src)[min(tid + (u) * ST_NT, tot4 - 1)];
#define WS_STU(var, u, nv)                                                                     \
  {                                                                                            \
    const int idx_ = min(tid + (u) * ST_NT, tot4 - 1);                                          \
    const int c_ = (idx_ >= per4) + (idx_ >= 2 * per4) + (idx_ >= 3 * per4);                   \
    const float4 e_ = ws_u8x4(var);                                                            \
    *reinterpret_cast<float4*>(img + c_ * p.PLANE1 + ((idx_ - c_ * per4) << 2)) =               \
        c_ < 4 - (nv) ? make_float4(0.f, 0.f, 0.f, 0.f) : e_;                                  \
  }
#define WS_LDI(var, u, src) var = *reinterpret_cast<const float4*>((src) + (min(tid + (u) * ST_NT, tot4 - 1) << 2));
#define WS_LDD(var, u, src) var = *reinterpret_cast<const float4*>((src) + (min(tid + (u) * ST_NT, dtot4 - 1) << 2));
#define WS_STI(var, u)                                                                         \
  {                                                                                            \
    const int idx_ = min(tid + (u) * ST_NT, tot4 - 1);                                          \
    const int c_ = (idx_ >= per4) + (idx_ >= 2 * per4) + (idx_ >= 3 * per4);                   \
    const int rem_ = idx_ - c_ * per4, row_ = rem_ / w4;                                       \
    *reinterpret_cast<float4*>(img + c_ * p.PLANE1 + row_ * p.WP + ((rem_ - row_ * w4) << 2)) = var; \
  }
#define WS_STD(var, u)                                                                         \
  {                                                                                            \
    const int idx_ = min(tid + (u) * ST_NT, dtot4 - 1);                                         \
    const int co_ = idx_ / dper4;                                                              \
    float* d_ = ldo + co_ * p.PLANEo + ((idx_ - co_ * dper4) << 2);        /* rows 2 mod 32 floats apart: 8 B aligned */ \
    *reinterpret_cast<float2*>(d_) = make_float2(var.x, var.y);                                \
    *reinterpret_cast<float2*>(d_ + 2) = make_float2(var.z, var.w);                            \
  }

template <bool U8>
__global__ __launch_bounds__(ST_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_stream_kernel(WstreamP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* __restrict__ img = lds;
  float* __restrict__ ldo = lds + 4 * p.PLANE1;       // dOut of the sample: [16][PLANEo], rows >= Cout stay 0
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kk = lane >> 4, j = lane & 15;
  const int q = w & 3, h = w >> 2;
  const int HW = p.H * p.W, W = p.W, NP = p.OH * p.OW;
  const int per4 = HW >> 2, tot4 = 4 * per4, w4 = W >> 2;
  const int dper4 = NP >> 2, dtot4 = p.Cout * dper4;
  for (int i = tid; i < 16 * p.PLANEo; i += ST_NT) ldo[i] = 0.f;
  // this lane's weight column group: (ci, ky, kxh), the 4 accumulators are kxl = 0..3
  const int cidx = q * 16 + j;
  const int kxh = cidx & 1, cy = cidx >> 1;
  const int ci = cy >> 3, ky = cy & 7;
  const int boff = ci * p.PLANE1 + ky * p.WP + kxh * 4 + kk * 4;
  const int knat = (ci * 8 + ky) * 8 + kxh * 4;
  f32x4 acc0 = (f32x4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  const int bco = tid & 15, bpart = tid >> 4;        // bias gradient: 32 partial sums per channel
  float dbacc = 0.f;
  float4 v0 = {}, v1 = {}, v2 = {}, v3 = {}, v4 = {}, v5 = {}, v6 = {}, v7 = {}, v8 = {}, v9 = {}, v10 = {}, v11 = {}, v12 = {}, v13 = {};
  float4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
  unsigned int g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, g5 = 0, g6 = 0, g7 = 0, g8 = 0, g9 = 0, g10 = 0, g11 = 0, g12 = 0, g13 = 0;
  int nv = 4;
  auto fsrc = [&](long nn_) { const long r_ = nn_ / p.T; return p.fstore + r_ * p.fs_slot_stride + (nn_ - r_ * p.T) * (long)HW; };
  long n = blockIdx.x;
  if (n < p.B) {
    const float* __restrict__ src = p.in + n * p.in_bs;
    const float* __restrict__ dsrc = p.dout + n * (long)p.Cout * NP;
    WS_LDD(d0, 0, dsrc) WS_LDD(d1, 1, dsrc) WS_LDD(d2, 2, dsrc) WS_LDD(d3, 3, dsrc)
    if (U8) {
      const unsigned char* __restrict__ us = fsrc(n);
      nv = p.nvalid[n];
      WS_LDU(g0, 0, us) WS_LDU(g1, 1, us) WS_LDU(g2, 2, us) WS_LDU(g3, 3, us) WS_LDU(g4, 4, us) WS_LDU(g5, 5, us) WS_LDU(g6, 6, us)
      WS_LDU(g7, 7, us) WS_LDU(g8, 8, us) WS_LDU(g9, 9, us) WS_LDU(g10, 10, us) WS_LDU(g11, 11, us) WS_LDU(g12, 12, us) WS_LDU(g13, 13, us)
    } else {
    WS_LDI(v0, 0, src) WS_LDI(v1, 1, src) WS_LDI(v2, 2, src) WS_LDI(v3, 3, src) WS_LDI(v4, 4, src) WS_LDI(v5, 5, src) WS_LDI(v6, 6, src)
    WS_LDI(v7, 7, src) WS_LDI(v8, 8, src) WS_LDI(v9, 9, src) WS_LDI(v10, 10, src) WS_LDI(v11, 11, src) WS_LDI(v12, 12, src) WS_LDI(v13, 13, src)
    }
  }
  const int nrow = (p.OH - h + 1) >> 1;              // output rows h, h+2, ...
  const int s1 = nrow / 3, s2 = (2 * nrow) / 3;
  const int c4n = p.OW >> 2;
  for (; n < p.B; n += gridDim.x) {
    const long nn = (n + gridDim.x < p.B) ? n + gridDim.x : n;        // past the end: re-read this sample (discarded)
    const float* __restrict__ nsrc = p.in + nn * p.in_bs;
    const float* __restrict__ ndsrc = p.dout + nn * (long)p.Cout * NP;
    __syncthreads();                                 // everyone is done with the previous sample
    WS_STD(d0, 0) WS_STD(d1, 1) WS_STD(d2, 2) WS_STD(d3, 3)
    if (U8) {
      WS_STU(g0, 0, nv) WS_STU(g1, 1, nv) WS_STU(g2, 2, nv) WS_STU(g3, 3, nv) WS_STU(g4, 4, nv) WS_STU(g5, 5, nv) WS_STU(g6, 6, nv)
      WS_STU(g7, 7, nv) WS_STU(g8, 8, nv) WS_STU(g9, 9, nv) WS_STU(g10, 10, nv) WS_STU(g11, 11, nv) WS_STU(g12, 12, nv) WS_STU(g13, 13, nv)
    } else {
    WS_STI(v0, 0) WS_STI(v1, 1) WS_STI(v2, 2) WS_STI(v3, 3) WS_STI(v4, 4) WS_STI(v5, 5) WS_STI(v6, 6)
    WS_STI(v7, 7) WS_STI(v8, 8) WS_STI(v9, 9) WS_STI(v10, 10) WS_STI(v11, 11) WS_STI(v12, 12) WS_STI(v13, 13)
    }
    __syncthreads();
    if (bco < p.Cout) {                              // bias gradient partials from the dOut tile
      const float* __restrict__ pl = ldo + bco * p.PLANEo;
      float sb = 0.f;
      for (int i = bpart; i < NP; i += ST_NT / 16) sb += pl[i];
      dbacc += sb;
    }
#define WS_ROWS(R0, R1)                                                                                     \
    for (int rr = (R0); rr < (R1); ++rr) {                                                                  \
      const int r = 2 * rr + h;                                                                             \
      const float* __restrict__ arow = ldo + j * p.PLANEo + r * p.OW + kk;                                  \
      const float* __restrict__ brow = img + r * 4 * p.WP + boff;                                              \
      for (int c0 = 0; c0 < c4n; c0 += 5) {                                                                 \
        float av[5];                                                                                        \
        float4 bv[5];                                                                                       \
        _Pragma("unroll") for (int u = 0; u < 5; ++u) {                                                     \
          const int c4 = min(c0 + u, c4n - 1);                                                              \
          av[u] = (c0 + u < c4n) ? arow[4 * c4] : 0.f;                                                      \
          bv[u] = *reinterpret_cast<const float4*>(brow + 16 * c4);                                         \
        }                                                                                                   \
        _Pragma("unroll") for (int u = 0; u < 5; ++u) {                                                     \
          acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].x, acc0, 0, 0, 0);                       \
          acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].y, acc1, 0, 0, 0);                       \
          acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].z, acc2, 0, 0, 0);                       \
          acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u].w, acc3, 0, 0, 0);                       \
        }                                                                                                   \
      }                                                                                                     \
    }
    if (U8) {
      const unsigned char* __restrict__ nus = fsrc(nn);
      nv = p.nvalid[nn];
      WS_LDD(d0, 0, ndsrc) WS_LDD(d1, 1, ndsrc) WS_LDD(d2, 2, ndsrc) WS_LDD(d3, 3, ndsrc) WS_LDU(g0, 0, nus) WS_LDU(g1, 1, nus)
      WS_ROWS(0, s1)
      WS_LDU(g2, 2, nus) WS_LDU(g3, 3, nus) WS_LDU(g4, 4, nus) WS_LDU(g5, 5, nus) WS_LDU(g6, 6, nus) WS_LDU(g7, 7, nus)
      WS_ROWS(s1, s2)
      WS_LDU(g8, 8, nus) WS_LDU(g9, 9, nus) WS_LDU(g10, 10, nus) WS_LDU(g11, 11, nus) WS_LDU(g12, 12, nus) WS_LDU(g13, 13, nus)
      WS_ROWS(s2, nrow)
    } else {
    WS_LDD(d0, 0, ndsrc) WS_LDD(d1, 1, ndsrc) WS_LDD(d2, 2, ndsrc) WS_LDD(d3, 3, ndsrc) WS_LDI(v0, 0, nsrc) WS_LDI(v1, 1, nsrc)
    WS_ROWS(0, s1)
    WS_LDI(v2, 2, nsrc) WS_LDI(v3, 3, nsrc) WS_LDI(v4, 4, nsrc) WS_LDI(v5, 5, nsrc) WS_LDI(v6, 6, nsrc) WS_LDI(v7, 7, nsrc)
    WS_ROWS(s1, s2)
    WS_LDI(v8, 8, nsrc) WS_LDI(v9, 9, nsrc) WS_LDI(v10, 10, nsrc) WS_LDI(v11, 11, nsrc) WS_LDI(v12, 12, nsrc) WS_LDI(v13, 13, nsrc)
    WS_ROWS(s2, nrow)
    }
  }
  // epilogue: add the two row parities in a fixed order, write this workgroup's slab
  __syncthreads();
  float* __restrict__ scr = lds;                      // [4 k-groups][4 acc][256]
  if (h == 1) {
    *reinterpret_cast<float4*>(scr + ((q * 4 + 0) * 64 + lane) * 4) = (float4){acc0[0], acc0[1], acc0[2], acc0[3]};
    *reinterpret_cast<float4*>(scr + ((q * 4 + 1) * 64 + lane) * 4) = (float4){acc1[0], acc1[1], acc1[2], acc1[3]};
    *reinterpret_cast<float4*>(scr + ((q * 4 + 2) * 64 + lane) * 4) = (float4){acc2[0], acc2[1], acc2[2], acc2[3]};
    *reinterpret_cast<float4*>(scr + ((q * 4 + 3) * 64 + lane) * 4) = (float4){acc3[0], acc3[1], acc3[2], acc3[3]};
  }
  float* __restrict__ red = lds + 4096;               // [32][16] bias partials
  red[bpart * 16 + bco] = dbacc;
  __syncthreads();
  float* __restrict__ sl = p.slab + (long)blockIdx.x * ((long)p.Cout * p.K + p.Cout);
  if (h == 0) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const float4 o = *reinterpret_cast<const float4*>(scr + ((q * 4 + x) * 64 + lane) * 4);
      const f32x4 a = x == 0 ? acc0 : x == 1 ? acc1 : x == 2 ? acc2 : acc3;
      const float t[4] = {a[0] + o.x, a[1] + o.y, a[2] + o.z, a[3] + o.w};
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int co = 4 * kk + rr;
        if (co < p.Cout) sl[(long)co * p.K + knat + x] = t[rr];
      }
    }
  }
  if (tid < p.Cout) {
    float sb = 0.f;
    for (int x = 0; x < ST_NT / 16; ++x) sb += red[x * 16 + tid];
    sl[(long)p.Cout * p.K + tid] = sb;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient from the single-frame uint8 store on the BF16 matrix pipe, with fp32 results (round 5).
// dW[co][k] = sum_px dOut[co][px] * x[k][px] where x are uint8 frame pixels (pong_prep / breakout_prep outputs,
// preprocessing.py:8-23): integers of at most 8 significant bits, EXACT in bf16.  Every fp32 dOut value is split, when its
// sample is committed to LDS, into three bf16 pieces d = hi + mid + lo (round to nearest each; exact: 3 x 8 >= 24 bits), so
// d * x = hi*x + mid*x + lo*x with every product exact and every sum in the MFMA's fp32 accumulator: the fp32 sum, re-
// associated.  v_mfma_f32_16x16x32_bf16 takes 32 pixels per instruction at ~16 cycles where the fp32 form takes 4 at 32:
// 3/16 of the matrix time of the kernel above (1.2 ms = 0.57 of the fp32 peak at N = 32,768).
//   A = dOut pieces [piece][co][pixel group of 8][8] bf16 (rows of OW pixels padded to whole groups, pads zero),
//   B = the input as PHASE planes P[ci][kxl][row][m] = x[ci][row][4 m + kxl] bf16: the 8 consecutive output pixels of a
//       group at tap column kx = 4 kxh + kxl are the 8 CONSECUTIVE elements m = c0 + kxh .. of one phase-plane row (two
//       ds_read_b64 + one b32; kxh = 1 shifts by one element with v_alignbit) -- no gather, no conversion in the loop;
//   wave (q, h): input plane ci = q (64 weight columns = 4 accumulator tiles kxl), K blocks of parity half h.
struct WsbGeo { int NG, NGT, NB, PA, PP, PLS, SK, dbg; };      // PLS: elements per phase plane, SK: skew per 4 rows      // dbg (A2C_WSB_DBG, timing only): 1 = no matrix phase, 2 = no commit
//   wave w: K blocks {w, w + 8, ...} (a block = 4 pixel groups = 32 pixels) for ALL 256 weight columns -- 16 accumulator
//   tiles, the block's three A fragments read once --; tile t = (ci, kxh, kp): lane j = (ky = j >> 1, kxl = 2 kp + (j & 1)),
//   so kxh is uniform per tile and only the kxh = 1 tiles pay the one-element shift; the eight waves' partial tiles are
//   added in wave order through LDS at the end of the launch.
template <int DUMMY>
__global__ __launch_bounds__(ST_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_stream_bf16_kernel(WstreamP p, WsbGeo gq) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  typedef __bf16 bf16x8w __attribute__((ext_vector_type(8)));
  unsigned short* __restrict__ A = reinterpret_cast<unsigned short*>(lds);                 // [3][16][PA]
  unsigned short* __restrict__ P = A + 3 * 16 * gq.PA;                                      // [16][PLS]: row y at y * PP + (y >> 2) * SK
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int HW = p.H * p.W, NP = p.OH * p.OW;
  const int w4 = p.W >> 2;
  const int dper4 = NP >> 2, dtot4 = p.Cout * dper4;
  const int NG = gq.NG, NGT = gq.NGT, NB = gq.NB, PA = gq.PA, PP = gq.PP, PLS = gq.PLS, SK = gq.SK;
  const int spr = (w4 + 1) >> 1, nslot = 4 * p.H * spr;        // frame slots: (plane, row, pair of dwords m = 2 s, 2 s + 1)
  {  // zeros that stay: pad pixels / pad groups / channels >= Cout of A, columns m >= W/4 of P
    u32x4w* z = reinterpret_cast<u32x4w*>(lds);
    const int n16 = (3 * 16 * PA * 2 + 16 * PLS * 2) >> 4;
    for (int i = tid; i < n16; i += ST_NT) z[i] = (u32x4w){0u, 0u, 0u, 0u};
  }
  const int ky = j >> 1;
  f32x4 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbv[4] = {0.f, 0.f, 0.f, 0.f};           // bias gradient: this thread's dOut quads belong to fixed channels
  // (ext vectors, not HIP's float4 STRUCT: the struct is split into four scalars, the register allocator scatters the
  // loop-carried scalars, and the loop then ends in `s_waitcnt vmcnt(0)` + v_mov copies of the quad a load has to write
  // contiguously -- a wait for the prefetch right behind its issue, in every version of this kernel up to round 5)
  f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
  uint2 g0 = {}, g1 = {}, g2 = {}, g3 = {}, g4 = {}, g5 = {}, g6 = {}, g7 = {};
  int nv = 4;
  auto fsrc = [&](long nn_) { const long r_ = nn_ / p.T; return p.fstore + r_ * p.fs_slot_stride + (nn_ - r_ * p.T) * (long)HW; };
  // slot u of this thread: plane c, row, dword pair s -> byte offset in the sample's 4 planes (dword aligned)
  int soff[8];
  int sdst[8];                                   // element offset of (phase 0, row, m = 2 s) in P, or -1
  unsigned int smask[8];                         // second dword of the pair lies inside the row
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = tid + u * ST_NT;
    const bool ok = idx < nslot;
    const int ii = ok ? idx : 0;
    const int c = ii / (p.H * spr), rem = ii - c * (p.H * spr), row = rem / spr, sp = rem - row * spr;
    soff[u] = (c * p.H + row) * p.W + 8 * sp;
    sdst[u] = ok ? (c * 4) * PLS + row * PP + (row >> 2) * SK + 2 * sp : -1;
    smask[u] = ((2 * sp + 1 < w4) ? 0xffffff00u : 0u) | (unsigned int)c;      // low byte: the plane (zero planes of a fresh episode)
  }
  // THE PLANES STAY IN LDS (round 6).  Sample n + 1 of a slot is sample n with its oldest plane dropped and one new frame
  // (utils.py:26-43), and a workgroup walks a contiguous run of samples: the four plane slots of P are a RING -- logical
  // plane c of the current sample lives in slot (c + rot) & 3 --, a successor sample loads and commits ONE frame (two slots
  // per thread instead of eight: 7 KB instead of 28 KB through the CU's load path, a quarter of the byte -> bf16 commit), and
  // the matrix phase reads plane c at pofs[c].  A sample that starts a slot or a run, or whose state has fewer than four
  // real planes (a fresh episode: nvalid < 4), or whose predecessor had fewer than three, is loaded and committed whole, as
  // before.  Same sums in the same order: bit-identical to A2C_WSB_NO_RING=1 (test_gpu_frames.py).
  int soff2[2], sdst2[2];                        // the NEWEST plane only: slot u = (row, dword pair) of logical plane 3
  unsigned int smask2[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int idx = tid + u * ST_NT;
    const bool ok = idx < p.H * spr;
    const int ii = ok ? idx : 0;
    const int row = ii / spr, sp = ii - row * spr;
    soff2[u] = (3 * p.H + row) * p.W + 8 * sp;
    sdst2[u] = ok ? row * PP + (row >> 2) * SK + 2 * sp : -1;
    smask2[u] = (2 * sp + 1 < w4) ? 1u : 0u;
  }
  int rot = 0;                                   // slot of logical plane 0
  int pofs0 = 0, pofs1 = 4 * PLS, pofs2 = 8 * PLS, pofs3 = 12 * PLS;     // element offset of logical plane c's four phase planes
  int adst[4];                                   // element offset of dOut quad u in piece image 0, or -1 (no division in the loop)
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int idx = tid + u * ST_NT;
    const bool ok = idx < dtot4;
    const int ii = ok ? idx : 0;
    const int co = ii / dper4, p0 = (ii - co * dper4) << 2;
    const int r = p0 / p.OW, c = p0 - r * p.OW;
    adst[u] = ok ? co * PA + (r * NG + (c >> 3)) * 8 + (c & 7) : -1;
  }
  // Samples in CONTIGUOUS runs per workgroup (round 6; was n = blockIdx.x + k * gridDim.x): sample n + 1 of a slot shares
  // three of its four frames with sample n, so the workgroup that just fetched them finds them in its XCD's L2 -- with the
  // strided deal the four samples that share a frame ran on four different XCDs and every frame crossed the fabric four times
  // (PMC: 1.77 GB per launch against 1.07 GB of frames + dOut).  A2C_WSB_STRIDED=1 (gq.dbg bit 2) keeps the strided deal.
  const bool strided = (gq.dbg & 4) != 0;
  const bool ring_off = strided || (gq.dbg & 8) != 0;          // A2C_WSB_NO_RING=1: every sample loaded and committed whole
  const long chunk = (p.B + gridDim.x - 1) / gridDim.x;
  const long nstep = strided ? (long)gridDim.x : 1L;
  long n = strided ? (long)blockIdx.x : (long)blockIdx.x * chunk;
  const long n_end = strided ? (long)p.B : (n + chunk < (long)p.B ? n + chunk : (long)p.B);
#define WSB_LDU(var, u, src)                                                                                   \
  {                                                                                                            \
    const unsigned int* s_ = reinterpret_cast<const unsigned int*>((src) + soff[u]);                            \
    var.x = s_[0];                                                                                             \
    var.y = s_[(smask[u] >> 8) ? 1 : 0];                                                                       \
  }
  // dOut rows are prefetched TWO samples ahead (two register sets d / e, the loop unrolled twice), the frames one: stamps of
  // the one-deep version (A2C_WSB_DBG=1/2/3, tools/a3c_wgrad_bench.py) -- loads alone 0.27 ms, + matrix 0.18, + commit 0.15 =
  // the whole kernel: a load issued at the head of a 1.4 us matrix phase comes back after ~3.5 us under this kernel's own
  // bursts, so every sample waited ~2 us for its dOut.  (The frames of sample n + 1 are mostly L2 hits since the contiguous
  // deal: one sample of lookahead covers them.)
#define WSB_LOADD(NN, D0, D1, D2, D3)                                                                          \
  {                                                                                                            \
    const float* __restrict__ ds_ = p.dout + (NN) * (long)p.Cout * NP;                                         \
    D0 = *reinterpret_cast<const f32x4*>(ds_ + (min(tid + 0 * ST_NT, dtot4 - 1) << 2));                        \
    D1 = *reinterpret_cast<const f32x4*>(ds_ + (min(tid + 1 * ST_NT, dtot4 - 1) << 2));                        \
    D2 = *reinterpret_cast<const f32x4*>(ds_ + (min(tid + 2 * ST_NT, dtot4 - 1) << 2));                        \
    D3 = *reinterpret_cast<const f32x4*>(ds_ + (min(tid + 3 * ST_NT, dtot4 - 1) << 2));                        \
  }
#define WSB_LDU2(var, u, src)                                                                                  \
  {                                                                                                            \
    const unsigned int* s_ = reinterpret_cast<const unsigned int*>((src) + soff2[u]);                           \
    var.x = s_[0];                                                                                             \
    var.y = s_[smask2[u] ? 1 : 0];                                                                             \
  }
  // frames of sample NN: INC_ = it continues the sample in LDS (its newest frame only), else all four planes
#define WSB_LOADG(NN, INC_)                                                                                    \
  {                                                                                                            \
    const unsigned char* __restrict__ us_ = fsrc(NN);                                                          \
    if (INC_) {                                                                                                \
      WSB_LDU2(g0, 0, us_) WSB_LDU2(g1, 1, us_)                                                                \
    } else {                                                                                                   \
      WSB_LDU(g0, 0, us_) WSB_LDU(g1, 1, us_) WSB_LDU(g2, 2, us_) WSB_LDU(g3, 3, us_)                          \
      WSB_LDU(g4, 4, us_) WSB_LDU(g5, 5, us_) WSB_LDU(g6, 6, us_) WSB_LDU(g7, 7, us_)                          \
    }                                                                                                          \
  }
  // dOut quad u of this thread (4 consecutive pixels of one output row, OW % 4 == 0) -> the three piece images
#define WSB_STD(var, u)                                                                                        \
  if (adst[u] >= 0) {                                                                                          \
    const float e_[4] = {var.x, var.y, var.z, var.w};                                                          \
    unsigned short pc_[3][4];                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
      const __bf16 h0_ = (__bf16)e_[i_];                                                                       \
      const float r1_ = e_[i_] - (float)h0_;                                                                   \
      const __bf16 h1_ = (__bf16)r1_;                                                                          \
      const float r2_ = r1_ - (float)h1_;                                                                      \
      pc_[0][i_] = __builtin_bit_cast(unsigned short, h0_);                                                    \
      pc_[1][i_] = __builtin_bit_cast(unsigned short, h1_);                                                    \
      pc_[2][i_] = __builtin_bit_cast(unsigned short, (__bf16)r2_);                                            \
    }                                                                                                          \
    dbv[u] += (e_[0] + e_[1]) + (e_[2] + e_[3]);                                                               \
    unsigned short* a_ = A + adst[u];                                                                          \
    _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                                          \
      *reinterpret_cast<uint2*>(a_ + pc * 16 * PA) = make_uint2((unsigned int)pc_[pc][0] | ((unsigned int)pc_[pc][1] << 16), \
                                                                (unsigned int)pc_[pc][2] | ((unsigned int)pc_[pc][3] << 16)); \
  }
  // frame slot u (two dwords = pixels 4 m .. 4 m + 7 of one row) -> phase plane kxl gets [bf16(byte kxl of m) | bf16(.. of m+1)]:
  // the upper halves of the exact floats, one 4-byte LDS store per phase
#define WSB_STU(var, u)                                                                                        \
  if (sdst[u] >= 0) {                                                                                          \
    const int c_ = (int)(smask[u] & 0xffu);                                                                    \
    const unsigned int x0_ = c_ < 4 - nv ? 0u : var.x, x1_ = (c_ < 4 - nv || !(smask[u] >> 8)) ? 0u : var.y;   \
    unsigned int* d_ = reinterpret_cast<unsigned int*>(P + sdst[u]);                                           \
    const int ps_ = PLS >> 1;                                                                                  \
    d_[0] = __builtin_amdgcn_perm(__float_as_uint((float)(x1_ & 0xffu)), __float_as_uint((float)(x0_ & 0xffu)), 0x07060302u); \
    d_[ps_] = __builtin_amdgcn_perm(__float_as_uint((float)((x1_ >> 8) & 0xffu)), __float_as_uint((float)((x0_ >> 8) & 0xffu)), 0x07060302u); \
    d_[2 * ps_] = __builtin_amdgcn_perm(__float_as_uint((float)((x1_ >> 16) & 0xffu)), __float_as_uint((float)((x0_ >> 16) & 0xffu)), 0x07060302u); \
    d_[3 * ps_] = __builtin_amdgcn_perm(__float_as_uint((float)(x1_ >> 24)), __float_as_uint((float)(x0_ >> 24)), 0x07060302u); \
  }
  auto matrix = [&]() {
    for (int b = (gq.dbg & 1) ? NB : w; b < NB; b += ST_NT / 64) {
      const int grp = 4 * b + g;
      const unsigned short* __restrict__ ap = A + j * PA + grp * 8;
      const bf16x8w ah = *reinterpret_cast<const bf16x8w*>(ap);
      const bf16x8w am = *reinterpret_cast<const bf16x8w*>(ap + 16 * PA);
      const bf16x8w al = *reinterpret_cast<const bf16x8w*>(ap + 32 * PA);
      const int gcl = min(grp, NGT - 1);             // (pad groups of the last block: A is zero there, B must only be finite)
      const int r = gcl / NG, gc = gcl - r * NG;
      const unsigned short* __restrict__ bp = P + (j & 1) * PLS + (4 * r + ky) * PP + (r + (ky >> 2)) * SK + 8 * gc;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int ci = t >> 2, kxh = (t >> 1) & 1, kp = t & 1;
        const int po = ci == 0 ? pofs0 : ci == 1 ? pofs1 : ci == 2 ? pofs2 : pofs3;       // (the ring: logical plane ci's slot)
        const unsigned short* __restrict__ bx = bp + po + 2 * kp * PLS;
        const uint2 lo = *reinterpret_cast<const uint2*>(bx), hi = *reinterpret_cast<const uint2*>(bx + 4);
        u32x4w o;
        if (kxh) {
          const unsigned int nx = *reinterpret_cast<const unsigned int*>(bx + 8);
          o[0] = __builtin_amdgcn_alignbit(lo.y, lo.x, 16u);
          o[1] = __builtin_amdgcn_alignbit(hi.x, lo.y, 16u);
          o[2] = __builtin_amdgcn_alignbit(hi.y, hi.x, 16u);
          o[3] = __builtin_amdgcn_alignbit(nx, hi.y, 16u);
        } else {
          o[0] = lo.x; o[1] = lo.y; o[2] = hi.x; o[3] = hi.y;
        }
        const bf16x8w bv = __builtin_bit_cast(bf16x8w, o);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bv, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bv, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bv, acc[t], 0, 0, 0);
      }
    }
  };
  // the newest frame of a successor sample -> the slot its oldest plane just left (= logical plane 3 after the rotation)
#define WSB_STU2(var, u, PB_)                                                                                  \
  if (sdst2[u] >= 0) {                                                                                         \
    const unsigned int x0_ = var.x, x1_ = smask2[u] ? var.y : 0u;                                              \
    unsigned int* d_ = reinterpret_cast<unsigned int*>(P + (PB_) + sdst2[u]);                                  \
    const int ps_ = PLS >> 1;                                                                                  \
    d_[0] = __builtin_amdgcn_perm(__float_as_uint((float)(x1_ & 0xffu)), __float_as_uint((float)(x0_ & 0xffu)), 0x07060302u); \
    d_[ps_] = __builtin_amdgcn_perm(__float_as_uint((float)((x1_ >> 8) & 0xffu)), __float_as_uint((float)((x0_ >> 8) & 0xffu)), 0x07060302u); \
    d_[2 * ps_] = __builtin_amdgcn_perm(__float_as_uint((float)((x1_ >> 16) & 0xffu)), __float_as_uint((float)((x0_ >> 16) & 0xffu)), 0x07060302u); \
    d_[3 * ps_] = __builtin_amdgcn_perm(__float_as_uint((float)(x1_ >> 24)), __float_as_uint((float)(x0_ >> 24)), 0x07060302u); \
  }
  // may sample b_ be built from sample a_ in LDS?  the next sample of the same slot, all four of its planes real frames
  // (... and the three planes it shares with a_ are real frames THERE too: nvalid[a_] >= 3.  The rollout's counts always
  // satisfy that -- the count grows by one per step --, an arbitrary nvalid array need not.)
  auto succ = [&](long a_, long b_, int nva_, int nvb_) {
    return !ring_off && b_ == a_ + 1 && (b_ % p.T) != 0 && nvb_ == 4 && nva_ >= 3;
  };
  // one sample: commit it (its dOut set D0..D3, its frames g0..g7), then -- in flight during the matrix phase -- the dOut of
  // the sample two steps on into the set just freed and the frames of the next sample; past the end: re-read this sample
#define WSB_ITER(N_, D0, D1, D2, D3)                                                                           \
  {                                                                                                            \
    const long n2_ = ((N_) + 2 * nstep < n_end) ? (N_) + 2 * nstep : (N_);                                     \
    const long n1_ = ((N_) + nstep < n_end) ? (N_) + nstep : (N_);                                             \
    __syncthreads();                                 /* everyone is done with the previous sample (and the zero fill) */ \
    if (inc_cur) rot = (rot + 1) & 3; else rot = 0;                                                            \
    pofs0 = rot * 4 * PLS; pofs1 = ((rot + 1) & 3) * 4 * PLS; pofs2 = ((rot + 2) & 3) * 4 * PLS; pofs3 = ((rot + 3) & 3) * 4 * PLS; \
    if (!(gq.dbg & 2)) {                                                                                       \
      WSB_STD(D0, 0) WSB_STD(D1, 1) WSB_STD(D2, 2) WSB_STD(D3, 3)                                              \
      if (inc_cur) {                                                                                           \
        WSB_STU2(g0, 0, pofs3) WSB_STU2(g1, 1, pofs3)                                                          \
      } else {                                                                                                 \
        WSB_STU(g0, 0) WSB_STU(g1, 1) WSB_STU(g2, 2) WSB_STU(g3, 3) WSB_STU(g4, 4) WSB_STU(g5, 5) WSB_STU(g6, 6) WSB_STU(g7, 7) \
      }                                                                                                        \
    }                                                                                                          \
    __syncthreads();                                                                                           \
    /* frames FIRST: loads retire in order, and the next commit needs them before this dOut */                 \
    inc_cur = n1_ != (N_) && succ((N_), n1_, nv, nv1);                                                         \
    nv = nv1;                                        /* nvalid of sample n1_ (fetched one iteration ago) */     \
    WSB_LOADG(n1_, inc_cur)                                                                                    \
    nv1 = p.nvalid[n2_];                                                                                       \
    WSB_LOADD(n2_, D0, D1, D2, D3)                                                                             \
    matrix();                                                                                                  \
  }
  f32x4 e0 = {}, e1 = {}, e2 = {}, e3 = {};
  bool inc_cur = false;                              // the sample about to be committed continues the one in LDS
  int nv1 = 4;
  if (n < n_end) {
    const long n1 = (n + nstep < n_end) ? n + nstep : n;
    WSB_LOADD(n, d0, d1, d2, d3)
    nv = p.nvalid[n];
    WSB_LOADG(n, false)
    nv1 = p.nvalid[n1];
    WSB_LOADD(n1, e0, e1, e2, e3)
  }
  for (; n < n_end; n += 2 * nstep) {
    WSB_ITER(n, d0, d1, d2, d3)
    if (n + nstep < n_end) WSB_ITER(n + nstep, e0, e1, e2, e3)
  }
#undef WSB_ITER
#undef WSB_STU2
#undef WSB_LDU2
#undef WSB_LOADD
#undef WSB_LOADG
#undef WSB_LDU
#undef WSB_STD
#undef WSB_STU
  // epilogue: the eight waves' partial tiles added in wave order (four tiles per round through LDS), this workgroup's slab;
  // bias sums per channel in thread order
  float* __restrict__ sl = p.slab + (long)blockIdx.x * ((long)p.Cout * p.K + p.Cout);
  float* __restrict__ scr = lds;                      // [8 waves][4 tiles][256]
#pragma unroll
  for (int t0 = 0; t0 < 16; t0 += 4) {
    __syncthreads();
#pragma unroll
    for (int x = 0; x < 4; ++x)
      *reinterpret_cast<float4*>(scr + ((w * 4 + x) * 64 + lane) * 4) = (float4){acc[t0 + x][0], acc[t0 + x][1], acc[t0 + x][2], acc[t0 + x][3]};
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 2; ++e) {                     // 4 tiles x 256 elements = 1024 sums for 512 threads
      const int o = tid + e * ST_NT, x = o >> 8, l = (o >> 2) & 63, rr = o & 3;
      float v = scr[((0 * 4 + x) * 64 + l) * 4 + rr];
#pragma unroll
      for (int ww = 1; ww < ST_NT / 64; ++ww) v += scr[((ww * 4 + x) * 64 + l) * 4 + rr];
      const int t = t0 + x, ci = t >> 2, kxh = (t >> 1) & 1, kp = t & 1;
      const int jj = l & 15, co = 4 * (l >> 4) + rr;
      const int k = ((ci * 8 + (jj >> 1)) * 8) + 4 * kxh + 2 * kp + (jj & 1);
      if (co < p.Cout) sl[(long)co * p.K + k] = v;
    }
  }
  __syncthreads();
  float* __restrict__ red = lds;                      // [4 * ST_NT] bias partials, index = dOut quad
#pragma unroll
  for (int u = 0; u < 4; ++u) red[tid + u * ST_NT] = dbv[u];
  __syncthreads();
  if (tid < p.Cout) {
    float sb = 0.f;
    for (int i = tid * dper4; i < (tid + 1) * dper4; ++i) sb += red[i];
    sl[(long)p.Cout * p.K + tid] = sb;
  }
}

static bool plan_wstream_bf16(const a2c_conv_desc* d, const WstreamP& p, WsbGeo& gq, size_t& lds) {
  gq.NG = (d->OW + 7) / 8;
  gq.dbg = getenv("A2C_WSB_DBG") ? atoi(getenv("A2C_WSB_DBG")) & 3 : 0;
  { const char* sd = getenv("A2C_WSB_STRIDED"); if (sd && sd[0] == '1') gq.dbg |= 4; }
  { const char* sd = getenv("A2C_WSB_NO_RING"); if (sd && sd[0] == '1') gq.dbg |= 8; }
  gq.NGT = d->OH * gq.NG;
  gq.NB = (gq.NGT + 3) / 4;
  const int need_dw = gq.NB * 16;                      // dwords of one channel's groups (8 bf16 = 4 dwords each)
  int pa_dw = ((need_dw + 31) / 32) * 32 + 4;          // channel pitch = 4 (mod 32) dwords: the 8 channels of a b128 pass hit 32 banks
  if (pa_dw - 32 >= need_dw) pa_dw -= 32;
  gq.PA = pa_dw * 2;
  // phase-plane image: rows of PP = 8 NG + 8 elements (64 B for OW = 20), every 4 rows (one output-row step) skewed by 4 more,
  // planes padded by 24: the two 8-byte reads of a B fragment average 1.33 bank passes (a brute-force search over pitch / skew /
  // plane padding; rows of 8 NG + 4 elements: 2.0; SQ_LDS_BANK_CONFLICT share of the first version 0.53)
  gq.PP = gq.NG * 8 + 8;
  gq.SK = 4;
  gq.PLS = d->H * gq.PP + (d->H / 4 + 1) * gq.SK + 24;
  if (d->W / 4 > gq.NG * 8 + 1 || d->OW % 4 || d->Cout > 16) return false;
  lds = (size_t)3 * 16 * gq.PA * 2 + (size_t)16 * gq.PLS * 2;
  lds = (lds + 15) / 16 * 16;
  if (lds < 4 * (size_t)(8 * 4 * 256) || 4 * d->H * ((d->W / 4 + 1) / 2) > 8 * ST_NT) return false;   // epilogue scratch reuses the images; 8 frame slots per thread
  if ((long)d->Cout * (d->OH * d->OW / 4) > 4L * ST_NT) return false;
  (void)p;
  return lds <= 160 * 1024 && (gq.PP * 2) % 8 == 0 && (gq.PLS * 2) % 8 == 0 && 4 * (d->OH - 1) + 7 < d->H;
}

static bool plan_wstream(const a2c_conv_desc* d, WstreamP& p) {
  if (!(d->ks == 8 && d->stride == 4 && d->pad == 0 && d->Cin == 4 && d->Cout <= 16 && d->W % 4 == 0 && d->OW % 4 == 0)) return false;
  const int NP = d->OH * d->OW;
  p.H = d->H; p.W = d->W; p.OH = d->OH; p.OW = d->OW; p.Cout = d->Cout; p.K = d->Cin * 64;
  p.WP = d->W;      // (no row padding: no pitch makes the 4 x 16 lane groups of the B reads (ds_read_b128) conflict-free, measured)
  p.PLANE1 = ((d->H * p.WP + 63) / 64) * 64;
  p.PLANEo = ((NP + 31) / 32) * 32 + 2;                // A reads: lanes (channel j, pixel kk) hit banks 2j + kk: conflict-free
  if (d->H * d->W > ST_NS * ST_NT || d->Cout * NP > 4 * ST_NT * 4 || d->OH < 6) return false;
  if (4 * p.PLANE1 < 4096 + 512) return false;         // epilogue scratch reuses the image region
  return 4 * (size_t)(4 * p.PLANE1 + 16 * p.PLANEo) <= 160 * 1024;
}
static int stream_grid() {
  static const int n_cu = []() { int dev = 0, n = 0; (void)hipGetDevice(&dev);
                                 (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  return n_cu;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slab, int nslab, long per,
                                                           long nW, float* __restrict__ dW, float* __restrict__ db) {
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

struct WgradPlan { SrcTile t; int OWp, PLANEo, MT, KTW, grid, run, rs, pf, ex; size_t lds; };

// prefetching instantiations: <MT, KTW, RS, NI, ND>
#define WGRAD_PF_A 1, 3, true, 4, 16, true
/* (the <2, 9, true, 8, 8, true> prefetching instance of the 84-wide 16-channel layers is gone: it needed 20 B/lane of
 * scratch at two workgroups per CU; those layers run conv3.hip's c3w_kernel, the RS variant below is their fallback) */
#define WGRAD_PF_C 2, 4, false, 12, 8, true
#define WGRAD_PF_D 2, 4, false, 14, 8, true, 2, 1       /* 42-wide input (8-B slots), 21-wide dOut (4-B slots) */
#define WGRAD_VARIANTS(X)                                                                                      \
  if (pl.pf == 2) { X(WGRAD_PF_D); }                                                                           \
  else if (pl.pf && pl.KTW == 3) { X(WGRAD_PF_A); }                                                            \
  else if (pl.pf) { X(WGRAD_PF_C); }                                                                           \
  else if (pl.rs && pl.KTW == 3) { X(1, 3, true, 0, 0, true); }                                                \
  else if (pl.rs) { X(2, 9, true, 0, 0, true); }                                                               \
  else if (pl.MT == 1 && pl.KTW == 1) { X(1, 1, false, 0, 0, true); }                                         \
  else if (pl.MT == 1 && pl.KTW == 4) { X(1, 4); }                                                            \
  else if (pl.MT == 2 && pl.KTW == 4 && pl.ex) { X(2, 4, false, 0, 0, true); }                                \
  else if (pl.MT == 2 && pl.KTW == 4) { X(2, 4); }                                                            \
  else if (pl.KTW == 5) { X(4, 5, false, 0, 0, true); }                                                       \
  else if (pl.ex) { X(4, 7, false, 0, 0, true); }                                                              \
  else { X(4, 7); }

static const void* wgrad_fn(const WgradPlan& pl) {
#define WGRAD_FN(...) return (const void*)wgrad_kernel<__VA_ARGS__>
  WGRAD_VARIANTS(WGRAD_FN)
#undef WGRAD_FN
}

// which wgrad_run_kernel instantiation (if any) fits the layer: 1 = <1,4,1,5>, 2 = <2,2,2,3>
static int wgrad_run_variant(const a2c_conv_desc* d) {
  if (!run_layout(d) || getenv("A2C_NO_PF")) return 0;
  const int mt = ceil_div(d->Cout, 16), c4n = ceil_div(d->OW, 4), rgs = d->Cin * d->ks * 2 / 16;
  if (d->stride == 4 && mt == 1 && rgs == 4 && c4n == 5) return 1;
  if (d->stride == 2 && mt == 2 && rgs == 8 && c4n == 3) return 2;
  return 0;
}

static bool plan_wgrad(const a2c_conv_desc* d, int B, WgradPlan& pl, bool allow_run = true) {
  SrcTile& t = pl.t;
  t.Cp = d->Cin; t.IH = d->H; t.IW = d->W; t.SY = d->stride; t.SX = d->stride;
  t.sy0 = -d->pad; t.sx0 = -d->pad; t.span_y = d->ks; t.span_x = d->ks; t.PH = d->OH;
  pl.OWp = ceil_div(d->OW, 4) * 4;
  // Pixel columns are padded to a multiple of 4 (OWp) with dOut = 0; their B reads run up to
  // (OWp-OW)*S <= 12 floats past a row end, i.e. into the next row / the >= 16-float plane slack,
  // all of which hold finite values (0 * finite = 0), so the image itself is planned unpadded.
  t.PW = d->OW;
  pl.run = allow_run ? wgrad_run_variant(d) : 0;
  if (pl.run) {
    pl.MT = ceil_div(d->Cout, 16); pl.KTW = 0;
    int tph = 0;
    for (int c = 1; c <= d->OH; ++c) {
      const int tih = (c - 1) * d->stride + d->ks;
      const int plane = ((tih * d->W + 63) / 64) * 64;
      const int planeo = ((c * pl.OWp + 31) / 32) * 32 + 2;
      const long bytes = 4L * ((long)d->Cin * plane + (long)pl.MT * 16 * planeo + 64);
      if ((long)d->Cin * tih * d->W <= 256L * PF_N * 4 && (long)d->Cout * c * d->OW <= 256L * PF_D &&
          bytes <= WGRAD_LDS_BUDGET) tph = c; else break;
    }
    if (tph >= 1) {
      t.WP = t.IW; t.TPH = tph; t.TIH = (tph - 1) * d->stride + d->ks;
      t.PLANE = ((t.TIH * t.WP + 63) / 64) * 64;
      t.tiles = ceil_div(t.PH, t.TPH);
      pl.PLANEo = ((t.TPH * pl.OWp + 31) / 32) * 32 + 2;
      pl.lds = 4 * ((size_t)t.Cp * t.PLANE + (size_t)pl.MT * 16 * pl.PLANEo + 64);
      const void* k = pl.run == 1 ? (const void*)wgrad_run_kernel<1, 4, 1, 5> : (const void*)wgrad_run_kernel<2, 2, 2, 3>;
      if (pl.lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
      pl.grid = resident_grid(k, pl.lds, (long)B * t.tiles);
      return true;
    }
    pl.run = 0;
  }
  const int mt = ceil_div(d->Cout, 16);
  const int nkt = ceil_div(d->Cin * d->ks * d->ks, 16);
  const int ktw = ceil_div(nkt, 4);
  pl.rs = 0;
  const bool rs_ok = !getenv("A2C_NO_WGRAD_RS");
  if (rs_ok && mt == 1 && nkt == 3) { pl.MT = 1; pl.KTW = 3; pl.rs = 1; }          // exact: every k-tile is real
  else if (rs_ok && mt == 2 && nkt == 9) { pl.MT = 2; pl.KTW = 9; pl.rs = 1; }
  else if (mt == 1 && ktw <= 1) { pl.MT = 1; pl.KTW = 1; }
  else if (mt == 1 && ktw <= 4) { pl.MT = 1; pl.KTW = 4; }
  else if (mt <= 2 && ktw <= 4) { pl.MT = 2; pl.KTW = 4; }
  else if (mt <= 4 && ktw == 5) { pl.MT = 4; pl.KTW = 5; }
  else if (mt <= 4 && ktw <= 7) { pl.MT = 4; pl.KTW = 7; }
  else return false;
  pl.ex = (ktw == pl.KTW) ? 1 : 0;
  // dOut tile floats per pixel row: MT*16 planes * OWp (+ plane padding, fixed)
  plan_src(t, pl.MT * 16 * pl.OWp, pl.MT * 16 * 34, WGRAD_LDS_BUDGET);
  if (t.tiles > 1 && !getenv("A2C_WGRAD_LDS_KB")) {     // a small plane: the whole sample as ONE tile beats two uneven ones
    SrcTile t1 = t;
    plan_src(t1, pl.MT * 16 * pl.OWp, pl.MT * 16 * 34, 128 * 1024);     // (whole samples are staged as float4 runs)
    if (t1.tiles == 1) t = t1;
  }
  pl.pf = 0;
  if (allow_run && !getenv("A2C_NO_WGRAD_PF") && d->W % 4 == 0 && d->OW % 2 == 0) {
    const int ni = (pl.rs && pl.KTW == 3) ? 4 : (pl.rs && pl.KTW == 9) ? 8 : (!pl.rs && pl.MT == 2 && pl.KTW == 4 && pl.ex) ? 12 : 0;
    const int nd = (pl.rs && pl.KTW == 3) ? 16 : 8;
    const long budget = env_kb("A2C_WGRAD_PF_LDS_KB", 64);
    int tph = 0;
    for (int c = 1; c <= d->OH && ni; ++c) {      // largest band whose loads fit the prefetch registers
      const int tih = (c - 1) * d->stride + d->ks;
      const long plane = ((tih * t.WP + 31) / 32) * 32 + 16, planeo = ((c * pl.OWp + 31) / 32) * 32 + 2;
      if ((long)d->Cin * tih * (d->W / 4) <= 256L * ni && (long)d->Cout * c * (d->OW / 2) <= 256L * nd && tih < 256 &&
          4 * (d->Cin * plane + pl.MT * 16 * planeo + 64) <= budget) tph = c; else break;
    }
    if (tph) {
      pl.pf = 1;
      t.TPH = tph; t.TIH = (tph - 1) * d->stride + d->ks;
      t.PLANE = ((t.TIH * t.WP + 31) / 32) * 32 + 16;
      t.tiles = ceil_div(t.PH, t.TPH);
    }
  }
  if (!pl.pf && allow_run && !getenv("A2C_NO_WGRAD_PF") && !getenv("A2C_NO_WGRAD_PF2") && d->W % 4 && d->W % 2 == 0 && pl.ex &&
      !pl.rs && pl.MT == 2 && pl.KTW == 4) {
    // narrow layers: 8-B input slots, 4-B dOut slots (WGRAD_PF_D / _E)
    const long budget = env_kb("A2C_WGRAD_PF_LDS_KB", 64);
    int tph = 0;
    for (int c = 1; c <= d->OH; ++c) {
      const int tih = (c - 1) * d->stride + d->ks;
      const long plane = ((tih * t.WP + 31) / 32) * 32 + 16, planeo = ((c * pl.OWp + 31) / 32) * 32 + 2;
      if ((long)d->Cin * tih * (d->W / 2) <= 256L * 14 && (long)d->Cout * c * d->OW <= 256L * 8 && tih < 256 &&
          4 * (d->Cin * plane + pl.MT * 16 * planeo + 64) <= budget) tph = c; else break;
    }
    if (tph) {
      pl.pf = 2;
      t.TPH = tph; t.TIH = (tph - 1) * d->stride + d->ks;
      t.PLANE = ((t.TIH * t.WP + 31) / 32) * 32 + 16;
      t.tiles = ceil_div(t.PH, t.TPH);
    }
  }
  pl.PLANEo = ((t.TPH * pl.OWp + 31) / 32) * 32 + 2;
  pl.lds = 4 * ((size_t)t.Cp * t.PLANE + (size_t)pl.MT * 16 * pl.PLANEo + 64);
  if (pl.rs && pl.lds < (size_t)pl.KTW * 4096) pl.lds = (size_t)pl.KTW * 4096;   // cross-wave reduction scratch
  if (pl.pf && pl.lds < 256 * 16 * 4) pl.lds = 256 * 16 * 4;                      // bias partials, one per dOut slot
  if (pl.lds > LDS_HARD_MAX) return false;
  const long total = (long)B * t.tiles;
  const void* k = wgrad_fn(pl);
  if (pl.lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  pl.grid = resident_grid(k, pl.lds, total);
  return true;
}

template <int MT, int KTW, bool RS = false, int NI = 0, int ND = 0, bool EX = false, int VI = 4, int VD = 2>
static void launch_wgrad_t(const WgradP& p, int grid, size_t lds, hipStream_t st) {
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute((const void*)wgrad_kernel<MT, KTW, RS, NI, ND, EX, VI, VD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((wgrad_kernel<MT, KTW, RS, NI, ND, EX, VI, VD>), dim3(grid), dim3(256), lds, st, p);
}
}  // namespace

extern "C" {
static size_t prep_floats_base(const a2c_conv_desc* d, int kind) {
  if (kind == 0) return (size_t)pad_steps(d->ks * d->ks * (d->Cin / 4)) * ceil_div(d->Cout, 16) * 64;
  return bwd_class_offset(d, d->stride * d->stride);
}
size_t a2c_conv2d_prep_floats(const a2c_conv_desc* d, int kind) {
  if (!desc_ok(d)) return 0;
  return prep_floats_base(d, kind) + c3_prep_floats(d, kind);      // [this file's fragments | conv3.hip's]
}

int a2c_conv2d_prep_weights(const a2c_conv_desc* d, int kind, const float* weight, float* wprep,
                            a2c_stream_t stream) {
  if (!desc_ok(d) || !weight || !wprep) return A2C_ERR_ARG;
  hipStream_t st = a2c_s(stream);
  if (c3_prep(d, kind, weight, wprep + prep_floats_base(d, kind), st) != A2C_OK) return A2C_ERR_LAUNCH;
  if (kind == 0) {
    const int MT = ceil_div(d->Cout, 16);
    const long total = (long)prep_floats_base(d, 0);
    hipLaunchKernelGGL(prep_fwd_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight, wprep, d->Cin, d->Cout,
                       d->ks, MT, (run_layout(d) || run3_layout(d)) ? 1 : 0, total);
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  const int S = d->stride, MTb = ceil_div(d->Cin, 16);
  for (int cls = 0; cls < S * S; ++cls) {
    const int ry = cls / S, rx = cls % S;
    const int na = ntaps_1d(d->ks, S, ry), nb = ntaps_1d(d->ks, S, rx);
    const long total = (long)pad_steps(na * nb * (d->Cout / 4)) * MTb * 64;
    hipLaunchKernelGGL(prep_bwd_kernel, dim3(a2c_grid_1d(total, 256)), dim3(256), 0, st, weight,
                       wprep + bwd_class_offset(d, cls), d->Cin, d->Cout, d->ks, S, ry, rx, na, nb > 0 ? nb : 1, MTb,
                       total);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

}  // extern "C"
namespace {
// fwd_kb > 0 replaces the LDS budget (KB) that sizes the forward tiles of the tiled kernels (the tuner below)
static int conv_fwd_impl(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* wprep_fwd,
                         const float* bias, int relu, float* out, int64_t out_bstride, int B, int fwd_kb,
                         a2c_stream_t stream) {
  SrcTile t;
  t.Cp = d->Cin; t.IH = d->H; t.IW = d->W; t.SY = d->stride; t.SX = d->stride;
  t.sy0 = -d->pad; t.sx0 = -d->pad; t.span_y = d->ks; t.span_x = d->ks; t.PH = d->OH; t.PW = d->OW;
  const bool staged_out = !getenv("A2C_NO_OUT_STAGE");
  plan_src(t, staged_out ? d->Cout * d->OW : 0, 16, fwd_kb > 0 ? fwd_kb * 1024 : IGEMM_LDS_BUDGET);
  IgemmP p;
  fill_stage(p.st, t, in, in_bstride);
  p.out_stage = 0; p.out_vec = 0;
  p.out = out; p.out_bs = out_bstride; p.Mch = d->Cout; p.OHf = d->OH; p.OWf = d->OW;
  p.wfrag = wprep_fwd; p.bias = bias; p.mask = nullptr; p.relu = relu;
  p.PH = d->OH; p.PW = d->OW; p.oy_mul = 1; p.oy_add = 0; p.ox_mul = 1; p.ox_add = 0;
  p.SY = t.SY; p.SX = t.SX; p.sy0 = t.sy0; p.TPH = t.TPH; p.tiles = t.tiles; p.B = B;
  const int c4n = d->Cin / 4, nsteps = d->ks * d->ks * c4n;
  const int MT = ceil_div(d->Cout, 16);
  p.nchunks = pad_steps(nsteps) / CH;
  // pipelined variant: unpadded layer, 16 B aligned rows, fragments <= 32 KB, image <= 32 KB
  const int nfrag = p.nchunks * CH * MT * 64;
  p.nsteps = nsteps; p.nb = d->ks; p.c4n = c4n; p.off0 = 0; p.step_b = 1;
  {  // streaming kernel: A3C conv1 class at large batch
    StreamP sp;
    const int n_cu = stream_grid();
    if (B >= 8 * n_cu && !getenv("A2C_NO_STREAM") && !getenv("A2C_NO_PF") && in_bstride % 4 == 0 && out_bstride % 4 == 0 &&
        ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && plan_stream(d, sp)) {
      sp.in = in; sp.in_bs = in_bstride; sp.wfrag = wprep_fwd; sp.bias = bias; sp.out = out; sp.out_bs = out_bstride;
      sp.B = B; sp.relu = relu;
      { const char* e = getenv("A2C_STREAM_DBG"); sp.dbg = e ? atoi(e) : 0; }
      const size_t lds = stream_lds(sp);
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute((const void*)conv_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return A2C_ERR_LAUNCH;
        attr = true;
      }
      hipLaunchKernelGGL(conv_stream_kernel, dim3(n_cu), dim3(ST_NT), lds, a2c_s(stream), sp);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  {  // streaming kernel: A3C conv2 class at large batch
    Stream2P sp;
    if (B >= 16 * stream_grid() && !getenv("A2C_NO_STREAM") && !getenv("A2C_NO_PF") && in_bstride % 4 == 0 && out_bstride % 4 == 0 &&
        ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && plan_stream2(d, sp)) {
      sp.in = in; sp.in_bs = in_bstride; sp.wfrag = wprep_fwd; sp.bias = bias; sp.out = out; sp.out_bs = out_bstride;
      sp.B = B; sp.relu = relu;
      const size_t lds = 4 * (size_t)(32 * sp.PLANE + 2 * d->Cout * d->OH * d->OW);
      if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)conv2_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const int grid = resident_grid((const void*)conv2_stream_kernel, lds, (B + 1) / 2, S2_NT);
      hipLaunchKernelGGL(conv2_stream_kernel, dim3(grid), dim3(S2_NT), lds, a2c_s(stream), sp);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  const bool run = run_layout(d);
  if (run && in_bstride % 4 == 0 && ((uintptr_t)in % 16 == 0) && nfrag * 4 <= 32 * 1024 && !getenv("A2C_NO_PF")) {
    // largest band whose image fits the prefetch registers; prefer bands whose pixel count fills
    // whole 32-pixel pairs
    int tph = 0;
    double best = -1.0;
    for (int c = 1; c <= d->OH; ++c) {
      if ((long)d->Cin * ((c - 1) * d->stride + d->ks) * d->W > 256L * PF_N * 4) break;
      const int ntl = ceil_div(d->OH, c);
      long slots = 0;
      for (int i = 0; i < ntl; ++i) {
        const int r = (i + 1 < ntl) ? c : d->OH - c * (ntl - 1);
        slots += (long)ceil_div(ceil_div(r * d->OW, 32), 4) * 4 * 32;      // 4 waves, pairs of 16-pixel tiles
      }
      const double eff = (double)d->OH * d->OW / slots - 0.02 * (double)((c - 1) * d->stride + d->ks) * ntl / d->H;
      if (eff > best) { best = eff; tph = c; }
    }
    if (tph >= 1) {
      t.TPH = tph; t.TIH = (tph - 1) * t.SY + t.span_y; t.WP = t.IW;
      t.PLANE = ((t.TIH * t.WP + 63) / 64) * 64 + (d->stride == 4 ? 0 : 32);
      t.tiles = ceil_div(t.PH, t.TPH);
      fill_stage(p.st, t, in, in_bstride);
      p.TPH = t.TPH; p.tiles = t.tiles;
      const size_t lds = 4 * ((size_t)nfrag + (size_t)t.Cp * t.PLANE + 64);
      const long total = (long)B * t.tiles;
      const void* k;
      if (d->ks == 8) k = MT == 1 ? (const void*)igemm_run_kernel<1, 8, 4> : (const void*)igemm_run_kernel<2, 8, 4>;
      else k = MT == 1 ? (const void*)igemm_run_kernel<1, 4, 2> : (const void*)igemm_run_kernel<2, 4, 2>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const int grid = resident_grid(k, lds, total);
      hipStream_t st = a2c_s(stream);
      if (d->ks == 8 && MT == 1) hipLaunchKernelGGL((igemm_run_kernel<1, 8, 4>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else if (d->ks == 8) hipLaunchKernelGGL((igemm_run_kernel<2, 8, 4>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else if (MT == 1) hipLaunchKernelGGL((igemm_run_kernel<1, 4, 2>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else hipLaunchKernelGGL((igemm_run_kernel<2, 4, 2>), dim3(grid), dim3(256), lds, st, p, nfrag);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  const bool run3 = run3_layout(d);
  if (run3 && in_bstride % 4 == 0 && ((uintptr_t)in % 16 == 0) && !getenv("A2C_NO_PF") && !getenv("A2C_NO_RUN3")) {
    const int S = d->stride;
    const bool ovec = (d->OW % 4 == 0) && (out_bstride % 4 == 0) && (((long)d->OH * d->OW) % 4 == 0) && ((uintptr_t)out % 16 == 0);
    const long budget = fwd_kb > 0 ? fwd_kb * 1024L : (long)env_kb("A2C_RUN3_LDS_KB", 80);
    int tph = 0, plane_best = 0;
    double best = -1.0;
    for (int c = 1; c <= d->OH; ++c) {
      const int tih = (c - 1) * S + 3;
      if ((long)d->Cin * tih * d->W > 256L * PF3 * 4) break;
      const int wp = d->W + 2;
      const int plane = ((tih * wp + 31) / 32) * 32 + (S == 1 ? 16 : 1);
      const long bytes = 4L * ((long)nfrag + (long)d->Cin * plane + 64 + (long)d->Cout * c * d->OW);
      if (bytes > budget) break;
      const int ntl = ceil_div(d->OH, c);
      long slots = 0;
      for (int i = 0; i < ntl; ++i) {
        const int r = (i + 1 < ntl) ? c : d->OH - c * (ntl - 1);
        slots += (long)ceil_div(ceil_div(r * d->OW, 32), 4) * 4 * 32;
      }
      const double eff = (double)d->OH * d->OW / slots - 0.02 * (double)tih * ntl / d->H;
      if (eff > best) { best = eff; tph = c; plane_best = plane; }
    }
    if (tph >= 1) {
      t.TPH = tph; t.TIH = (tph - 1) * S + 3; t.WP = d->W + 2; t.PLANE = plane_best;
      t.tiles = ceil_div(t.PH, t.TPH);
      fill_stage(p.st, t, in, in_bstride);
      p.TPH = t.TPH; p.tiles = t.tiles; p.out_vec = ovec ? 1 : 0;
      const size_t lds = 4 * ((size_t)nfrag + (size_t)t.Cp * t.PLANE + 64 + (size_t)d->Cout * t.TPH * d->OW);
      const long total = (long)B * t.tiles;
      const void* k = S == 1 ? (MT == 1 ? (const void*)igemm_run3_kernel<1, 1> : (const void*)igemm_run3_kernel<2, 1>)
                             : (MT == 1 ? (const void*)igemm_run3_kernel<1, 2> : (const void*)igemm_run3_kernel<2, 2>);
      if (lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const int grid = resident_grid(k, lds, total);
      hipStream_t st = a2c_s(stream);
      if (S == 1 && MT == 1) hipLaunchKernelGGL((igemm_run3_kernel<1, 1>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else if (S == 1) hipLaunchKernelGGL((igemm_run3_kernel<2, 1>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else if (MT == 1) hipLaunchKernelGGL((igemm_run3_kernel<1, 2>), dim3(grid), dim3(256), lds, st, p, nfrag);
      else hipLaunchKernelGGL((igemm_run3_kernel<2, 2>), dim3(grid), dim3(256), lds, st, p, nfrag);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  if (staged_out) {            // generic kernel: output tile assembled in LDS, flushed coalesced
    p.out_stage = t.Cp * t.PLANE + 64;
    p.out_vec = (d->OW % 4 == 0) && (out_bstride % 4 == 0) && (((long)d->OH * d->OW) % 4 == 0) && ((uintptr_t)out % 16 == 0);
    if (!p.out_vec && t.tiles == 1 && t.TPH == d->OH && out_bstride % 4 == 0 && ((long)d->Cout * d->OH * d->OW) % 4 == 0 &&
        ((uintptr_t)out % 16 == 0) && (p.out_stage % 4 == 0) && !getenv("A2C_NO_FLAT_STAGE"))
      p.out_vec = 2;
  }
  if (run || run3) {   // generic kernel on a run-ordered layer: steps walk (c4, ky, kx)
    p.nb = d->ks; p.c4n = d->ks;                      // walker levels: outer c4, mid ky, inner kx
    p.step_a = 4 * t.PLANE; p.step_b = t.WP; p.step_c = 1;
    return launch_igemm(p, MT, a2c_s(stream));
  }
  p.step_a = t.WP; p.step_c = 4 * t.PLANE;
  return launch_igemm(p, MT, a2c_s(stream));
}

// Forward tile height by measurement, like bwd_band_tuned: the tile size changes neither the taps nor their order
// for any output element, so every candidate gives bit-identical results.  At rollout batch sizes (a few hundred
// tiles for 256 CUs) the best height is a trade of tiles per CU against the per-workgroup prologue that no rule
// predicted (measured 1.2-1.5x between budgets on the 24- and 32-channel layers at n_envs = 32).
static int conv_fwd_tuned(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* wprep_fwd,
                          const float* bias, int relu, float* out, int64_t out_bstride, int B, a2c_stream_t stream) {
  static std::mutex mu;
  static std::map<BandKey, int> cache;
  const long work = (long)B * d->Cout * d->OH * d->OW;
  if (getenv("A2C_NO_TUNE") || getenv("A2C_IGEMM_LDS_KB") || getenv("A2C_RUN3_LDS_KB") || work < (1L << 16))
    return conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, 0, stream);
  const BandKey key = {{d->Cin, d->H, d->W, d->Cout, d->ks, d->stride, d->pad, B, (int)(in_bstride % 4 == 0) | ((int)(out_bstride % 4 == 0) << 1)}};
  hipStream_t st = a2c_s(stream);
  std::lock_guard<std::mutex> g(mu);
  auto it = cache.find(key);
  if (it != cache.end()) return conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, it->second, stream);
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
    return conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, 0, stream);
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return A2C_ERR_LAUNCH;
  static const int cand[] = {0, 24, 32, 48, 64, 96, 128};
  const int reps = work < (1L << 22) ? 4 : 1;
  int best = 0;
  float best_ms = -1.f;
  for (int c : cand) {
    if (conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, c, stream) != A2C_OK) { (void)hipGetLastError(); continue; }
    (void)hipEventRecord(e0, st);
    int rc = A2C_OK;
    for (int r = 0; r < reps && rc == A2C_OK; ++r) rc = conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, c, stream);
    (void)hipEventRecord(e1, st);
    float ms = 0.f;
    if (rc != A2C_OK || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
    if (getenv("A2C_TUNE_LOG")) fprintf(stderr, "a2c fwd tune (%d,%d,%d)->%d s%d B=%d: budget %d KB %.4f ms\n", d->Cin, d->H, d->W, d->Cout, d->stride, B, c, ms / reps);
    if (best_ms < 0.f || ms < 0.97f * best_ms) { best = c; best_ms = ms; }      // later candidates must win by 3 %
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  cache[key] = best;
  return conv_fwd_impl(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, best, stream);
}
}  // namespace
extern "C" {
int a2c_conv2d_fwd(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* wprep_fwd,
                   const float* bias, int relu, float* out, int64_t out_bstride, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!in || !wprep_fwd || !out) return A2C_ERR_ARG;
  if (c3_supported(d, 0) && in_bstride % 4 == 0 && out_bstride % 4 == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0)
    return c3_fwd(d, in, (long)in_bstride, wprep_fwd + prep_floats_base(d, 0), bias, relu, out, (long)out_bstride, nullptr, 0, B, a2c_s(stream));
  return conv_fwd_tuned(d, in, in_bstride, wprep_fwd, bias, relu, out, out_bstride, B, stream);
}

int64_t a2c_conv2d_sign_words(const a2c_conv_desc* d) {
  if (!desc_ok(d)) return 0;
  return (int64_t)c3_sign_words(d);
}

int a2c_conv2d_fwd_signs(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* wprep_fwd, const float* bias,
                         int relu, float* out, int64_t out_bstride, uint32_t* signs, int64_t signs_bstride, int B,
                         a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!in || !wprep_fwd || !out || !signs || c3_sign_words(d) == 0 || signs_bstride < c3_sign_words(d)) return A2C_ERR_ARG;
  if (in_bstride % 4 || out_bstride % 4 || ((uintptr_t)in % 16) || ((uintptr_t)out % 16)) return A2C_ERR_ARG;
  return c3_fwd(d, in, (long)in_bstride, wprep_fwd + prep_floats_base(d, 0), bias, relu, out, (long)out_bstride, signs,
                (long)signs_bstride, B, a2c_s(stream));
}

int a2c_conv2d_fwd_chain_supported(const a2c_conv_desc* d, int n_layers) {
  if (!d || n_layers < 1 || n_layers > A2C_CONV_CHAIN_MAX) return 0;
  for (int i = 0; i < n_layers; ++i)
    if (!desc_ok(d + i) || (i > 0 && (d[i].Cin != d[i - 1].Cout || d[i].H != d[i - 1].OH || d[i].W != d[i - 1].OW))) return 0;
  return c3_chain_supported(d, n_layers) ? 1 : 0;
}

int a2c_conv2d_fwd_chain(const a2c_conv_desc* d, int n_layers, const float* in, int64_t in_bstride, const float* const* wprep_fwd,
                         const float* const* bias, int relu, float* const* out, const int64_t* out_bstride, uint32_t* const* signs,
                         const int64_t* signs_bstride, int B, a2c_stream_t stream) {
  if (!a2c_conv2d_fwd_chain_supported(d, n_layers) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!in || !wprep_fwd || !bias || !out || !out_bstride || in_bstride % 4 || ((uintptr_t)in % 16)) return A2C_ERR_ARG;
  const float* frag[A2C_CONV_CHAIN_MAX];
  long obs[A2C_CONV_CHAIN_MAX], sbs[A2C_CONV_CHAIN_MAX];
  unsigned* sg[A2C_CONV_CHAIN_MAX];
  for (int i = 0; i < n_layers; ++i) {
    if (!wprep_fwd[i] || !out[i] || out_bstride[i] % 4 || ((uintptr_t)out[i] % 16) ||
        out_bstride[i] < (int64_t)d[i].Cout * d[i].OH * d[i].OW)
      return A2C_ERR_ARG;
    frag[i] = wprep_fwd[i] + prep_floats_base(d + i, 0);
    obs[i] = (long)out_bstride[i];
    sg[i] = signs ? signs[i] : nullptr;
    sbs[i] = (sg[i] && signs_bstride) ? (long)signs_bstride[i] : 0;
    if (sg[i] && (c3_sign_words(d + i) == 0 || sbs[i] < c3_sign_words(d + i))) return A2C_ERR_ARG;
  }
  return c3_chain_fwd(d, n_layers, in, (long)in_bstride, frag, bias, relu, out, obs, sg, sbs, B, a2c_s(stream));
}

int a2c_conv2d_fwd_frames_supported(const a2c_conv_desc* d) { return desc_ok(d) && c3_fwd_frames_supported(d) ? 1 : 0; }

int a2c_conv2d_fwd_frames(const a2c_conv_desc* d, const uint8_t* frame_store, int64_t sample_stride, int64_t T,
                          const int* nvalid, int64_t nvalid_stride, const float* wprep_fwd, const float* bias, int relu,
                          float* out, int64_t out_bstride, uint32_t* signs, int64_t signs_bstride, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0 || T < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!frame_store || !wprep_fwd || !out || !c3_fwd_frames_supported(d)) return A2C_ERR_ARG;
  if (((uintptr_t)frame_store % 4) || sample_stride % 4 || sample_stride < (T + 3) * (int64_t)d->H * d->W || out_bstride % 4 ||
      ((uintptr_t)out % 16))
    return A2C_ERR_ARG;
  if (signs && (c3_sign_words(d) == 0 || signs_bstride < c3_sign_words(d))) return A2C_ERR_ARG;
  return c3_fwd_frames(d, frame_store, (long)sample_stride, (long)T, nvalid, (long)nvalid_stride, wprep_fwd + prep_floats_base(d, 0),
                       bias, relu, out, (long)out_bstride, signs, (long)signs_bstride, B, a2c_s(stream));
}

}  // extern "C"
namespace {
constexpr int BAND_NA = -1000;
// fills q for band height ty_force (> 0) or the tallest band inside the LDS budget (0); false: no such band
static bool band_setup(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask, float* din,
                       int B, int ty_force, BwdBandP& q, int& TIH, int& PLANEo) {
  const int S = d->stride, P = d->pad;
  // band kernel: all classes fused, dX band assembled in LDS and flushed coalesced
  const int c4n = d->Cout / 4;
  q.din = din; q.mask = mask; q.wfrag = wprep_bwd;
  q.Cin = d->Cin; q.H = d->H; q.W = d->W; q.S = S; q.P = P; q.ks = d->ks; q.B = B; q.ncls = S * S; q.c4n = c4n;
  const int ox_lo = (P - (d->ks - 1)) >= 0 ? (P - (d->ks - 1)) / S : -((-(P - (d->ks - 1)) + S - 1) / S);
  const int ox_hi = (d->W - 1 + P) / S;
  q.ox_lo = ox_lo;
  const int WPo = ox_hi - ox_lo + 1;
  int TY = 0;
  TIH = 0; PLANEo = 0;
  for (int ty = S; ty <= ((d->H + S - 1) / S) * S; ty += S) {
    const int tih = (ty - 1 + d->ks - 1) / S + 2;
    const int plane = ((tih * WPo + 31) / 32) * 32 + 16;
    const long bytes = 4L * ((long)d->Cin * ty * d->W + (long)d->Cout * plane + 64);
    if (ty_force > 0 ? ty == ty_force : (bytes <= IGEMM_LDS_BUDGET || TY == 0)) { TY = ty; TIH = tih; PLANEo = plane; }
    if (ty_force > 0 ? ty >= ty_force : bytes > IGEMM_LDS_BUDGET) break;
  }
  if (TY == 0) return false;
  if (TY > d->H) TY = d->H;        // one band = the whole sample: the dX block in LDS is laid out exactly like HBM
  q.TY = TY; q.bands = ceil_div(d->H, TY); q.out_floats = d->Cin * TY * d->W;
  q.mgroups = 1; q.mt_total = ceil_div(d->Cin, 16);
  q.st.src = dout; q.st.bstride = (long)d->Cout * d->OH * d->OW; q.st.Cp = d->Cout; q.st.IH = d->OH; q.st.IW = d->OW;
  q.st.TIH = TIH; q.st.WP = WPo; q.st.PLANE = PLANEo; q.st.sx0 = ox_lo; q.st.fast = 0;
  // one band = the whole sample (TY == H): dOut is staged and dX flushed as contiguous float4 runs (bwd_band_kernel)
  q.st.flat = (TY == d->H && q.st.vec != 4 && ((long)d->Cout * d->OH * d->OW) % 4 == 0 && ((long)d->Cin * d->H * d->W) % 4 == 0 &&
               ((uintptr_t)dout % 16 == 0) && !getenv("A2C_NO_FLAT_STAGE")) ? 1 : 0;
  q.st.vec = stage_vec(dout, q.st.bstride, d->OH, d->OW);
  for (int cls = 0; cls < S * S; ++cls) {
    const int ry = cls / S, rx = cls % S;
    BandClass& k = q.cls[cls];
    k.ry = ry; k.rx = rx; k.na = ntaps_1d(d->ks, S, ry); k.nb = ntaps_1d(d->ks, S, rx);
    if (k.nb < 1) k.nb = 1;
    k.frag_off = (int)bwd_class_offset(d, cls);
    k.nsteps = ntaps_1d(d->ks, S, ry) * ntaps_1d(d->ks, S, rx) * c4n;
    k.nchunks = pad_steps(k.nsteps) / CH;
    k.p0 = (P - rx) > 0 ? ceil_div(P - rx, S) : 0;
    k.PWc = (d->W - 1 + P - rx) >= 0 ? (d->W - 1 + P - rx) / S - k.p0 + 1 : 0;
    if (k.PWc < 0) k.PWc = 0;
  }
  return true;
}

// Band kernels (bwd_band2_kernel / bwd_band_kernel) for one launch; ty_force > 0 fixes the band height (a multiple
// of the stride), 0 = the tallest band inside the LDS budget.  BAND_NA: no band kernel fits this call.
static int launch_bwd_band(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask,
                           float* din, int B, int ty_force, a2c_stream_t stream) {
  const int S = d->stride;
  const int MTb = ceil_div(d->Cin, 16), c4n = d->Cout / 4;
  BwdBandP q;
  int TIH = 0, PLANEo = 0;
  if (!band_setup(d, dout, wprep_bwd, mask, din, B, ty_force, q, TIH, PLANEo)) return BAND_NA;
  const int TY = q.TY;
  (void)S;
  {  // pipelined band kernel: fragments in LDS, next band's dOut and this band's mask in registers
    const size_t nfrag = a2c_conv2d_prep_floats(d, 1);
    const int vecs = q.st.vec;
    const long tot_v = (long)d->Cout * TIH * (d->OW / vecs);
    const size_t lds2 = 4 * (nfrag + (size_t)q.out_floats + (size_t)d->Cout * PLANEo + 64);
    if (!getenv("A2C_NO_BAND2") && nfrag * 4 <= 64 * 1024 && d->OW % vecs == 0 && tot_v <= 256L * PFB2_MAX && MTb == 1 &&
        c4n % 2 == 0 && lds2 <= LDS_HARD_MAX) {      // MTb > 1 measured slower: fragments + band leave one workgroup per CU
      BwdBand2P q2;
      q2.b = q; q2.nfrag = (int)nfrag;
      q2.mask_pf = (d->W % 4 == 0 && (long)d->Cin * TY * d->W <= 256L * PFM2 * 4) ? 1 : 0;
      const void* kf2;
      hipStream_t st2 = a2c_s(stream);
      const int pfb = tot_v <= 256L * 8 ? 8 : tot_v <= 256L * 12 ? 12 : 16;      // fewest prefetch registers that hold a band
#define BAND2_CASE(V_, P_)                                                                                               \
      if (vecs == V_ && pfb == P_) {                                                                                    \
        kf2 = (const void*)bwd_band2_kernel<1, V_, P_>;                                                                 \
        if (lds2 > 64 * 1024) (void)hipFuncSetAttribute(kf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);    \
        const int grid2 = resident_grid(kf2, lds2, (long)B * q.bands);                                                  \
        hipLaunchKernelGGL((bwd_band2_kernel<1, V_, P_>), dim3(grid2), dim3(256), lds2, st2, q2, FuseW1P());                        \
        A2C_CHECK_LAUNCH();                                                                                             \
        return A2C_OK;                                                                                                  \
      }
      BAND2_CASE(1, 8) BAND2_CASE(1, 12) BAND2_CASE(1, 16) BAND2_CASE(2, 8) BAND2_CASE(2, 12) BAND2_CASE(2, 16)
      BAND2_CASE(4, 8) BAND2_CASE(4, 12) BAND2_CASE(4, 16)
#undef BAND2_CASE
    }
  }
  size_t lds = 4 * ((size_t)q.out_floats + (size_t)d->Cout * PLANEo + 64);
  int MTk = MTb;
  // whole-sample bands that leave ONE workgroup per CU: the input channels in groups of 16, one workgroup each, when two
  // of those fit a CU (GRUModel conv4 backward-data, 32 <- 48 @21: 96 KB -> 2 x 68 KB)
  if (q.st.flat && q.bands == 1 && MTb >= 2 && MTb <= 4 && lds > 80 * 1024 && d->Cin % 16 == 0 && (16 * d->H * d->W) % 4 == 0 &&
      !getenv("A2C_NO_BAND_GROUPS")) {
    const size_t lds1 = 4 * ((size_t)16 * TY * d->W + (size_t)d->Cout * PLANEo + 64);
    if (lds1 <= 80 * 1024) {
      q.mgroups = MTb; MTk = 1; q.out_floats = 16 * TY * d->W; lds = lds1;
    }
  }
  if (lds <= LDS_HARD_MAX && MTb <= 4) {
    // (NTU = 4, four tiles per A fragment, measured on GRUModel's 21 x 21 / 11 x 11 layers: 477 vs 475 us and 212 vs 162 us at
    // N = 4096 -- the fragment loads are not what these launches wait for)
    hipStream_t st = a2c_s(stream);
#define BAND_RUN(M_, N_)                                                                                          \
    {                                                                                                             \
      const void* kf = (const void*)bwd_band_kernel<M_, N_>;                                                      \
      if (lds > 64 * 1024) (void)hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   \
      const int grid = resident_grid(kf, lds, (long)B * q.bands * q.mgroups);                                     \
      hipLaunchKernelGGL((bwd_band_kernel<M_, N_>), dim3(grid), dim3(256), lds, st, q);                           \
    }
    if (MTk == 1) BAND_RUN(1, 2) else if (MTk == 2) BAND_RUN(2, 2) else if (MTk == 3) BAND_RUN(3, 2) else BAND_RUN(4, 2)
#undef BAND_RUN
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  return BAND_NA;
}

// ---- backward-data of layer 2 + weight gradient of layer 1 in one pass (bwd_band2_kernel<.., W1 = true>) ----------
struct BandW1Plan { BwdBand2P q2; FuseW1P fw; int vec, pfb, grid; size_t lds; };

// d2 = the upper layer (its dX is the lower layer's dOut), d1 = the lower (first) layer.  The band height is a fixed
// rule (tallest band with two workgroups per CU), NOT tuned by timing: it partitions the sums of dW1.
static bool plan_band_w1(const a2c_conv_desc* d2, const a2c_conv_desc* d1, int B, const float* dout, BandW1Plan& pl) {
  // OPT-IN (A2C_FUSE_W1=1): measured on MI355X the fused pass is SLOWER than the two separate ones (GRUModel layers at
  // N = 32 768: 21.7 vs 11.4 + 6.9 ms; ConvModel at N = 2 048: 3.0-4.1 vs 2.35 + 0.47 ms) although it moves 43 % fewer
  // bytes: with the input-row prefetch, the k offsets and the extra accumulators it needs 250-300 VGPRs (one workgroup
  // per CU), and the extra per-band phases (mask in place, barrier, bias partials, 3-MFMA steps) are latency-bound.
  const char* on = getenv("A2C_FUSE_W1");
  if (!on || on[0] != '1' || !desc_ok(d2) || !desc_ok(d1) || getenv("A2C_NO_BAND2")) return false;
  if (d1->ks != 3 || d1->stride != 1 || d1->pad != 1 || d1->Cout != d2->Cin || d1->OH != d2->H || d1->OW != d2->W) return false;
  if (d2->Cin > 16 || d1->Cin * 9 > 48 || d2->W % 4 || d2->stride * d2->stride > MAX_CLS || (d2->Cout / 4) % 2) return false;
  const size_t nfrag = a2c_conv2d_prep_floats(d2, 1);
  if (nfrag * 4 > 64 * 1024) return false;
  const int S = d2->stride;
  int best = 0;
  for (int ty = S; ty <= ((d2->H + S - 1) / S) * S; ty += S) {
    int TIH = 0, PLANEo = 0;
    BwdBandP q;
    if (!band_setup(d2, dout, nullptr, nullptr, nullptr, B, ty, q, TIH, PLANEo)) break;
    const int vec = q.st.vec;
    if (d2->OW % vec) break;
    const long tot_v = (long)d2->Cout * TIH * (d2->OW / vec);
    const int tihx = ty + 2, wpx = d2->W + 2;
    const long planex = ((tihx * wpx + 31) / 32) * 32 + 16;
    const long lds = 4L * ((long)nfrag + q.out_floats + (long)d2->Cout * PLANEo + 64 + d1->Cin * planex + 64);
    if (tot_v > 256L * PFB2_MAX || (long)d2->Cin * ty * d2->W > 256L * PFM2_W1 * 4 ||
        (long)d1->Cin * tihx * (d2->W / 4) > 256L * NIX || tihx > 255 || lds > env_kb("A2C_W1_LDS_KB", 80)) break;
    best = ty;
  }
  if (!best) return false;
  int TIH = 0, PLANEo = 0;
  band_setup(d2, dout, nullptr, nullptr, nullptr, B, best, pl.q2.b, TIH, PLANEo);
  pl.q2.nfrag = (int)nfrag; pl.q2.mask_pf = 1;
  pl.vec = pl.q2.b.st.vec;
  const long tot_v = (long)d2->Cout * TIH * (d2->OW / pl.vec);
  pl.pfb = tot_v <= 256L * 8 ? 8 : tot_v <= 256L * 12 ? 12 : 16;
  FuseW1P& fw = pl.fw;
  fw.Cin1 = d1->Cin; fw.K1 = d1->Cin * 9; fw.TIHx = best + 2; fw.WPx = d2->W + 2;
  fw.PLANEx = ((fw.TIHx * fw.WPx + 31) / 32) * 32 + 16;
  fw.xoff = (int)nfrag + pl.q2.b.out_floats + d2->Cout * PLANEo + 64;
  pl.lds = 4 * ((size_t)fw.xoff + (size_t)fw.Cin1 * fw.PLANEx + 64);
  if (pl.lds < 4 * (4 * 3 * 64 * 4 + 256)) pl.lds = 4 * (4 * 3 * 64 * 4 + 256);       // end-of-kernel reduction scratch
  return true;
}

#define BAND_W1_CASES(X) X(1, 8) X(1, 12) X(1, 16) X(2, 8) X(2, 12) X(2, 16) X(4, 8) X(4, 12) X(4, 16)
static const void* band_w1_fn(int vec, int pfb) {
#define W1_FN(V_, P_) if (vec == V_ && pfb == P_) return (const void*)bwd_band2_kernel<1, V_, P_, true>;
  BAND_W1_CASES(W1_FN)
#undef W1_FN
  return nullptr;
}
static int band_w1_grid(BandW1Plan& pl, int B) {
  const void* k = band_w1_fn(pl.vec, pl.pfb);
  if (!k) return 0;
  if (pl.lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  return resident_grid(k, pl.lds, (long)B * pl.q2.b.bands);
}

// The band height changes the tiling only (every dX element keeps its tap / channel summation order), so the
// result is bit-identical for every height and the height can be chosen by measurement: the first eager call of a
// (layer, batch) pair times each candidate once (after one untimed run) and caches the fastest.  Measured spread
// between heights on the reference's 3x3 layers: 1.3-1.7x (pixel-pair quantisation of the class rows, workgroups
// per CU, bands per sample), which no closed-form rule predicted.  Calls made while the stream is being captured
// into a hipGraph use the cached height, or the LDS-budget rule when the pair has not been seen yet.
static int bwd_band_tuned(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask,
                          float* din, int B, a2c_stream_t stream) {
  static std::mutex mu;
  static std::map<BandKey, int> cache;
  if (getenv("A2C_BAND_TY")) return launch_bwd_band(d, dout, wprep_bwd, mask, din, B, atoi(getenv("A2C_BAND_TY")), stream);
  const long work = (long)B * d->Cin * d->H * d->W;
  if (getenv("A2C_NO_TUNE") || work < (1L << 24)) return launch_bwd_band(d, dout, wprep_bwd, mask, din, B, 0, stream);
  const BandKey key = {{d->Cin, d->H, d->W, d->Cout, d->ks, d->stride, d->pad, B, mask ? 1 : 0}};
  hipStream_t st = a2c_s(stream);
  {
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return launch_bwd_band(d, dout, wprep_bwd, mask, din, B, it->second, stream);
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone)
      return launch_bwd_band(d, dout, wprep_bwd, mask, din, B, 0, stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return A2C_ERR_LAUNCH;
    const int S = d->stride;
    int best = 0;
    float best_ms = 0.f;
    const int ty_all = ((d->H + S - 1) / S) * S;                    // one band = the whole sample
    for (int ty = S; ty <= ty_all; ty += S) {
      if (ty > 16 && ty % (4 * S) && ty != ty_all) continue;       // coarser steps for tall bands
      if (launch_bwd_band(d, dout, wprep_bwd, mask, din, B, ty, stream) != A2C_OK) { (void)hipGetLastError(); continue; }
      (void)hipEventRecord(e0, st);
      const int rc = launch_bwd_band(d, dout, wprep_bwd, mask, din, B, ty, stream);
      (void)hipEventRecord(e1, st);
      float ms = 0.f;
      if (rc != A2C_OK || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) continue;
      if (getenv("A2C_TUNE_LOG")) fprintf(stderr, "a2c band tune (%d,%d,%d)<-%d s%d B=%d: TY %d %.3f ms\n", d->Cin, d->H, d->W, d->Cout, S, B, ty, ms);
      if (best == 0 || ms < best_ms) { best = ty; best_ms = ms; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    cache[key] = best;                                              // 0 (nothing ran) = the budget rule
    return launch_bwd_band(d, dout, wprep_bwd, mask, din, B, best, stream);
  }
}
// rank != nullptr: dOut formed inside the kernel from (dl, Wc, mask bits of the layer above): bwd_x6_kernel<2, true> only
struct RankSrc { const float* dl; long ldl; int nlog; const float* Wc; const unsigned char* a2b; long a2b_row; };
int conv_bwd_data_generic(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask, float* din, int B,
                          a2c_stream_t stream, const unsigned long long* lmask = nullptr, bool probe_only = false,
                          const RankSrc* rank = nullptr);
}  // namespace
extern "C" {
int a2c_conv2d_bwd_data(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask,
                        float* din, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dout || !wprep_bwd || !din) return A2C_ERR_ARG;
  if (c3_supported(d, 1) && (!mask || c3_bwd_mask_supported(d)) && ((uintptr_t)dout % 16) == 0 && ((uintptr_t)din % 16) == 0 &&
      (!mask || ((uintptr_t)mask % 16) == 0))
    return c3_bwd_data(d, dout, wprep_bwd + prep_floats_base(d, 1), mask, nullptr, 0, din, B, a2c_s(stream));
  return conv_bwd_data_generic(d, dout, wprep_bwd, mask, din, B, stream);
}

int a2c_conv2d_bwd_data_signs_supported(const a2c_conv_desc* d) { return desc_ok(d) && c3_bwd_signs_supported(d) ? 1 : 0; }

/* debug: device buffer of 8 waves x 8 uint64 that receives the summed phase stamps (shader clocks; [7] = samples) of workgroup
 * 0 of every following bwd_stream2_kernel launch; NULL switches it off.  Not part of the drop-in boundary. */
int a2c_debug_bwd_stream_timing(unsigned long long* dev_buf) {
  g_bs2_dbg = dev_buf;
  return A2C_OK;
}
/* include/a2c_mi355x.h: lane masks of an activation tensor, and the backward-data pass that takes them as its ReLU mask */
int a2c_lanemask_from_act(const float* act, uint64_t* lanemask, int64_t n_floats, a2c_stream_t stream) {
  if (n_floats < 0 || n_floats % 256) return A2C_ERR_ARG;
  if (n_floats == 0) return A2C_OK;
  if (!act || !lanemask || ((uintptr_t)act % 16) || ((uintptr_t)lanemask % 8)) return A2C_ERR_ARG;
  const long n4 = n_floats / 4;
  const int grid = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
  hipLaunchKernelGGL(lanemask_kernel, dim3(grid), dim3(256), 0, a2c_s(stream), act, (unsigned char*)lanemask, n4);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
int a2c_conv2d_bwd_data_lanemask_supported(const a2c_conv_desc* d, int B) {
  if (!desc_ok(d) || B < 1 || (d->Cin * d->H * d->W) % 256) return 0;
  if (c3_supported(d, 1)) return 0;          // (the 3x3 stacks have their own sign-word kernels)
  static const unsigned long long probe = 0;
  return conv_bwd_data_generic(d, (const float*)16, (const float*)16, nullptr, (float*)16, B, nullptr, &probe, true) == A2C_OK ? 1 : 0;
}
int a2c_conv2d_bwd_data_lanemask(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const uint64_t* lanemask,
                                 float* din, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dout || !wprep_bwd || !din || !lanemask || ((uintptr_t)lanemask % 8)) return A2C_ERR_ARG;
  if (!a2c_conv2d_bwd_data_lanemask_supported(d, B)) return A2C_ERR_ARG;
  return conv_bwd_data_generic(d, dout, wprep_bwd, nullptr, din, B, stream, (const unsigned long long*)lanemask);
}

/* include/a2c_mi355x.h: the two backward passes of the layer below a rank-n_logits head with dOut formed in the kernels */
int a2c_conv2d_bwd_rank_supported(const a2c_conv_desc* d, int n_logits, int B) {
  if (!desc_ok(d) || B < 1 || n_logits < 1 || n_logits > 4 || (d->Cin * d->H * d->W) % 256 || c3_supported(d, 1)) return 0;
  { const char* e = getenv("A2C_WGRAD_X6"); if (e && e[0] == '0') return 0; }
  static const unsigned long long probe = 0;
  static const RankSrc rprobe = {(const float*)16, 4, 1, (const float*)16, (const unsigned char*)16, 324};
  if (conv_bwd_data_generic(d, (const float*)16, (const float*)16, nullptr, (float*)16, B, nullptr, &probe, true, &rprobe) != A2C_OK) return 0;
  WgradPlan pl;
  if (!plan_wgrad(d, B, pl, true) || pl.run != 2) return 0;
  const int grid = stream_grid() < pl.grid ? stream_grid() : pl.grid;
  return B >= 8 * grid ? 1 : 0;
}
int a2c_conv2d_bwd_data_lanemask_rank(const a2c_conv_desc* d, const float* dl, int64_t ld_dl, int n_logits, const float* Wc,
                                      const uint8_t* maskbits, int64_t mask_row_bytes, const float* wprep_bwd,
                                      const uint64_t* lanemask, float* din, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dl || !Wc || !maskbits || !wprep_bwd || !din || !lanemask || ((uintptr_t)lanemask % 8) || ((uintptr_t)din % 16)) return A2C_ERR_ARG;
  if (ld_dl < n_logits || mask_row_bytes < (int64_t)d->Cout * d->OH * d->OW / 8 || !a2c_conv2d_bwd_rank_supported(d, n_logits, B)) return A2C_ERR_ARG;
  const RankSrc rk = {dl, (long)ld_dl, n_logits, Wc, maskbits, (long)mask_row_bytes};
  return conv_bwd_data_generic(d, nullptr, wprep_bwd, nullptr, din, B, stream, (const unsigned long long*)lanemask, false, &rk);
}

int a2c_conv2d_bwd_data_signs(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const uint32_t* signs,
                              int64_t signs_bstride, float* din, int B, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dout || !wprep_bwd || !din || !c3_bwd_signs_supported(d)) return A2C_ERR_ARG;
  if (((uintptr_t)dout % 16) || ((uintptr_t)din % 16)) return A2C_ERR_ARG;
  if (signs && signs_bstride < (int64_t)d->Cin * d->H * ((d->W + 31) / 32)) return A2C_ERR_ARG;
  return c3_bwd_data(d, dout, wprep_bwd + prep_floats_base(d, 1), nullptr, signs, (long)signs_bstride, din, B, a2c_s(stream));
}

/* include/a2c_mi355x.h: layer 2's backward-data fused with the first layer's weight gradient from the uint8 frame store */
size_t a2c_conv2d_bwd_data_w1_frames_ws_bytes(const a2c_conv_desc* d2, const a2c_conv_desc* d1, int B) {
  if (!desc_ok(d2) || !desc_ok(d1) || B < 1) return 0;
  return c3_bwd_data_w1_frames_ws_bytes(d2, d1, B);
}
int a2c_conv2d_bwd_data_w1_frames(const a2c_conv_desc* d2, const float* dout, const float* wprep_bwd, const uint32_t* signs,
                                  int64_t signs_bstride, const a2c_conv_desc* d1, const uint8_t* frame_store, int64_t slot_stride,
                                  int64_t T, const int32_t* nvalid, float* dW1, float* db1, int B, void* ws, size_t ws_bytes,
                                  a2c_stream_t stream) {
  if (!desc_ok(d2) || !desc_ok(d1) || B < 0) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!dout || !wprep_bwd || !signs || !frame_store || !nvalid || !dW1 || !db1 || !ws || T < 1) return A2C_ERR_ARG;
  if (!c3_bwd_data_w1_frames_supported(d2, d1)) return A2C_ERR_ARG;
  if (((uintptr_t)dout % 16) || ((uintptr_t)frame_store % 16) || slot_stride % 16 || slot_stride < (T + 3) * (int64_t)d1->H * d1->W ||
      ((uintptr_t)ws % 16))
    return A2C_ERR_ARG;
  if (signs_bstride < (int64_t)d2->Cin * d2->H * ((d2->W + 31) / 32)) return A2C_ERR_ARG;
  return c3_bwd_data_w1_frames(d2, d1, dout, wprep_bwd + prep_floats_base(d2, 1), signs, (long)signs_bstride, frame_store, (long)slot_stride,
                               (long)T, nvalid, dW1, db1, B, ws, ws_bytes, a2c_s(stream));
}
}  // extern "C"
namespace {
// lmask != nullptr: the mask as lane masks (bwd_stream2_kernel); only the streaming path reads them: A2C_ERR_ARG otherwise
int conv_bwd_data_generic(const a2c_conv_desc* d, const float* dout, const float* wprep_bwd, const float* mask, float* din, int B,
                          a2c_stream_t stream, const unsigned long long* lmask, bool probe_only, const RankSrc* rank) {
  const int S = d->stride, P = d->pad;
  {  // fused-class pipelined path (unpadded ks = 2S layers whose dOut sample fits the prefetch registers)
    const int MTb = ceil_div(d->Cin, 16), c4n = d->Cout / 4;
    const size_t nfrag = bwd_class_offset(d, S * S);
    const int nel = d->Cout * d->OH * d->OW;
    if (run_layout(d) && S * S <= MAX_CLS && MTb <= 2 && nel <= 256 * 4 * PF_B && (4 * c4n) % CH == 0 &&
        nfrag * 4 <= 48 * 1024 && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)din % 16 == 0) &&
        (!mask || (uintptr_t)mask % 16 == 0) && (d->H * d->W) % 4 == 0 && !getenv("A2C_NO_PF")) {
      BwdFusedP q;
      q.dout = dout; q.din = din; q.mask = mask; q.wfrag = wprep_bwd;
      q.Cout = d->Cout; q.OH = d->OH; q.OW = d->OW; q.Cin = d->Cin; q.H = d->H; q.W = d->W; q.S = S; q.B = B;
      q.WP = d->OW + 2;
      q.PLANE = (((d->OH + 2) * q.WP + 31) / 32) * 32 + 16;
      q.nfrag = (int)nfrag; q.ncls = S * S; q.nb = 2; q.c4n = c4n;
      q.off0 = q.WP + 1; q.step_a = -q.WP; q.step_b = -1; q.step_c = 4 * q.PLANE;
      for (int cls = 0; cls < S * S; ++cls) {
        const int ry = cls / S, rx = cls % S;
        q.cls[cls].frag_off = (int)bwd_class_offset(d, cls);
        q.cls[cls].nsteps = 4 * c4n;
        q.cls[cls].PH = (d->H - 1 - ry) / S + 1;
        q.cls[cls].PW = (d->W - 1 - rx) / S + 1;
        q.cls[cls].oy_add = ry; q.cls[cls].ox_add = rx;
      }
      if (S == 2 && d->ks == 4 && MTb == 1 && c4n == 8 && nel % 4 == 0 && nel <= BS_PD * BS_NT * 4 &&
          d->Cin * d->H * d->W <= BS_PM * BS_NT * 4 && B >= 8 * stream_grid() && !getenv("A2C_NO_STREAM")) {
        BstreamP sp;
        sp.dout = dout; sp.din = din; sp.mask = mask; sp.wfrag = wprep_bwd;
        sp.Cout = d->Cout; sp.OH = d->OH; sp.OW = d->OW; sp.Cin = d->Cin; sp.H = d->H; sp.W = d->W; sp.B = B;
        sp.WP = q.WP; sp.PLANE = q.PLANE; sp.off0 = q.off0; sp.step_a = q.step_a; sp.step_b = q.step_b; sp.step_c = q.step_c;
        for (int cls = 0; cls < 4; ++cls) sp.cls[cls] = q.cls[cls];
        const int n4 = d->Cin * d->H * d->W / 4;
        // second form (two dX images, flush under the next sample's matrix phase, lane masks): A2C_BWD_STREAM_V1=1 keeps the first
        const size_t slds2 = 4 * ((size_t)d->Cout * q.PLANE + 64 + 2 * (size_t)d->Cin * d->H * d->W);
        const char* v1 = getenv("A2C_BWD_STREAM_V1");
        const bool form2 = slds2 <= LDS_HARD_MAX && !(v1 && v1[0] == '1') && (!lmask || n4 % 64 == 0);
        if (lmask && !form2) return A2C_ERR_ARG;
        if (probe_only && !rank) return A2C_OK;
        {  // the bf16 x 6 form (bwd_x6_kernel): exactly A3CModel's conv2, mask as bits or none; A2C_BWD_X6=0 keeps the fp32 MFMA kernels
          const char* x6 = getenv("A2C_BWD_X6");
          if (d->Cout == 32 && d->Cin == 16 && d->OH == 9 && d->OW == 9 && d->H == 20 && d->W == 20 && P == 0 && (lmask || !mask) &&
              !(x6 && x6[0] == '0')) {
            if (probe_only) return A2C_OK;
            BwdX6P xp;
            xp.dout = dout; xp.din = din; xp.wfrag = wprep_bwd; xp.lmask = lmask; xp.lmw = n4 / 64 * 4; xp.B = B;
            for (int cls = 0; cls < 4; ++cls) xp.frag_off[cls] = q.cls[cls].frag_off;
            xp.dl = nullptr; xp.ldl = 0; xp.nlog = 0; xp.Wc = nullptr; xp.a2b = nullptr; xp.a2b_row = 0;
            if (rank) { xp.dl = rank->dl; xp.ldl = rank->ldl; xp.nlog = rank->nlog; xp.Wc = rank->Wc; xp.a2b = rank->a2b; xp.a2b_row = rank->a2b_row; }
            const size_t xlds = 2 * (size_t)BX_IMG + 2 * 4 * (size_t)d->Cin * d->H * d->W + (rank ? 16 * 2592 : 0);
            const void* xk = rank ? (const void*)bwd_x6_kernel<2, true> : lmask ? (const void*)bwd_x6_kernel<2> : (const void*)bwd_x6_kernel<0>;
            if (xlds > 64 * 1024) (void)hipFuncSetAttribute(xk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xlds);
            const int xgrid = resident_grid(xk, xlds, B, BX_NT);
            if (rank) hipLaunchKernelGGL((bwd_x6_kernel<2, true>), dim3(xgrid), dim3(BX_NT), xlds, a2c_s(stream), xp);
            else if (lmask) hipLaunchKernelGGL(bwd_x6_kernel<2>, dim3(xgrid), dim3(BX_NT), xlds, a2c_s(stream), xp);
            else hipLaunchKernelGGL(bwd_x6_kernel<0>, dim3(xgrid), dim3(BX_NT), xlds, a2c_s(stream), xp);
            A2C_CHECK_LAUNCH();
            return A2C_OK;
          }
        }
        if (rank) return A2C_ERR_ARG;
        // third form (one barrier per sample, both images double buffered, the two waves of a SIMD out of step): A2C_BWD_STREAM_FORM=3
        const size_t slds3 = 4 * (2 * ((size_t)d->Cout * q.PLANE + 64) + 2 * (size_t)d->Cin * d->H * d->W);
        const char* fm = getenv("A2C_BWD_STREAM_FORM");
        if (form2 && slds3 <= LDS_HARD_MAX && fm && fm[0] == '3') {        // OPT-IN: measured slower (0.855-0.913 vs 0.74 ms)
          Bstream2P s3;
          s3.s = sp; s3.lmask = lmask; s3.lmw = n4 / 64 * 4;
          { const char* o = getenv("A2C_BS3_ORDER"); s3.order = o ? atoi(o) : 0; }
          s3.dbg = nullptr;
          const void* sk = lmask ? (const void*)bwd_stream3_kernel<2> : mask ? (const void*)bwd_stream3_kernel<1> : (const void*)bwd_stream3_kernel<0>;
          if (slds3 > 64 * 1024) (void)hipFuncSetAttribute(sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds3);
          const int sgrid = resident_grid(sk, slds3, B, BS_NT);
          if (lmask) hipLaunchKernelGGL(bwd_stream3_kernel<2>, dim3(sgrid), dim3(BS_NT), slds3, a2c_s(stream), s3);
          else if (mask) hipLaunchKernelGGL(bwd_stream3_kernel<1>, dim3(sgrid), dim3(BS_NT), slds3, a2c_s(stream), s3);
          else hipLaunchKernelGGL(bwd_stream3_kernel<0>, dim3(sgrid), dim3(BS_NT), slds3, a2c_s(stream), s3);
          A2C_CHECK_LAUNCH();
          return A2C_OK;
        }
        if (form2) {
          Bstream2P s2;
          s2.s = sp; s2.lmask = lmask; s2.lmw = n4 / 64 * 4; s2.order = 0; s2.dbg = g_bs2_dbg;
          const void* sk = lmask ? (const void*)bwd_stream2_kernel<2> : mask ? (const void*)bwd_stream2_kernel<1> : (const void*)bwd_stream2_kernel<0>;
          if (slds2 > 64 * 1024) (void)hipFuncSetAttribute(sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds2);
          const int sgrid = resident_grid(sk, slds2, B, BS_NT);
          if (lmask && s2.dbg) hipLaunchKernelGGL((bwd_stream2_kernel<2, true>), dim3(sgrid), dim3(BS_NT), slds2, a2c_s(stream), s2);
          else if (lmask) hipLaunchKernelGGL(bwd_stream2_kernel<2>, dim3(sgrid), dim3(BS_NT), slds2, a2c_s(stream), s2);
          else if (mask) hipLaunchKernelGGL(bwd_stream2_kernel<1>, dim3(sgrid), dim3(BS_NT), slds2, a2c_s(stream), s2);
          else hipLaunchKernelGGL(bwd_stream2_kernel<0>, dim3(sgrid), dim3(BS_NT), slds2, a2c_s(stream), s2);
          A2C_CHECK_LAUNCH();
          return A2C_OK;
        }
        const size_t slds = 4 * ((size_t)d->Cout * q.PLANE + 64 + (size_t)d->Cin * d->H * d->W);
        const void* sk = mask ? (const void*)bwd_stream_kernel<true> : (const void*)bwd_stream_kernel<false>;
        if (slds > 64 * 1024) (void)hipFuncSetAttribute(sk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slds);
        const int sgrid = resident_grid(sk, slds, B, BS_NT);
        if (mask) hipLaunchKernelGGL(bwd_stream_kernel<true>, dim3(sgrid), dim3(BS_NT), slds, a2c_s(stream), sp);
        else hipLaunchKernelGGL(bwd_stream_kernel<false>, dim3(sgrid), dim3(BS_NT), slds, a2c_s(stream), sp);
        A2C_CHECK_LAUNCH();
        return A2C_OK;
      }
      if (lmask) return A2C_ERR_ARG;
      const size_t lds = 4 * (nfrag + (size_t)d->Cout * q.PLANE + 64 + (size_t)d->Cin * d->H * d->W);
      if (lds > LDS_HARD_MAX) return A2C_ERR_ARG;
      const void* k = MTb == 1 ? (const void*)bwd_fused_kernel<1> : (const void*)bwd_fused_kernel<2>;
      if (lds > 64 * 1024) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const int grid = resident_grid(k, lds, B);
      if (MTb == 1) hipLaunchKernelGGL(bwd_fused_kernel<1>, dim3(grid), dim3(256), lds, a2c_s(stream), q);
      else hipLaunchKernelGGL(bwd_fused_kernel<2>, dim3(grid), dim3(256), lds, a2c_s(stream), q);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  if (lmask) return A2C_ERR_ARG;           // only the streaming kernel reads lane masks
  if (S * S <= MAX_CLS && !getenv("A2C_NO_BAND") && ((uintptr_t)din % 16 == 0) && (!mask || (uintptr_t)mask % 16 == 0)) {
    const int rc = bwd_band_tuned(d, dout, wprep_bwd, mask, din, B, stream);
    if (rc != BAND_NA) return rc;
  }
  for (int cls = 0; cls < S * S; ++cls) {
    const int ry = cls / S, rx = cls % S;
    const int na = ntaps_1d(d->ks, S, ry), nb = ntaps_1d(d->ks, S, rx);
    // dX row y = S*q + ry - P, q in [qy0, qy1]
    const int qy0 = (P - ry) > 0 ? ceil_div(P - ry, S) : 0;
    const int qx0 = (P - rx) > 0 ? ceil_div(P - rx, S) : 0;
    const int ny = (d->H - 1 + P - ry) >= 0 ? (d->H - 1 + P - ry) / S - qy0 + 1 : 0;
    const int nx = (d->W - 1 + P - rx) >= 0 ? (d->W - 1 + P - rx) / S - qx0 + 1 : 0;
    if (ny <= 0 || nx <= 0) continue;
    SrcTile t;
    t.Cp = d->Cout; t.IH = d->OH; t.IW = d->OW; t.SY = 1; t.SX = 1;
    const int sa = na > 0 ? na : 1, sb = nb > 0 ? nb : 1;
    t.sy0 = qy0 - (sa - 1); t.sx0 = qx0 - (sb - 1); t.span_y = sa; t.span_x = sb; t.PH = ny; t.PW = nx;
    plan_src(t, 0, 0, IGEMM_LDS_BUDGET);
    IgemmP p;
    fill_stage(p.st, t, dout, (long)d->Cout * d->OH * d->OW);
    p.out = din; p.out_bs = (long)d->Cin * d->H * d->W; p.Mch = d->Cin; p.OHf = d->H; p.OWf = d->W;
    p.wfrag = wprep_bwd + bwd_class_offset(d, cls); p.bias = nullptr; p.mask = mask; p.relu = 0;
    p.out_stage = 0; p.out_vec = 0;
    p.PH = ny; p.PW = nx; p.oy_mul = S; p.oy_add = S * qy0 + ry - P; p.ox_mul = S; p.ox_add = S * qx0 + rx - P;
    p.SY = 1; p.SX = 1; p.sy0 = t.sy0; p.TPH = t.TPH; p.tiles = t.tiles; p.B = B;
    const int c4n = d->Cout / 4;
    p.nchunks = pad_steps(na * nb * c4n) / CH;
    p.nsteps = na * nb * c4n; p.nb = sb; p.c4n = c4n;
    p.off0 = (sa - 1) * t.WP + (sb - 1); p.step_a = -t.WP; p.step_b = -1; p.step_c = 4 * t.PLANE;
    const int rc = launch_igemm(p, ceil_div(d->Cin, 16), a2c_s(stream));
    if (rc != A2C_OK) return rc;
  }
  return A2C_OK;
}
}  // namespace
extern "C" {

size_t a2c_conv2d_bwd_data_w1_ws_bytes(const a2c_conv_desc* d2, const a2c_conv_desc* d1, int B) {
  BandW1Plan pl;
  if (B < 1 || !plan_band_w1(d2, d1, B, nullptr, pl)) return 0;
  // the vector width of the dOut staging depends on the pointer alignment, known at launch only: size for any of them
  int grid = 0;
  for (int v : {1, 2, 4}) {
    if (d2->OW % v) continue;
    pl.vec = v;
    const int g = band_w1_grid(pl, B);
    if (g > grid) grid = g;
  }
  return (size_t)grid * ((size_t)d1->Cout * d1->Cin * 9 + d1->Cout) * sizeof(float);
}

int a2c_conv2d_bwd_data_w1(const a2c_conv_desc* d2, const float* dout, const float* wprep_bwd, const float* mask,
                           const a2c_conv_desc* d1, const float* x, int64_t x_bstride, float* dW1, float* db1, int B,
                           void* ws, size_t ws_bytes, a2c_stream_t stream) {
  if (B < 1 || !dout || !wprep_bwd || !mask || !x || !dW1 || !db1) return A2C_ERR_ARG;
  BandW1Plan pl;
  if (((uintptr_t)mask % 16) || ((uintptr_t)x % 16) || x_bstride % 4 || !plan_band_w1(d2, d1, B, dout, pl)) return A2C_ERR_ARG;
  pl.q2.b.wfrag = wprep_bwd; pl.q2.b.mask = mask; pl.q2.b.din = nullptr;
  pl.fw.x = x; pl.fw.x_bs = x_bstride; pl.fw.slab = (float*)ws;
  const int grid = band_w1_grid(pl, B);
  const size_t per = (size_t)d1->Cout * pl.fw.K1 + d1->Cout;
  if (grid < 1) return A2C_ERR_ARG;
  if (!ws || ws_bytes < (size_t)grid * per * sizeof(float)) return A2C_ERR_WORKSPACE;
  hipStream_t st = a2c_s(stream);
#define W1_LAUNCH(V_, P_)                                                                                         \
  if (pl.vec == V_ && pl.pfb == P_)                                                                               \
    hipLaunchKernelGGL((bwd_band2_kernel<1, V_, P_, true>), dim3(grid), dim3(256), pl.lds, st, pl.q2, pl.fw);
  BAND_W1_CASES(W1_LAUNCH)
#undef W1_LAUNCH
  A2C_CHECK_LAUNCH();
  const long nW = (long)d1->Cout * pl.fw.K1;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d((long)per, 256)), dim3(256), 0, st, (const float*)ws, grid,
                     (long)per, nW, dW1, db1);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

size_t a2c_conv2d_bwd_weight_ws_bytes(const a2c_conv_desc* d, int B) {
  WgradPlan pl, pg;
  if (!desc_ok(d) || B < 0 || !plan_wgrad(d, B, pl) || !plan_wgrad(d, B, pg, false)) return 0;
  int grid = pl.grid > pg.grid ? pl.grid : pg.grid;           // any of the kernels may be picked at launch
  WstreamP wp;
  if (plan_wstream(d, wp) && stream_grid() > grid) grid = stream_grid();
  const size_t base = (size_t)grid * ((size_t)d->Cout * d->Cin * d->ks * d->ks + d->Cout) * sizeof(float);
  const size_t c3 = c3w_ws_bytes(d);                           // conv3.hip's streaming kernel: one slab per pixel group
  return c3 > base ? c3 : base;
}

int a2c_conv2d_bwd_weight_frames(const a2c_conv_desc* d, const uint8_t* fstore, int64_t slot_stride, int64_t T,
                                 const int* nvalid, const float* dout, float* dW, float* db, int B, void* ws, size_t ws_bytes,
                                 a2c_stream_t stream) {
  if (!desc_ok(d) || B < 1 || !fstore || !nvalid || !dout || !dW || T < 1) return A2C_ERR_ARG;
  if (c3w_supported(d) && d->Cin == 4) {          // first layer of the 3x3 stacks (ConvModel / GRUModel): conv3.hip's streaming kernel
    if (((uintptr_t)fstore % 4) || slot_stride % 4 || ((uintptr_t)dout % 16) || ((uintptr_t)dW % 16) || ((uintptr_t)ws % 16) ||
        slot_stride < (T + 3) * (int64_t)d->H * d->W)
      return A2C_ERR_ARG;
    if (!ws || ws_bytes < a2c_conv2d_bwd_weight_ws_bytes(d, B)) return A2C_ERR_WORKSPACE;
    return c3w_bwd_weight_frames(d, fstore, (long)slot_stride, (long)T, nvalid, dout, dW, db, B, ws, ws_bytes, a2c_s(stream));
  }
  WstreamP wp;
  if (!plan_wstream(d, wp) || (d->H * d->W) % 16 || ((uintptr_t)fstore % 16) || slot_stride % 16 || ((uintptr_t)dout % 16) ||
      slot_stride < (T + 3) * (int64_t)d->H * d->W)
    return A2C_ERR_ARG;
  if (!ws || ws_bytes < a2c_conv2d_bwd_weight_ws_bytes(d, B)) return A2C_ERR_WORKSPACE;
  hipStream_t st = a2c_s(stream);
  const int grid = stream_grid();
  wp.in = nullptr; wp.in_bs = 0; wp.dout = dout; wp.slab = (float*)ws; wp.B = B;
  wp.fstore = fstore; wp.fs_slot_stride = (long)slot_stride; wp.T = (int)T; wp.nvalid = nvalid;
  {  // the bf16-pipe form (exact 3-way split of dOut; uint8 pixels are exact in bf16); A2C_WGRAD_F32=1: the fp32 MFMAs below
    const char* e32 = getenv("A2C_WGRAD_F32");       // (read per call: A/B runs, tests)
    WsbGeo gq;
    size_t ldsb = 0;
    if (!(e32 != nullptr && e32[0] == '1') && plan_wstream_bf16(d, wp, gq, ldsb)) {
      static bool attrb = false;
      if (!attrb) {
        if (hipFuncSetAttribute((const void*)wgrad_stream_bf16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return A2C_ERR_LAUNCH;
        attrb = true;
      }
      const int gb = grid < B ? grid : B;
      hipLaunchKernelGGL(wgrad_stream_bf16_kernel<0>, dim3(gb), dim3(ST_NT), ldsb, st, wp, gq);
      A2C_CHECK_LAUNCH();
      const long nWb = (long)wp.Cout * wp.K, perb = nWb + wp.Cout;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(perb, 256)), dim3(256), 0, st, (const float*)ws, gb, perb, nWb, dW, db);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  const size_t lds = 4 * (size_t)(4 * wp.PLANE1 + 16 * wp.PLANEo);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute((const void*)wgrad_stream_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return A2C_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(wgrad_stream_kernel<true>, dim3(grid < B ? grid : B), dim3(ST_NT), lds, st, wp);
  A2C_CHECK_LAUNCH();
  const int g = grid < B ? grid : B;
  const long nWs = (long)wp.Cout * wp.K, pers = nWs + wp.Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(pers, 256)), dim3(256), 0, st, (const float*)ws, g, pers, nWs, dW, db);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_conv2d_bwd_weight_rank(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* dl, int64_t ld_dl,
                               int n_logits, const float* Wc, const uint8_t* maskbits, int64_t mask_row_bytes, float* dW, float* db,
                               int B, void* ws, size_t ws_bytes, a2c_stream_t stream) {
  if (!desc_ok(d) || B < 1 || !in || !dl || !Wc || !maskbits || !dW || !db) return A2C_ERR_ARG;
  if (in_bstride % 4 || ((uintptr_t)in % 16) || ld_dl < n_logits || mask_row_bytes < (int64_t)d->Cout * d->OH * d->OW / 8 ||
      in_bstride < (int64_t)d->Cin * d->H * d->W || !a2c_conv2d_bwd_rank_supported(d, n_logits, B))
    return A2C_ERR_ARG;
  if (!ws || ws_bytes < a2c_conv2d_bwd_weight_ws_bytes(d, B)) return A2C_ERR_WORKSPACE;
  WgradPlan pl;
  if (!plan_wgrad(d, B, pl, true)) return A2C_ERR_ARG;
  const int grid = stream_grid() < pl.grid ? stream_grid() : pl.grid;
  hipStream_t st = a2c_s(stream);
  static bool attrr = false;
  if (!attrr) {
    if (hipFuncSetAttribute((const void*)wgrad_x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wx::LDS_BYTES) != hipSuccess)
      return A2C_ERR_LAUNCH;
    attrr = true;
  }
  WgradX6P xp;
  xp.in = in; xp.in_bs = (long)in_bstride; xp.dout = nullptr; xp.slab = (float*)ws; xp.B = B; xp.dbg = 0;
  xp.dl = dl; xp.ldl = (long)ld_dl; xp.nlog = n_logits; xp.Wc = Wc; xp.a2b = maskbits; xp.a2b_row = (long)mask_row_bytes;
  const char* x6 = getenv("A2C_WGRAD_X6");
  if (x6 && x6[0] == '2') {
    static bool attrq = false;
    if (!attrq) {
      if (hipFuncSetAttribute((const void*)wgrad_x6p_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wxp::LDS_BYTES) != hipSuccess)
        return A2C_ERR_LAUNCH;
      attrq = true;
    }
    hipLaunchKernelGGL(wgrad_x6p_kernel<true>, dim3(grid), dim3(wxp::NT), wxp::LDS_BYTES, st, xp);
  } else
  hipLaunchKernelGGL(wgrad_x6_kernel<true>, dim3(grid), dim3(wx::NT), wx::LDS_BYTES, st, xp);
  A2C_CHECK_LAUNCH();
  const long nWx = 32L * 256, perx = nWx + 32;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(perx, 256)), dim3(256), 0, st, (const float*)ws, grid, perx, nWx, dW, db);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_conv2d_bwd_weight(const a2c_conv_desc* d, const float* in, int64_t in_bstride, const float* dout, float* dW,
                          float* db, int B, void* ws, size_t ws_bytes, a2c_stream_t stream) {
  WgradPlan pl;
  if (!desc_ok(d) || B < 1 || !in || !dout || !dW) return A2C_ERR_ARG;
  const bool aligned = (in_bstride % 4 == 0) && ((uintptr_t)in % 16 == 0) && ((uintptr_t)dout % 8 == 0);
  if (!plan_wgrad(d, B, pl, aligned)) return A2C_ERR_ARG;
  if (!ws || ws_bytes < a2c_conv2d_bwd_weight_ws_bytes(d, B)) return A2C_ERR_WORKSPACE;
  hipStream_t st = a2c_s(stream);
  if (c3w_supported(d) && aligned && ((uintptr_t)dout % 16 == 0) && ((uintptr_t)dW % 16 == 0) && ((uintptr_t)ws % 16 == 0))
    return c3w_bwd_weight(d, in, (long)in_bstride, dout, dW, db, B, ws, ws_bytes, st);
  {  // streaming kernel: A3C conv1 class at large batch
    WstreamP wp;
    const int grid = stream_grid();
    if (aligned && B >= 8 * grid && ((uintptr_t)dout % 16 == 0) && !getenv("A2C_NO_STREAM") && !getenv("A2C_NO_PF") &&
        plan_wstream(d, wp)) {
      wp.in = in; wp.in_bs = in_bstride; wp.dout = dout; wp.slab = (float*)ws; wp.B = B;
      wp.fstore = nullptr; wp.fs_slot_stride = 0; wp.T = 1; wp.nvalid = nullptr;
      const size_t lds = 4 * (size_t)(4 * wp.PLANE1 + 16 * wp.PLANEo);
      static bool attr = false;
      if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad_stream_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
          return A2C_ERR_LAUNCH;
        attr = true;
      }
      hipLaunchKernelGGL(wgrad_stream_kernel<false>, dim3(grid), dim3(ST_NT), lds, st, wp);
      A2C_CHECK_LAUNCH();
      const long nWs = (long)wp.Cout * wp.K, pers = nWs + wp.Cout;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(pers, 256)), dim3(256), 0, st, (const float*)ws, grid, pers, nWs,
                         dW, db);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  {  // the bf16 x 6 form (wgrad_x6_kernel): exactly A3CModel's conv2 at streaming batch; A2C_WGRAD_X6=0 keeps the fp32 MFMA kernel
    const char* x6 = getenv("A2C_WGRAD_X6");
    const int grid = stream_grid() < pl.grid ? stream_grid() : pl.grid;
    if (pl.run == 2 && d->Cin == 16 && d->Cout == 32 && d->H == 20 && d->W == 20 && d->ks == 4 && d->stride == 2 && d->pad == 0 &&
        aligned && ((uintptr_t)dout % 16 == 0) && B >= 8 * grid && !(x6 && x6[0] == '0')) {
      static bool attrx = false;
      if (!attrx) {
        if (hipFuncSetAttribute((const void*)wgrad_x6_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wx::LDS_BYTES) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_x6_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, wx::LDS_BYTES) != hipSuccess)
          return A2C_ERR_LAUNCH;
        attrx = true;
      }
      WgradX6P xp;
      xp.in = in; xp.in_bs = (long)in_bstride; xp.dout = dout; xp.slab = (float*)ws; xp.B = B;
      xp.dl = nullptr; xp.ldl = 0; xp.nlog = 0; xp.Wc = nullptr; xp.a2b = nullptr; xp.a2b_row = 0;
      { const char* dg = getenv("A2C_WGRAD_X6_DBG"); xp.dbg = dg ? atoi(dg) : 0; }      // timing experiments only (wrong sums)
      if (x6 && x6[0] == '2') {              // OPT-IN (A2C_WGRAD_X6=2): conversion under the matrix phase -- measured slower, see the kernel
        static bool attrp = false;
        if (!attrp) {
          if (hipFuncSetAttribute((const void*)wgrad_x6p_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, wxp::LDS_BYTES) != hipSuccess)
            return A2C_ERR_LAUNCH;
          attrp = true;
        }
        hipLaunchKernelGGL(wgrad_x6p_kernel<false>, dim3(grid), dim3(wxp::NT), wxp::LDS_BYTES, st, xp);
      } else
      hipLaunchKernelGGL(wgrad_x6_kernel<false>, dim3(grid), dim3(wx::NT), wx::LDS_BYTES, st, xp);
      A2C_CHECK_LAUNCH();
      const long nWx = 32L * 256, perx = nWx + 32;
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(perx, 256)), dim3(256), 0, st, (const float*)ws, grid, perx, nWx, dW, db);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  if (pl.run) {
    WrunP q;
    fill_stage(q.st, pl.t, in, in_bstride);
    q.dout = dout; q.slab = (float*)ws;
    q.Cout = d->Cout; q.K = d->Cin * d->ks * d->ks; q.ks = d->ks; q.OH = d->OH; q.OW = d->OW; q.OWp = pl.OWp;
    q.TPH = pl.t.TPH; q.tiles = pl.t.tiles; q.B = B; q.PLANEo = pl.PLANEo;
    if (pl.run == 1) hipLaunchKernelGGL((wgrad_run_kernel<1, 4, 1, 5>), dim3(pl.grid), dim3(256), pl.lds, st, q);
    else hipLaunchKernelGGL((wgrad_run_kernel<2, 2, 2, 3>), dim3(pl.grid), dim3(256), pl.lds, st, q);
    A2C_CHECK_LAUNCH();
    const long nWr = (long)q.Cout * q.K, perr = nWr + q.Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(perr, 256)), dim3(256), 0, st, (const float*)ws, pl.grid,
                       perr, nWr, dW, db);
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  WgradP p;
  fill_stage(p.st, pl.t, in, in_bstride);
  p.dout = dout; p.slab = (float*)ws;
  p.Cout = d->Cout; p.K = d->Cin * d->ks * d->ks; p.ks = d->ks; p.OH = d->OH; p.OW = d->OW; p.OWp = pl.OWp;
  p.S = d->stride; p.sy0 = pl.t.sy0; p.TPH = pl.t.TPH; p.tiles = pl.t.tiles; p.B = B;
  p.PLANEo = pl.PLANEo; p.nkt = ceil_div(p.K, 16);
  p.dvec = ((d->OW % 4 == 0) && ((uintptr_t)dout % 16 == 0)) ? 4 : ((d->OW % 2 == 0) && ((uintptr_t)dout % 8 == 0)) ? 2 : 0;
  p.dflat = (pl.t.tiles == 1 && ((long)d->Cout * d->OH * d->OW) % 4 == 0 && ((uintptr_t)dout % 16 == 0) &&
             !getenv("A2C_NO_FLAT_STAGE")) ? 1 : 0;
#define WGRAD_LAUNCH(...) launch_wgrad_t<__VA_ARGS__>(p, pl.grid, pl.lds, st)
  WGRAD_VARIANTS(WGRAD_LAUNCH)
#undef WGRAD_LAUNCH
  A2C_CHECK_LAUNCH();
  const long nW = (long)p.Cout * p.K, per = nW + p.Cout;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(a2c_grid_1d(per, 256)), dim3(256), 0, st, (const float*)ws, pl.grid, per,
                     nW, dW, db);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
