// fp32 GEMM on the CDNA4 matrix cores for the dense layers of the model classes
// (proj_matrx / resize_emb / pi / value / GRU gates: models.py:40-47,246-264,472-475,626-636)
// and their backward passes.
//
// v_mfma_f32_32x32x2_f32: fp32 in, fp32 accumulate, bit-for-bit a k-ordered fmaf chain
// (no reduced precision; gfx950 has no xf32).  Workgroup = 256 threads = 4 waves (2x2), block
// tile 128x128x16, each wave a 64x64 sub-tile = 2x2 MFMA tiles (64 accumulator VGPRs).
// Operands are staged global -> registers -> LDS as As[k][m], Bs[k][n] so that an MFMA operand
// read is one conflict-free ds_read_b32 per lane (lane l: A[i=l&31][k=l>>5]); the next K-tile's
// global loads are issued before the current tile's MFMAs (register prefetch).
// Split-K writes partial slabs that a second kernel sums in fixed order (deterministic), which
// also applies the epilogue (bias, ReLU, ReLU-derivative mask of the layer below).
#include <mutex>
#include "a2c_common.h"

namespace {
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LD_T = 130;  // operand arrives k-contiguous: transposing scalar LDS stores, 4*LD_T % 32 == 8
constexpr int LD_D = 132;  // operand arrives m/n-contiguous: float4 LDS stores (16 B aligned rows)

struct Frag { float4 v[2]; };

// KC = true : rows of the block tile are k-contiguous in memory: elem(i,k) = P[(r0+i)*ld + k]
// KC = false: k-rows are i-contiguous in memory:                 elem(i,k) = P[k*ld + r0+i]
template <bool KC>
__device__ __forceinline__ void load_tile(Frag& f, const float* __restrict__ P, long ld, long r0, long R, long k0,
                                          long kend, bool vec) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = t + 256 * it;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (KC) {
      const long i = r0 + (q >> 2), k = k0 + (q & 3) * 4;
      if (i < R) {
        const float* p = P + i * ld + k;
        if (vec && k + 3 < kend) v = *reinterpret_cast<const float4*>(p);
        else {
          if (k + 0 < kend) v.x = p[0];
          if (k + 1 < kend) v.y = p[1];
          if (k + 2 < kend) v.z = p[2];
          if (k + 3 < kend) v.w = p[3];
        }
      }
    } else {
      const long k = k0 + (q >> 5), i = r0 + (q & 31) * 4;
      if (k < kend) {
        const float* p = P + k * ld + i;
        if (vec && i + 3 < R) v = *reinterpret_cast<const float4*>(p);
        else {
          if (i + 0 < R) v.x = p[0];
          if (i + 1 < R) v.y = p[1];
          if (i + 2 < R) v.z = p[2];
          if (i + 3 < R) v.w = p[3];
        }
      }
    }
    f.v[it] = v;
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(const Frag& f, float* __restrict__ S) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int q = t + 256 * it;
    const float4 v = f.v[it];
    if (KC) {
      const int i = q >> 2, k = (q & 3) * 4;
      S[(k + 0) * LD_T + i] = v.x;
      S[(k + 1) * LD_T + i] = v.y;
      S[(k + 2) * LD_T + i] = v.z;
      S[(k + 3) * LD_T + i] = v.w;
    } else {
      const int k = q >> 5, i = (q & 31) * 4;
      *reinterpret_cast<float4*>(&S[k * LD_D + i]) = v;
    }
  }
}

template <bool A_KC, bool B_KC, bool SKINNY>
__global__ __launch_bounds__(256) void gemm_kernel(long M, long N, long K, const float* __restrict__ A, long lda,
                                                   const float* __restrict__ B, long ldb, float* __restrict__ C,
                                                   long ldc, const float* __restrict__ bias, int relu,
                                                   const float* __restrict__ mask, long ldmask, int accumulate,
                                                   long k_per_split, float* __restrict__ slab, int vecA, int vecB, int vec_epi) {
  constexpr int LDA = A_KC ? LD_T : LD_D;
  constexpr int LDB = B_KC ? LD_T : LD_D;
  __shared__ __attribute__((aligned(16))) float smem[BK * LDA + BK * LDB];     // >= 4 x 32 x 32 floats: reused by the epilogue
  float* __restrict__ As = smem;
  float* __restrict__ Bs = smem + BK * LDA;
  const long m0 = (long)blockIdx.y * BM, n0 = (long)blockIdx.x * BN;
  const long kbeg = (long)blockIdx.z * k_per_split;
  const long kend = min(K, kbeg + k_per_split);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const bool rows0 = m0 + wr * 64 < M, rows1 = m0 + wr * 64 + 32 < M;
  Frag fa, fb;
  load_tile<A_KC>(fa, A, lda, m0, M, kbeg, kend, vecA);
  load_tile<B_KC>(fb, B, ldb, n0, N, kbeg, kend, vecB);
  for (long k0 = kbeg; k0 < kend; k0 += BK) {
    store_tile<A_KC>(fa, As);
    store_tile<B_KC>(fb, Bs);
    __syncthreads();
    if (k0 + BK < kend) {
      load_tile<A_KC>(fa, A, lda, m0, M, k0 + BK, kend, vecA);
      load_tile<B_KC>(fb, B, ldb, n0, N, k0 + BK, kend, vecB);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const float a0 = As[(kk + lk) * LDA + wr * 64 + li];
      const float a1 = As[(kk + lk) * LDA + wr * 64 + 32 + li];
      const float b0 = Bs[(kk + lk) * LDB + wc * 64 + li];
      const float b1 = Bs[(kk + lk) * LDB + wc * 64 + 32 + li];
      if (!SKINNY || rows0) {   // SKINNY (M <= 96, e.g. M = n_envs of a small rollout): wave-uniform skip of 32-row blocks that are all padding
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      }
      if (!SKINNY || rows1) {
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool direct = (slab == nullptr);
  if (direct && vec_epi) {
    // Row-major epilogue: each 32x32 block goes through this wave's 4 KB of LDS and leaves as
    // float4 rows, so bias / accumulate / ReLU-mask are 16 B loads and the stores full 128 B rows:
    // 4 + 4 vector memory instructions per block and lane instead of 16 + 16 dependent dword ones.
    float* __restrict__ stg = smem + w * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[i][j][r];
        const int c4 = (lane & 7) * 4;
        const long n = n0 + wc * 64 + j * 32 + c4;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && n < N) bv4 = *reinterpret_cast<const float4*>(bias + n);
        float4 mk[4], od[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long m = min(m0 + wr * 64 + i * 32 + q * 8 + (lane >> 3), M - 1);
          const long nc = min(n, N - 4);
          mk[q] = mask ? *reinterpret_cast<const float4*>(mask + m * ldmask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
          od[q] = accumulate ? *reinterpret_cast<const float4*>(C + m * ldc + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = q * 8 + (lane >> 3);
          const long m = m0 + wr * 64 + i * 32 + row;
          float4 v = *reinterpret_cast<const float4*>(stg + row * 32 + c4);
          v.x = (v.x + od[q].x) + bv4.x; v.y = (v.y + od[q].y) + bv4.y;      // same order as the scalar path
          v.z = (v.z + od[q].z) + bv4.z; v.w = (v.w + od[q].w) + bv4.w;
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (!(mk[q].x > 0.f)) v.x = 0.f;
          if (!(mk[q].y > 0.f)) v.y = 0.f;
          if (!(mk[q].z > 0.f)) v.z = 0.f;
          if (!(mk[q].w > 0.f)) v.w = 0.f;
          if (m < M && n < N) *reinterpret_cast<float4*>(C + m * ldc + n) = v;
        }
      }
    return;
  }
  float* out = direct ? C : slab + (long)blockIdx.z * M * N;
  const long ldo = direct ? ldc : N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long n = n0 + wc * 64 + j * 32 + li;
      if (n >= N) continue;
      const float bv = (direct && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (direct) {
          if (accumulate) v += out[m * ldo + n];
          v += bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mask && !(mask[m * ldmask + n] > 0.f)) v = 0.f;
        }
        out[m * ldo + n] = v;
      }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same block-tiled GEMM on the BF16 matrix pipe with fp32 results ("bf16 x 9", round 5).  Every fp32 operand element is
// split into three bf16 pieces a = a1 + a2 + a3 (round to nearest each; exact: 3 x 8 significant bits cover the 24 of an
// fp32), so a * b = sum over the 9 piece pairs of a_i * b_j with EVERY product exact (8 x 8 bits) and every sum in the MFMA's
// fp32 accumulator -- the real-number sum of the fp32 FMA chain, re-associated; against fp64 it is no less accurate than
// the fp32 MFMA form (test).  9 v_mfma_f32_32x32x16_bf16 (32 cycles each) per 16-deep k-step replace 8
// v_mfma_f32_32x32x2_f32 (64 cycles each): 0.56 of the matrix time -- on paper; measured it only ties the fp32 kernels on the
// 28224 x 2000 layers of ConvModel (see x9_eligible below), so the path is opt-in (A2C_GEMM_X9=1).
//   x9_split_kernel   one pass over each operand: fp32 (rows x k, either orientation) -> three bf16 images [piece][row][k],
//                     k-contiguous, rows padded to 128 and k to 32 with zeros (splitting inside the GEMM's staging phase
//                     cost more than the matrix phase it fed: 1.6 of 2.7 ms);
//   gemm_x9_kernel    C = A B^T over those images: tile 128 x 128 x 32, 4 waves of 64 x 64, 16-byte global -> register
//                     -> LDS staging with the next tile in flight, LDS rows of 80 bytes (a lane's 8 consecutive k = one
//                     conflict-free ds_read_b128), no bounds checks in the loop.
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4g __attribute__((ext_vector_type(4)));
constexpr int X9_BK = 32, X9_LD = 40;                 // bf16 elements per LDS row (80 B: rows 0..15 hit 16 distinct 16-byte bank groups)
constexpr int X9_PSZ = 128 * X9_LD;                   // one piece image of a 128-row operand tile

__device__ __forceinline__ void split1_x9(float e, unsigned short pc[3]) {
  const __bf16 h0 = (__bf16)e;
  const float r1 = e - (float)h0;                     // exact
  const __bf16 h1 = (__bf16)r1;
  const float r2 = r1 - (float)h1;                    // exact, at most 8 significant bits
  pc[0] = __builtin_bit_cast(unsigned short, h0);
  pc[1] = __builtin_bit_cast(unsigned short, h1);
  pc[2] = __builtin_bit_cast(unsigned short, (__bf16)r2);
}
// dst[q][r][k] (r < Rp, k < Kp; piece stride Rp * Kp) <- split(src), zero outside R x K.  Every thread produces 8 consecutive
// k of one row: three 16-byte stores.  TRANS = false: src[r * ld + k] (two 16-byte loads where aligned);
// TRANS = true: src[k * ld + r]: a tile of 32 k x 64 rows goes through LDS (loads coalesced along r).
__device__ __forceinline__ void split8_store(const float e[8], unsigned short* __restrict__ d, long pst) {
  unsigned int o[3][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    unsigned short a[3], b[3];
    split1_x9(e[2 * i], a);
    split1_x9(e[2 * i + 1], b);
#pragma unroll
    for (int q = 0; q < 3; ++q) o[q][i] = (unsigned int)a[q] | ((unsigned int)b[q] << 16);
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) *reinterpret_cast<u32x4g*>(d + q * pst) = (u32x4g){o[q][0], o[q][1], o[q][2], o[q][3]};
}
template <bool TRANS>
__global__ __launch_bounds__(256) void x9_split_kernel(const float* __restrict__ src, long ld, long R, long K, unsigned short* __restrict__ dst,
                                                       long Rp, long Kp, int vec) {
  const long pst = Rp * Kp;
  if (!TRANS) {
    const long nchunk = Rp * (Kp >> 3);
    for (long c = blockIdx.x * 256L + threadIdx.x; c < nchunk; c += gridDim.x * 256L) {
      const long r = c / (Kp >> 3), k = (c - r * (Kp >> 3)) << 3;
      float e[8];
      if (r < R && vec && k + 7 < K) {
        const float4 v0 = *reinterpret_cast<const float4*>(src + r * ld + k), v1 = *reinterpret_cast<const float4*>(src + r * ld + k + 4);
        e[0] = v0.x; e[1] = v0.y; e[2] = v0.z; e[3] = v0.w; e[4] = v1.x; e[5] = v1.y; e[6] = v1.z; e[7] = v1.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (r < R && k + i < K) ? src[r * ld + k + i] : 0.f;
      }
      split8_store(e, dst + r * Kp + k, pst);
    }
  } else {
    __shared__ float tile[32][65];
    const long r0 = (long)blockIdx.y * 64, k0 = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;       // 64 x 4
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long k = k0 + ty + 4 * i, r = r0 + tx;
      tile[ty + 4 * i][tx] = (k < K && r < R) ? src[k * ld + r] : 0.f;
    }
    __syncthreads();
    const int rr = threadIdx.x >> 2, kc = (threadIdx.x & 3) * 8;     // row 0..63, chunk of 8 k
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[kc + i][rr];
    if (r0 + rr < Rp) split8_store(e, dst + (r0 + rr) * Kp + k0 + kc, pst);
  }
}

struct Frag9 { u32x4g v[6]; };
// operand tile: 3 pieces x 128 rows x 32 k = 1536 chunks of 8 bf16 (16 B), six per thread
__device__ __forceinline__ void load_tile9(Frag9& f, const unsigned short* __restrict__ P, long pst, long Kp, long r0, long k0) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = t + 256 * it, q = c >> 9, rem = c & 511;
    f.v[it] = *reinterpret_cast<const u32x4g*>(P + q * pst + (r0 + (rem >> 2)) * Kp + k0 + 8 * (rem & 3));
  }
}
__device__ __forceinline__ void store_tile9(const Frag9& f, unsigned short* __restrict__ S) {
  const int t = threadIdx.x;
#pragma unroll
  for (int it = 0; it < 6; ++it) {
    const int c = t + 256 * it, q = c >> 9, rem = c & 511;
    *reinterpret_cast<u32x4g*>(S + q * X9_PSZ + (rem >> 2) * X9_LD + 8 * (rem & 3)) = f.v[it];
  }
}

__global__ __launch_bounds__(256) void gemm_x9_kernel(long M, long N, long K, const unsigned short* __restrict__ Ap, long apst,
                                                      const unsigned short* __restrict__ Bp, long bpst, long Kp, float* __restrict__ C,
                                                      long ldc, const float* __restrict__ bias, int relu,
                                                      const float* __restrict__ mask, long ldmask, int accumulate,
                                                      long k_per_split, float* __restrict__ slab, int vec_epi) {
  __shared__ __attribute__((aligned(16))) unsigned short smem9[2 * 3 * X9_PSZ];          // 61,440 B; >= 4 x 32 x 32 floats for the epilogue
  unsigned short* __restrict__ As = smem9;
  unsigned short* __restrict__ Bs = smem9 + 3 * X9_PSZ;
  float* __restrict__ smem = reinterpret_cast<float*>(smem9);
  // XCD-aware tile order: workgroup ids go round robin over the 8 XCDs (each with its own L2); XCD x walks ITS column
  // blocks x, x + 8, ... and, per column block, all row blocks back to back -- the 128-row B image tile (K x 768 B) is
  // fetched from HBM once and reused from that XCD's L2 by every row block; the A image streams from the infinity cache.
  // (The plain x-fastest order re-read B from HBM once per row block: 5.5 GB per 2048 x 28224 x 2000 product, 2.1 ms.)
  const long nmb = gridDim.y, nnb = gridDim.x;
  long bm, bn;
  {
    const long bid = (long)blockIdx.y * gridDim.x + blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const long ncol_x = (nnb - xcd + 7) >> 3;          // column blocks of this XCD
    if (nnb >= 8 && idx < ncol_x * nmb) { bm = idx % nmb; bn = (idx / nmb) * 8 + xcd; }
    else { bm = blockIdx.y; bn = blockIdx.x; }
    if (nnb >= 8) {
      // ids beyond an XCD's share (nnb % 8 != 0 leaves the XCDs unequal): the leftover tiles in plain order
      const long per = (nnb >> 3) * nmb;               // tiles every XCD surely owns
      if (idx >= per) {
        // leftover: column blocks 8 * (nnb >> 3) .. nnb - 1, all row blocks; leftover ids are (idx - per) * 8 + xcd
        const long l = (idx - per) * 8 + xcd, nleft = (nnb & 7) * nmb;
        if (l < nleft) { bm = l % nmb; bn = (nnb >> 3) * 8 + l / nmb; }
        else return;
      } else { bm = idx % nmb; bn = (idx / nmb) * 8 + xcd; }
    }
  }
  const long m0 = bm * BM, n0 = bn * BN;
  const long kbeg = (long)blockIdx.z * k_per_split;
  const long kend = min(Kp, kbeg + k_per_split);      // (k_per_split is a multiple of 32; the images are zero beyond K)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int wr = w >> 1, wc = w & 1;
  const int li = lane & 31, lk = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  Frag9 fa, fb;
  load_tile9(fa, Ap, apst, Kp, m0, kbeg);
  load_tile9(fb, Bp, bpst, Kp, n0, kbeg);
  const unsigned short* __restrict__ ar = As + (wr * 64 + li) * X9_LD + 8 * lk;
  const unsigned short* __restrict__ br = Bs + (wc * 64 + li) * X9_LD + 8 * lk;
  for (long k0 = kbeg; k0 < kend; k0 += X9_BK) {
    store_tile9(fa, As);
    store_tile9(fb, Bs);
    __syncthreads();
    if (k0 + X9_BK < kend) {
      load_tile9(fa, Ap, apst, Kp, m0, k0 + X9_BK);
      load_tile9(fb, Bp, bpst, Kp, n0, k0 + X9_BK);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {                  // two 16-deep k-steps per staged tile
      bf16x8g a[2][3], b[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          a[i][q] = *reinterpret_cast<const bf16x8g*>(ar + q * X9_PSZ + i * 32 * X9_LD + 16 * s2);
          b[i][q] = *reinterpret_cast<const bf16x8g*>(br + q * X9_PSZ + i * 32 * X9_LD + 16 * s2);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          // smallest pairs first (lo x lo ... hi x hi): the accumulator meets the terms in rising magnitude
#pragma unroll
          for (int sidx = 4; sidx >= 0; --sidx)
#pragma unroll
            for (int qa = 2; qa >= 0; --qa) {
              const int qb = sidx - qa;
              if (qb < 0 || qb > 2) continue;
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][qa], b[j][qb], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
  }

  // C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool direct = (slab == nullptr);
  if (direct && vec_epi) {
    // Row-major epilogue: each 32x32 block goes through this wave's 4 KB of LDS and leaves as
    // float4 rows, so bias / accumulate / ReLU-mask are 16 B loads and the stores full 128 B rows:
    // 4 + 4 vector memory instructions per block and lane instead of 16 + 16 dependent dword ones.
    float* __restrict__ stg = smem + w * 1024;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) stg[((r & 3) + 8 * (r >> 2) + 4 * lk) * 32 + li] = acc[i][j][r];
        const int c4 = (lane & 7) * 4;
        const long n = n0 + wc * 64 + j * 32 + c4;
        float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias && n < N) bv4 = *reinterpret_cast<const float4*>(bias + n);
        float4 mk[4], od[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long m = min(m0 + wr * 64 + i * 32 + q * 8 + (lane >> 3), M - 1);
          const long nc = min(n, N - 4);
          mk[q] = mask ? *reinterpret_cast<const float4*>(mask + m * ldmask + nc) : make_float4(1.f, 1.f, 1.f, 1.f);
          od[q] = accumulate ? *reinterpret_cast<const float4*>(C + m * ldc + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int row = q * 8 + (lane >> 3);
          const long m = m0 + wr * 64 + i * 32 + row;
          float4 v = *reinterpret_cast<const float4*>(stg + row * 32 + c4);
          v.x = (v.x + od[q].x) + bv4.x; v.y = (v.y + od[q].y) + bv4.y;      // same order as the scalar path
          v.z = (v.z + od[q].z) + bv4.z; v.w = (v.w + od[q].w) + bv4.w;
          if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (!(mk[q].x > 0.f)) v.x = 0.f;
          if (!(mk[q].y > 0.f)) v.y = 0.f;
          if (!(mk[q].z > 0.f)) v.z = 0.f;
          if (!(mk[q].w > 0.f)) v.w = 0.f;
          if (m < M && n < N) *reinterpret_cast<float4*>(C + m * ldc + n) = v;
        }
      }
    return;
  }
  float* out = direct ? C : slab + (long)blockIdx.z * M * N;
  const long ldo = direct ? ldc : N;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long n = n0 + wc * 64 + j * 32 + li;
      if (n >= N) continue;
      const float bv = (direct && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (direct) {
          if (accumulate) v += out[m * ldo + n];
          v += bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mask && !(mask[m * ldmask + n] > 0.f)) v = 0.f;
        }
        out[m * ldo + n] = v;
      }
    }
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, long M,
                                                            long N, float* __restrict__ C, long ldc,
                                                            const float* __restrict__ bias, int relu,
                                                            const float* __restrict__ mask, long ldmask,
                                                            int accumulate) {
  const long tot = M * N;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += gridDim.x * 256L) {
    float v = 0.f;
    int z = 0;
    for (; z + 8 <= splits; z += 8) {      // 8 independent loads in flight, summed in fixed order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = slab[(long)(z + u) * tot + i];
#pragma unroll
      for (int u = 0; u < 8; ++u) v += t[u];
    }
    for (; z < splits; ++z) v += slab[(long)z * tot + i];
    const long m = i / N, n = i - m * N;
    if (accumulate) v += C[m * ldc + n];
    if (bias) v += bias[n];
    if (relu) v = fmaxf(v, 0.f);
    if (mask && !(mask[m * ldmask + n] > 0.f)) v = 0.f;
    C[m * ldc + n] = v;
  }
}

// Skinny-M streaming variant for the rollout batch (M = n_envs <= 64) against a LARGE k-contiguous weight matrix
// (ConvModel: 32 x 28224 times 2000 x 28224, 226 MB of weights per step): the block-tiled kernel above moves the
// weights through LDS 16 k-columns (64 B per row) at a time behind two barriers per step and runs at ~2.2 TB/s.
// Here nothing goes through LDS and nothing synchronises: a wave owns 32 weight rows and one k-split; lane
// (i = l & 31, h = l >> 5) loads 16 B of ITS row at k + 4h (weights) and of activation row i (L2 resident), and
// register j of both quads is one v_mfma_f32_32x32x2_f32 (the k-pair {k + j, k + 4 + j} is the same on both
// operands, which is all the instruction needs).  A row is read as one forward stream of 32 B pieces, SK_U loads
// in flight per lane.  Partials go to the split-K slabs; splitk_reduce_kernel applies the epilogue as before.
template <int MB, int SK_U>
__global__ __launch_bounds__(256) void skinny_stream_kernel(long M, long N, long K, const float* __restrict__ X, long ldx,
                                                            const float* __restrict__ W, long ldw, long k_per_split,
                                                            float* __restrict__ slab) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long n0 = ((long)blockIdx.x * 4 + w) * 32;
  if (n0 >= N) return;
  const long kbeg = (long)blockIdx.y * k_per_split, kend = min(K, kbeg + k_per_split);
  const float* __restrict__ wrow = W + min(n0 + li, N - 1) * ldw + 4 * lk;
  const float* __restrict__ xrow[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b) xrow[b] = X + min((long)b * 32 + li, M - 1) * ldx + 4 * lk;
  f32x16 acc[MB];
#pragma unroll
  for (int b = 0; b < MB; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  long k = kbeg;
  for (; k + 8 * SK_U <= kend; k += 8 * SK_U) {
    float4 wv[SK_U], xv[MB][SK_U];
#pragma unroll
    for (int u = 0; u < SK_U; ++u) {
      wv[u] = *reinterpret_cast<const float4*>(wrow + k + 8 * u);
#pragma unroll
      for (int b = 0; b < MB; ++b) xv[b][u] = *reinterpret_cast<const float4*>(xrow[b] + k + 8 * u);
    }
#pragma unroll
    for (int u = 0; u < SK_U; ++u)
#pragma unroll
      for (int b = 0; b < MB; ++b) {
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[b][u].x, wv[u].x, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[b][u].y, wv[u].y, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[b][u].z, wv[u].z, acc[b], 0, 0, 0);
        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[b][u].w, wv[u].w, acc[b], 0, 0, 0);
      }
  }
  for (; k < kend; k += 8) {          // K % 8 == 0 and splits start at multiples of 8
    const float4 wv = *reinterpret_cast<const float4*>(wrow + k);
#pragma unroll
    for (int b = 0; b < MB; ++b) {
      const float4 xv = *reinterpret_cast<const float4*>(xrow[b] + k);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.x, wv.x, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.y, wv.y, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.z, wv.z, acc[b], 0, 0, 0);
      acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv.w, wv.w, acc[b], 0, 0, 0);
    }
  }
  // D map of the 32x32 tile: col (n) = lane & 31, row (m) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* __restrict__ out = slab + (long)blockIdx.y * M * N;
  const long n = n0 + li;
  if (n < N) {
#pragma unroll
    for (int b = 0; b < MB; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = (long)b * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m < M) out[m * N + n] = acc[b][r];
      }
  }
}

// the skinny streaming kernel writes the same [split][M][N] slabs as gemm_kernel; true = launched
static bool launch_skinny_stream(long M, long N, long K, const float* A, long lda, const float* B, long ldb, long kps,
                                 int splits, float* slab, int vecA, int vecB, hipStream_t st) {
  if (M > 64 || splits < 2 || !vecA || !vecB || K % 8 || kps % 8 || N * K < (1L << 22) || getenv("A2C_NO_SKINNY_STREAM"))
    return false;
  dim3 grid((unsigned)((N + 127) / 128), (unsigned)splits);
  if (M <= 32) hipLaunchKernelGGL((skinny_stream_kernel<1, 4>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, kps, slab);
  else hipLaunchKernelGGL((skinny_stream_kernel<2, 4>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, kps, slab);
  return true;
}

// Rollout-batch forward of the big dense layers (64 < M <= 256 rows against a k-contiguous weight matrix: ConvModel's
// 256 x 28224 times 2000 x 28224 is 29 GFLOP per env step): matrix-bound, and the block-tiled kernel above spends its
// time between two barriers per 16-deep step with register staging (83 TF).  Here:
//   * a workgroup owns ALL rows (256 x 128 tile: the weights are read once), K split over the grid (slabs as above);
//   * both operands arrive k-contiguous, so a lane's MFMA operands for FOUR instructions are one 16-byte LDS read: lane
//     (i, h) holds row i, columns k0 + 4h .. k0 + 4h + 3, and component j of both operands is the k-pair
//     {k0 + j, k0 + 4 + j} of one v_mfma_f32_32x32x2_f32 -- no transposing stores, 4 ds_read_b128 per 16 MFMAs;
//   * 32-deep K tiles come in by LDS-DMA (global_load_lds_dwordx4, full 128-byte lines, 8 rows per instruction) from
//     two loader waves into a ring of three 48 KB stages, two tiles ahead of the eight computing waves, counted vmcnt,
//     one raw barrier per tile; the 16-byte pieces of a row are XOR-swizzled by (row >> 1) & 7 on the SOURCE address, so
//     the LDS image is lane-linear for the DMA and conflict-free for the ds_read_b128 lane groups;
//   * sums per output element: k ascending inside a slab, slabs in fixed order -- the order of gemm_kernel's split-K.
// Measured at 256 x 28224 x 2000, 16 slabs (tools/gemm_check.py): 307 us with the reduce (94 TF) against gemm_kernel's 337;
// with the DMA switched off after the first tiles (wrong sums, same MFMA stream) 259 us -- the fp32 matrix rate this
// chip sustains (111 TF of the nominal 157) -- and no faster with every DMA reading one cache-resident tile: what is left
// is what 48 LDS-DMA instructions per tile cost the CU beside 128 MFMAs per wave, not memory.  Four loader waves at raised
// priority instead of two at default: 337 -> 307 us.  M <= 32 (the 32 x 256 shape, 226 MB of weights per step): 61 us against
// skinny_stream_kernel's 64 on the same box, 3.7 TB/s; reading the same bytes from a tile-major copy of the weights (32 KB
// contiguous per tile instead of 256 rows x 128 B) changed nothing -- it is not the access pattern.
namespace nt {
constexpr int BKT = 32, NW = 8, NL = 4;
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;
template <int N_>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }
__device__ __forceinline__ void bar() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// TM x TN block tile, the 8 computing waves WM (M) x WN (N): 256 x 128 as 4 x 2 (64 x 64 per wave) for the rollout batch of
// configs 4 / 5; 32 x 256 as 1 x 8 (one 32 x 32 tile per wave) for M <= 32, where the product is the weight STREAM
// (ConvModel at 32 envs: 226 MB per env step) and the x rows ride in the same ring (4 KB per tile).
template <int TM_, int TN_, int WM_, int DEPTH_>
struct Shape {
  static constexpr int TM = TM_, TN = TN_, WM = WM_, WN = NW / WM_, DEPTH = DEPTH_;
  static constexpr int IM = TM / WM / 32, JN = TN / WN / 32;            // 32 x 32 tiles per wave
  static constexpr int AF = TM * BKT, BF = TN * BKT, STG = AF + BF;     // floats per stage
  static constexpr size_t LDS_BYTES = (size_t)DEPTH * STG * 4;
  static constexpr int NIA = TM / 8, NIB = TN / 8;                      // DMA instructions per tile (8 rows x 128 B each)
  static constexpr int NI = (NIA + NIB) / NL;                           // ... per loader wave
  static_assert((NIA + NIB) % NL == 0 && (DEPTH - 2) * NI <= 63 && DEPTH >= 3 && DEPTH <= 4 && LDS_BYTES <= 160 * 1024, "shape");
};
using Big = Shape<256, 128, 4, 3>;
using Skinny = Shape<32, 256, 1, 4>;
}  // namespace nt

template <class S>
__global__ __launch_bounds__(64 * (nt::NW + nt::NL)) void gemm_nt_kernel(long M, long N, long K, const float* __restrict__ X, long ldx,
                                                                        const float* __restrict__ W, long ldw, long k_per_split,
                                                                        float* __restrict__ slab, const float* __restrict__ zero) {
  using namespace nt;
  constexpr int LOOK = S::DEPTH - 1;                             // tiles in flight ahead of the one being computed
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long n0 = (long)blockIdx.x * S::TN;
  const long kbeg = (long)blockIdx.y * k_per_split, kend = min(K, kbeg + k_per_split);
  const int nt_ = (int)((kend - kbeg + BKT - 1) / BKT);          // K tiles of this split
  if (w >= NW) {
    // ------------------------------------------------------------------ loader waves
    const int lw = w - NW;
    __builtin_amdgcn_s_setprio(3);                               // (the computing waves never wait on anything but these)
    const int r8 = lane >> 3, s8 = lane & 7;
    // piece s8 of LDS row r holds the row's 16-byte piece s8 ^ ((r >> 1) & 7); r = 8 q + r8: (r >> 1) & 7 = ((q & 1) * 4 + (r8 >> 1)) & 7
    const int ce = s8 ^ (r8 >> 1), co = s8 ^ (4 + (r8 >> 1));    // source piece for even / odd q
    auto dma = [&](int t) {
      float* __restrict__ st = lds + (t % S::DEPTH) * S::STG;
      const long k0 = kbeg + (long)t * BKT;
      const int kval = (int)min((long)BKT, kend - k0);           // valid columns of this tile (a multiple of 4)
#pragma unroll
      for (int q = 0; q < S::NIA + S::NIB; ++q) {
        if (q % NL != lw) continue;
        const bool isA = q < S::NIA;
        const int qq = isA ? q : q - S::NIA;
        const long row = (isA ? 0 : n0) + 8 * qq + r8;
        const int c = (qq & 1) ? co : ce;
        const bool ok = row < (isA ? M : N) && 4 * c + 4 <= kval;
        const float* gsrc = ok ? (isA ? X + row * ldx : W + row * ldw) + k0 + 4 * c : zero;
        __builtin_amdgcn_global_load_lds((gptr_t)gsrc, (lptr_t)(st + (isA ? 0 : S::AF) + qq * 256), 16, 0, 0);
      }
    };
    // before the barrier that opens tile t1: everything up to t1 has landed, the younger tiles stay in flight
    auto wait_for = [&](int t1) {
      int m = min(t1 + LOOK - 1, nt_ - 1) - t1;
      if (LOOK >= 3 && m >= 2) wait_vm<(LOOK >= 3 ? 2 : 0) * S::NI>();
      else if (m >= 1) wait_vm<S::NI>();
      else wait_vm<0>();
    };
    for (int t = 0; t < LOOK; ++t)
      if (t < nt_) dma(t);
    wait_for(0);
    bar();
    for (int t = 0; t < nt_; ++t) {
      if (t + LOOK < nt_) dma(t + LOOK);                         // into the stage tile t - 1 left
      wait_for(t + 1);
      bar();
    }
    return;
  }
  // -------------------------------------------------------------------- computing waves: WM (M) x WN (N)
  const int wm = w / S::WN, wn = w % S::WN;
  const int li = lane & 31, lh = lane >> 5;
  int aoff[S::IM], asw[S::IM], boff[S::JN], bsw[S::JN];
#pragma unroll
  for (int i = 0; i < S::IM; ++i) {
    const int ra = (wm * S::IM + i) * 32 + li;
    aoff[i] = ra * BKT; asw[i] = (ra >> 1) & 7;
  }
#pragma unroll
  for (int j = 0; j < S::JN; ++j) {
    const int rb = (wn * S::JN + j) * 32 + li;
    boff[j] = S::AF + rb * BKT; bsw[j] = (rb >> 1) & 7;
  }
  f32x16 acc[S::IM][S::JN];
#pragma unroll
  for (int i = 0; i < S::IM; ++i)
#pragma unroll
    for (int j = 0; j < S::JN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bar();
  for (int t = 0; t < nt_; ++t) {
    const float* __restrict__ st = lds + (t % S::DEPTH) * S::STG;
#pragma unroll
    for (int q = 0; q < BKT / 8; ++q) {
      const int c = 2 * q + lh;
      float4 a[S::IM], b[S::JN];
#pragma unroll
      for (int i = 0; i < S::IM; ++i) a[i] = *reinterpret_cast<const float4*>(st + aoff[i] + 4 * (c ^ asw[i]));
#pragma unroll
      for (int j = 0; j < S::JN; ++j) b[j] = *reinterpret_cast<const float4*>(st + boff[j] + 4 * (c ^ bsw[j]));
#define NT_STEP(J)                                                                                     \
      _Pragma("unroll") for (int i = 0; i < S::IM; ++i)                                                \
        _Pragma("unroll") for (int j = 0; j < S::JN; ++j)                                              \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i].J, b[j].J, acc[i][j], 0, 0, 0);
      NT_STEP(x) NT_STEP(y) NT_STEP(z) NT_STEP(w)
#undef NT_STEP
    }
    bar();
  }
  // D map of a 32x32 tile: col (n) = lane & 31, row (m) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* __restrict__ out = slab + (long)blockIdx.y * M * N;
#pragma unroll
  for (int i = 0; i < S::IM; ++i)
#pragma unroll
    for (int j = 0; j < S::JN; ++j) {
      const long n = n0 + (wn * S::JN + j) * 32 + li;
      if (n >= N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = (wm * S::IM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < M) out[m * N + n] = acc[i][j][r];
      }
    }
}

// the same [split][M][N] slabs as gemm_kernel; true = launched
template <class S>
static bool launch_gemm_nt_shape(long M, long N, long K, const float* A, long lda, const float* B, long ldb, long kps, int splits,
                                 float* slab, hipStream_t st) {
  static float* zero = nullptr;
  static bool ready = false;
  static std::once_flag once;
  std::call_once(once, [] {
    if (hipMalloc(&zero, 256) != hipSuccess || hipMemset(zero, 0, 256) != hipSuccess) return;
    if (hipFuncSetAttribute((const void*)gemm_nt_kernel<S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::LDS_BYTES) != hipSuccess)
      return;
    ready = true;
  });
  if (!ready) return false;                         // the caller falls back to gemm_kernel
  dim3 grid((unsigned)((N + S::TN - 1) / S::TN), (unsigned)splits);
  hipLaunchKernelGGL(gemm_nt_kernel<S>, grid, dim3(64 * (nt::NW + nt::NL)), S::LDS_BYTES, st, M, N, K, A, lda, B, ldb, kps, slab,
                     (const float*)zero);
  return hipGetLastError() == hipSuccess;           // a refused launch (LDS attribute, block size) -> gemm_kernel, not garbage slabs
}

static bool launch_gemm_nt(long M, long N, long K, const float* A, long lda, const float* B, long ldb, long kps, int splits,
                           float* slab, int vecA, int vecB, hipStream_t st) {
  if (M > nt::Big::TM || splits < 2 || !vecA || !vecB || K % 4 || kps % 4 || N * K < (1L << 22) || getenv("A2C_NO_GEMM_NT"))
    return false;
  if (M <= nt::Skinny::TM) return launch_gemm_nt_shape<nt::Skinny>(M, N, K, A, lda, B, ldb, kps, splits, slab, st);
  if (M <= 64) return false;                       // (33 .. 64 rows: skinny_stream_kernel<2>)
  return launch_gemm_nt_shape<nt::Big>(M, N, K, A, lda, B, ldb, kps, splits, slab, st);
}

// ---------------------------------------------------------------------------------------------------------------
// "bf16 x 6" (round 6): the large dense products of ConvModel's update (2048 .. 32768 rows against the 28224 x 2000 layer) on
// the bf16 pipe with fp32 results.  Both fp32 operands are split into three bf16 pieces a = a0 + a1 + a2 (exact, see
// split1_x9); of the nine piece products the SIX with qa + qb <= 2 are issued: the dropped ones are below 2^-24 of |a b| --
// under the rounding of the fp32 product itself -- and every issued product is exact (8 x 8 bits) with the sums in the MFMA's
// fp32 accumulator.  6 v_mfma_f32_32x32x16_bf16 (32 cycles) per 16-deep k-step replace 8 v_mfma_f32_32x32x2_f32 (64 cycles):
// 0.375 of the fp32 matrix time.  What kept the x 9 form above at a tie was the operand traffic (1.5 x the bytes through L2 and
// LDS, register staging, two barriers per 32-deep step); here:
//   * x6_split_kernel writes the piece images in PANEL order img[q][row / 256][k / 16][(k / 8) & 1][row % 256][8]: the
//     256 rows x 16 k of one (piece, row block, k block) are 8 KB contiguous, in the order the LDS wants them;
//   * a workgroup (8 waves, 2 x 4, 128 x 64 per wave) owns a 256 x 256 tile: 16 B of operand per cycle and CU at full MFMA
//     rate instead of 32; a stage (16 k of 512 rows, three pieces: 48 KB) arrives by LDS-DMA (global_load_lds_dwordx4, six
//     instructions per wave and stage, no staging registers) into a ring of three, two stages ahead, counted vmcnt, ONE raw
//     barrier per stage; an MFMA operand is one conflict-free ds_read_b128 (lanes of a k-half are 512 contiguous bytes);
//   * work items (split, group of 8 row blocks, column block, row block) are dealt to the XCDs in contiguous runs (workgroup id
//     & 7 = XCD), so that the 32 workgroups an XCD runs at a time are 8 row blocks x 4 column blocks reading the same panels
//     at the same k: each panel crosses the fabric once per round, not once per tile.
namespace x6 {
constexpr int TM = 256, TN = 256, BK = 16, DEPTH = 3;
constexpr int CH = 256 * BK;                           // bf16 elements of one chunk [k-half][row][8] = 8 KB
constexpr int STG = 6 * CH;                            // A pieces 0..2, B pieces 0..2
constexpr size_t LDS_BYTES = (size_t)DEPTH * STG * 2;  // 147,456
}  // namespace x6

template <bool TRANS>
__global__ __launch_bounds__(256) void x6_split_kernel(const float* __restrict__ src, long ld, long R, long K, unsigned short* __restrict__ dst,
                                                       long Rp, long Kp) {
  const long pst = Rp * Kp, ng = Kp >> 3, ngb = (ng + 3) >> 2, nblk = (Rp >> 6) * ngb;
  for (long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long r0 = (blk / ngb) << 6, g0 = (blk % ngb) << 2;
    long r, g;
    float e[8];
    if (!TRANS) {                                      // src[r * ld + k]: four lanes read 128 B of one row
      r = r0 + (threadIdx.x >> 2); g = g0 + (threadIdx.x & 3);
      const long k = g << 3;
      if (r < R && k + 7 < K && (ld & 3) == 0 && ((uintptr_t)src & 15) == 0) {
        const float4 v0 = *reinterpret_cast<const float4*>(src + r * ld + k), v1 = *reinterpret_cast<const float4*>(src + r * ld + k + 4);
        e[0] = v0.x; e[1] = v0.y; e[2] = v0.z; e[3] = v0.w; e[4] = v1.x; e[5] = v1.y; e[6] = v1.z; e[7] = v1.w;
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (r < R && k + i < K) ? src[r * ld + k + i] : 0.f;
      }
    } else {                                           // src[k * ld + r]: a wave reads 256 B of one k, eight times
      r = r0 + (threadIdx.x & 63); g = g0 + (threadIdx.x >> 6);
      const long k = g << 3;
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = (r < R && k + i < K) ? src[(k + i) * ld + r] : 0.f;
    }
    if (g < ng) split8_store(e, dst + (r >> 8) * (Kp << 8) + (g >> 1) * x6::CH + (g & 1) * 2048 + (r & 255) * 8, pst);
  }
}

__global__ __launch_bounds__(512) void gemm_x6_kernel(long M, long N, const unsigned short* __restrict__ Ap, long apst,
                                                      const unsigned short* __restrict__ Bp, long bpst, long Kp, int nmb, int nnb,
                                                      int nsplit, int st_per_split, float* __restrict__ C, long ldc,
                                                      const float* __restrict__ bias, int relu, const float* __restrict__ mask,
                                                      long ldmask, int accumulate, float* __restrict__ slab) {
  using namespace x6;
  using nt::gptr_t;
  using nt::lptr_t;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds6[];
  long bm, bn, sp;
  {
    const long T = (long)nmb * nnb, Wk = T * nsplit, per = (Wk + 7) >> 3;
    const long id = blockIdx.x, xcd = id & 7, idx = id >> 3;
    const long j = xcd * per + idx;
    if (idx >= per || j >= Wk) return;
    sp = j / T;
    const long rem = j - sp * T, full = (long)(nmb >> 3) * 8 * nnb;
    if (rem < full) { const long g = rem / (8L * nnb), r2 = rem - g * 8L * nnb; bn = r2 >> 3; bm = 8 * g + (r2 & 7); }
    else { const long tr = nmb & 7, r2 = rem - full; bn = r2 / tr; bm = (long)(nmb >> 3) * 8 + r2 % tr; }
  }
  const int nkb = (int)(Kp >> 4);
  const int kb0 = (int)sp * st_per_split, nst = min(nkb, kb0 + st_per_split) - kb0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3, li = lane & 31, lk = lane >> 5;
  // this wave's 1 KB of every chunk: [w * 64 + lane] * 16 B
  const unsigned short* __restrict__ ga = Ap + bm * (Kp << 8) + (long)kb0 * CH + w * 512 + lane * 8;
  const unsigned short* __restrict__ gb = Bp + bn * (Kp << 8) + (long)kb0 * CH + w * 512 + lane * 8;
  auto dma = [&](int t, int buf) {
    unsigned short* __restrict__ st = lds6 + buf * STG + w * 512;
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(ga + q * apst + (long)t * CH), (lptr_t)(st + q * CH), 16, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) __builtin_amdgcn_global_load_lds((gptr_t)(gb + q * bpst + (long)t * CH), (lptr_t)(st + (3 + q) * CH), 16, 0, 0);
  };
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  if (nst > 0) dma(0, 0);
  if (nst > 1) dma(1, 1);
  const int aoff = lk * 2048 + (wr * 128 + li) * 8, boff = 3 * CH + lk * 2048 + (wc * 64 + li) * 8;
  int buf = 0;
  for (int t = 0; t < nst; ++t) {
    if (t + 1 < nst) nt::wait_vm<6>(); else nt::wait_vm<0>();
    nt::bar();
    int b2 = buf + 2; if (b2 >= DEPTH) b2 -= DEPTH;
    if (t + 2 < nst) dma(t + 2, b2);
    const unsigned short* __restrict__ st = lds6 + buf * STG;
    bf16x8g a[4][3], b[2][3];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 3; ++q) b[j][q] = *reinterpret_cast<const bf16x8g*>(st + boff + q * CH + j * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 3; ++q) a[i][q] = *reinterpret_cast<const bf16x8g*>(st + aoff + q * CH + i * 256);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        // rising magnitude: (2,0) (1,1) (0,2) | (1,0) (0,1) | (0,0)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], acc[i][j], 0, 0, 0);
      }
    if (++buf == DEPTH) buf = 0;
  }
  // C/D map of a 32 x 32 tile: col (n) = lane & 31, row (m) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  const bool direct = (slab == nullptr);
  float* __restrict__ out = direct ? C : slab + sp * M * N;
  const long ldo = direct ? ldc : N;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long n = bn * TN + wc * 64 + j * 32 + li;
      if (n >= N) continue;
      const float bv = (direct && bias) ? bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = bm * TM + wr * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (m >= M) continue;
        float v = acc[i][j][r];
        if (direct) {
          if (accumulate) v += out[m * ldo + n];
          v += bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mask && !(mask[m * ldmask + n] > 0.f)) v = 0.f;
        }
        out[m * ldo + n] = v;
      }
    }
}

// Small products (the GRU cell's h x h and x x 3h GEMMs at rollout / BPTT batch, M = n_envs: a handful of
// 128 x 128 tiles) used to run as split-K + slab reduce: two launches and a few MB of slab traffic for a few
// MFLOP, 18 us per product, ~9 products per env step.  Here ONE launch: a workgroup owns a 32 x 32 tile of C, its
// four waves split K into quarters, operands go global -> register -> MFMA with no LDS staging (lane (i, h) reads
// 16 B of row i at k + 4h when the operand is k-contiguous, else four coalesced dwords), the four partial tiles are
// added through LDS in wave order (deterministic) and the epilogue (accumulate, bias, ReLU, mask) is applied.
template <bool B_KC, int NW>
__global__ __launch_bounds__(64 * NW) void small_gemm_kernel(long M, long N, long K, const float* __restrict__ A, long lda,
                                                         const float* __restrict__ B, long ldb, float* __restrict__ C,
                                                         long ldc, const float* __restrict__ bias, int relu,
                                                         const float* __restrict__ mask, long ldmask, int accumulate) {
  __shared__ __attribute__((aligned(16))) float red[NW][16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int li = lane & 31, lk = lane >> 5;
  const long m0 = (long)blockIdx.y * 32, n0 = (long)blockIdx.x * 32;
  const long kq = ((K / 8 + NW - 1) / NW) * 8;         // K % 8 == 0: parts of whole 8-column groups, the last ones shorter
  const long kbeg = min(K, w * kq), kend = min(K, kbeg + kq);
  const float* __restrict__ arow = A + min(m0 + li, M - 1) * lda + 4 * lk;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if (B_KC) {
    const float* __restrict__ brow = B + min(n0 + li, N - 1) * ldb + 4 * lk;
    long k = kbeg;
    for (; k + 8 <= kend && ((kend - k) & 31); k += 8) {   // head: bring the rest to a multiple of 32
      const float4 a1 = *reinterpret_cast<const float4*>(arow + k), b1 = *reinterpret_cast<const float4*>(brow + k);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, b1.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, b1.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, b1.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, b1.w, acc, 0, 0, 0);
    }
    if (k < kend) {                                    // four 8-column groups per trip, the next trip's loads in flight
      float4 av[4], bv[4], an[4], bn[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        av[u] = *reinterpret_cast<const float4*>(arow + k + 8 * u);
        bv[u] = *reinterpret_cast<const float4*>(brow + k + 8 * u);
      }
      for (; k < kend; k += 32) {
        const long kn = (k + 32 < kend) ? k + 32 : k;  // last trip: re-read (valid addresses, values unused)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          an[u] = *reinterpret_cast<const float4*>(arow + kn + 8 * u);
          bn[u] = *reinterpret_cast<const float4*>(brow + kn + 8 * u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].x, bv[u].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].y, bv[u].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].z, bv[u].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u].w, bv[u].w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { av[u] = an[u]; bv[u] = bn[u]; }
      }
    }
  } else {
    const float* __restrict__ bcol = B + min(n0 + li, N - 1) + (long)(4 * lk) * ldb;
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
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) red[w][r][lane] = acc[r];
  __syncthreads();
  // wave w finishes registers w*16/NW .. of every lane: col n = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int q = 0; q < 16 / NW; ++q) {
    const int r = w * (16 / NW) + q;
    float v = red[0][r][lane];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) v += red[ww][r][lane];
    const long m = m0 + (r & 3) + 8 * (r >> 2) + 4 * lk, n = n0 + li;
    if (m < M && n < N) {
      if (accumulate) v += C[m * ldc + n];
      if (bias) v += bias[n];
      if (relu) v = fmaxf(v, 0.f);
      if (mask && !(mask[m * ldmask + n] > 0.f)) v = 0.f;
      C[m * ldc + n] = v;
    }
  }
}

// out[n] = sum_m x[m*ld+n]: stage 1 = per-workgroup partial column sums over a row band,
// stage 2 = fixed-order sum of the partials.
constexpr int CS_BANDS = 256;
__global__ __launch_bounds__(256) void colsum_stage1(const float* __restrict__ x, long ld, long M, long N,
                                                     float* __restrict__ part) {
  const long rows_per = (M + gridDim.y - 1) / gridDim.y;
  const long r0 = (long)blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  const long n = blockIdx.x * 256L + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  long m = r0;
  for (; m + 8 <= r1; m += 8) {              // 8 independent loads in flight, fixed summation order
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = x[(m + u) * ld + n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; m < r1; ++m) s += x[m * ld + n];
  part[(long)blockIdx.y * N + n] = s;
}
__global__ __launch_bounds__(256) void colsum_stage2(const float* __restrict__ part, int bands, long N,
                                                     float* __restrict__ out) {
  const long n = blockIdx.x * 256L + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  int b = 0;
  for (; b + 8 <= bands; b += 8) {
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = part[(long)(b + u) * N + n];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += t[u];
  }
  for (; b < bands; ++b) s += part[(long)b * N + n];
  out[n] = s;
}

// ---- skinny layers (policy / value heads: N <= 8 outputs).  y[m, n] = x[m,:] . W[n,:] + b[n]:
// one wave per row, 16 B per lane, butterfly reduce.  Memory-bound on x.
constexpr int SN_MAX = 8;
__global__ __launch_bounds__(256) void small_n_fwd_kernel(const float* __restrict__ x, long ldx,
                                                          const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ y, long ldy, long M, int N, int K) {
  const int lane = threadIdx.x & 63;
  const long row0 = blockIdx.x * 4L + (threadIdx.x >> 6);
  for (long m = row0; m < M; m += gridDim.x * 4L) {
    float acc[SN_MAX];
#pragma unroll
    for (int n = 0; n < SN_MAX; ++n) acc[n] = 0.f;
    for (int k = lane * 4; k < K; k += 256) {           // K % 4 == 0, 16 B aligned rows (checked by the launcher)
      const float4 xv = *reinterpret_cast<const float4*>(x + m * ldx + k);
#pragma unroll
      for (int n = 0; n < SN_MAX; ++n)
        if (n < N) {
          const float4 wv = *reinterpret_cast<const float4*>(W + (long)n * K + k);
          acc[n] += xv.x * wv.x + xv.y * wv.y + xv.z * wv.z + xv.w * wv.w;
        }
    }
#pragma unroll
    for (int n = 0; n < SN_MAX; ++n)
      if (n < N) {
        const float v = wave_sum(acc[n]);
        if (lane == 0) y[m * ldy + n] = v + (b ? b[n] : 0.f);
      }
  }
}

// dx[m, k] (+)= sum_n dy[m, n] W[n, k]   (n < N <= 8), optional ReLU-derivative mask
__global__ __launch_bounds__(256) void small_n_bwd_data_kernel(const float* __restrict__ dy, long ldy,
                                                               const float* __restrict__ W, float* __restrict__ dx,
                                                               long ldx, const float* __restrict__ mask, long ldmask,
                                                               long M, int N, int K, int accumulate) {
  const long tot = M * (K >> 2);
  for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += gridDim.x * 256L) {
    const long m = i / (K >> 2);
    const int k = (int)(i - m * (K >> 2)) << 2;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int n = 0; n < N; ++n) {
      const float g = dy[m * ldy + n];
      const float4 wv = *reinterpret_cast<const float4*>(W + (long)n * K + k);
      a.x += g * wv.x; a.y += g * wv.y; a.z += g * wv.z; a.w += g * wv.w;
    }
    float4* o = reinterpret_cast<float4*>(dx + m * ldx + k);
    if (accumulate) { const float4 c = *o; a.x += c.x; a.y += c.y; a.z += c.z; a.w += c.w; }
    if (mask) {
      const float4 mk = *reinterpret_cast<const float4*>(mask + m * ldmask + k);
      if (!(mk.x > 0.f)) a.x = 0.f;
      if (!(mk.y > 0.f)) a.y = 0.f;
      if (!(mk.z > 0.f)) a.z = 0.f;
      if (!(mk.w > 0.f)) a.w = 0.f;
    }
    *o = a;
  }
}

// the same product with the mask as one bit per activation (include/a2c_mi355x.h: a2c_small_n_bwd_data_bits): a thread owns 8
// consecutive columns = one mask byte; per column the FMA chain of small_n_bwd_data_kernel (n ascending): bit-identical
__global__ __launch_bounds__(256) void small_n_bwd_data_bits_kernel(const float* __restrict__ dy, long ldy,
                                                                    const float* __restrict__ W, float* __restrict__ dx,
                                                                    long ldx, const unsigned char* __restrict__ mb,
                                                                    long mb_row, long M, int N, int K) {
  const int k8 = K >> 3;
  const long tot = M * k8;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += gridDim.x * 256L) {
    const long m = i / k8;
    const int kb = (int)(i - m * k8), k = kb << 3;
    const unsigned int bits = mb[m * mb_row + kb];
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    for (int n = 0; n < N; ++n) {
      const float g = dy[m * ldy + n];
      const float4 w0 = *reinterpret_cast<const float4*>(W + (long)n * K + k);
      const float4 w1 = *reinterpret_cast<const float4*>(W + (long)n * K + k + 4);
      a.x += g * w0.x; a.y += g * w0.y; a.z += g * w0.z; a.w += g * w0.w;
      b.x += g * w1.x; b.y += g * w1.y; b.z += g * w1.z; b.w += g * w1.w;
    }
    if (!(bits & 1u)) a.x = 0.f;
    if (!(bits & 2u)) a.y = 0.f;
    if (!(bits & 4u)) a.z = 0.f;
    if (!(bits & 8u)) a.w = 0.f;
    if (!(bits & 16u)) b.x = 0.f;
    if (!(bits & 32u)) b.y = 0.f;
    if (!(bits & 64u)) b.z = 0.f;
    if (!(bits & 128u)) b.w = 0.f;
    float4* o = reinterpret_cast<float4*>(dx + m * ldx + k);
    o[0] = a;
    o[1] = b;
  }
}

// dW[n, k] = sum_m dy[m, n] x[m, k]: each workgroup owns a band of rows and all of (N, K<=1024)
// with thread t holding columns k = 4t..4t+3; partial slabs [band][N][K] + fixed-order reduce.
constexpr int SN_BANDS = 512;
__global__ __launch_bounds__(256) void small_n_bwd_weight_kernel(const float* __restrict__ dy, long ldy,
                                                                 const float* __restrict__ x, long ldx,
                                                                 float* __restrict__ part, long M, int N, int K) {
  // threads: kq = column quad, rp = row sub-band (K/4 < 256 leaves threads for several rows at once)
  // blockIdx.y = column chunk: K columns starting at blockIdx.y * K of x, slabs [chunk][band][N][K]
  __shared__ float4 sm[256];
  x += (long)blockIdx.y * K;
  part += (long)blockIdx.y * gridDim.x * N * K;
  const long rows_per = (M + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
  const int kq4 = K >> 2;                                  // float4 columns
  const int nrp = kq4 >= 256 ? 1 : 256 / kq4;              // row sub-bands
  const int kq = threadIdx.x % kq4, rp = threadIdx.x / kq4;
  const int k = kq * 4;
  const bool live = rp < nrp && (kq4 <= 256 ? true : threadIdx.x * 4 < K);
  float4 acc[SN_MAX];
#pragma unroll
  for (int n = 0; n < SN_MAX; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
  long m = r0 + rp;
  // eight rows in flight per thread (the loop is latency-bound on its 16-B loads otherwise: 2.0 TB/s with four); rows are
  // still accumulated in increasing m, so the sums are bit-identical to the one-row-at-a-time loop
  constexpr int RU = 8;
  if (nrp == 1 && (int)(threadIdx.x & ~63u) < kq4) {
    // every live thread walks the SAME rows: the RU x N gradient values of a pass are ONE coalesced load per wave
    // (lane l holds row l / N, head l % N) handed out by v_readlane, instead of RU x N same-address vector loads that
    // kept the load pipe busier than the x rows themselves did.  v_readlane ignores EXEC, so this block is entered per
    // WAVE (the predicate above is wave-uniform) and ALL 64 lanes load their gradient value: in a partly live wave
    // (K/4 not a multiple of 64) the dead lanes carry gradient values too, read x from column 0 and never write a result.
    const int lane = threadIdx.x & 63;
    const int ur = lane / N, un = lane - ur * N;
    const int kx = live ? k : 0;
    m = r0;                                // (dead lanes have rp == 1: same trip count for all 64 lanes)
    for (; m + (long)(RU - 1) < r1; m += RU) {
      float4 xv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) xv[u] = *reinterpret_cast<const float4*>(x + (m + u) * ldx + kx);
      const float dyv = ur < RU ? dy[(m + ur) * ldy + un] : 0.f;
#pragma unroll
      for (int u = 0; u < RU; ++u)
#pragma unroll
        for (int n = 0; n < SN_MAX; ++n)
          if (n < N) {
            const float g = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dyv), u * N + n));
            acc[n].x += g * xv[u].x; acc[n].y += g * xv[u].y; acc[n].z += g * xv[u].z; acc[n].w += g * xv[u].w;
          }
    }
  }
  if (live) {
    for (; m + (long)(RU - 1) * nrp < r1; m += (long)RU * nrp) {
      float4 xv[RU];
#pragma unroll
      for (int u = 0; u < RU; ++u) xv[u] = *reinterpret_cast<const float4*>(x + (m + (long)u * nrp) * ldx + k);
#pragma unroll
      for (int u = 0; u < RU; ++u)
#pragma unroll
        for (int n = 0; n < SN_MAX; ++n)
          if (n < N) {
            const float g = dy[(m + (long)u * nrp) * ldy + n];
            acc[n].x += g * xv[u].x; acc[n].y += g * xv[u].y; acc[n].z += g * xv[u].z; acc[n].w += g * xv[u].w;
          }
    }
    for (; m < r1; m += nrp) {
      const float4 xv = *reinterpret_cast<const float4*>(x + m * ldx + k);
#pragma unroll
      for (int n = 0; n < SN_MAX; ++n)
        if (n < N) {
          const float g = dy[m * ldy + n];
          acc[n].x += g * xv.x; acc[n].y += g * xv.y; acc[n].z += g * xv.z; acc[n].w += g * xv.w;
        }
    }
  }
  // fixed-order sum of the row sub-bands through LDS, one head at a time
#pragma unroll
  for (int n = 0; n < SN_MAX; ++n) {
    if (n >= N) break;
    __syncthreads();
    sm[threadIdx.x] = acc[n];
    __syncthreads();
    if (rp == 0 && live) {
      float4 t = sm[kq];
      for (int q = 1; q < nrp; ++q) {
        const float4 o = sm[q * kq4 + kq];
        t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
      }
      *reinterpret_cast<float4*>(part + ((long)blockIdx.x * N + n) * K + k) = t;
    }
  }
}
// dW[n][chunk * Kc + k] = sum over the bands of part[chunk][band][n][k], bands in increasing order.  32 outputs x 8 band
// groups per workgroup (a thread sums bands g, g + 8, ...; the eight partial sums are added through LDS in group order):
// the one-thread-per-output form walked 512 slabs with 14 workgroups, 23 us per call against the 30 us of the main kernel
__global__ __launch_bounds__(256) void small_n_bwd_weight_reduce(const float* __restrict__ part, int bands, int N, int Kc,
                                                                 int nchunk, float* __restrict__ dW) {
  __shared__ float sm[8][32];
  const int o = threadIdx.x & 31, gq = threadIdx.x >> 5;
  const long per = (long)N * Kc, tot = per * nchunk;
  const long i = blockIdx.x * 32L + o;                  // (chunk, n, k) flattened chunk-major
  float v = 0.f;
  if (i < tot) {
    const long ch = i / per, r = i - ch * per;
    const float* __restrict__ src = part + ch * bands * per + r;
    int z = gq;
    for (; z + 24 < bands; z += 32) {                  // four loads in flight
      const float t0 = src[(long)z * per], t1 = src[(long)(z + 8) * per], t2 = src[(long)(z + 16) * per], t3 = src[(long)(z + 24) * per];
      v += t0; v += t1; v += t2; v += t3;
    }
    for (; z < bands; z += 8) v += src[(long)z * per];
  }
  sm[gq][o] = v;
  __syncthreads();
  if (gq == 0 && i < tot) {
    float t = sm[0][o];
#pragma unroll
    for (int q = 1; q < 8; ++q) t += sm[q][o];
    const long ch = i / per, r = i - ch * per;
    const long n = r / Kc, k = r - n * Kc;
    dW[n * (long)Kc * nchunk + ch * Kc + k] = t;
  }
}

// C[m, n] (+)= sum_{k < K <= 8} A[k, m] B[k, n]: both operands k-major (rank-K outer-product sum).  A3CModel's rank-A
// backward (models.py:73, 85: no activation behind proj_matrx and a detached value head make the embedding gradient
// demb = dl . W_pi a rank-A matrix): G(proj_matrx.weight) = W_pi^T . (dl^T a2) with K = A actions.  Memory-bound on C.
template <int VEC>
__global__ __launch_bounds__(256) void small_k_tn_kernel(const float* __restrict__ A, long lda, const float* __restrict__ B,
                                                         long ldb, float* __restrict__ C, long ldc, long M, long N, int K,
                                                         int accumulate) {
  const long nq = (N + VEC - 1) / VEC, tot = M * nq;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += gridDim.x * 256L) {
    const long m = i / nq;
    const long n = (i - m * nq) * VEC;
    float a[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) a[v] = 0.f;
    for (int k = 0; k < K; ++k) {
      const float g = A[k * lda + m];
      if (VEC == 4) {
        const float4 bv = *reinterpret_cast<const float4*>(B + k * ldb + n);
        a[0] += g * bv.x; a[1 % VEC] += g * bv.y; a[2 % VEC] += g * bv.z; a[3 % VEC] += g * bv.w;
      } else {
        a[0] += g * B[k * ldb + n];
      }
    }
    if (VEC == 4) {
      float4* o = reinterpret_cast<float4*>(C + m * ldc + n);
      float4 r = make_float4(a[0], a[1 % VEC], a[2 % VEC], a[3 % VEC]);
      if (accumulate) { const float4 c = *o; r.x += c.x; r.y += c.y; r.z += c.z; r.w += c.w; }
      *o = r;
    } else {
      C[m * ldc + n] = accumulate ? C[m * ldc + n] + a[0] : a[0];
    }
  }
}

// fused forward tail: slab sum + bias (+ReLU) -> [emb] -> N<=8 heads -> [softmax + inverse-CDF sample]
// One WORKGROUP per row: its 4 waves split K (16 B per lane per pass), butterfly + LDS reduce in a
// fixed order (deterministic), thread 0 finishes the row.
__global__ __launch_bounds__(256) void heads_fused_kernel(const float* __restrict__ xs, int nslab, long slab_stride,
                                                          long ldx, const float* __restrict__ bias_in, int relu_in,
                                                          float* __restrict__ emb_out, long ld_emb,
                                                          const float* __restrict__ W, const float* __restrict__ b,
                                                          float* __restrict__ heads, long ldh, long M, int N, int K,
                                                          const float* __restrict__ u, int n_logits,
                                                          int64_t* __restrict__ actions, long act_stride,
                                                          unsigned long long* __restrict__ cmd,
                                                          const unsigned int* __restrict__ seq_base, unsigned int seq_off) {
  __shared__ float red[4][SN_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (long m = blockIdx.x; m < M; m += gridDim.x) {
    float acc[SN_MAX];
#pragma unroll
    for (int n = 0; n < SN_MAX; ++n) acc[n] = 0.f;
    for (int k = tid * 4; k < K; k += 1024) {
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      int z = 0;
      for (; z + 8 <= nslab; z += 8) {                 // 8 slab loads in flight, fixed summation order
        float4 t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = *reinterpret_cast<const float4*>(xs + (z + q) * slab_stride + m * ldx + k);
#pragma unroll
        for (int q = 0; q < 8; ++q) { xv.x += t[q].x; xv.y += t[q].y; xv.z += t[q].z; xv.w += t[q].w; }
      }
      for (; z < nslab; ++z) {
        const float4 t = *reinterpret_cast<const float4*>(xs + z * slab_stride + m * ldx + k);
        xv.x += t.x; xv.y += t.y; xv.z += t.z; xv.w += t.w;
      }
      if (bias_in) {
        const float4 bv = *reinterpret_cast<const float4*>(bias_in + k);
        xv.x += bv.x; xv.y += bv.y; xv.z += bv.z; xv.w += bv.w;
      }
      if (relu_in) { xv.x = fmaxf(xv.x, 0.f); xv.y = fmaxf(xv.y, 0.f); xv.z = fmaxf(xv.z, 0.f); xv.w = fmaxf(xv.w, 0.f); }
      if (emb_out) *reinterpret_cast<float4*>(emb_out + m * ld_emb + k) = xv;
#pragma unroll
      for (int n = 0; n < SN_MAX; ++n)
        if (n < N) {
          const float4 wv4 = *reinterpret_cast<const float4*>(W + (long)n * K + k);
          acc[n] += xv.x * wv4.x + xv.y * wv4.y + xv.z * wv4.z + xv.w * wv4.w;
        }
    }
#pragma unroll
    for (int n = 0; n < SN_MAX; ++n)
      if (n < N) {
        const float v = wave_sum(acc[n]);
        if (lane == 0) red[wv][n] = v;
      }
    __syncthreads();
    if (tid == 0) {
      float h[SN_MAX];
#pragma unroll
      for (int n = 0; n < SN_MAX; ++n) {
        h[n] = 0.f;
        if (n < N) {
          h[n] = ((red[0][n] + red[1][n]) + (red[2][n] + red[3][n])) + (b ? b[n] : 0.f);
          heads[m * ldh + n] = h[n];
        }
      }
      if (u != nullptr) {      // same maths as sample_kernel<true>: softmax, running fp32 cumsum, first >= u
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < SN_MAX; ++n)
          if (n < n_logits) mx = fmaxf(mx, h[n]);
        float den = 0.f;
#pragma unroll
        for (int n = 0; n < SN_MAX; ++n)
          if (n < n_logits) den += expf(h[n] - mx);
        const float ub = u[m];
        float cs = 0.f;
        int pick = -1;
#pragma unroll
        for (int n = 0; n < SN_MAX; ++n)
          if (n < n_logits) {
            cs = cs + expf(h[n] - mx) / den;
            if (pick < 0 && cs >= ub) pick = n;
          }
        if (pick < 0) pick = n_logits - 1;      // fp32 cumsum short of u: the last action (see sample_kernel)
        actions[m * act_stride] = (int64_t)pick;
        if (cmd != nullptr)      // the device relay's action hand-off (pool_publish_kernel) from the thread that sampled it
          __hip_atomic_store(cmd + m, ((unsigned long long)(seq_base[0] + seq_off) << 32) | (unsigned int)pick, __ATOMIC_RELAXED,
                             __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    __syncthreads();
  }
}

// Inference-only composition of two stacked linear layers with no activation in between
// (A3CModel: proj_matrx then [pi; value], models.py:73,84-85):
//   Wc[n][f] = sum_h Wh[n][h] * Wp[h][f]      bc[n] = sum_h Wh[n][h] * bp[h] + bh[n]
// 64 columns f per workgroup (Wp rows read coalesced), the hidden index h split over the 4 waves and
// summed through LDS in a fixed order; all N <= 8 heads at once.
// (32 feature columns x 4 h-quarters per workgroup, 16 rows of Wp in flight per thread: the kernel sits on the path from the
// optimiser step to the next rollout and is bound by the latency of its dependent load batches -- 41 workgroups of 64 columns
// with 8 rows in flight took 29 us; same sums in the same order)
__global__ __launch_bounds__(128) void compose_heads_kernel(const float* __restrict__ Wh, const float* __restrict__ bh,
                                                            const float* __restrict__ Wp, const float* __restrict__ bp,
                                                            float* __restrict__ Wc, float* __restrict__ bc, int N, int H,
                                                            int F) {
  __shared__ float sm[4][SN_MAX][32];
  const int fq = threadIdx.x & 31, hq = threadIdx.x >> 5;
  const int f = blockIdx.x * 32 + fq;                   // f == F: the bias column
  const int hper = (H + 3) / 4, h_lo = hq * hper, h_hi = min(H, h_lo + hper);
  float acc[SN_MAX];
#pragma unroll
  for (int n = 0; n < SN_MAX; ++n) acc[n] = 0.f;
  if (f <= F)
    for (int h0 = h_lo; h0 < h_hi; h0 += 16) {
      float xv[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int h = min(h0 + u, H - 1);
        xv[u] = (f < F) ? Wp[(long)h * F + f] : bp[h];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (h0 + u < h_hi) {
#pragma unroll
          for (int n = 0; n < SN_MAX; ++n)
            if (n < N) acc[n] += Wh[n * H + h0 + u] * xv[u];
        }
    }
#pragma unroll
  for (int n = 0; n < SN_MAX; ++n) sm[hq][n][fq] = acc[n];
  __syncthreads();
  if (hq == 0 && f <= F) {
#pragma unroll
    for (int n = 0; n < SN_MAX; ++n)
      if (n < N) {
        const float v = ((sm[0][n][fq] + sm[1][n][fq]) + sm[2][n][fq]) + sm[3][n][fq];
        if (f < F) Wc[(long)n * F + f] = v;
        else bc[n] = v + bh[n];
      }
  }
}

template <bool A_KC, bool B_KC>
void launch_gemm(dim3 grid, hipStream_t st, long M, long N, long K, const float* A, long lda, const float* B, long ldb,
                 float* C, long ldc, const float* bias, int relu, const float* mask, long ldmask, int acc, long kps,
                 float* slab, int vecA, int vecB) {
  // row-major float4 epilogue when every row segment is 16 B aligned
  const int vec_epi = slab == nullptr && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16 == 0) &&
                      (!mask || (ldmask % 4 == 0 && (uintptr_t)mask % 16 == 0)) && (!bias || (uintptr_t)bias % 16 == 0) &&
                      !getenv("A2C_GEMM_SCALAR_EPILOGUE");
  if (false) {}
  else if (M <= 96)
    hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, true>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu,
                       mask, ldmask, acc, kps, slab, vecA, vecB, vec_epi);
  else
    hipLaunchKernelGGL((gemm_kernel<A_KC, B_KC, false>), grid, dim3(256), 0, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu,
                       mask, ldmask, acc, kps, slab, vecA, vecB, vec_epi);
}
}  // namespace

// ---- the x 6 path: images, split pass, launch (shared by a2c_gemm_f32 and the a2c_gemm_x6_* entry points)
static bool x6_ready() {
  static std::once_flag once6;
  static bool ready6 = false;
  std::call_once(once6, [] {
    ready6 = hipFuncSetAttribute((const void*)gemm_x6_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)x6::LDS_BYTES) == hipSuccess;
  });
  return ready6;
}
static void x6_split(const float* src, long ld, long R, long K, bool k_contiguous, unsigned short* img, hipStream_t st) {
  const long Rp = (R + 255) / 256 * 256, Kp = (K + 15) / 16 * 16, ngb = (Kp / 8 + 3) / 4;
  const int g = (int)std::min<long>((Rp / 64) * ngb, 1L << 20);
  if (k_contiguous) hipLaunchKernelGGL((x6_split_kernel<false>), dim3(g), dim3(256), 0, st, src, ld, R, K, img, Rp, Kp);
  else hipLaunchKernelGGL((x6_split_kernel<true>), dim3(g), dim3(256), 0, st, src, ld, R, K, img, Rp, Kp);
}
// C = A B^T from two panel images; K splits only where the tiles alone leave CUs idle, and only as many as `splitk` (the
// caller's slabs) allows
static int x6_run(long M, long N, long K, const unsigned short* Ap, const unsigned short* Bp, float* C, long ldc, const float* bias,
                  int relu, const float* mask, long ldmask, int accumulate, int splitk, float* slab, hipStream_t st) {
  const long Mp = (M + 255) / 256 * 256, Np = (N + 255) / 256 * 256, Kp = (K + 15) / 16 * 16;
  const int nmb = (int)(Mp / 256), nnb = (int)(Np / 256), nkb = (int)(Kp / 16);
  int want = 1;
  if ((long)nmb * nnb < 192) want = (int)std::min<long>(256 / ((long)nmb * nnb), nkb / 32);
  if (want > splitk || !slab) want = slab ? splitk : 1;
  if (want < 1) want = 1;
  const int sps = (nkb + want - 1) / want, s6 = (nkb + sps - 1) / sps;
  const long wk = (long)nmb * nnb * s6, per = (wk + 7) / 8;
  hipLaunchKernelGGL(gemm_x6_kernel, dim3((unsigned)(per * 8)), dim3(512), x6::LDS_BYTES, st, M, N, Ap, Mp * Kp, Bp, Np * Kp, Kp, nmb,
                     nnb, s6, sps, C, ldc, bias, relu, mask, ldmask, accumulate, s6 > 1 ? slab : nullptr);
  A2C_CHECK_LAUNCH();
  if (s6 > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(a2c_grid_1d(M * N, 256)), dim3(256), 0, st, slab, s6, M, N, C, ldc, bias, relu, mask,
                       ldmask, accumulate);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

extern "C" {
size_t a2c_gemm_ws_bytes(int64_t M, int64_t N, int splitk) {
  size_t need = splitk > 1 ? (size_t)splitk * (size_t)M * (size_t)N * sizeof(float) : 0;
  if (M <= SN_MAX) {                      // the skinny weight-gradient path wants SN_BANDS slabs
    const size_t sn = (size_t)SN_BANDS * (size_t)M * (size_t)N * sizeof(float);
    if (sn > need) need = sn;
  }
  return need;
}

// bf16 x 9 path of a2c_gemm_f32 (large products): bytes of the three-piece bf16 images of both operands, rows padded to 128
// and k to 32; 0 when the product does not take that path.  The caller's workspace must hold a2c_gemm_ws_bytes(M, N, splitk)
// FOLLOWED by this (a2c_gemm_f32 falls back to the fp32 MFMA kernel when it does not).
static int x9_env() {                        // A2C_GEMM_X9 (read per call): -1 unset, else its first digit
  const char* e9 = getenv("A2C_GEMM_X9");
  return (e9 != nullptr && e9[0] >= '0' && e9[0] <= '9') ? e9[0] - '0' : -1;
}
static bool x6_mode() { return x9_env() != 1; }      // the six-product panel kernel (gemm_x6_kernel) unless the x 9 form is asked for
static bool x9_eligible(int64_t M, int64_t N, int64_t K) {
  // Measured on MI355X (tools/gemm_x9_bench.py, ConvModel's 28224 x 2000 layers, split passes included):
  //   x 6 (gemm_x6_kernel, DEFAULT for large products): 2048 rows NN 1.23 vs 2.26 ms fp32, TN 1.23 vs 2.20; 32,768 rows
  //     16.8 vs 31.3 and 18.9 vs 32.1 (196-220 TF fp32-equivalent);
  //   x 9 (gemm_x9_kernel, A2C_GEMM_X9=1): ties the fp32 MFMA kernels (2.33 / 2.18 ms; 30.2 / 34.1 ms) -- its 9/16 of the matrix
  //     time is eaten by 1.5 x the operand bytes through L2 / LDS with 128 x 128 tiles and register staging.
  // A2C_GEMM_X9=0: the fp32 MFMA kernels everywhere; =2: x 6 from the small threshold on (tests).
  const int e = x9_env();
  if (e == 0) return false;
  if (e == 1 || e == 2) return M >= 256 && N >= 256 && K >= 256 && (double)M * (double)N * (double)K >= 2.5e8;
  return M >= 1024 && N >= 1024 && K >= 1024 && (double)M * (double)N * (double)K >= 2e10;
}
size_t a2c_gemm_x9_ws_bytes(int64_t M, int64_t N, int64_t K) {
  if (M < 1 || N < 1 || K < 1 || !x9_eligible(M, N, K)) return 0;
  // (rows to 256, k to 32: room for either image layout)
  const size_t Mp = (size_t)(M + 255) / 256 * 256, Np = (size_t)(N + 255) / 256 * 256, Kp = (size_t)(K + 31) / 32 * 32;
  return 3 * 2 * (Mp + Np) * Kp + 256;
}

size_t a2c_gemm_x6_image_bytes(int64_t rows, int64_t K) {
  if (rows < 1 || K < 1) return 0;
  return 3 * 2 * (size_t)((rows + 255) / 256 * 256) * (size_t)((K + 15) / 16 * 16);
}

int a2c_gemm_x6_split(const float* src, int64_t ld, int64_t rows, int64_t K, int k_contiguous, void* image, a2c_stream_t stream) {
  if (rows < 1 || K < 1 || !src || !image || ((uintptr_t)image % 16) || ld < (k_contiguous ? K : rows)) return A2C_ERR_ARG;
  x6_split(src, (long)ld, (long)rows, (long)K, k_contiguous != 0, reinterpret_cast<unsigned short*>(image), a2c_s(stream));
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gemm_x6_images(int64_t M, int64_t N, int64_t K, const void* image_a, const void* image_b, float* C, int64_t ldc,
                       const float* bias, int relu, const float* mask, int64_t ldmask, int accumulate, int splitk, void* ws,
                       size_t ws_bytes, a2c_stream_t stream) {
  if (M < 1 || N < 1 || K < 1 || !image_a || !image_b || !C || ldc < N || (mask && ldmask < N)) return A2C_ERR_ARG;
  if (((uintptr_t)image_a % 16) || ((uintptr_t)image_b % 16)) return A2C_ERR_ARG;
  if (splitk < 1) splitk = 1;
  if (splitk > 1 && (!ws || ws_bytes < (size_t)splitk * (size_t)M * (size_t)N * sizeof(float))) return A2C_ERR_WORKSPACE;
  if (!x6_ready()) return A2C_ERR_LAUNCH;
  return x6_run((long)M, (long)N, (long)K, reinterpret_cast<const unsigned short*>(image_a),
                reinterpret_cast<const unsigned short*>(image_b), C, (long)ldc, bias, relu, mask, (long)ldmask, accumulate, splitk,
                splitk > 1 ? (float*)ws : nullptr, a2c_s(stream));
}

int a2c_small_n_bwd_data_bits(const float* dy, int64_t ldy, const float* W, float* dx, int64_t ldx, const uint8_t* maskbits,
                              int64_t mask_row_bytes, int64_t M, int N, int64_t K, a2c_stream_t stream) {
  if (M < 0 || N < 1 || N > SN_MAX || K < 0 || K % 8 || ldx % 4 || ldx < K || ldy < N || mask_row_bytes < K / 8) return A2C_ERR_ARG;
  if (M == 0 || K == 0) return A2C_OK;
  if (!dy || !W || !dx || !maskbits || ((uintptr_t)W % 16) || ((uintptr_t)dx % 16)) return A2C_ERR_ARG;
  hipLaunchKernelGGL(small_n_bwd_data_bits_kernel, dim3(a2c_grid_1d(M * (K / 8), 256)), dim3(256), 0, a2c_s(stream), dy, (long)ldy,
                     W, dx, (long)ldx, maskbits, (long)mask_row_bytes, (long)M, N, (int)K);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                 const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, int relu, const float* mask,
                 int64_t ldmask, int accumulate, int splitk, void* ws, size_t ws_bytes, a2c_stream_t stream) {
  if (M < 0 || N < 0 || K < 0) return A2C_ERR_ARG;
  if (M == 0 || N == 0) return A2C_OK;
  if (!A || !B || !C || K == 0) return A2C_ERR_ARG;
  hipStream_t st0 = a2c_s(stream);
  const bool al16 = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0);
  // skinny fast paths (heads): exact same maths, fp32 FMA chains in a different (fixed) order
  if (transA == 0 && transB == 1 && N <= SN_MAX && K % 4 == 0 && lda % 4 == 0 && ldb == K && !relu && !mask &&
      !accumulate && ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0)) {
    hipLaunchKernelGGL(small_n_fwd_kernel, dim3(a2c_grid_1d(M, 4, 4096)), dim3(256), 0, st0, A, (long)lda, B, bias, C,
                       (long)ldc, (long)M, (int)N, (int)K);
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  if (transA == 0 && transB == 0 && K <= SN_MAX && N % 4 == 0 && ldb == N && ldc % 4 == 0 && !relu && !bias &&
      (!mask || ldmask % 4 == 0) && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) &&
      (!mask || (uintptr_t)mask % 16 == 0)) {
    hipLaunchKernelGGL(small_n_bwd_data_kernel, dim3(a2c_grid_1d(M * (N / 4), 256)), dim3(256), 0, st0, A, (long)lda, B,
                       C, (long)ldc, mask, (long)ldmask, (long)M, (int)K, (int)N, accumulate);
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  if (transA == 1 && transB == 0 && M <= SN_MAX && N % 4 == 0 && N <= 8192 && ldb % 4 == 0 && ldc == N && !relu &&
      !bias && !mask && !accumulate && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0) && ws &&
      ws_bytes >= (size_t)SN_BANDS * M * N * sizeof(float)) {
    // columns in chunks of <= 1024 (a thread owns 4 columns of its chunk): blockIdx.y walks the chunks of ONE launch
    int nch = (int)((N + 1023) / 1024);
    while (nch <= N && !(N % nch == 0 && (N / nch) % 4 == 0 && N / nch <= 1024)) ++nch;
    if (nch <= N) {
      const int Kc = (int)(N / nch);
      // fewer bands when there are several chunks: the slabs are what the reduce reads
      const int bcap = nch > 1 ? SN_BANDS / 2 : SN_BANDS;
      const int bands = (int)(K < bcap ? K : bcap);
      hipLaunchKernelGGL(small_n_bwd_weight_kernel, dim3(bands, nch), dim3(256), 0, st0, A, (long)lda, B, (long)ldb, (float*)ws,
                         (long)K, (int)M, Kc);
      A2C_CHECK_LAUNCH();
      const long tot = (long)M * N;
      hipLaunchKernelGGL(small_n_bwd_weight_reduce, dim3((unsigned)((tot + 31) / 32)), dim3(256), 0, st0, (const float*)ws, bands,
                         (int)M, Kc, nch, C);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  if (transA == 1 && transB == 0 && K <= SN_MAX && !relu && !bias && !mask) {      // rank-K outer-product sum
    const bool v4 = N % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)B % 16 == 0) && ((uintptr_t)C % 16 == 0);
    if (v4)
      hipLaunchKernelGGL((small_k_tn_kernel<4>), dim3(a2c_grid_1d(M * (N / 4), 256)), dim3(256), 0, st0, A, (long)lda, B,
                         (long)ldb, C, (long)ldc, (long)M, (long)N, (int)K, accumulate);
    else
      hipLaunchKernelGGL((small_k_tn_kernel<1>), dim3(a2c_grid_1d(M * N, 256)), dim3(256), 0, st0, A, (long)lda, B, (long)ldb,
                         C, (long)ldc, (long)M, (long)N, (int)K, accumulate);
    A2C_CHECK_LAUNCH();
    return A2C_OK;
  }
  (void)al16;
  {  // small products: one launch, no slabs (see small_gemm_kernel)
    const long t128 = ((M + 127) / 128) * ((N + 127) / 128), t32 = ((M + 31) / 32) * ((N + 31) / 32);
    const bool vA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0), vB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
    if (transA == 0 && K % 8 == 0 && K >= 32 && K <= 4096 && t128 < 64 && (t32 >= (K > 1024 ? 128 : 16) || (M <= 32 && t32 >= 4 && K <= 1024)) && t32 <= 4096 && vA &&
        (transB == 0 || vB) &&
        !getenv("A2C_NO_SMALL_GEMM")) {
      dim3 grid((unsigned)((N + 31) / 32), (unsigned)((M + 31) / 32));
      const bool deep = K >= 1024 && t32 <= 512;                     // long K, few tiles: eight waves share it
      if (transB && deep) hipLaunchKernelGGL((small_gemm_kernel<true, 8>), grid, dim3(512), 0, st0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate);
      else if (transB) hipLaunchKernelGGL((small_gemm_kernel<true, 4>), grid, dim3(256), 0, st0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate);
      else if (deep) hipLaunchKernelGGL((small_gemm_kernel<false, 8>), grid, dim3(512), 0, st0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate);
      else hipLaunchKernelGGL((small_gemm_kernel<false, 4>), grid, dim3(256), 0, st0, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate);
      A2C_CHECK_LAUNCH();
      return A2C_OK;
    }
  }
  if (splitk < 1) splitk = 1;
  long kps = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;  // multiple of BK keeps 16 B alignment of k0
  splitk = (int)((K + kps - 1) / kps);
  float* slab = nullptr;
  if (splitk > 1) {
    if (!ws || ws_bytes < a2c_gemm_ws_bytes(M, N, splitk)) return A2C_ERR_WORKSPACE;
    slab = (float*)ws;
  }
  const int vecA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const int vecB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)splitk);
  hipStream_t st = a2c_s(stream);
  const bool a_kc = (transA == 0), b_kc = (transB != 0);
  {  // large products: the bf16 x 9 form (exact 3-way split of both operands, fp32 accumulation) when the workspace holds the images
    const size_t base = a2c_gemm_ws_bytes(M, N, splitk), need9 = a2c_gemm_x9_ws_bytes(M, N, K);
    const size_t off9 = (base + 255) / 256 * 256;
    if (need9 && ws && ws_bytes >= off9 + need9 && x6_mode() && x6_ready()) {
      const long Mp = (M + 255) / 256 * 256, Kp = (K + 15) / 16 * 16;
      unsigned short* Ap = reinterpret_cast<unsigned short*>((char*)ws + off9);
      unsigned short* Bp = Ap + 3 * Mp * Kp;
      x6_split(A, lda, M, K, a_kc, Ap, st);
      A2C_CHECK_LAUNCH();
      x6_split(B, ldb, N, K, b_kc, Bp, st);
      A2C_CHECK_LAUNCH();
      return x6_run(M, N, K, Ap, Bp, C, ldc, bias, relu, mask, ldmask, accumulate, splitk, slab, st);
    }
    if (need9 && ws && ws_bytes >= off9 + need9 && !x6_mode()) {
      const long Mp = (M + 127) / 128 * 128, Np = (N + 127) / 128 * 128, Kp = (K + 31) / 32 * 32;
      unsigned short* Ap = reinterpret_cast<unsigned short*>((char*)ws + off9);
      unsigned short* Bp = Ap + 3 * Mp * Kp;
      dim3 ga((unsigned)(Kp / 32), (unsigned)(Mp / 64)), gb((unsigned)(Kp / 32), (unsigned)(Np / 64));
      const int g1a = a2c_grid_1d(Mp * (Kp / 8), 256), g1b = a2c_grid_1d(Np * (Kp / 8), 256);
      if (a_kc) hipLaunchKernelGGL((x9_split_kernel<false>), dim3(g1a), dim3(256), 0, st, A, (long)lda, (long)M, (long)K, Ap, Mp, Kp, vecA);
      else hipLaunchKernelGGL((x9_split_kernel<true>), ga, dim3(256), 0, st, A, (long)lda, (long)M, (long)K, Ap, Mp, Kp, 0);
      A2C_CHECK_LAUNCH();
      if (b_kc) hipLaunchKernelGGL((x9_split_kernel<false>), dim3(g1b), dim3(256), 0, st, B, (long)ldb, (long)N, (long)K, Bp, Np, Kp, vecB);
      else hipLaunchKernelGGL((x9_split_kernel<true>), gb, dim3(256), 0, st, B, (long)ldb, (long)N, (long)K, Bp, Np, Kp, 0);
      A2C_CHECK_LAUNCH();
      const long kps9 = (kps + 31) / 32 * 32;
      const int sk9 = (int)((Kp + kps9 - 1) / kps9);
      if (sk9 <= splitk) {
        const int vec_epi = sk9 == 1 && N % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)C % 16 == 0) &&
                            (!mask || (ldmask % 4 == 0 && (uintptr_t)mask % 16 == 0)) && (!bias || (uintptr_t)bias % 16 == 0);
        dim3 g9((unsigned)(Np / 128), (unsigned)(Mp / 128), (unsigned)sk9);
        hipLaunchKernelGGL(gemm_x9_kernel, g9, dim3(256), 0, st, (long)M, (long)N, (long)K, Ap, Mp * Kp, Bp, Np * Kp, Kp, C,
                           (long)ldc, bias, relu, mask, (long)ldmask, accumulate, kps9, sk9 > 1 ? slab : nullptr, vec_epi);
        A2C_CHECK_LAUNCH();
        if (sk9 > 1) {
          hipLaunchKernelGGL(splitk_reduce_kernel, dim3(a2c_grid_1d(M * N, 256)), dim3(256), 0, st, slab, sk9, (long)M,
                             (long)N, C, (long)ldc, bias, relu, mask, (long)ldmask, accumulate);
          A2C_CHECK_LAUNCH();
        }
        return A2C_OK;
      }
    }
  }
  if (a_kc && b_kc && launch_gemm_nt(M, N, K, A, lda, B, ldb, kps, splitk, slab, vecA, vecB, st)) {}
  else if (a_kc && b_kc && launch_skinny_stream(M, N, K, A, lda, B, ldb, kps, splitk, slab, vecA, vecB, st)) {}
  else if (a_kc && b_kc) launch_gemm<true, true>(grid, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate, kps, slab, vecA, vecB);
  else if (a_kc && !b_kc) launch_gemm<true, false>(grid, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate, kps, slab, vecA, vecB);
  else if (!a_kc && b_kc) launch_gemm<false, true>(grid, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate, kps, slab, vecA, vecB);
  else launch_gemm<false, false>(grid, st, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, mask, ldmask, accumulate, kps, slab, vecA, vecB);
  A2C_CHECK_LAUNCH();
  if (splitk > 1) {
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(a2c_grid_1d(M * N, 256)), dim3(256), 0, st, slab, splitk, (long)M,
                       (long)N, C, (long)ldc, bias, relu, mask, (long)ldmask, accumulate);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

static long gemm_kps(int64_t K, int splitk) {
  if (splitk < 1) splitk = 1;
  return ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
}

int a2c_gemm_splits(int64_t K, int splitk) {
  if (K <= 0) return 0;
  const long kps = gemm_kps(K, splitk);
  return (int)((K + kps - 1) / kps);
}

int a2c_gemm_f32_partial(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda,
                         const float* B, int64_t ldb, int splitk, void* ws, size_t ws_bytes, a2c_stream_t stream) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !ws) return A2C_ERR_ARG;
  const long kps = gemm_kps(K, splitk);
  const int splits = a2c_gemm_splits(K, splitk);
  if (ws_bytes < (size_t)splits * M * N * sizeof(float)) return A2C_ERR_WORKSPACE;
  const int vecA = (lda % 4 == 0) && ((uintptr_t)A % 16 == 0);
  const int vecB = (ldb % 4 == 0) && ((uintptr_t)B % 16 == 0);
  dim3 grid((unsigned)((N + BN - 1) / BN), (unsigned)((M + BM - 1) / BM), (unsigned)splits);
  hipStream_t st = a2c_s(stream);
  float* slab = (float*)ws;
  const bool a_kc = (transA == 0), b_kc = (transB != 0);
  if (a_kc && b_kc && launch_gemm_nt(M, N, K, A, lda, B, ldb, kps, splits, slab, vecA, vecB, st)) {}
  else if (a_kc && b_kc && launch_skinny_stream(M, N, K, A, lda, B, ldb, kps, splits, slab, vecA, vecB, st)) {}
  else if (a_kc && b_kc) launch_gemm<true, true>(grid, st, M, N, K, A, lda, B, ldb, nullptr, 0, nullptr, 0, nullptr, 0, 0, kps, slab, vecA, vecB);
  else if (a_kc && !b_kc) launch_gemm<true, false>(grid, st, M, N, K, A, lda, B, ldb, nullptr, 0, nullptr, 0, nullptr, 0, 0, kps, slab, vecA, vecB);
  else if (!a_kc && b_kc) launch_gemm<false, true>(grid, st, M, N, K, A, lda, B, ldb, nullptr, 0, nullptr, 0, nullptr, 0, 0, kps, slab, vecA, vecB);
  else launch_gemm<false, false>(grid, st, M, N, K, A, lda, B, ldb, nullptr, 0, nullptr, 0, nullptr, 0, 0, kps, slab, vecA, vecB);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_heads_fused_publish(const float* xs, int nslab, int64_t slab_stride, int64_t ldx, const float* bias_in, int relu_in,
                            float* emb_out, int64_t ld_emb, const float* W, const float* b, float* heads, int64_t ldh, int64_t M,
                            int N, int K, const float* u, int n_logits, int64_t* actions, int64_t act_stride, uint64_t* cmd,
                            const uint32_t* seq_base, uint32_t seq_off, a2c_stream_t stream) {
  if (M < 0 || N < 1 || N > SN_MAX || K < 4 || K % 4 || nslab < 1) return A2C_ERR_ARG;
  if (M == 0) return A2C_OK;
  if (!xs || !W || !heads || (u && (!actions || n_logits < 1 || n_logits > N))) return A2C_ERR_ARG;
  if (cmd && (!u || !seq_base || (uintptr_t)cmd % 8)) return A2C_ERR_ARG;
  if (ldx % 4 || slab_stride % 4 || (emb_out && ld_emb % 4)) return A2C_ERR_ARG;
  if (((uintptr_t)xs | (uintptr_t)W | (uintptr_t)(bias_in ? bias_in : W) | (uintptr_t)(emb_out ? emb_out : (float*)W)) % 16)
    return A2C_ERR_ARG;
  hipLaunchKernelGGL(heads_fused_kernel, dim3(a2c_grid_1d(M, 1, 8192)), dim3(256), 0, a2c_s(stream), xs, nslab,
                     (long)slab_stride, (long)ldx, bias_in, relu_in, emb_out, (long)ld_emb, W, b, heads, (long)ldh,
                     (long)M, N, K, u, n_logits, actions, (long)act_stride, (unsigned long long*)cmd, seq_base, seq_off);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_heads_fused(const float* xs, int nslab, int64_t slab_stride, int64_t ldx, const float* bias_in, int relu_in,
                    float* emb_out, int64_t ld_emb, const float* W, const float* b, float* heads, int64_t ldh, int64_t M,
                    int N, int K, const float* u, int n_logits, int64_t* actions, int64_t act_stride,
                    a2c_stream_t stream) {
  return a2c_heads_fused_publish(xs, nslab, slab_stride, ldx, bias_in, relu_in, emb_out, ld_emb, W, b, heads, ldh, M, N, K, u,
                                 n_logits, actions, act_stride, nullptr, nullptr, 0, stream);
}

int a2c_gemm_f32_nt(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* bias, int relu, a2c_stream_t stream) {
  return a2c_gemm_f32(0, 1, M, N, K, A, lda, B, ldb, C, ldc, bias, relu, nullptr, 0, 0, 1, nullptr, 0, stream);
}
int a2c_gemm_f32_nn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, const float* mask, int64_t ldmask, a2c_stream_t stream) {
  return a2c_gemm_f32(0, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, mask, ldmask, 0, 1, nullptr, 0, stream);
}
int a2c_gemm_f32_tn(int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                    float* C, int64_t ldc, int splitk, void* ws, size_t ws_bytes, a2c_stream_t stream) {
  return a2c_gemm_f32(1, 0, M, N, K, A, lda, B, ldb, C, ldc, nullptr, 0, nullptr, 0, 0, splitk, ws, ws_bytes, stream);
}

int a2c_compose_heads(const float* Wh, const float* bh, const float* Wp, const float* bp, float* Wc, float* bc, int N,
                      int H, int F, a2c_stream_t stream) {
  if (N < 1 || N > SN_MAX || H < 1 || F < 1 || !Wh || !bh || !Wp || !bp || !Wc || !bc) return A2C_ERR_ARG;
  hipLaunchKernelGGL(compose_heads_kernel, dim3((unsigned)((F + 1 + 31) / 32)), dim3(128), 0, a2c_s(stream), Wh, bh, Wp, bp,
                     Wc, bc, N, H, F);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

size_t a2c_colsum_ws_bytes(int64_t N) { return (size_t)CS_BANDS * (size_t)(N > 0 ? N : 0) * sizeof(float); }

int a2c_colsum(const float* x, int64_t ld, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes,
               a2c_stream_t stream) {
  if (M < 0 || N < 0) return A2C_ERR_ARG;
  if (N == 0) return A2C_OK;
  if (!out || (M > 0 && !x)) return A2C_ERR_ARG;
  if (!ws || ws_bytes < a2c_colsum_ws_bytes(N)) return A2C_ERR_WORKSPACE;
  int bands = (int)(M < CS_BANDS ? (M > 0 ? M : 1) : CS_BANDS);
  dim3 g1((unsigned)((N + 255) / 256), (unsigned)bands);
  hipLaunchKernelGGL(colsum_stage1, g1, dim3(256), 0, a2c_s(stream), x, (long)ld, (long)M, (long)N, (float*)ws);
  A2C_CHECK_LAUNCH();
  hipLaunchKernelGGL(colsum_stage2, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, a2c_s(stream), (const float*)ws,
                     bands, (long)N, out);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
