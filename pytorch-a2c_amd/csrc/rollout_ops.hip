// Per-env-step kernels of Runner.rollout (runner.py:174-248), batched over B slots.
// All memory-bound / latency-bound; the rollout buffers keep the reference's rollout-major
// layout (element e = slot*T + t, training.py:88-101), so a "row of step t" is a strided
// gather over slots: each kernel takes explicit strides instead of staging copies.
#include "a2c_common.h"

namespace {

// uint8 frames (the host pool's transport format, preprocessing.py:11-17 yields uint8): 4 pixels per dword,
// expanded to fp32 on the way into the state (runner.py:199 `FloatTensor(state)`)
__device__ __forceinline__ float4 u8x4_to_f32(unsigned int w) {
  return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
}

// plane C-1 of the new state from a uint8 frame: `fstride` bytes between envs, HW % 4 == 0
__device__ __forceinline__ void stack_u8_plane(const uint8_t* __restrict__ f, float* __restrict__ o, int HW) {
  const unsigned int* __restrict__ w = reinterpret_cast<const unsigned int*>(f);
  const int n4 = HW >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) reinterpret_cast<float4*>(o)[i] = u8x4_to_f32(w[i]);
}

// a1: frame stack.  One workgroup column per (env, plane); 16 B per lane when HW % 4 == 0.
template <bool VEC4>
__global__ __launch_bounds__(256) void frame_stack_kernel(const float* __restrict__ frame_new,
                                                          const uint8_t* __restrict__ frame_u8, long fstride,
                                                          const float* __restrict__ reset_mask,
                                                          const float* __restrict__ prev, long prev_stride,
                                                          float* __restrict__ out, long out_stride, int C,
                                                          int HW) {
  const int b = blockIdx.y;
  const int c = blockIdx.z;
  const bool rst = reset_mask != nullptr && reset_mask[b] != 0.f;
  float* o = out + (long)b * out_stride + (long)c * HW;
  const float* src = nullptr;  // nullptr => zeros
  if (c == C - 1) {
    if (frame_u8) { stack_u8_plane(frame_u8 + (long)b * fstride, o, HW); return; }
    src = frame_new + (long)b * HW;
  } else if (!rst) src = prev + (long)b * prev_stride + (long)(c + 1) * HW;
  if (VEC4) {
    const int n4 = HW >> 2;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (src) v = reinterpret_cast<const float4*>(src)[i];
      reinterpret_cast<float4*>(o)[i] = v;
    }
  } else {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) o[i] = src ? src[i] : 0.f;
  }
}

// a2: softmax + inverse-CDF sampling, one lane per env (A is small: 2..18).
template <bool SOFTMAX>
__global__ __launch_bounds__(256) void sample_kernel(const float* __restrict__ logits, long ld,
                                                     const float* __restrict__ u, int64_t* __restrict__ act_i,
                                                     long act_stride, float* __restrict__ act_f,
                                                     float* __restrict__ probs, long B, int A) {
  for (long b = blockIdx.x * 256L + threadIdx.x; b < B; b += gridDim.x * 256L) {
    const float* row = logits + b * ld;
    float mx = -INFINITY, den = 1.f;
    if (SOFTMAX) {
      for (int a = 0; a < A; ++a) mx = fmaxf(mx, row[a]);
      den = 0.f;
      for (int a = 0; a < A; ++a) den += expf(row[a] - mx);
    }
    const float ub = u[b];
    float cs = 0.f;
    int pick = -1;
    for (int a = 0; a < A; ++a) {
      const float p = SOFTMAX ? expf(row[a] - mx) / den : row[a];
      if (probs) probs[b * A + a] = p;
      cs = __fadd_rn(cs, p);
      if (pick < 0 && cs >= ub) pick = a;
    }
    // Rounding can leave the fp32 cumsum of a softmax below a uniform close to 1 (utils.py:58 then returns
    // -1, which indexes the LAST action in the loss, updater.py:104, but is not a valid env action): the
    // rollout sampler returns that last action.  a2c_sample_probs keeps the reference's -1.
    if (SOFTMAX && pick < 0) pick = A - 1;
    if (act_i) act_i[b * act_stride] = (int64_t)pick;
    if (act_f) act_f[b] = (float)pick;
  }
}

// a3: rewards / dones / TD deltas of one env step (runner.py:212-232)
__global__ __launch_bounds__(256) void record_kernel(const float* __restrict__ rew, const float* __restrict__ done,
                                                     const float* __restrict__ val, long vstride,
                                                     float* __restrict__ val_prev,
                                                     float* __restrict__ rewards, float* __restrict__ dones,
                                                     float* __restrict__ deltas, float* __restrict__ done_eff_out,
                                                     int B, long T, long t, long slot0, float gamma, int pong) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const long e = (slot0 + b) * T + t;
  const float r = rew[b];
  float d = done[b] != 0.f ? 1.f : 0.f;
  if (pong && r != 0.f) d = 1.f;
  rewards[e] = r;
  dones[e] = d;
  if (done_eff_out) done_eff_out[b] = d;
  const float v = val[b * vstride];
  if (t > 0) {
    // delta = prev_rew + gamma*val*(1-prev_done) - prev_val, left to right, each op rounded
    const float pr = rewards[e - 1], pd = dones[e - 1];
    const float gv = __fmul_rn(gamma, v);
    deltas[e - 1] = __fsub_rn(__fadd_rn(pr, __fmul_rn(gv, __fsub_rn(1.f, pd))), val_prev[b]);
  }
  val_prev[b] = v;
}

// record + frame stack in one launch: blocks with blockIdx.z < C stack plane z, the z == C layer
// (only its x == 0 blocks, one per 256 envs) does the bookkeeping of record_kernel
__global__ __launch_bounds__(256) void post_kernel(const float* __restrict__ rew, const float* __restrict__ done,
                                                   const float* __restrict__ val, long vstride,
                                                   float* __restrict__ val_prev, float* __restrict__ rewards,
                                                   float* __restrict__ dones, float* __restrict__ deltas, long T, long t,
                                                   long slot0, float gamma, int pong,
                                                   const float* __restrict__ frame_new,
                                                   const uint8_t* __restrict__ frame_u8, long fstride,
                                                   const float* __restrict__ reset_mask, const float* __restrict__ prev,
                                                   long prev_stride, float* __restrict__ out, long out_stride, int B,
                                                   int C, int HW, float* __restrict__ done_eff_out, float* h,
                                                   int hdim, float* __restrict__ h_rows, long h_rows_stride,
                                                   const float* h_src, int* __restrict__ nv_rows = nullptr,
                                                   int* __restrict__ nv_carry = nullptr) {
  if ((int)blockIdx.z == C + 1) {      // recurrent nets (runner.py:201,219-221): h = 0 where the episode ended, then
    if (blockIdx.x != 0) return;       // h_states[row] = h for the step that follows
    const int b = blockIdx.y;
    const bool d = (done[b] != 0.f) || (pong && rew[b] != 0.f);
    for (int i = threadIdx.x; i < hdim; i += 256) {      // h_src: where the previous step left its new hidden rows
      const float v = d ? 0.f : h_src[(long)b * hdim + i];
      if (d || h_src != h) h[(long)b * hdim + i] = v;
      if (h_rows) h_rows[(long)b * h_rows_stride + i] = v;
    }
    return;
  }
  if ((int)blockIdx.z == C) {
    if (blockIdx.x != 0) return;
    const int b = blockIdx.y * 256 + threadIdx.x;
    if (blockIdx.y * 256 >= B || b >= B) return;
    const long e = (slot0 + b) * T + t;
    const float r = rew[b];
    float d = done[b] != 0.f ? 1.f : 0.f;
    if (pong && r != 0.f) d = 1.f;
    rewards[e] = r;
    dones[e] = d;
    if (done_eff_out) done_eff_out[b] = d;
    const float v = val[b * vstride];
    if (t > 0) {
      const float pr = rewards[e - 1], pd = dones[e - 1];
      const float gv = gamma * v;
      deltas[e - 1] = (pr + gv * (1.f - pd)) - val_prev[b];
    }
    if (nv_carry) {      // single-frame store: how many planes of the NEXT state are real frames (utils.py:37-42: a reset
      const int nv = done[b] != 0.f ? 1 : min(nv_carry[b] + 1, 4);      // leaves the reset frame behind zeros)
      nv_carry[b] = nv;
      if (t + 1 < T) nv_rows[e + 1] = nv;
    }
    val_prev[b] = v;
    return;
  }
  const int b = blockIdx.y;
  const int c = blockIdx.z;
  const bool rst = reset_mask != nullptr && reset_mask[b] != 0.f;
  float* o = out + (long)b * out_stride + (long)c * HW;
  const float* src = nullptr;
  if (c == C - 1) {
    if (frame_u8) { stack_u8_plane(frame_u8 + (long)b * fstride, o, HW); return; }
    src = frame_new + (long)b * HW;
  } else if (!rst) src = prev + (long)b * prev_stride + (long)(c + 1) * HW;
  const int n4 = HW >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src) v = reinterpret_cast<const float4*>(src)[i];
    reinterpret_cast<float4*>(o)[i] = v;
  }
}

__global__ __launch_bounds__(256) void zero_done_rows_kernel(float* __restrict__ h, int hdim,
                                                             const float* __restrict__ done_eff, int B) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)B * hdim) return;
  if (done_eff[i / hdim] != 0.f) h[i] = 0.f;
}

__global__ __launch_bounds__(256) void bootstrap_kernel(const float* __restrict__ val_boot, long vstride,
                                                        const float* __restrict__ val_prev,
                                                        float* __restrict__ rewards, float* __restrict__ dones,
                                                        float* __restrict__ deltas, int B, long T, long slot0,
                                                        float gamma) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const long e = (slot0 + b) * T + T - 1;
  float r = rewards[e];
  if (dones[e] == 0.f) {
    r = __fadd_rn(r, __fmul_rn(gamma, val_boot[b * vstride]));
    rewards[e] = r;
    dones[e] = 1.f;
  }
  deltas[e] = __fsub_rn(r, val_prev[b]);
}

__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ src, long ss,
                                                        float* __restrict__ dst, long ds, long n, long B) {
  for (long b = blockIdx.y; b < B; b += gridDim.y)
    for (long j = blockIdx.x * 256L + threadIdx.x; j < n; j += gridDim.x * 256L) dst[b * ds + j] = src[b * ss + j];
}

__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, long ld,
                                                        const float* __restrict__ dones, long dstride, int B,
                                                        int n) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= (long)B * n) return;
  const long b = i / n, j = i - b * n;
  x[b * ld + j] *= (1.f - dones[b * dstride]);
}
__global__ __launch_bounds__(256) void permute_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                           long R, long T, long n) {
  const long tot = R * T * n;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < tot; i += gridDim.x * 256L) {
    const long row = i / n, j = i - row * n;      // dst row = t*R + r
    const long t = row / R, r = row - t * R;
    dst[i] = src[(r * T + t) * n + j];
  }
}

// ---- device relay of the host env pool (include/a2c_hostpool.h) for the per-step rollout of ANY model -------
// The persistent A3C kernel talks to the env workers itself; every other model runs one segment of kernels per env
// step.  With these two kernels the segments need no host in between either: the tail of segment k publishes the
// sampled actions as cmd granules in the pinned region (system-scope 8-byte stores), the head of segment k+1 waits
// for the workers' rec granules and pulls the frames over PCIe with 16-byte system-scope loads (what the host's
// D2H copy / post_actions / wait_frames / H2D copies did, at ~50 us of host round trip + launch gaps per step).
// The step number comes from device memory (seq_base[0] + seq_off) so that a captured segment can be replayed by
// later rollouts.  A worker that does not answer within timeout_ticks (100 MHz) sets *err; later waits return at once.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;

__global__ __launch_bounds__(256) void pool_publish_kernel(unsigned long long* __restrict__ cmd,
                                                           const int64_t* __restrict__ actions, long stride, int n,
                                                           const unsigned int* __restrict__ seq_base, unsigned int seq_off) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned long long seq = (unsigned long long)(seq_base[0] + seq_off);
  __hip_atomic_store(cmd + i, (seq << 32) | (unsigned int)actions[i * stride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// 16 packed pixels (bit q = pixel q) -> 16 uint8 pixels
__device__ __forceinline__ u32x4 expand_bits16(unsigned int b) {
  u32x4 v;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const unsigned int w = b >> (4 * q);
    v[q] = (w & 1u) | ((w & 2u) << 7) | ((w & 4u) << 14) | ((w & 8u) << 21);
  }
  return v;
}

// packed frames (A2C_FRAME_BITS transport, staged in HBM by a memcpy) -> uint8 frames; n_pixels % 16 == 0
__global__ __launch_bounds__(256) void unpack_bits_kernel(const uint8_t* __restrict__ src, long sstride,
                                                          uint8_t* __restrict__ dst, long dstride, int n_pixels) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;           // group of 16 pixels
  if (i * 16 >= n_pixels) return;
  const unsigned int bits = reinterpret_cast<const unsigned short*>(src + (long)b * sstride)[i];
  *reinterpret_cast<u32x4*>(dst + (long)b * dstride + i * 16) = expand_bits16(bits);
}

__global__ void store_u32_system_kernel(unsigned int* p, unsigned int v) {
  if (threadIdx.x == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// POST: the workgroup that fetched env b's answer also does env b's share of a2c_rollout_post_frames (the bookkeeping of the
// env step the answer belongs to and, for recurrent nets, the hidden row of the next step): one launch fewer per env step,
// same arithmetic (post_kernel's z == C / C + 1 layers).
struct IngestPost {
  const float* val; long vstride;
  float* val_prev; float* rewards; float* dones; float* deltas;
  long T, t, slot0;
  float gamma; int pong;
  float* done_eff; float* h; int hdim; float* h_rows; long h_rows_stride; const float* h_src;
  int* nv_rows; int* nv_carry;
};

template <bool BITS, bool POST = false>
__global__ __launch_bounds__(256) void pool_ingest_kernel(const unsigned long long* __restrict__ rec,
                                                          const uint8_t* __restrict__ frames, long fstride, int fbytes,
                                                          const unsigned int* __restrict__ seq_base, unsigned int seq_off,
                                                          long timeout_ticks, int* __restrict__ err, float* __restrict__ rew,
                                                          float* __restrict__ done, uint8_t* __restrict__ out, long ostride,
                                                          IngestPost q = IngestPost{}) {
  __shared__ unsigned int sh[3];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) {
    const unsigned int want = (seq_base[0] + seq_off) & 0x7fffffffu;      // rec carries the step number modulo 2^31
    unsigned long long gr = 0ULL;
    unsigned int dead = 1u;                          // its own flag: every 64-bit value is a legal granule
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      const unsigned long long t0 = wall_clock64();
      for (;;) {
        gr = __hip_atomic_load(rec + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned int)(gr >> 33) == want) { dead = 0u; break; }
        if ((long)(wall_clock64() - t0) > timeout_ticks) {
          __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(16);
      }
    }
    sh[0] = (unsigned int)gr;
    sh[1] = (unsigned int)(gr >> 32);
    sh[2] = dead;
  }
  __syncthreads();
  const unsigned int lo = sh[0], hi = sh[1];
  if (sh[2] != 0u) return;                            // timeout (now or earlier): the error flag is set
  if (tid == 0) {
    rew[b] = __uint_as_float(lo);
    done[b] = (hi & 1u) ? 1.f : 0.f;
  }
  if (POST) {
    const float r = __uint_as_float(lo);
    const bool reset = (hi & 1u) != 0u;
    if (tid == 64) {                                    // (a wave of its own: tid 0's wave starts the frame loads)
      const long e = (q.slot0 + b) * q.T + q.t;
      float d = reset ? 1.f : 0.f;
      if (q.pong && r != 0.f) d = 1.f;
      q.rewards[e] = r;
      q.dones[e] = d;
      if (q.done_eff) q.done_eff[b] = d;
      const float v = q.val[b * q.vstride];
      if (q.t > 0) {
        const float pr = q.rewards[e - 1], pd = q.dones[e - 1];
        const float gv = q.gamma * v;
        q.deltas[e - 1] = (pr + gv * (1.f - pd)) - q.val_prev[b];
      }
      const int nv = reset ? 1 : min(q.nv_carry[b] + 1, 4);
      q.nv_carry[b] = nv;
      if (q.t + 1 < q.T) q.nv_rows[e + 1] = nv;
      q.val_prev[b] = v;
    }
    if (q.h != nullptr) {
      const bool d = reset || (q.pong && r != 0.f);
      for (int i = tid; i < q.hdim; i += 256) {
        const float v = d ? 0.f : q.h_src[(long)b * q.hdim + i];
        if (d || q.h_src != q.h) q.h[(long)b * q.hdim + i] = v;
        if (q.h_rows) q.h_rows[(long)b * q.h_rows_stride + i] = v;
      }
    }
  }
  // the frame was written before its rec granule (release): 16-byte system-scope loads straight from pinned host memory
  __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(frames + (long)b * fstride), 0, fbytes, 0x00020000);
  uint8_t* __restrict__ o = out + (long)b * ostride;
  if (BITS) {     // fbytes = ceil(n_pixels / 8) packed bytes: 2 bytes per lane -> 16 uint8 pixels in HBM
    for (int off = tid * 2; off < fbytes; off += 256 * 2) {
      const unsigned int bits = __builtin_amdgcn_raw_buffer_load_b16(fr, off, 0, 1 | 16);
      *reinterpret_cast<u32x4*>(o + off * 8) = expand_bits16(bits);
    }
    return;
  }
  for (int off = tid * 16; off < fbytes; off += 256 * 16) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(fr, off, 0, 1 | 16);   // sc0 sc1
    *reinterpret_cast<u32x4*>(o + off) = v;
  }
}

// ---- row f4 of SURVEY.md section 8: preprocessing on the device and the single-frame uint8 store -----------------
// pong_prep / breakout_prep (preprocessing.py:11-23) on raw (H, W, C) uint8 frames: crop rows y0:y1 and columns x0:x1,
// keep every `step`-th pixel of channel 0; binarise (Pong: 144 and 109 are background -> 0, everything else that is
// not 0 -> 1) or keep the byte (Breakout: skimage's rgb2grey returns the 2-D slice unchanged, see a2c_amd/preprocessing.py)
__global__ __launch_bounds__(256) void frame_prep_kernel(const uint8_t* __restrict__ raw, long raw_stride, int W, int C,
                                                         int y0, int x0, int step, int OH, int OW, int binarise,
                                                         uint8_t* __restrict__ out, long out_stride) {
  const int b = blockIdx.y;
  const uint8_t* __restrict__ r = raw + (long)b * raw_stride;
  uint8_t* __restrict__ o = out + (long)b * out_stride;
  const int n4 = (OH * OW) >> 2;                                     // 4 output pixels per thread (OW % 4 == 0)
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256) {
    const int oy = (i * 4) / OW, ox = i * 4 - oy * OW;
    unsigned int w = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unsigned int v = r[((long)(y0 + oy * step) * W + (x0 + (ox + q) * step)) * C];
      if (binarise) v = (v == 144u || v == 109u || v == 0u) ? 0u : 1u;
      w |= v << (8 * q);
    }
    reinterpret_cast<unsigned int*>(o)[i] = w;
  }
}

// single-frame store -> fp32 states (the reference's layout, runner.py:199): state k of slot r is the window
// frames[r][k .. k+C-1], planes older than the last reset (c < C - nvalid) are zero
__global__ __launch_bounds__(256) void frames_to_states_kernel(const uint8_t* __restrict__ f, long slot_stride,
                                                               const int* __restrict__ nvalid, long nv_slot_stride,
                                                               float* __restrict__ out, long out_slot_stride, int nt, int C,
                                                               int HW) {
  const int r = blockIdx.y / nt, k = blockIdx.y - r * nt, c = blockIdx.z;
  const int nv = nvalid ? nvalid[(long)r * nv_slot_stride + k] : C;
  const unsigned int* __restrict__ src = reinterpret_cast<const unsigned int*>(f + (long)r * slot_stride + (long)(k + c) * HW);
  float4* __restrict__ o = reinterpret_cast<float4*>(out + (long)r * out_slot_stride + ((long)k * C + c) * HW);
  const bool live = c >= C - nv;
  const int n4 = HW >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
    o[i] = live ? u8x4_to_f32(src[i]) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// start of a slot: the C frames the previous slot ended with move to the head of the window (frames[r][0 .. C-1] =
// frames[r][T .. T+C-1]) and state 0's valid-plane count is the carried one
__global__ __launch_bounds__(256) void frame_store_begin_kernel(uint8_t* __restrict__ f, long slot_stride, long T, int C, int HW,
                                                                int* __restrict__ nv_rows, const int* __restrict__ nv_carry) {
  const int b = blockIdx.y;
  uint8_t* __restrict__ fr = f + (long)b * slot_stride;
  const int n4 = (C * HW) >> 2;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n4; i += gridDim.x * 256)
    reinterpret_cast<unsigned int*>(fr)[i] = reinterpret_cast<const unsigned int*>(fr + T * HW)[i];
  if (blockIdx.x == 0 && threadIdx.x == 0) nv_rows[(long)b * T] = nv_carry[b];
}
}  // namespace

extern "C" {
int a2c_permute_rows(const float* src, float* dst, int64_t R, int64_t T, int64_t n, a2c_stream_t stream) {
  if (R < 0 || T < 0 || n < 0) return A2C_ERR_ARG;
  if (R * T * n == 0) return A2C_OK;
  if (!src || !dst) return A2C_ERR_ARG;
  hipLaunchKernelGGL(permute_rows_kernel, dim3(a2c_grid_1d(R * T * n, 256)), dim3(256), 0, a2c_s(stream), src, dst,
                     (long)R, (long)T, (long)n);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

static int frame_stack_push_any(const float* frame_new, const uint8_t* frame_u8, int64_t fstride,
                                const float* reset_mask, const float* prev, int64_t prev_stride, float* out,
                                int64_t out_stride, int B, int C, int HW, a2c_stream_t stream) {
  if (B < 0 || C < 1 || HW < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if ((!frame_new && !frame_u8) || !out || (C > 1 && !prev)) return A2C_ERR_ARG;
  const bool vec = (HW % 4 == 0) && (prev_stride % 4 == 0) && (out_stride % 4 == 0) &&
                   (((uintptr_t)frame_new | (uintptr_t)prev | (uintptr_t)out) % 16 == 0);
  if (frame_u8 && (!vec || fstride % 4 || (uintptr_t)frame_u8 % 4 || fstride < HW)) return A2C_ERR_ARG;
  const int work = vec ? HW / 4 : HW;
  dim3 grid((work + 255) / 256 > 8 ? 8 : (work + 255) / 256, B, C);
  if (vec)
    hipLaunchKernelGGL(frame_stack_kernel<true>, grid, dim3(256), 0, a2c_s(stream), frame_new, frame_u8, (long)fstride,
                       reset_mask, prev, (long)prev_stride, out, (long)out_stride, C, HW);
  else
    hipLaunchKernelGGL(frame_stack_kernel<false>, grid, dim3(256), 0, a2c_s(stream), frame_new, frame_u8, (long)fstride,
                       reset_mask, prev, (long)prev_stride, out, (long)out_stride, C, HW);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_frame_stack_push(const float* frame_new, const float* reset_mask, const float* prev, int64_t prev_stride,
                         float* out, int64_t out_stride, int B, int C, int HW, a2c_stream_t stream) {
  if (!frame_new && B > 0) return A2C_ERR_ARG;
  return frame_stack_push_any(frame_new, nullptr, 0, reset_mask, prev, prev_stride, out, out_stride, B, C, HW, stream);
}

int a2c_frame_stack_push_u8(const uint8_t* frame_u8, int64_t frame_stride, const float* reset_mask, const float* prev,
                            int64_t prev_stride, float* out, int64_t out_stride, int B, int C, int HW,
                            a2c_stream_t stream) {
  if (!frame_u8 && B > 0) return A2C_ERR_ARG;
  return frame_stack_push_any(nullptr, frame_u8, frame_stride, reset_mask, prev, prev_stride, out, out_stride, B, C, HW,
                              stream);
}

int a2c_softmax_sample(const float* logits, int64_t ld_logits, const float* u, int64_t* actions,
                       int64_t act_stride, float* probs, int B, int A, a2c_stream_t stream) {
  if (B < 0 || A < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!logits || !u || !actions) return A2C_ERR_ARG;
  hipLaunchKernelGGL(sample_kernel<true>, dim3(a2c_grid_1d(B, 256)), dim3(256), 0, a2c_s(stream), logits,
                     (long)ld_logits, u, actions, (long)act_stride, (float*)nullptr, probs, (long)B, A);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_sample_probs(const float* probs, const float* u, float* actions, int64_t B, int A, a2c_stream_t stream) {
  if (B < 0 || A < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!probs || !u || !actions) return A2C_ERR_ARG;
  hipLaunchKernelGGL(sample_kernel<false>, dim3(a2c_grid_1d(B, 256)), dim3(256), 0, a2c_s(stream), probs, (long)A, u,
                     (int64_t*)nullptr, 0L, actions, (float*)nullptr, (long)B, A);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_rollout_record(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                       float* rewards,
                       float* dones, float* deltas, float* done_eff_out, float* h, int hdim, int B, int64_t T,
                       int64_t t, int64_t slot0, float gamma, int pong, a2c_stream_t stream) {
  if (B < 0 || T < 1 || t < 0 || t >= T) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!rew || !done || !val || !val_prev || !rewards || !dones || !deltas) return A2C_ERR_ARG;
  if (h && (!done_eff_out || hdim < 1)) return A2C_ERR_ARG;
  hipLaunchKernelGGL(record_kernel, dim3((B + 255) / 256), dim3(256), 0, a2c_s(stream), rew, done, val,
                     (long)val_stride, val_prev, rewards, dones, deltas, done_eff_out, B, (long)T, (long)t, (long)slot0,
                     gamma, pong);
  A2C_CHECK_LAUNCH();
  if (h) {
    const long n = (long)B * hdim;
    hipLaunchKernelGGL(zero_done_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, a2c_s(stream), h, hdim,
                       done_eff_out, B);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

static int rollout_post_any(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                            float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma,
                            int pong, const float* frame_new, const uint8_t* frame_u8, int64_t fstride,
                            const float* reset_mask, const float* prev, int64_t prev_stride, float* out,
                            int64_t out_stride, int B, int C, int HW, a2c_stream_t stream, float* done_eff_out = nullptr,
                            float* h = nullptr, int hdim = 0, float* h_rows = nullptr, int64_t h_rows_stride = 0,
                            const float* h_src = nullptr) {
  if (B < 0 || T < 1 || t < 0 || t >= T || C < 1 || HW < 4 || HW % 4) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!rew || !done || !val || !val_prev || !rewards || !dones || !deltas || (!frame_new && !frame_u8) || !out ||
      (C > 1 && !prev))
    return A2C_ERR_ARG;
  if (prev_stride % 4 || out_stride % 4 || (((uintptr_t)frame_new | (uintptr_t)prev | (uintptr_t)out) % 16))
    return A2C_ERR_ARG;
  if (frame_u8 && (fstride % 4 || (uintptr_t)frame_u8 % 4 || fstride < HW)) return A2C_ERR_ARG;
  if (h && hdim < 1) return A2C_ERR_ARG;
  const int work = HW / 4;
  dim3 grid((work + 255) / 256 > 8 ? 8 : (work + 255) / 256, B, C + (h ? 2 : 1));
  hipLaunchKernelGGL(post_kernel, grid, dim3(256), 0, a2c_s(stream), rew, done, val, (long)val_stride, val_prev, rewards,
                     dones, deltas, (long)T, (long)t, (long)slot0, gamma, pong, frame_new, frame_u8, (long)fstride,
                     reset_mask, prev, (long)prev_stride, out, (long)out_stride, B, C, HW, done_eff_out, h, hdim, h_rows,
                     (long)h_rows_stride, h_src ? h_src : h);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_rollout_post(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                     float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma, int pong,
                     const float* frame_new, const float* reset_mask, const float* prev, int64_t prev_stride, float* out,
                     int64_t out_stride, int B, int C, int HW, a2c_stream_t stream) {
  if (!frame_new && B > 0) return A2C_ERR_ARG;
  return rollout_post_any(rew, done, val, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, frame_new,
                          nullptr, 0, reset_mask, prev, prev_stride, out, out_stride, B, C, HW, stream);
}

int a2c_rollout_post_u8(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                        float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma,
                        int pong, const uint8_t* frame_u8, int64_t frame_stride, const float* reset_mask, const float* prev,
                        int64_t prev_stride, float* out, int64_t out_stride, int B, int C, int HW, a2c_stream_t stream) {
  if (!frame_u8 && B > 0) return A2C_ERR_ARG;
  return rollout_post_any(rew, done, val, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, nullptr,
                          frame_u8, frame_stride, reset_mask, prev, prev_stride, out, out_stride, B, C, HW, stream);
}

int a2c_rollout_post_rec(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                         float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma,
                         int pong, const float* frame_new, const uint8_t* frame_u8, int64_t frame_stride,
                         const float* reset_mask, const float* prev, int64_t prev_stride, float* out, int64_t out_stride,
                         int B, int C, int HW, float* done_eff_out, float* h, int hdim, float* h_rows,
                         int64_t h_rows_stride, const float* h_src, a2c_stream_t stream) {
  if (B > 0 && (!h || (!frame_new == !frame_u8))) return A2C_ERR_ARG;
  return rollout_post_any(rew, done, val, val_stride, val_prev, rewards, dones, deltas, T, t, slot0, gamma, pong, frame_new,
                          frame_u8, frame_stride, reset_mask, prev, prev_stride, out, out_stride, B, C, HW, stream,
                          done_eff_out, h, hdim, h_rows, h_rows_stride, h_src);
}

int a2c_rollout_bootstrap(const float* val_boot, int64_t val_stride, const float* val_prev, float* rewards,
                          float* dones, float* deltas,
                          int B, int64_t T, int64_t slot0, float gamma, a2c_stream_t stream) {
  if (B < 0 || T < 1) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!val_boot || !val_prev || !rewards || !dones || !deltas) return A2C_ERR_ARG;
  hipLaunchKernelGGL(bootstrap_kernel, dim3((B + 255) / 256), dim3(256), 0, a2c_s(stream), val_boot,
                     (long)val_stride, val_prev, rewards, dones, deltas, B, (long)T, (long)slot0, gamma);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_copy_rows(const float* src, int64_t src_stride, float* dst, int64_t dst_stride, int B, int64_t n,
                  a2c_stream_t stream) {
  if (B < 0 || n < 0) return A2C_ERR_ARG;
  if (B == 0 || n == 0) return A2C_OK;
  if (!src || !dst) return A2C_ERR_ARG;
  dim3 grid((unsigned)((n + 255) / 256 > 64 ? 64 : (n + 255) / 256), B > 32768 ? 32768 : B);
  hipLaunchKernelGGL(copy_rows_kernel, grid, dim3(256), 0, a2c_s(stream), src, (long)src_stride, dst,
                     (long)dst_stride, (long)n, (long)B);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_mask_rows(float* x, int64_t ld, const float* dones, int64_t done_stride, int B, int n,
                  a2c_stream_t stream) {
  if (B < 0 || n < 0) return A2C_ERR_ARG;
  if (B == 0 || n == 0) return A2C_OK;
  if (!x || !dones) return A2C_ERR_ARG;
  const long tot = (long)B * n;
  hipLaunchKernelGGL(mask_rows_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, a2c_s(stream), x, (long)ld,
                     dones, (long)done_stride, B, n);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_pool_publish_actions(uint64_t* cmd, const int64_t* actions, int64_t act_stride, int n, const uint32_t* seq_base,
                             uint32_t seq_off, a2c_stream_t stream) {
  if (n < 0 || !cmd || !actions || !seq_base || act_stride < 1) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL(pool_publish_kernel, dim3((n + 255) / 256), dim3(256), 0, a2c_s(stream), (unsigned long long*)cmd,
                     actions, (long)act_stride, n, seq_base, seq_off);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_pool_ingest(const uint64_t* rec, const uint8_t* frames, int64_t frame_stride, int frame_bytes, int n,
                    const uint32_t* seq_base, uint32_t seq_off, int64_t timeout_ticks, int* err, float* rew, float* done,
                    uint8_t* frames_out, int64_t out_stride, a2c_stream_t stream) {
  if (n < 0 || !rec || !frames || !seq_base || !err || !rew || !done || !frames_out) return A2C_ERR_ARG;
  if (frame_bytes < 16 || frame_bytes % 16 || frame_stride % 16 || out_stride % 16 || frame_stride < frame_bytes ||
      out_stride < frame_bytes || timeout_ticks < 1 || ((uintptr_t)frames % 16) || ((uintptr_t)frames_out % 16))
    return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL((pool_ingest_kernel<false, false>), dim3(n), dim3(256), 0, a2c_s(stream), (const unsigned long long*)rec, frames,
                     (long)frame_stride, frame_bytes, seq_base, seq_off, (long)timeout_ticks, err, rew, done, frames_out,
                     (long)out_stride, IngestPost{});
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_pool_ingest_bits(const uint64_t* rec, const uint8_t* frames, int64_t frame_stride, int n_pixels, int n,
                         const uint32_t* seq_base, uint32_t seq_off, int64_t timeout_ticks, int* err, float* rew, float* done,
                         uint8_t* frames_out, int64_t out_stride, a2c_stream_t stream) {
  if (n < 0 || !rec || !frames || !seq_base || !err || !rew || !done || !frames_out) return A2C_ERR_ARG;
  if (n_pixels < 16 || n_pixels % 16 || frame_stride % 16 || out_stride % 16 || frame_stride * 8 < n_pixels ||
      out_stride < n_pixels || timeout_ticks < 1 || ((uintptr_t)frames % 16) || ((uintptr_t)frames_out % 16))
    return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL((pool_ingest_kernel<true, false>), dim3(n), dim3(256), 0, a2c_s(stream), (const unsigned long long*)rec, frames,
                     (long)frame_stride, n_pixels / 8, seq_base, seq_off, (long)timeout_ticks, err, rew, done, frames_out,
                     (long)out_stride, IngestPost{});
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_pool_ingest_post(int packed_bits, const uint64_t* rec, const uint8_t* frames, int64_t frame_stride, int frame_elems, int n,
                         const uint32_t* seq_base, uint32_t seq_off, int64_t timeout_ticks, int* err, float* rew, float* done,
                         uint8_t* frames_out, int64_t out_stride, const float* val, int64_t val_stride, float* val_prev,
                         float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma, int pong,
                         float* done_eff, float* h, int hdim, float* h_rows, int64_t h_rows_stride, const float* h_src,
                         int* nvalid_rows, int* nvalid_carry, a2c_stream_t stream) {
  if (n < 0 || !rec || !frames || !seq_base || !err || !rew || !done || !frames_out) return A2C_ERR_ARG;
  if (T < 1 || t < 0 || t >= T || !val || !val_prev || !rewards || !dones || !deltas || !nvalid_rows || !nvalid_carry) return A2C_ERR_ARG;
  if (h && hdim < 1) return A2C_ERR_ARG;
  if (frame_stride % 16 || out_stride % 16 || out_stride < frame_elems || timeout_ticks < 1 || ((uintptr_t)frames % 16) ||
      ((uintptr_t)frames_out % 16) || frame_elems < 16 || frame_elems % 16)
    return A2C_ERR_ARG;
  if (packed_bits ? frame_stride * 8 < frame_elems : frame_stride < frame_elems) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  const IngestPost q{val, (long)val_stride, val_prev, rewards, dones, deltas, (long)T, (long)t, (long)slot0, gamma, pong,
                     done_eff, h, hdim, h_rows, (long)h_rows_stride, h_src ? h_src : h, nvalid_rows, nvalid_carry};
  if (packed_bits)
    hipLaunchKernelGGL((pool_ingest_kernel<true, true>), dim3(n), dim3(256), 0, a2c_s(stream), (const unsigned long long*)rec, frames,
                       (long)frame_stride, frame_elems / 8, seq_base, seq_off, (long)timeout_ticks, err, rew, done, frames_out,
                       (long)out_stride, q);
  else
    hipLaunchKernelGGL((pool_ingest_kernel<false, true>), dim3(n), dim3(256), 0, a2c_s(stream), (const unsigned long long*)rec, frames,
                       (long)frame_stride, frame_elems, seq_base, seq_off, (long)timeout_ticks, err, rew, done, frames_out,
                       (long)out_stride, q);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_store_u32_system(uint32_t* dev_ptr, uint32_t value, a2c_stream_t stream) {
  if (!dev_ptr || ((uintptr_t)dev_ptr % 4)) return A2C_ERR_ARG;
  hipLaunchKernelGGL(store_u32_system_kernel, dim3(1), dim3(64), 0, a2c_s(stream), (unsigned int*)dev_ptr, (unsigned int)value);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_unpack_bits(const uint8_t* src, int64_t src_stride, uint8_t* dst, int64_t dst_stride, int n, int n_pixels,
                    a2c_stream_t stream) {
  if (n < 0 || !src || !dst || n_pixels < 16 || n_pixels % 16 || src_stride % 2 || src_stride * 8 < n_pixels ||
      dst_stride % 16 || dst_stride < n_pixels || ((uintptr_t)src % 2) || ((uintptr_t)dst % 16))
    return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL(unpack_bits_kernel, dim3((n_pixels / 16 + 255) / 256, n), dim3(256), 0, a2c_s(stream), src,
                     (long)src_stride, dst, (long)dst_stride, n_pixels);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
int a2c_frame_prep_u8(const uint8_t* raw, int64_t raw_stride, int H, int W, int C, int y0, int y1, int x0, int x1, int step,
                      int binarise, uint8_t* out, int64_t out_stride, int n, a2c_stream_t stream) {
  if (n < 0 || H < 1 || W < 1 || C < 1 || step < 1 || y0 < 0 || y1 > H || y0 >= y1 || x0 < 0 || x1 > W || x0 >= x1) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  const int OH = (y1 - y0 + step - 1) / step, OW = (x1 - x0 + step - 1) / step;
  if (!raw || !out || OW % 4 || ((uintptr_t)out % 4) || out_stride % 4 || out_stride < (int64_t)OH * OW ||
      raw_stride < (int64_t)H * W * C)
    return A2C_ERR_ARG;
  hipLaunchKernelGGL(frame_prep_kernel, dim3((unsigned)((OH * OW / 4 + 255) / 256), (unsigned)n), dim3(256), 0, a2c_s(stream), raw,
                     (long)raw_stride, W, C, y0, x0, step, OH, OW, binarise, out, (long)out_stride);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_frames_to_states(const uint8_t* frame_store, int64_t slot_stride, const int* nvalid, int64_t nvalid_slot_stride,
                         float* out, int64_t out_slot_stride, int R, int nt, int C, int HW, a2c_stream_t stream) {
  if (R < 0 || nt < 0 || C < 1 || HW < 4 || HW % 4) return A2C_ERR_ARG;
  if (R == 0 || nt == 0) return A2C_OK;
  if (!frame_store || !out || ((uintptr_t)frame_store % 4) || slot_stride % 4 || ((uintptr_t)out % 16) || out_slot_stride % 4 ||
      (long)R * nt > 65535L * 8)
    return A2C_ERR_ARG;
  // gridDim.y <= 65535: slots in batches
  const int per = 65535 / nt;
  if (per < 1) return A2C_ERR_ARG;
  for (int r0 = 0; r0 < R; r0 += per) {
    const int nr = R - r0 < per ? R - r0 : per;
    hipLaunchKernelGGL(frames_to_states_kernel, dim3((unsigned)((HW / 4 + 255) / 256), (unsigned)(nr * nt), (unsigned)C), dim3(256), 0,
                       a2c_s(stream), frame_store + (long)r0 * slot_stride, (long)slot_stride,
                       nvalid ? nvalid + (long)r0 * nvalid_slot_stride : nullptr, (long)nvalid_slot_stride,
                       out + (long)r0 * out_slot_stride, (long)out_slot_stride, nt, C, HW);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

int a2c_frame_store_begin(uint8_t* frame_store, int64_t slot_stride, int64_t T, int C, int HW, int* nvalid_rows,
                          const int* nvalid_carry, int B, a2c_stream_t stream) {
  if (B < 0 || T < C || C < 1 || HW % 4 || slot_stride < (T + C) * (int64_t)HW || slot_stride % 4) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!frame_store || !nvalid_rows || !nvalid_carry || ((uintptr_t)frame_store % 4)) return A2C_ERR_ARG;
  hipLaunchKernelGGL(frame_store_begin_kernel, dim3((unsigned)((C * HW / 4 + 255) / 256), (unsigned)B), dim3(256), 0, a2c_s(stream),
                     frame_store, (long)slot_stride, (long)T, C, HW, nvalid_rows, nvalid_carry);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_rollout_post_frames(const float* rew, const float* done, const float* val, int64_t val_stride, float* val_prev,
                            float* rewards, float* dones, float* deltas, int64_t T, int64_t t, int64_t slot0, float gamma,
                            int pong, int B, float* done_eff, float* h, int hdim, float* h_rows, int64_t h_rows_stride,
                            const float* h_src, int* nvalid_rows, int* nvalid_carry, a2c_stream_t stream) {
  if (B < 0 || T < 1 || t < 0 || t >= T) return A2C_ERR_ARG;
  if (B == 0) return A2C_OK;
  if (!rew || !done || !val || !val_prev || !rewards || !dones || !deltas || !nvalid_rows || !nvalid_carry) return A2C_ERR_ARG;
  if (h && hdim < 1) return A2C_ERR_ARG;
  // post_kernel with no planes to stack (C = 0): z = 0 is the bookkeeping layer, z = 1 the recurrent nets' hidden rows
  const dim3 grid(1, (unsigned)(h ? B : (B + 255) / 256), h ? 2u : 1u);
  hipLaunchKernelGGL(post_kernel, grid, dim3(256), 0, a2c_s(stream), rew, done, val, (long)val_stride, val_prev, rewards, dones,
                     deltas, (long)T, (long)t, (long)slot0, gamma, pong, (const float*)nullptr, (const uint8_t*)nullptr, 0L,
                     (const float*)nullptr, (const float*)nullptr, 0L, (float*)nullptr, 0L, B, 0, 0, done_eff, h, hdim, h_rows,
                     (long)h_rows_stride, h_src ? h_src : h, nvalid_rows, nvalid_carry);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
