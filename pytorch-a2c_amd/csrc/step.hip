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
//
// Three instantiation axes of the same kernel body:
//   U8       the new frame arrives as uint8 pixels (the host pool's transport format) and is expanded
//            in registers: one 16-byte load per thread brings the whole plane;
//   PERSIST  the WHOLE slot (T+1 iterations) in one launch (a2c_a3c_rollout): each workgroup loops
//            over the time steps of its env(s), waits for the env worker's `rec` granule in pinned
//            host memory, loads the uint8 frame straight from the pinned pool slot (system-scope loads
//            over PCIe) and hands the sampled action back with one 8-byte system-scope store to `cmd`
//            (include/a2c_hostpool.h) -- no kernel boundary and no host code between env steps.
#include <stdlib.h>
#include "a2c_common.h"

namespace {
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NT = 512;        // threads per workgroup (8 waves, 2 per SIMD)
constexpr int NCH = 3;         // the state arrives in 3 row chunks; conv1 runs on chunk k while k+1.. are in flight
constexpr int SL0 = 6, SL1 = 5, SL2 = 5;   // float4 staging registers per thread and chunk
constexpr int HN = 8;          // max heads (n_actions + 1)
constexpr int NF1 = 64 * 64;   // conv1 fragments (64 steps x 64 lanes)
constexpr int NF2 = 64 * 2 * 64;
// ring kernel, conv1 on the bf16 matrix pipe: 8 K-blocks (plane, tap-row quad) x 3 weight pieces x 64 lanes x 8 bf16 (16 B)
constexpr int NF1B = 8 * 3 * 64 * 4;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifdef A2C_STEP_TIMING
__device__ unsigned long long a2c_step_ts[16];
__device__ int a2c_step_skip;       // debug: bit 0 no row stores, 1 no conv1, 2 no conv2, 3 no heads, 4 no state loads
#define TS(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) a2c_step_ts[i] = wall_clock64(); } while (0)
#define SKIP(bit) (a2c_step_skip & (1 << (bit)))
#else
#define TS(i)
#define SKIP(bit) false
#endif

struct RolloutX {              // what a2c_a3c_rollout adds to the per-step arguments
  float* states;
  float* bookmark;
  const float* u; long u_stride;
  int64_t* actions;
  unsigned long long* cmd;
  const unsigned long long* rec;
  unsigned int seq0;
  int env0;
  int* err;
  long timeout_ticks;
  float* a1_rows;
  float* a2_rows;
  unsigned long long* a1_lm;                       // ring kernel: lane masks of the a1 stash rows (a2c_conv2d_bwd_data_lanemask), or nullptr
  unsigned char* a2_mb;                            // ring kernel: mask bits of the a2 stash rows (a2c_small_n_bwd_data_bits), or nullptr
  float* heads_rows; long heads_rows_ld;
  unsigned char* fstore; long fs_slot_stride;      // single-frame uint8 store (T+4 frames per slot)
  int* nvalid; int* nvalid_carry;
  int states_lazy;                                 // ring kernel: keep the frame store only, no fp32 state rows (the bookmark stays)
  const unsigned char* tagged; long tagged_stride; int tagged_nd;   // self-validating frame mirror (nd data chunks + 1 record chunk)
  int frame_bits;                                  // the pool publishes one bit per pixel (A2C_FRAME_BITS)
  unsigned long long* dbg;                         // phase stamps of workgroup 0 (a2c_debug_ring_timing), or nullptr
  int poll_gap;                                    // 64-cycle sleeps between two polls of the rec granule
  int early_poll;                                  // ring kernel: first poll behind plane 0 (A2C_RING_EARLY=0: behind plane 1 only)
  int bstride, boff;                               // ring kernel: workgroup i plays env b = i * bstride + boff (blocks of a larger launch)
};

struct StepP {
  a2c_a3c_step_args a;
  int OH1, OW1, OH2, OW2, PLANE1, PLANE2, F, F4;
  int row_end[NCH];            // chunk k = input rows [row_end[k-1], row_end[k])
  int tile_end[NCH];           // conv1 16-pixel tiles computable once chunk k is in LDS
  RolloutX x;
};

// what one iteration works on (uniform over the workgroup)
struct It {
  const float* prev;           // state row the frame is pushed onto (or, without a frame, the state itself)
  const float* frame32;        // fp32 new frame of this env, or nullptr
  bool frame8;                 // the new frame is the uint8 one (loaded separately)
  float* out;                  // row that receives the state (nullptr: none)
  float* a1o;                  // stash rows for the conv activations of this state (nullptr: none)
  float* a2o;
  float* ho;                   // stash row for [logits | value]
};

// plane c of state_t comes from: the new frame (c == 3) / plane c+1 of the previous state (frame
// stack, utils.py:26-43) or, without a new frame, plane c of `prev` itself.  Rows of a reset env
// are read anyway (valid memory) and zeroed at commit time, so no load waits on the reset flag.
// Every thread executes the SAME number of global loads/stores (out-of-range slots are clamped onto
// the chunk's last element: a benign duplicate), so the s_waitcnt vmcnt(n) the compiler derives for
// "chunk k has landed" is exact and never drains the younger loads and the row stores behind it.
// With a uint8 frame the plane-3 slots re-load plane 2's bytes (a cache hit) and are dropped at commit.
template <int N>
__device__ __forceinline__ void issue_chunk(float4 (&pf)[N], const It& it, int r0, int q4, int HW, int W, int tid) {
  const bool has_frame = it.frame32 != nullptr || it.frame8;
  const float* __restrict__ pb = it.prev + (has_frame ? HW : 0);
  const float* __restrict__ p3 = it.frame32 ? it.frame32 : pb + (it.frame8 ? 2L : 3L) * HW;
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int idx = min(tid + u * NT, 4 * q4 - 1);
    const int c = (idx >= q4) + (idx >= 2 * q4) + (idx >= 3 * q4);
    const int rem = idx - c * q4;
    const long off = (c == 3) ? (long)(p3 - pb) : (long)c * HW;       // uniform select, no branch
    // streaming load: every state row is read once per launch
    const f32x4 t_ = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pb + off + r0 * W + (rem << 2)));
    pf[u] = SKIP(4) ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(t_[0], t_[1], t_[2], t_[3]);
  }
}

template <int N>
__device__ __forceinline__ void commit_chunk(const float4 (&pf)[N], const It& it, int r0, int q4, int W, int tid,
                                             bool zero_old, float* __restrict__ img, int PLANE1) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int idx = min(tid + u * NT, 4 * q4 - 1);
    const int c = (idx >= q4) + (idx >= 2 * q4) + (idx >= 3 * q4);
    const int rem = idx - c * q4;
    float4 v = pf[u];
    if (zero_old && c != 3) v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!(it.frame8 && c == 3)) *reinterpret_cast<float4*>(img + c * PLANE1 + r0 * W + (rem << 2)) = v;
  }
}

// the same chunk to the rollout buffer row (states[e] / bookmark).  Issued one matrix phase AFTER
// the chunk was committed to LDS: at the start of the kernel HBM belongs to the state loads (the
// critical path); the row stores drain under the conv phases instead of competing with them.
template <int N>
__device__ __forceinline__ void store_chunk(const float4 (&pf)[N], const It& it, int r0, int q4, int HW, int W, int tid,
                                            bool zero_old) {
#pragma unroll
  for (int u = 0; u < N; ++u) {
    const int idx = min(tid + u * NT, 4 * q4 - 1);
    const int c = (idx >= q4) + (idx >= 2 * q4) + (idx >= 3 * q4);
    const int rem = idx - c * q4;
    float4 v = pf[u];
    if (zero_old && c != 3) v = make_float4(0.f, 0.f, 0.f, 0.f);
    // streaming store: the rows are not re-read by this kernel, and a write-back line left dirty in
    // the XCD's L2 would have to be flushed at the kernel boundary (serialised after the last wave)
    if (!SKIP(0) && !(it.frame8 && c == 3))
      __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w},
                                  reinterpret_cast<f32x4*>(it.out + (long)c * HW + r0 * W + (rem << 2)));
  }
}

__device__ __forceinline__ float4 u8x4(unsigned int w) {
  return make_float4((float)(w & 0xffu), (float)((w >> 8) & 0xffu), (float)((w >> 16) & 0xffu), (float)(w >> 24));
}

// [bf16(a) | bf16(b) << 16] of two floats with at most 8 significant bits (uint8 pixels): the upper halves, exact
__device__ __forceinline__ unsigned int bf16_pair_exact(float a, float b) {
  return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// eight uint8 pixels (two dwords) -> eight bf16, element j = byte j: the B operand fragment of v_mfma_f32_16x16x32_bf16
__device__ __forceinline__ bf16x8 u8x8_bf16(unsigned int lo, unsigned int hi) {
  u32x4 r;
  r[0] = bf16_pair_exact((float)(lo & 0xffu), (float)((lo >> 8) & 0xffu));
  r[1] = bf16_pair_exact((float)((lo >> 16) & 0xffu), (float)(lo >> 24));
  r[2] = bf16_pair_exact((float)(hi & 0xffu), (float)((hi >> 8) & 0xffu));
  r[3] = bf16_pair_exact((float)((hi >> 16) & 0xffu), (float)(hi >> 24));
  return __builtin_bit_cast(bf16x8, r);
}
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

template <bool OUT, bool U8, bool PERSIST, int HNT = HN>      // HNT: heads kept in registers (n_actions + 1 <= HNT)
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void a3c_step_kernel(StepP p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (SKIP(6)) return;
  const a2c_a3c_step_args& a = p.a;
  float* __restrict__ fr1 = lds;                     // conv1 fragments
  float* __restrict__ img = fr1 + NF1;               // state; later conv2 fragments | a2
  float* __restrict__ a1 = img + 4 * p.PLANE1;
  float* __restrict__ red = a1 + 16 * p.PLANE2;
  float* __restrict__ a2 = img + NF2;
  const int tid = threadIdx.x, lane = tid & 63;       // (the loop body works on its own copies, see there)
  const int g = lane >> 4;
  const int HW = a.H * a.W, W = a.W;
  const int N = a.n_actions + 1;
  const bool tail = tid == NT - 64;                  // lane 0 of the last wave (fewest conv tiles): bookkeeping + sampling
  // One batch of kernarg loads: left alone the compiler fetches the ~290 B argument block piecemeal,
  // in six dependent s_load -> s_waitcnt rounds before the first state load can issue.
  asm volatile("" ::"s"(a.prev), "s"(a.prev_stride), "s"(a.frame_new), "s"(a.reset_mask), "s"(a.out), "s"(a.out_stride),
               "s"(a.wfrag1), "s"(a.bias1), "s"(a.bias2), "s"(a.heads), "s"(a.ldh), "s"(a.rew), "s"(a.done),
               "s"(a.val_prev), "s"(a.rewards), "s"(a.dones), "s"(a.T), "s"(a.t_rec), "s"(a.slot0), "s"(a.H), "s"(a.W),
               "s"(a.n_actions), "s"(p.PLANE1), "s"(p.row_end[0]), "s"(p.row_end[1]), "s"(p.row_end[2]));
  TS(0);

  // ---- once per launch: conv1 fragments and the biases
  const float4 b1v = *reinterpret_cast<const float4*>(a.bias1 + 4 * g);
  const float b1[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
  const float ld_b2 = a.bias2[tid & 31];
  const float4* __restrict__ wf1v = reinterpret_cast<const float4*>(a.wfrag1);
  const int q0 = (p.row_end[0] * W) >> 2, q1 = ((p.row_end[1] - p.row_end[0]) * W) >> 2, q2 = ((p.row_end[2] - p.row_end[1]) * W) >> 2;
  const int NP1 = p.OH1 * p.OW1;
  const float* __restrict__ la1 = fr1 + lane;
  if (PERSIST) {
    float4* __restrict__ fv = reinterpret_cast<float4*>(fr1);
    fv[tid] = wf1v[tid]; fv[tid + NT] = wf1v[tid + NT];
    if (tid < 32) red[HN + tid] = ld_b2;
    if (tid == 0) reinterpret_cast<unsigned int*>(red)[HN + 37] = 0u;      // "the env worker timed out" flag of the polls below
  }

  // PERSIST: this workgroup plays envs blockIdx.x, blockIdx.x + gridDim.x, ... in turn, time step by time step
  const int nb = PERSIST ? (a.B - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 1;
  const int n_it = PERSIST ? ((int)a.T + 1) * nb : 1;
  const long S = 4L * HW;
  for (int iter = 0; iter < n_it; ++iter) {
    const int t = PERSIST ? iter / nb : 0;
    const int b = PERSIST ? (int)blockIdx.x + (iter - t * nb) * (int)gridDim.x : (int)blockIdx.x;
    // The persistent body with more than 4 heads sits at the 256-register cap: the compiler hoists a dozen LDS addresses
    // derived from the thread index out of this loop and then spills them (60 B/lane of scratch).  Behind an opaque copy
    // of the index they are three VALU instructions per use and stay in the loop.
    int tid_op = tid;
    if (PERSIST && HNT > 4) asm volatile("" : "+v"(tid_op));      // (4 heads: fits without; measured the same either way)
    const int tid = tid_op, lane = tid & 63, w = tid >> 6, g = lane >> 4, j = lane & 15;
    // ---- what this iteration is (uniform)
    It it;
    bool rec, boot, sample;
    long t_rec;
    const float* u_ptr;
    int64_t* act_ptr;
    if (PERSIST) {
      const long row = (a.slot0 + b) * a.T;
      it.prev = t == 0 ? p.x.bookmark + (long)b * S : p.x.states + (row + t - 1) * S;
      it.frame32 = nullptr;
      it.frame8 = t > 0;
      it.out = t == (int)a.T ? p.x.bookmark + (long)b * S : p.x.states + (row + t) * S;
      rec = t > 0; t_rec = t - 1; boot = t == (int)a.T; sample = t < (int)a.T;
      u_ptr = p.x.u + (long)t * p.x.u_stride + b;
      act_ptr = p.x.actions + row + t;
      it.a1o = (p.x.a1_rows && t < (int)a.T) ? p.x.a1_rows + (row + t) * (16L * NP1) : nullptr;
      it.a2o = (p.x.a2_rows && t < (int)a.T) ? p.x.a2_rows + (row + t) * (long)p.F : nullptr;
      it.ho = (p.x.heads_rows && t < (int)a.T) ? p.x.heads_rows + (row + t) * p.x.heads_rows_ld : nullptr;
    } else {
      it.prev = a.prev + (long)b * a.prev_stride;
      it.frame32 = a.frame_new ? a.frame_new + (long)b * HW : nullptr;
      it.frame8 = U8 && a.frame_u8 != nullptr;
      it.out = OUT ? a.out + (long)b * a.out_stride : nullptr;
      rec = a.rew != nullptr; t_rec = a.t_rec; boot = a.bootstrap != 0; sample = a.u != nullptr;
      u_ptr = a.u + b;
      act_ptr = a.actions + (long)b * a.act_stride;
      it.a1o = a.a1_out ? a.a1_out + (long)b * a.a1_stride : nullptr;
      it.a2o = a.a2_out ? a.a2_out + (long)b * a.a2_stride : nullptr;
      it.ho = a.heads_out ? a.heads_out + (long)b * a.heads_out_stride : nullptr;
    }
    const bool has_frame = it.frame32 != nullptr || it.frame8;

    // ---- every global load of the iteration is issued here, in the order it is consumed.
    // Bookkeeping inputs first (oldest in the in-order vmcnt queue), from always-valid addresses.
    const long bk_e = rec ? (a.slot0 + b) * a.T + t_rec : 0;
    const float* __restrict__ safe = a.heads + (long)b * a.ldh;
    float ld_r, ld_d, ld_v, ld_pr, ld_pd, ld_vp, ld_rst;
    int ld_nvc = 4;
    unsigned char* __restrict__ fs_slot = nullptr;
    float4 f1a, f1b;
    u32x4 f8 = (u32x4){0u, 0u, 0u, 0u};
    float4 pf0[SL0], pf1[SL1], pf2[SL2];
    if (PERSIST) {
      // values this workgroup stored in its previous iteration (same cache lines, rewritten every step):
      // agent-scope loads are served by L2, never by a stale line of this CU's vector L1
      ld_v = __hip_atomic_load(safe + a.n_actions, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ld_pr = __hip_atomic_load((rec && t_rec > 0) ? a.rewards + bk_e - 1 : safe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ld_pd = __hip_atomic_load((rec && t_rec > 0) ? a.dones + bk_e - 1 : safe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ld_vp = __hip_atomic_load(rec ? a.val_prev + b : safe, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (p.x.fstore != nullptr) {
        fs_slot = p.x.fstore + (a.slot0 + b) * p.x.fs_slot_stride;
        ld_nvc = __hip_atomic_load(p.x.nvalid_carry + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) {      // the slot starts in the state the previous one ended in: frames[0..3] = frames[T..T+3]
          const int g16 = (4 * HW) >> 4;
          for (int q = tid; q < g16; q += NT)
            reinterpret_cast<u32x4*>(fs_slot)[q] = reinterpret_cast<const u32x4*>(fs_slot + a.T * (long)HW)[q];
        }
      }
      // the 3 planes the state shares with its predecessor do not depend on the host: in flight during the wait
      issue_chunk(pf0, it, 0, q0, HW, W, tid);
      issue_chunk(pf1, it, p.row_end[0], q1, HW, W, tid);
      ld_r = 0.f; ld_d = 0.f; ld_rst = 0.f;
      if (t > 0) {
        // wait for the env worker: rec granule = ((seq << 1 | done) << 32) | float_bits(reward), frame written before it
        if (tid == 0) {
          const unsigned int want = (p.x.seq0 + (unsigned int)t) & 0x7fffffffu;     // rec carries seq modulo 2^31
          const unsigned long long t0 = wall_clock64();
          unsigned long long gr;
          for (;;) {
            gr = __hip_atomic_load(p.x.rec + p.x.env0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned int)(gr >> 33) == want) break;
            if ((long)(wall_clock64() - t0) > p.x.timeout_ticks) {
              __hip_atomic_store(p.x.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              reinterpret_cast<unsigned int*>(red)[HN + 37] = 1u;      // its own flag: every granule value is a legal answer
              break;
            }
            __builtin_amdgcn_s_sleep(16);
          }
          reinterpret_cast<unsigned int*>(red)[HN + 32] = (unsigned int)gr;
          reinterpret_cast<unsigned int*>(red)[HN + 33] = (unsigned int)(gr >> 32);
        }
        __syncthreads();
        const unsigned int g_lo = reinterpret_cast<const unsigned int*>(red)[HN + 32];
        const unsigned int g_hi = reinterpret_cast<const unsigned int*>(red)[HN + 33];
        if (reinterpret_cast<const unsigned int*>(red)[HN + 37] != 0u) break;   // host timeout: the error flag is set, give up on this slot
        ld_r = __uint_as_float(g_lo);
        ld_d = (g_hi & 1u) ? 1.f : 0.f;
        ld_rst = ld_d;
        // the frame: the whole uint8 plane in one 16-byte system-scope load per thread, over PCIe
        if (p.x.frame_bits) {
          // packed transport: this thread's 16 pixels are 2 bytes of the slot (HW / 8 bytes per env cross PCIe)
          __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.frame_u8 + (long)(p.x.env0 + b) * a.frame_stride), 0, HW >> 3, 0x00020000);
          const unsigned int bits = __builtin_amdgcn_raw_buffer_load_b16(fr, tid * 2, 0, 1 | 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned int wq = bits >> (4 * q);
            f8[q] = (wq & 1u) | ((wq & 2u) << 7) | ((wq & 4u) << 14) | ((wq & 8u) << 21);
          }
        } else {
          __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.frame_u8 + (long)(p.x.env0 + b) * a.frame_stride), 0, HW, 0x00020000);
          f8 = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 1 | 16);   // sc0 sc1
        }
        // single-frame store: the newest frame of state t is frames[slot][t + 3]
        if (fs_slot != nullptr && tid * 16 < HW) *reinterpret_cast<u32x4*>(fs_slot + (long)(t + 3) * HW + tid * 16) = f8;
      }
    } else {
      ld_r = *(rec ? a.rew + b : safe);
      ld_d = *(rec ? a.done + b : safe);
      ld_v = safe[a.n_actions];                      // value of the state the env step left
      ld_pr = *((rec && t_rec > 0) ? a.rewards + bk_e - 1 : safe);
      ld_pd = *((rec && t_rec > 0) ? a.dones + bk_e - 1 : safe);
      ld_vp = *(rec ? a.val_prev + b : safe);
      ld_rst = *((has_frame && a.reset_mask) ? a.reset_mask + b : safe);
      f1a = SKIP(5) ? make_float4(0.f, 0.f, 0.f, 0.f) : wf1v[tid];
      f1b = SKIP(5) ? make_float4(0.f, 0.f, 0.f, 0.f) : wf1v[tid + NT];
      if (U8) {
        if (it.frame8) {
          __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(a.frame_u8 + (long)b * a.frame_stride), 0, HW, 0x00020000);
          f8 = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 0);
        }
      }
      issue_chunk(pf0, it, 0, q0, HW, W, tid);
      issue_chunk(pf1, it, p.row_end[0], q1, HW, W, tid);
    }
    const bool zero_old = has_frame && (PERSIST || a.reset_mask != nullptr) && ld_rst != 0.f;
    TS(1);

    // ---- conv1 (16 x OH1*OW1, K = 256), chunk by chunk as the state lands in LDS.  The loads of
    // chunk k+2 / the later weights are issued right before the matrix phase of chunk k, so the
    // vector-memory pipe works while the MFMAs run.
    if (!PERSIST) {
      float4* __restrict__ fv = reinterpret_cast<float4*>(fr1);
      fv[tid] = f1a; fv[tid + NT] = f1b;
      if (tid < 32) red[HN + tid] = ld_b2;             // conv2 bias for the K-half epilogue
    }
    float4 w2a = make_float4(0.f, 0.f, 0.f, 0.f), w2b = w2a, w2c = w2a, w2d = w2a;
    float4 wc[2][HNT];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      if (k == 0) {
        commit_chunk(pf0, it, 0, q0, W, tid, zero_old, img, p.PLANE1);
        if (U8 && it.frame8 && tid * 16 < HW) {         // plane 3 from the uint8 frame, all rows at once
          float4* __restrict__ d = reinterpret_cast<float4*>(img + 3 * p.PLANE1 + tid * 16);
          d[0] = u8x4(f8[0]); d[1] = u8x4(f8[1]); d[2] = u8x4(f8[2]); d[3] = u8x4(f8[3]);
        }
      } else if (k == 1) commit_chunk(pf1, it, p.row_end[0], q1, W, tid, zero_old, img, p.PLANE1);
      else commit_chunk(pf2, it, p.row_end[1], q2, W, tid, zero_old, img, p.PLANE1);
      __syncthreads();
      TS(2 + k);
      if (k == 0) issue_chunk(pf2, it, p.row_end[1], q2, HW, W, tid);
#ifdef A2C_STEP_EARLY_STORES
      if (OUT && k == 0) store_chunk(pf0, it, 0, q0, HW, W, tid, zero_old);
      if (OUT && k == 1) store_chunk(pf1, it, p.row_end[0], q1, HW, W, tid, zero_old);
      if (OUT && k == 2) store_chunk(pf2, it, p.row_end[1], q2, HW, W, tid, zero_old);
#else
      if (OUT && k == 1) {
        store_chunk(pf0, it, 0, q0, HW, W, tid, zero_old);
        if (U8 && it.frame8 && tid * 16 < HW && !SKIP(0)) {
          f32x4* __restrict__ d = reinterpret_cast<f32x4*>(it.out + 3L * HW + tid * 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = u8x4(f8[q]);
            __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, d + q);
          }
        }
      }
      if (OUT && k == 2) store_chunk(pf1, it, p.row_end[0], q1, HW, W, tid, zero_old);
#endif
      if (k == 1 && !SKIP(5)) {
        const float4* __restrict__ wf2v = reinterpret_cast<const float4*>(a.wfrag2);
        w2a = wf2v[tid]; w2b = wf2v[tid + NT]; w2c = wf2v[tid + 2 * NT]; w2d = wf2v[tid + 3 * NT];
      }
      for (int tile = (k ? p.tile_end[k - 1] : 0) + w; tile < p.tile_end[k] && !SKIP(1); tile += NT / 64) {
        const int idx = tile * 16 + j;
        const bool ok = idx < NP1;
        const int i = ok ? idx : 0;
        const int r = i / p.OW1, c = i - r * p.OW1;
        const float* __restrict__ l = img + r * 4 * W + c * 4 + g * p.PLANE1;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
          const float4 t0 = *reinterpret_cast<const float4*>(l + ky * W);
          const float4 t1 = *reinterpret_cast<const float4*>(l + ky * W + 4);
          float av[8];
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) av[kx] = la1[(ky * 8 + kx) * 64];
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], t0.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], t0.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], t0.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], t0.w, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[4], t1.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[5], t1.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[6], t1.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[7], t1.w, acc, 0, 0, 0);
        }
        if (ok) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) a1[(4 * g + rr) * p.PLANE2 + i] = fmaxf(acc[rr] + b1[rr], 0.f);
        }
      }
    }
    __syncthreads();                                   // img consumed, a1 complete
    TS(5);
#ifndef A2C_STEP_EARLY_STORES
    if (OUT) store_chunk(pf2, it, p.row_end[1], q2, HW, W, tid, zero_old);
#endif
    if (it.a1o != nullptr) {       // stash conv1's activations (16 x NP1, dense) for the update: LDS -> HBM, streaming
      const int n4 = NP1 >> 2;     // float4 per channel
      for (int q = tid; q < 16 * n4; q += NT) {
        const int ch = q / n4, o4 = q - ch * n4;
        const float4 v = *reinterpret_cast<const float4*>(a1 + ch * p.PLANE2 + (o4 << 2));
        __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(it.a1o) + q);
      }
    }
    // head weights for this thread's K slices: issued here (not with conv2's fragments before phase 2:
    // 16 more float4 per thread on the vector-memory pipe delayed that matrix phase), in flight during conv2
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int kk = (tid << 2) + q * (NT * 4);
#pragma unroll
      for (int n = 0; n < HNT; ++n)
        wc[q][n] = *reinterpret_cast<const float4*>(a.Wc + (long)min(n, N - 1) * p.F + min(kk, p.F - 4));
    }
    {
      float4* __restrict__ iv = reinterpret_cast<float4*>(img);
      iv[tid] = w2a; iv[tid + NT] = w2b; iv[tid + 2 * NT] = w2c; iv[tid + 3 * NT] = w2d;
    }
    __syncthreads();

    // ---- conv2: 32 x (OH2*OW2), K = 256 split in two halves so that the 4*ntile units spread
    // evenly over the 8 waves; unit = (16-pixel tile, 16-channel half m, K half kh) writes its raw
    // partial sums to LDS, the epilogue adds the halves in a fixed order (+ bias, ReLU).
    const int NP2 = p.OH2 * p.OW2, ntile2 = (NP2 + 15) >> 4;
    float* __restrict__ part = a2 + p.F4;              // [2][ntile2*2][256]
    {
      const float* __restrict__ la = img + lane;
      for (int unit = w; unit < ntile2 * 4 && !SKIP(2); unit += NT / 64) {
        const int kh = unit & 1, m = (unit >> 1) & 1, tile = unit >> 2;
        const int idx = tile * 16 + j;
        const int i = idx < NP2 ? idx : 0;
        const int r = i / p.OW2, c = i - r * p.OW2;
        const float* __restrict__ l = a1 + r * 2 * p.OW1 + c * 2 + g * p.PLANE2;
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c4 = kh * 2 + cc;
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
        *reinterpret_cast<float4*>(part + ((kh * ntile2 + tile) * 2 + m) * 256 + lane * 4) = (float4){acc[0], acc[1], acc[2], acc[3]};
      }
    }
    __syncthreads();
    {  // a2 flat (co, pixel) = relu((half0 + half1) + bias): thread = (channel, pixel lane), no division
      const int co = tid >> 4, pl = tid & 15;
      const float bias2 = red[HN + co];
      const int cslot = (co >> 4) * 256 + 16 * ((co & 15) >> 2) * 4 + (co & 3);
      for (int px = pl; px < NP2; px += 16) {
        const int slot = (px >> 4) * 512 + (px & 15) * 4 + cslot;
        a2[co * NP2 + px] = fmaxf((part[slot] + part[ntile2 * 512 + slot]) + bias2, 0.f);
      }
    }
    __syncthreads();
    if (it.a2o != nullptr) {       // stash conv2's activations (flat (c, y, x) = the proj_matrx input row)
      for (int q = tid; q < (p.F >> 2); q += NT) {
        const float4 v = *reinterpret_cast<const float4*>(a2 + (q << 2));
        __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(it.a2o) + q);
      }
    }
    TS(6);

    // ---- heads: N dot products of length F.  Per-thread partials are transposed through LDS so
    // that wave n reduces head n with ONE butterfly; summation order is fixed (deterministic).
    float* __restrict__ hp = part + ntile2 * 1024;      // [HN][NT]
    {
      float acc[HNT];
#pragma unroll
      for (int n = 0; n < HNT; ++n) acc[n] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = (tid << 2) + q * (NT * 4);
        if (k < p.F && !SKIP(3)) {
          const float4 x = *reinterpret_cast<const float4*>(a2 + k);
#pragma unroll
          for (int n = 0; n < HNT; ++n)
            if (n < N) acc[n] += x.x * wc[q][n].x + x.y * wc[q][n].y + x.z * wc[q][n].z + x.w * wc[q][n].w;
        }
      }
#pragma unroll
      for (int n = 0; n < HNT; ++n)
        if (n < N) hp[n * NT + tid] = acc[n];
    }
    __syncthreads();
    if (w < N) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) v += hp[w * NT + q * 64 + lane];
      v = wave_sum(v);
      if (lane == 0) red[w] = v;
    }
    __syncthreads();
    TS(7);
    if (tail) {
      // bookkeeping of the env step that produced this state (runner.py:212-232)
      const float bk_r = ld_r, bk_v = ld_v;
      float bk_d = ld_d != 0.f ? 1.f : 0.f;
      if (a.pong && bk_r != 0.f) bk_d = 1.f;
      if (rec) {
        a.rewards[bk_e] = bk_r;
        a.dones[bk_e] = bk_d;
        if (t_rec > 0) {
          const float gv = a.gamma * bk_v;
          a.deltas[bk_e - 1] = (ld_pr + gv * (1.f - ld_pd)) - ld_vp;
        }
        a.val_prev[b] = bk_v;
      }
      if (PERSIST && p.x.nvalid != nullptr) {     // how many planes of this state are real frames (not pre-reset zeros)
        const int nvs = t == 0 ? ld_nvc : (ld_rst != 0.f ? 1 : min(ld_nvc + 1, 4));
        if (t < (int)a.T) p.x.nvalid[(a.slot0 + b) * a.T + t] = nvs;
        p.x.nvalid_carry[b] = nvs;
      }
      float h[HNT], vboot = 0.f;
#pragma unroll
      for (int n = 0; n < HNT; ++n) {
        h[n] = 0.f;
        if (n < N) {
          h[n] = red[n] + a.bc[n];
          a.heads[(long)b * a.ldh + n] = h[n];
          if (it.ho != nullptr) it.ho[n] = h[n];
          if (n == a.n_actions) vboot = h[n];
        }
      }
      if (sample) {              // same maths as sample_kernel<true>: softmax, running fp32 cumsum, first >= u
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) mx = fmaxf(mx, h[n]);
        float den = 0.f;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) den += expf(h[n] - mx);
        const float ub = *u_ptr;
        float cs = 0.f;
        int pick = -1;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) {
            cs = cs + expf(h[n] - mx) / den;
            if (pick < 0 && cs >= ub) pick = n;
          }
        if (pick < 0) pick = a.n_actions - 1;          // fp32 cumsum short of u: the last action (see sample_kernel)
        *act_ptr = (int64_t)pick;
        if (PERSIST)     // hand the action to the env worker: one 8-byte granule {step number, action} in pinned host memory
          __hip_atomic_store(p.x.cmd + p.x.env0 + b,
                             ((unsigned long long)(p.x.seq0 + (unsigned int)t) << 32) | (unsigned int)pick,
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      if (boot && rec) {  // runner.py:236-245 on the step recorded above (e = slot*T + T-1)
        float r = bk_r;
        if (bk_d == 0.f) {
          r = r + a.gamma * vboot;
          a.rewards[bk_e] = r;
          a.dones[bk_e] = 1.f;
        }
        a.deltas[bk_e] = r - bk_v;
      }
    }
    TS(8);
    if (PERSIST) {
      // the next iteration re-reads what this one stored (state row, heads, records): every wave drains its
      // stores, then the workgroup meets; this barrier also frees the LDS regions the heads phase read
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The persistent rollout when every env has a workgroup of its own (B <= CU count): the "ring" kernel.
//
// What the per-step body above cannot do -- it is also the one-launch-per-step kernel, so every iteration starts from
// HBM -- and this one does:
//   * the state lives in LDS for the WHOLE slot as a ring of 4 uint8 planes (28 KB): a new state is the previous one
//     with its oldest plane replaced by the frame the env worker just published (utils.py:26-43).  No plane is ever
//     re-read from HBM (85 KB per env-step), the fp32 state row of the rollout buffer is written from the ring;
//   * conv1's K is ordered PLANE-major (plane p, tap row quad, tap column): the three planes a state shares with
//     its predecessor are 3/4 of the sum and do not depend on the host.  Their partial sums for state t+1 (48 of
//     the 64 MFMA steps per tile, accumulators kept in registers), the state-row / stash stores of state t and the
//     env worker's turn-around (cmd granule over PCIe -> env.step -> rec granule + frame back) all run CONCURRENTLY;
//     when the frame lands only the newest plane's 16 steps, conv2, the heads and the sampler are left;
//   * two 16-pixel tiles per wave run as two interleaved accumulator chains sharing the A operand
//     (v_mfma_f32_16x16x4_f32: 32-cycle issue, 40-cycle dependent latency), A fragments as ds_read_b128 of 4 steps;
//   * conv2's fragments (32 KB), the head weights and conv1's fragments are loaded ONCE per launch (LDS / registers),
//     and the values the bookkeeping carries from step to step stay in registers.
// An env reset (real done) zeroes the older planes of the ring and the partial sums (a sum over zero planes is 0).
// Results equal the per-step kernel's up to the fp32 re-association of conv1's sum (plane-major instead of tap-major).
//
// C1BF (round 5): conv1 on the BF16 matrix pipe with fp32 results.  Its input is a uint8 frame stack (pong_prep /
// breakout_prep outputs, preprocessing.py:8-23; runner.py:199 casts them to float): integers of at most 8 significant
// bits, i.e. EXACT as bf16.  Every fp32 weight is split once per launch into three bf16 pieces w = hi + mid + lo (round to
// nearest each; 3 x 8 significant bits cover the 24 of an fp32 exactly), so that w * x = hi*x + mid*x + lo*x with every
// product exact (8 x 8 bits) and all sums in the MFMA's fp32 accumulator: the same real-number sum as the fp32 FMA chain,
// re-associated (within an instruction's 32-deep dot product, and piece by piece).  One v_mfma_f32_16x16x32_bf16 takes the
// whole (plane, tap-row quad) block a lane's 8-byte window read covers -- 3 instructions of ~16 cycles where the fp32 form
// issues 8 of 32: conv1 is 1,600 of the step's 2,368 fp32 MFMAs.  C1BF = false keeps the fp32 form (A2C_RING_F32=1).
template <int HNT, bool C1BF>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void a3c_ring_kernel(StepP p, const float* __restrict__ w1raw) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const a2c_a3c_step_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int g = lane >> 4, j = lane & 15;
  const int HW = a.H * a.W, W = a.W;
  const int N = a.n_actions + 1;
  const int NP1 = p.OH1 * p.OW1, ntile1 = (NP1 + 15) >> 4;
  const int NP2 = p.OH2 * p.OW2, ntile2 = (NP2 + 15) >> 4;
  const bool tail = tid == NT - 64;
  float* __restrict__ fr1 = lds;                        // conv1 fragments, [step/4][lane][step%4]
  float* __restrict__ fr2 = fr1 + (C1BF ? NF1B : NF1);  // conv2 fragments (the per-step kernel's layout)
  float* __restrict__ a1 = fr2 + NF2;                   // 16 x PLANE2
  float* __restrict__ a2 = a1 + 16 * p.PLANE2;          // flat (c, y, x)
  float* __restrict__ part = a2 + p.F4;                 // [2][ntile2*2][256] conv2 K-half partials
  float* __restrict__ hp = part;                        // [HNT][NT] head partials: over the K-half partials (dead by then)
  float* __restrict__ red = part + max(ntile2 * 1024, HNT * NT);   // heads | conv2 bias | granule, flags
  // the state: a ring of FIVE uint8 plane slots.  State t owns slots base .. base+3 (mod 5); the fifth receives the
  // frame of state t+1 while state t's row is still being written out
  unsigned char* __restrict__ ring = reinterpret_cast<unsigned char*>(red + HN + 32 + 8);
  const int b = (int)blockIdx.x * p.x.bstride + p.x.boff;      // (B > CU count: block boff of bstride interleaved blocks)
  const long S = 4L * HW, row = (a.slot0 + b) * a.T;
  const int T = (int)a.T;

  // ---- once per launch
  if constexpr (C1BF) {
    // fragment (block = plane * 2 + kyq, piece, lane l): the eight kx of W1[co = l&15][plane][ky = 4*kyq + (l>>4)] as bf16
    for (int q = tid; q < 8 * 64; q += NT) {
      const int blk = q >> 6, l = q & 63, pl = blk >> 1, kyq = blk & 1;
      const float* __restrict__ wr = w1raw + (((l & 15) * 4 + pl) * 8 + 4 * kyq + (l >> 4)) * 8;
      unsigned int pk[3][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        unsigned short hb[3][2];
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          const float wv = wr[2 * e + h2];
          const __bf16 h0 = (__bf16)wv;
          const float r1 = wv - (float)h0;              // exact: the low 16 bits of the mantissa (sign folded in)
          const __bf16 h1 = (__bf16)r1;
          const float r2 = r1 - (float)h1;              // exact: at most 8 significant bits are left
          hb[0][h2] = __builtin_bit_cast(unsigned short, h0);
          hb[1][h2] = __builtin_bit_cast(unsigned short, h1);
          hb[2][h2] = bf16_bits(r2);
        }
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) pk[pc][e] = (unsigned int)hb[pc][0] | ((unsigned int)hb[pc][1] << 16);
      }
#pragma unroll
      for (int pc = 0; pc < 3; ++pc)
        reinterpret_cast<u32x4*>(fr1)[(blk * 3 + pc) * 64 + l] = (u32x4){pk[pc][0], pk[pc][1], pk[pc][2], pk[pc][3]};
    }
  } else {
    for (int q = tid; q < NF1; q += NT) {               // fragment element (step s, lane l) <- W1[co = l&15][p][ky = 4*kyq + (l>>4)][kx]
      const int sq = q >> 8, l = (q >> 2) & 63, s = sq * 4 + (q & 3);
      const int pl = s >> 4, kyq = (s >> 3) & 1, kx = s & 7;
      fr1[q] = w1raw[(((l & 15) * 4 + pl) * 8 + 4 * kyq + (l >> 4)) * 8 + kx];
    }
  }
  {
    const float4* __restrict__ wf2v = reinterpret_cast<const float4*>(a.wfrag2);
    float4* __restrict__ f2 = reinterpret_cast<float4*>(fr2);
#pragma unroll
    for (int q = 0; q < 4; ++q) f2[tid + q * NT] = wf2v[tid + q * NT];
  }
  if (tid < 32) red[HN + tid] = a.bias2[tid];
  if (tid == 0) reinterpret_cast<unsigned int*>(red)[HN + 37] = 0u;        // "the env worker timed out" flag of the polls below
  const float4 b1v = *reinterpret_cast<const float4*>(a.bias1 + 4 * g);
  const float b1[4] = {b1v.x, b1v.y, b1v.z, b1v.w};
  float4 wc[2][HNT];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int kk = (tid << 2) + q * (NT * 4);
#pragma unroll
    for (int n = 0; n < HNT; ++n)
      wc[q][n] = *reinterpret_cast<const float4*>(a.Wc + (long)min(n, N - 1) * p.F + min(kk, p.F - 4));
  }
  float bcv[HNT];
#pragma unroll
  for (int n = 0; n < HNT; ++n) bcv[n] = a.bc[min(n, N - 1)];
  // this wave's conv1 tiles: w, w+8, w+16, w+24 (16 pixels each); byte offset of pixel j's window row g in a plane
  int boff[4], pix[4];
  bool okp[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = (w + 8 * q) * 16 + j;
    okp[q] = (w + 8 * q) < ntile1 && idx < NP1;
    const int i = okp[q] ? idx : 0;
    const int r = i / p.OW1, c = i - r * p.OW1;
    pix[q] = i;
    boff[q] = (4 * r + g) * W + 4 * c;
  }
  const bool three = (w + 16) < ntile1, four = (w + 24) < ntile1;   // tiles beyond the first pair (wave-uniform)
  f32x4 acc[4];

  // steps of planes [p0, p1) of the state whose plane 0 sits in ring slot `base`, added to acc[]: per (plane, tap
  // row quad) ONE 8-byte LDS read per tile gives the 8 tap columns of this lane's window row, two ds_read_b128 the
  // A operands of the 8 steps; the two tiles of a pair are two independent accumulator chains
  auto conv1_planes = [&](int base, int p0, int p1) {
    for (int pl = p0; pl < p1; ++pl) {
      const unsigned char* __restrict__ plane = ring + ((base + pl) % 5) * HW;
#pragma unroll
      for (int kyq = 0; kyq < 2; ++kyq) {
        if constexpr (C1BF) {
          // one 8-byte window read per tile -> 8 bf16 = the lane's share of a 32-deep K block; hi / mid / lo weight pieces
          const bf16x8* __restrict__ fa = reinterpret_cast<const bf16x8*>(fr1) + ((pl * 2 + kyq) * 3) * 64 + lane;
          const bf16x8 wh = fa[0], wm = fa[64], wl = fa[128];
          {
            const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[0] + 4 * kyq * W);
            const unsigned int* __restrict__ q1 = reinterpret_cast<const unsigned int*>(plane + boff[1] + 4 * kyq * W);
            const bf16x8 bx = u8x8_bf16(q0[0], q0[1]), by = u8x8_bf16(q1[0], q1[1]);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, by, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, by, acc[1], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bx, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, by, acc[1], 0, 0, 0);
          }
          if (four) {
            const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[2] + 4 * kyq * W);
            const unsigned int* __restrict__ q1 = reinterpret_cast<const unsigned int*>(plane + boff[3] + 4 * kyq * W);
            const bf16x8 bx = u8x8_bf16(q0[0], q0[1]), by = u8x8_bf16(q1[0], q1[1]);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bx, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, by, acc[3], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bx, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, by, acc[3], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bx, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, by, acc[3], 0, 0, 0);
          } else if (three) {
            const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[2] + 4 * kyq * W);
            const bf16x8 bx = u8x8_bf16(q0[0], q0[1]);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bx, acc[2], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wm, bx, acc[2], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bx, acc[2], 0, 0, 0);
          }
        } else {
        const float4* __restrict__ fa = reinterpret_cast<const float4*>(fr1) + ((pl * 2 + kyq) * 2) * 64 + lane;
        const float4 av0 = fa[0], av1 = fa[64];
        const float av[8] = {av0.x, av0.y, av0.z, av0.w, av1.x, av1.y, av1.z, av1.w};
        {
          const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[0] + 4 * kyq * W);
          const unsigned int* __restrict__ q1 = reinterpret_cast<const unsigned int*>(plane + boff[1] + 4 * kyq * W);
          const float4 f0 = u8x4(q0[0]), f1 = u8x4(q0[1]), h0 = u8x4(q1[0]), h1 = u8x4(q1[1]);
          const float bx[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
          const float by[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], bx[kx], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], by[kx], acc[1], 0, 0, 0);
          }
        }
        if (four) {
          const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[2] + 4 * kyq * W);
          const unsigned int* __restrict__ q1 = reinterpret_cast<const unsigned int*>(plane + boff[3] + 4 * kyq * W);
          const float4 f0 = u8x4(q0[0]), f1 = u8x4(q0[1]), h0 = u8x4(q1[0]), h1 = u8x4(q1[1]);
          const float bx[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
          const float by[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) {
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], bx[kx], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], by[kx], acc[3], 0, 0, 0);
          }
        } else if (three) {
          const unsigned int* __restrict__ q0 = reinterpret_cast<const unsigned int*>(plane + boff[2] + 4 * kyq * W);
          const float4 f0 = u8x4(q0[0]), f1 = u8x4(q0[1]);
          const float bx[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kx], bx[kx], acc[2], 0, 0, 0);
        }
        }
      }
    }
  };

  // ---- state 0 = the bookmark (fp32 rows in HBM) -> ring slots 0..3
  {
    const float4* __restrict__ bm = reinterpret_cast<const float4*>(p.x.bookmark + (long)b * S);
    unsigned int* __restrict__ r32 = reinterpret_cast<unsigned int*>(ring);
    for (int q = tid; q < (int)(S >> 2); q += NT) {
      const float4 v = bm[q];
      r32[q] = (unsigned int)v.x | ((unsigned int)v.y << 8) | ((unsigned int)v.z << 16) | ((unsigned int)v.w << 24);
    }
  }
  __syncthreads();
  // single-frame uint8 store: frames[r][0..3] = the planes of state 0 (ring slots 0..3), then one
  // frame per env step; c_nv (tail lane) = how many planes of the current state are real frames (utils.py:37-42)
  unsigned char* __restrict__ fs_slot = p.x.fstore != nullptr ? p.x.fstore + (a.slot0 + b) * p.x.fs_slot_stride : nullptr;
  int c_nv = 4;
  if (fs_slot != nullptr) {
    for (int q = tid; q < (4 * HW) >> 4; q += NT) reinterpret_cast<u32x4*>(fs_slot)[q] = reinterpret_cast<const u32x4*>(ring)[q];
    if (tail) c_nv = __hip_atomic_load(p.x.nvalid_carry + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int base = 0;
  conv1_planes(base, 0, 3);
  // values the bookkeeping carries from iteration to iteration (tail lane only)
  float c_val = 0.f, c_rew = 0.f, c_done = 0.f, c_vprev = 0.f;
  // phase stamps (debug): wave 1's view of workgroup 0, summed over the iterations, 100 MHz ticks
  const bool stamp = p.x.dbg != nullptr && b == 0 && tid == 64;
  unsigned long long ts_prev = 0, ts_sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define RING_TS(i) do { if (stamp) { const unsigned long long n_ = wall_clock64(); ts_sum[i] += n_ - ts_prev; ts_prev = n_; } } while (0)
  if (stamp) ts_prev = wall_clock64();

  bool pre_ok = false;            // wave 1: granule and frame of the coming state are already in LDS (prefetched)
  // ---- self-validating mirror (a2c_hostpool.h): lane l < nd holds 112 packed pixels + tag, lane nd the record + tag
  const bool tg = p.x.tagged != nullptr;
  const int nd = p.x.tagged_nd;
  __amdgpu_buffer_rsrc_t tg_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(tg ? p.x.tagged + (long)(p.x.env0 + b) * p.x.tagged_stride : (const unsigned char*)p.x.rec), 0,
      tg ? 16 * (nd + 1) : 0, 0x00020000);
  auto tg_fetch = [&]() { return __builtin_amdgcn_raw_buffer_load_b128(tg_rsrc, min(lane, nd) * 16, 0, 1 | 16); };
  auto tg_ok = [&](const u32x4 c, unsigned int want) {        // every chunk carries the awaited step number's low 16 bits
    return __builtin_amdgcn_ballot_w64(lane > nd || (c[3] >> 16) == (want & 0xffffu)) == ~0ULL;
  };
  auto tg_unpack = [&](const u32x4 c, int slot) {             // chunks -> plane `slot` of the ring + the record in LDS
    if (lane < nd) {
      unsigned char* __restrict__ dst = ring + slot * HW + lane * 112;
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const unsigned int bits = c[q >> 1] >> (16 * (q & 1));
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int wq = bits >> (4 * e);
          v[e] = (wq & 1u) | ((wq & 2u) << 7) | ((wq & 4u) << 14) | ((wq & 8u) << 21);
        }
        if (lane * 112 + q * 16 < HW) *reinterpret_cast<u32x4*>(dst + q * 16) = v;
      }
    } else if (lane == nd) {                                  // {reward, done, seq}: the rec granule's two words
      reinterpret_cast<unsigned int*>(red)[HN + 32] = c[0];
      reinterpret_cast<unsigned int*>(red)[HN + 33] = (c[2] << 1) | (c[1] & 1u);
      reinterpret_cast<unsigned int*>(red)[HN + 36] = 1u;    // the frame is in the ring
    }
  };
  for (int t = 0; t <= T; ++t) {
    float ld_r = 0.f, ld_d = 0.f;
    // the sampler's uniform of this step: in flight from here, consumed by the tail lane behind the heads
    const float ub = (tail && t < T) ? p.x.u[(long)t * p.x.u_stride + b] : 0.f;
    if (t > 0) {
      // ---- the env worker's answer: rec granule = ((seq << 1 | done) << 32) | float_bits(reward), frame written before it.
      // Usually wave 1 fetched both during the previous iteration's off-critical-path phase (below); otherwise it
      // polls here and every thread then loads its part of the frame.
      if (w == 1 && !pre_ok && tg) {
        // the early fetch came too soon: re-fetch the chunks until every tag is the awaited step's
        const unsigned int want = p.x.seq0 + (unsigned int)t;
        const unsigned long long t0 = wall_clock64();
        u32x4 c;
        bool dead = false;
        for (;;) {
          c = tg_fetch();
          if (stamp) ts_sum[9] += 1;
          if (tg_ok(c, want)) break;
          if ((long)(wall_clock64() - t0) > p.x.timeout_ticks) {
            __hip_atomic_store(p.x.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane == 0) reinterpret_cast<unsigned int*>(red)[HN + 37] = 1u;
            dead = true;
            break;
          }
          for (int q = 0; q < p.x.poll_gap; ++q) __builtin_amdgcn_s_sleep(1);
        }
        if (!dead) tg_unpack(c, (base + 3) % 5);
      } else if (w == 1 && !pre_ok) {
        const unsigned int want = (p.x.seq0 + (unsigned int)t) & 0x7fffffffu;
        const unsigned long long t0 = wall_clock64();
        unsigned long long gr;
        for (;;) {
          gr = __hip_atomic_load(p.x.rec + p.x.env0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (stamp) ts_sum[9] += 1;
          if ((unsigned int)(gr >> 33) == want) break;
          if ((long)(wall_clock64() - t0) > p.x.timeout_ticks) {
            __hip_atomic_store(p.x.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (lane == 0) reinterpret_cast<unsigned int*>(red)[HN + 37] = 1u;   // its own flag: every granule value is a legal answer
            break;
          }
          for (int q = 0; q < p.x.poll_gap; ++q) __builtin_amdgcn_s_sleep(1);
        }
        if (lane == 0) {
          reinterpret_cast<unsigned int*>(red)[HN + 32] = (unsigned int)gr;
          reinterpret_cast<unsigned int*>(red)[HN + 33] = (unsigned int)(gr >> 32);
          reinterpret_cast<unsigned int*>(red)[HN + 36] = 0u;        // the frame is still to be loaded
        }
      }
      if (stamp) {                // turn-around: cmd store (tail lane's stamp, in LDS) -> answer in hand
        RING_TS(0);
        const unsigned long long tc = (unsigned long long)reinterpret_cast<const unsigned int*>(red)[HN + 34] |
                                      ((unsigned long long)reinterpret_cast<const unsigned int*>(red)[HN + 35] << 32);
        ts_sum[8] += ts_prev - tc;
      }
      __syncthreads();          // (also: every read of the slot the NEXT frame will go to has returned)
      RING_TS(1);
      const unsigned int g_lo = reinterpret_cast<const unsigned int*>(red)[HN + 32];
      const unsigned int g_hi = reinterpret_cast<const unsigned int*>(red)[HN + 33];
      const bool in_ring = reinterpret_cast<const unsigned int*>(red)[HN + 36] != 0u;
      if (reinterpret_cast<const unsigned int*>(red)[HN + 37] != 0u) break;   // host timeout: the error flag is set, give up on this slot
      ld_r = __uint_as_float(g_lo);
      ld_d = (g_hi & 1u) ? 1.f : 0.f;
      if (!in_ring) {
        // the frame: packed (2 bytes = this thread's 16 pixels) or uint8 (16 bytes), system-scope loads over PCIe
        u32x4 f8 = (u32x4){0u, 0u, 0u, 0u};
        if (p.x.frame_bits) {
          __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.frame_u8 + (long)(p.x.env0 + b) * a.frame_stride), 0, HW >> 3, 0x00020000);
          const unsigned int bits = __builtin_amdgcn_raw_buffer_load_b16(fr, tid * 2, 0, 1 | 16);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const unsigned int wq = bits >> (4 * q);
            f8[q] = (wq & 1u) | ((wq & 2u) << 7) | ((wq & 4u) << 14) | ((wq & 8u) << 21);
          }
        } else {
          __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(
              (void*)(a.frame_u8 + (long)(p.x.env0 + b) * a.frame_stride), 0, HW, 0x00020000);
          f8 = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 1 | 16);
        }
        // state t: planes 0..2 = planes 1..3 of state t-1 (in place: base advanced), plane 3 = the frame
        if (tid * 16 < HW) *reinterpret_cast<u32x4*>(ring + ((base + 3) % 5) * HW + tid * 16) = f8;
      }
      if (ld_d != 0.f) {        // real done: the frame stack restarts from zeros (utils.py:37-42)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int pl = 0; pl < 3; ++pl)
          if (tid * 16 < HW) *reinterpret_cast<u32x4*>(ring + ((base + pl) % 5) * HW + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
      }
      if (!in_ring) __syncthreads();
      RING_TS(2);                 // frame over PCIe into the ring (when it was not prefetched)
    }
    // ---- conv1: the newest plane's 16 steps on top of the partial sums, then bias + ReLU -> a1
    conv1_planes(base, 3, 4);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (okp[q]) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) a1[(4 * g + rr) * p.PLANE2 + pix[q]] = fmaxf(acc[q][rr] + b1[rr], 0.f);
      }
    __syncthreads();
    RING_TS(3);                   // conv1, newest plane
    // ---- conv2: 32 x (OH2*OW2), K = 256 split in two halves (see the per-step kernel)
    {
      const float* __restrict__ la = fr2 + lane;
      // units (tile, channel half m, K half kh) = w, w + 8, w + 16, ...: the units of a wave share kh and m, i.e. the A
      // fragments.  Three units at a time as three INDEPENDENT accumulator chains (a single chain issues one MFMA per
      // 40-cycle dependent latency instead of one per 32): fragments read once for the three, 48 MFMAs back to back
      // per channel quad.  Per output element the sum order is the one-unit loop's: bit-identical results.
      int unit = w;
      for (; unit + 16 < ntile2 * 4; unit += 24) {
        const int kh = unit & 1, m = (unit >> 1) & 1;
        const float* __restrict__ l3[3];
        f32x4 ac3[3];
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3) {
          const int idx = ((unit + 8 * i3) >> 2) * 16 + j;
          const int i = idx < NP2 ? idx : 0;
          const int r = i / p.OW2, c = i - r * p.OW2;
          l3[i3] = a1 + r * 2 * p.OW1 + c * 2 + g * p.PLANE2;
          ac3[i3] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c4 = kh * 2 + cc;
          float av[16], bv3[3][16];
#pragma unroll
          for (int u = 0; u < 16; ++u) av[u] = la[((c4 * 16 + u) * 2 + m) * 64];
#pragma unroll
          for (int i3 = 0; i3 < 3; ++i3)
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
              const int off = c4 * 4 * p.PLANE2 + ky * p.OW1;
              const float2 t0 = *reinterpret_cast<const float2*>(l3[i3] + off);
              const float2 t1 = *reinterpret_cast<const float2*>(l3[i3] + off + 2);
              bv3[i3][ky * 4 + 0] = t0.x; bv3[i3][ky * 4 + 1] = t0.y; bv3[i3][ky * 4 + 2] = t1.x; bv3[i3][ky * 4 + 3] = t1.y;
            }
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i3 = 0; i3 < 3; ++i3) ac3[i3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv3[i3][u], ac3[i3], 0, 0, 0);
        }
#pragma unroll
        for (int i3 = 0; i3 < 3; ++i3)
          *reinterpret_cast<float4*>(part + ((kh * ntile2 + ((unit + 8 * i3) >> 2)) * 2 + m) * 256 + lane * 4) =
              (float4){ac3[i3][0], ac3[i3][1], ac3[i3][2], ac3[i3][3]};
      }
      for (; unit < ntile2 * 4; unit += NT / 64) {
        const int kh = unit & 1, m = (unit >> 1) & 1, tile = unit >> 2;
        const int idx = tile * 16 + j;
        const int i = idx < NP2 ? idx : 0;
        const int r = i / p.OW2, c = i - r * p.OW2;
        const float* __restrict__ l = a1 + r * 2 * p.OW1 + c * 2 + g * p.PLANE2;
        f32x4 ac2 = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c4 = kh * 2 + cc;
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
          for (int u = 0; u < 16; ++u) ac2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[u], bv[u], ac2, 0, 0, 0);
        }
        *reinterpret_cast<float4*>(part + ((kh * ntile2 + tile) * 2 + m) * 256 + lane * 4) = (float4){ac2[0], ac2[1], ac2[2], ac2[3]};
      }
    }
    __syncthreads();
    {
      const int co = tid >> 4, pl = tid & 15;
      const float bias2 = red[HN + co];
      const int cslot = (co >> 4) * 256 + 16 * ((co & 15) >> 2) * 4 + (co & 3);
      for (int px = pl; px < NP2; px += 16) {
        const int slot = (px >> 4) * 512 + (px & 15) * 4 + cslot;
        a2[co * NP2 + px] = fmaxf((part[slot] + part[ntile2 * 512 + slot]) + bias2, 0.f);
      }
    }
    __syncthreads();
    RING_TS(4);                   // conv2 + epilogue
    // ---- heads (same summation order as the per-step kernel)
    {
      float hacc[HNT];
#pragma unroll
      for (int n = 0; n < HNT; ++n) hacc[n] = 0.f;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int k = (tid << 2) + q * (NT * 4);
        if (k < p.F) {
          const float4 x = *reinterpret_cast<const float4*>(a2 + k);
#pragma unroll
          for (int n = 0; n < HNT; ++n)
            if (n < N) hacc[n] += x.x * wc[q][n].x + x.y * wc[q][n].y + x.z * wc[q][n].z + x.w * wc[q][n].w;
        }
      }
#pragma unroll
      for (int n = 0; n < HNT; ++n)
        if (n < N) hp[n * NT + tid] = hacc[n];
    }
    __syncthreads();
    if (w < N) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < NT / 64; ++q) v += hp[w * NT + q * 64 + lane];
      v = wave_sum(v);
      if (lane == 0) red[w] = v;
    }
    __syncthreads();
    RING_TS(5);                   // heads
    if (tail) {
      // The action goes to the env worker FIRST: its turn-around is the critical path of the step, and this lane shares
      // its SIMD with a wave that is already computing the next state's partial sums (hence the priority).
      __builtin_amdgcn_s_setprio(3);
      float h[HNT], vboot = 0.f;
#pragma unroll
      for (int n = 0; n < HNT; ++n) {
        h[n] = 0.f;
        if (n < N) {
          h[n] = red[n] + bcv[n];
          if (n == a.n_actions) vboot = h[n];
        }
      }
      int pick = -1;
      if (t < T) {               // softmax + running fp32 cumsum, first >= u (utils.py:45-60)
        float mx = -INFINITY;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) mx = fmaxf(mx, h[n]);
        float den = 0.f;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) den += expf(h[n] - mx);
        float cs = 0.f;
#pragma unroll
        for (int n = 0; n < HNT; ++n)
          if (n < a.n_actions) {
            cs = cs + expf(h[n] - mx) / den;
            if (pick < 0 && cs >= ub) pick = n;
          }
        if (pick < 0) pick = a.n_actions - 1;
        __hip_atomic_store(p.x.cmd + p.x.env0 + b,
                           ((unsigned long long)(p.x.seq0 + (unsigned int)t) << 32) | (unsigned int)pick,
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (p.x.dbg != nullptr && b == 0) {
          const unsigned long long tc = wall_clock64();
          reinterpret_cast<unsigned int*>(red)[HN + 34] = (unsigned int)tc;
          reinterpret_cast<unsigned int*>(red)[HN + 35] = (unsigned int)(tc >> 32);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      // bookkeeping of the env step that produced this state (runner.py:212-232), like the per-step kernel's tail
      const bool rec = t > 0;
      const long bk_e = row + t - 1;
      const float bk_r = ld_r, bk_v = c_val;            // c_val: value of state t-1
      float bk_d = ld_d != 0.f ? 1.f : 0.f;
      if (a.pong && bk_r != 0.f) bk_d = 1.f;
      if (rec) {
        a.rewards[bk_e] = bk_r;
        a.dones[bk_e] = bk_d;
        if (t > 1) a.deltas[bk_e - 1] = (c_rew + (a.gamma * bk_v) * (1.f - c_done)) - c_vprev;
        c_vprev = bk_v;
        c_rew = bk_r;
        c_done = bk_d;
      }
      float* __restrict__ ho = (p.x.heads_rows && t < T) ? p.x.heads_rows + (row + t) * p.x.heads_rows_ld : nullptr;
#pragma unroll
      for (int n = 0; n < HNT; ++n)
        if (n < N) {
          a.heads[(long)b * a.ldh + n] = h[n];
          if (ho != nullptr) ho[n] = h[n];
        }
      c_val = vboot;
      if (fs_slot != nullptr) {  // valid planes of state t: 1 behind a real done, else one more than its predecessor's
        if (t > 0) c_nv = ld_d != 0.f ? 1 : min(c_nv + 1, 4);
        if (t < T) p.x.nvalid[row + t] = c_nv;
        else p.x.nvalid_carry[b] = c_nv;
      }
      if (t < T) {
        p.x.actions[row + t] = (int64_t)pick;
      } else if (rec) {          // t == T: bootstrap on the step recorded above (runner.py:236-245)
        float r = bk_r;
        if (bk_d == 0.f) {
          r = r + a.gamma * vboot;
          a.rewards[bk_e] = r;
          a.dones[bk_e] = 1.f;
        }
        a.deltas[bk_e] = r - bk_v;
        a.val_prev[b] = bk_v;
      }
    }
    // ---- off the critical path (the env worker is stepping): partial sums of state t+1, state row + stash of state t,
    // and -- wave 1 -- the env worker's answer: ONE early poll issued between the partial sums (its PCIe round trip runs
    // under the MFMAs), then the packed frame (one 16-byte load per lane covers it) under the stores
    const bool pf = w == 1 && t < T && p.x.frame_bits != 0;
    const unsigned int want_n = (p.x.seq0 + (unsigned int)(t + 1)) & 0x7fffffffu;
    unsigned long long gr_pf = ~0ULL;
    u32x4 f_pf = (u32x4){0u, 0u, 0u, 0u};
    pre_ok = false;
    const int nbase = (base + 1) % 5;
    if (t < T) {
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // planes 0..2 of state t+1 = planes 1..3 of state t.  Wave 1 fetches the env worker's answer UNDER them: an early
      // poll of the rec granule behind plane 0 (its PCIe round trip runs under plane 1's MFMAs); when that one already
      // shows the awaited step the packed frame's load goes out behind plane 1 and lands under plane 2 -- both round
      // trips hidden --, else a second poll goes out there and the frame is loaded behind the partial sums (one exposed).
      // (Earlier the answer cannot have landed; later -- behind this CU's row stores -- the loads queue behind them.)
      __amdgpu_buffer_rsrc_t fr_pf = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.frame_u8 + (long)(p.x.env0 + b) * a.frame_stride), 0, (int)a.frame_stride, 0x00020000);
      unsigned long long gr_e = ~0ULL;
      bool early = false;
      conv1_planes(nbase, 0, 1);
      if (pf && !tg && p.x.early_poll) gr_e = __hip_atomic_load(p.x.rec + p.x.env0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      conv1_planes(nbase, 1, 2);
      if (pf && tg) f_pf = tg_fetch();     // poll + frame + record in ONE load per lane (validated by the chunk tags below)
      else if (pf) {
        early = p.x.early_poll && (unsigned int)(gr_e >> 33) == want_n;
        if (early) {
          gr_pf = gr_e;
          f_pf = __builtin_amdgcn_raw_buffer_load_b128(fr_pf, lane * 16, 0, 1 | 16);      // pixels 128*lane .. +127
        } else {
          gr_pf = __hip_atomic_load(p.x.rec + p.x.env0 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
      }
      conv1_planes(nbase, 2, 3);
      if (pf && !tg) {
        // wave 1 issues no store here: loads and stores retire through ONE in-order counter, a frame load behind the
        // row stores would wait for their HBM acknowledgements.  The packed frame is one 16-byte load per lane.
        pre_ok = (unsigned int)(gr_pf >> 33) == want_n;
        if (pre_ok && !early) f_pf = __builtin_amdgcn_raw_buffer_load_b128(fr_pf, lane * 16, 0, 1 | 16);
      }
    }
    RING_TS(6);                   // partial sums of the next state
    if (pf && tg) {
      pre_ok = tg_ok(f_pf, p.x.seq0 + (unsigned int)(t + 1));
      if (pre_ok) tg_unpack(f_pf, (nbase + 3) % 5);
    } else if (pf) {
      // (the answer's loads were issued above)
    } else {
      // state t: plane pl sits in slot (base + pl) % 5.  The stores are dealt to the waves other than the fetching one.
      const bool dealt = t < T && p.x.frame_bits != 0;          // wave 1 is busy with the env worker's answer
      const int sid = dealt ? (w == 0 ? tid : tid - 64) : tid, sn = dealt ? NT - 64 : NT;
      float* __restrict__ out = t == T ? p.x.bookmark + (long)b * S : p.x.states + (row + t) * S;
      const int hw4 = HW >> 2;
      if (t == T || !p.x.states_lazy)       // (lazy: the fp32 rows are expanded from the frame store on demand; the bookmark stays)
        for (int q = sid; q < 4 * hw4; q += sn) {
          const int pl = q / hw4, o = q - pl * hw4;
          const unsigned int x = reinterpret_cast<const unsigned int*>(ring + ((base + pl) % 5) * HW)[o];
          const float4 v = u8x4(x);
          __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(out) + q);
        }
      if (fs_slot != nullptr && t > 0) {    // the newest frame of state t -> frames[r][t+3] (7 KB instead of the 113 KB row)
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(ring + ((base + 3) % 5) * HW);
        u32x4* __restrict__ dst = reinterpret_cast<u32x4*>(fs_slot + (long)(t + 3) * HW);
        for (int q = sid; q < (HW >> 4); q += sn) dst[q] = src[q];
      }
      if (p.x.a1_rows != nullptr && t < T) {
        float* __restrict__ a1o = p.x.a1_rows + (row + t) * (16L * NP1);
        const int n4 = NP1 >> 2;
        // mask bits beside the stash (the update's conv2 backward-data reads 800 B of them per sample instead of the 25.6 KB
        // row as its ReLU mask): bit e of the row = (a1[e] > 0) -- a lane's float4 is one nibble, lane pairs make the bytes
        unsigned char* __restrict__ lmo = (p.x.a1_lm != nullptr && (16 * n4) % 64 == 0)
                                              ? reinterpret_cast<unsigned char*>(p.x.a1_lm) + (row + t) * (long)(16 * n4 / 2) : nullptr;
        for (int q = sid; q < 16 * n4; q += sn) {
          const int ch = q / n4, o4 = q - ch * n4;
          const float4 v = *reinterpret_cast<const float4*>(a1 + ch * p.PLANE2 + (o4 << 2));
          __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(a1o) + q);
          if (lmo != nullptr) {
            const unsigned int nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            const unsigned int hi = __shfl_xor(nib, 1);           // (whole waves run this loop: sid, sn, 16 * n4 are multiples of 64)
            if (!(q & 1)) lmo[q >> 1] = (unsigned char)(nib | (hi << 4));
          }
        }
      }
      if (p.x.a2_rows != nullptr && t < T) {
        float* __restrict__ a2o = p.x.a2_rows + (row + t) * (long)p.F;
        // (+ its mask bits: the update's da2 = (dl . Wc) * (a2 > 0) reads F / 8 bytes per sample instead of the row)
        unsigned char* __restrict__ mbo = (p.x.a2_mb != nullptr && (p.F & 7) == 0) ? p.x.a2_mb + (row + t) * (long)(p.F >> 3) : nullptr;
        for (int q = sid; q < (p.F >> 2); q += sn) {
          const float4 v = *reinterpret_cast<const float4*>(a2 + (q << 2));
          __builtin_nontemporal_store((f32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<f32x4*>(a2o) + q);
          if (mbo != nullptr) {
            const unsigned int nib = (v.x > 0.f ? 1u : 0u) | (v.y > 0.f ? 2u : 0u) | (v.z > 0.f ? 4u : 0u) | (v.w > 0.f ? 8u : 0u);
            const unsigned int hi = __shfl_xor(nib, 1);           // (F / 4 is even: a lane and its partner run the same trips)
            if (!(q & 1)) mbo[q >> 1] = (unsigned char)(nib | (hi << 4));
          }
        }
      }
    }
    if (pf && pre_ok && !tg) {
      // the prefetched frame -> plane 3 of state t+1 = the ring's free fifth slot, 128 pixels per lane
      unsigned char* __restrict__ dst = ring + ((nbase + 3) % 5) * HW + lane * 128;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const unsigned int bits = f_pf[q >> 1] >> (16 * (q & 1));
        u32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const unsigned int wq = bits >> (4 * e);
          v[e] = (wq & 1u) | ((wq & 2u) << 7) | ((wq & 4u) << 14) | ((wq & 8u) << 21);
        }
        if (lane * 128 + q * 16 < HW) *reinterpret_cast<u32x4*>(dst + q * 16) = v;
      }
      if (lane == 0) {
        reinterpret_cast<unsigned int*>(red)[HN + 32] = (unsigned int)gr_pf;
        reinterpret_cast<unsigned int*>(red)[HN + 33] = (unsigned int)(gr_pf >> 32);
        reinterpret_cast<unsigned int*>(red)[HN + 36] = 1u;
      }
    }
    if (t < T) base = nbase;
    RING_TS(7);                   // state row + stash stores issued
  }
  if (stamp)
    for (int q = 0; q < 10; ++q) p.x.dbg[q] = ts_sum[q];
#undef RING_TS
}

// conv1 of the ring kernel on the bf16 pipe (exact 3-way split of the weights, see a3c_ring_kernel); A2C_RING_F32=1: fp32 MFMAs
static bool ring_bf16() {
  const char* e = getenv("A2C_RING_F32");       // (read per call: A/B runs)
  return !(e != nullptr && e[0] == '1');
}
static size_t ring_lds(const StepP& p, int hnt) {
  const int ntile2 = (p.OH2 * p.OW2 + 15) / 16;
  const size_t part = (size_t)ntile2 * 1024 > (size_t)hnt * NT ? (size_t)ntile2 * 1024 : (size_t)hnt * NT;
  const size_t nf1 = ring_bf16() ? (size_t)NF1B : (size_t)NF1;
  return 4 * (nf1 + NF2 + (size_t)16 * p.PLANE2 + p.F4 + part + HN + 32 + 8) + (size_t)5 * p.a.H * p.a.W + 64;
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
  const int ntile2 = (p.OH2 * p.OW2 + 15) / 16;
  if (4 * p.PLANE1 < NF2 + p.F4 + ntile2 * 1024 + HN * NT) return false;   // conv2 fragments | a2 | K-half partials | head partials reuse the image region
  // chunks: output rows split 7/20, 14/20, rest -> input rows 4*r+8 (the last chunk takes the remainder)
  const int or0 = (p.OH1 * 7 + 19) / 20, or1 = (p.OH1 * 14 + 19) / 20;
  p.row_end[0] = 4 * (or0 - 1) + 8; p.row_end[1] = 4 * (or1 - 1) + 8; p.row_end[2] = H;
  if (!(0 < p.row_end[0] && p.row_end[0] < p.row_end[1] && p.row_end[1] < H)) return false;
  const int ntile = (p.OH1 * p.OW1 + 15) / 16;
  p.tile_end[0] = or0 * p.OW1 / 16; p.tile_end[1] = or1 * p.OW1 / 16; p.tile_end[2] = ntile;
  const int cap[NCH] = {SL0, SL1, SL2};
  for (int k = 0; k < NCH; ++k) {
    const int rows = p.row_end[k] - (k ? p.row_end[k - 1] : 0);
    if (rows * W > cap[k] * NT) return false;          // 4 planes * rows * W / 4 float4 over NT threads
  }
  return true;
}
// uint8 frames: one 16-byte load per thread must cover the plane
static bool u8_shapes(int H, int W) { return (H * W) % 16 == 0 && H * W <= 16 * NT; }
static size_t step_lds(const StepP& p) { return 4 * ((size_t)NF1 + (size_t)4 * p.PLANE1 + (size_t)16 * p.PLANE2 + 8 * HN + 16 + 32); }

static bool set_lds_attr() {
  static bool attr_set = false;
  if (!attr_set) {
    const void* ks[] = {(const void*)a3c_step_kernel<true, false, false>, (const void*)a3c_step_kernel<false, false, false>,
                        (const void*)a3c_step_kernel<true, true, false>, (const void*)a3c_step_kernel<false, true, false>,
                        (const void*)a3c_step_kernel<true, true, true, 4>, (const void*)a3c_step_kernel<true, true, true, 8>,
                        (const void*)a3c_ring_kernel<4, true>, (const void*)a3c_ring_kernel<8, true>,
                        (const void*)a3c_ring_kernel<4, false>, (const void*)a3c_ring_kernel<8, false>};
    for (const void* k : ks)
      if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    attr_set = true;
  }
  return true;
}
}  // namespace

/* the ring kernel's preconditions in ONE place: a2c_a3c_rollout launches it exactly when this holds, and
 * a2c_a3c_ring_supported answers the same question for the caller that wants to leave the fp32 state rows out
 * (states_lazy).  The environment switches are read per call (tests and A/B scripts flip them). */
static bool ring_applies(const StepP& p, int B, int cus, int hnt, const float* conv1_weight) {
  const char* nr = getenv("A2C_NO_RING");
  if (nr != nullptr && nr[0] == '1') return false;
  const char* rb = getenv("A2C_RING_BLOCKS");
  const bool ring_blocks = !(rb != nullptr && rb[0] == '0');
  return conv1_weight != nullptr && ((uintptr_t)conv1_weight % 4) == 0 && (B <= cus || ring_blocks) &&
         ring_lds(p, hnt) <= 160 * 1024 && (p.OH1 * p.OW1 + 15) / 16 <= 32;
}

static unsigned long long* g_ring_dbg = nullptr;
extern "C" {
/* debug: device buffer of 10 x uint64 that receives the summed phase stamps (100 MHz ticks) of workgroup 0 of every
 * following ring-kernel launch (see a3c_ring_kernel); NULL switches it off.  Not part of the drop-in boundary. */
int a2c_debug_ring_timing(unsigned long long* dev_buf) {
  g_ring_dbg = dev_buf;
  return A2C_OK;
}
#ifdef A2C_STEP_TIMING
int a2c_debug_step_skip(int mask) {
  return hipMemcpyToSymbol(HIP_SYMBOL(a2c_step_skip), &mask, sizeof(int)) == hipSuccess ? 0 : -1;
}
int a2c_debug_step_ts(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(a2c_step_ts), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : -1;
}
#endif
int a2c_a3c_step_supported(int C, int H, int W, int n_actions) {
  StepP p;
  return step_shapes(C, H, W, n_actions, p) && step_lds(p) <= 160 * 1024 ? 1 : 0;
}

int a2c_a3c_ring_supported(int B, int C, int H, int W, int n_actions, const float* conv1_weight) {
  StepP p;
  p.a = a2c_a3c_step_args{};
  p.x = RolloutX{};
  if (B < 1 || !step_shapes(C, H, W, n_actions, p) || step_lds(p) > 160 * 1024 || !u8_shapes(H, W)) return 0;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  return ring_applies(p, B, cus, n_actions + 1 <= 4 ? 4 : 8, conv1_weight) ? 1 : 0;
}

int a2c_a3c_step(const a2c_a3c_step_args* args, a2c_stream_t stream) {
  if (!args) return A2C_ERR_ARG;
  StepP p;
  p.a = *args;
  p.x = RolloutX{};
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
  if (a.frame_u8 && (a.frame_new || !u8_shapes(a.H, a.W) || a.frame_stride % 16 || (uintptr_t)a.frame_u8 % 16 ||
                     a.frame_stride < a.H * a.W))
    return A2C_ERR_ARG;
  if ((a.a1_out || a.a2_out) && ((p.OH1 * p.OW1) % 4 || a.a1_stride % 4 || a.a2_stride % 4 ||
                                 (((uintptr_t)a.a1_out | (uintptr_t)a.a2_out) % 16)))
    return A2C_ERR_ARG;
  const size_t lds = step_lds(p);
  if (!set_lds_attr()) return A2C_ERR_LAUNCH;
  if (a.frame_u8) {
    if (a.out) hipLaunchKernelGGL((a3c_step_kernel<true, true, false>), dim3(a.B), dim3(NT), lds, a2c_s(stream), p);
    else hipLaunchKernelGGL((a3c_step_kernel<false, true, false>), dim3(a.B), dim3(NT), lds, a2c_s(stream), p);
  } else {
    if (a.out) hipLaunchKernelGGL((a3c_step_kernel<true, false, false>), dim3(a.B), dim3(NT), lds, a2c_s(stream), p);
    else hipLaunchKernelGGL((a3c_step_kernel<false, false, false>), dim3(a.B), dim3(NT), lds, a2c_s(stream), p);
  }
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_a3c_rollout(const a2c_a3c_rollout_args* r, a2c_stream_t stream) {
  if (!r) return A2C_ERR_ARG;
  if (r->B < 0) return A2C_ERR_ARG;
  if (r->B == 0) return A2C_OK;
  StepP p;
  p.a = a2c_a3c_step_args{};
  a2c_a3c_step_args& a = p.a;
  a.B = r->B; a.C = r->C; a.H = r->H; a.W = r->W; a.n_actions = r->n_actions;
  a.wfrag1 = r->wfrag1; a.bias1 = r->bias1; a.wfrag2 = r->wfrag2; a.bias2 = r->bias2; a.Wc = r->Wc; a.bc = r->bc;
  a.heads = r->heads; a.ldh = r->ldh;
  a.val_prev = r->val_prev; a.rewards = r->rewards; a.dones = r->dones; a.deltas = r->deltas;
  a.T = r->T; a.slot0 = r->slot0; a.gamma = r->gamma; a.pong = r->pong;
  a.frame_u8 = r->frames; a.frame_stride = r->frame_stride;
  p.x.states = r->states; p.x.bookmark = r->bookmark; p.x.u = r->u; p.x.u_stride = (long)r->u_stride;
  p.x.actions = r->actions; p.x.cmd = (unsigned long long*)r->cmd; p.x.rec = (const unsigned long long*)r->rec;
  p.x.seq0 = r->seq0; p.x.env0 = r->env0; p.x.err = r->err; p.x.timeout_ticks = (long)r->timeout_ticks;
  p.x.a1_rows = r->a1_rows; p.x.a2_rows = r->a2_rows;
  p.x.a1_lm = (r->a1_rows != nullptr) ? (unsigned long long*)r->a1_lanemask_rows : nullptr;
  p.x.a2_mb = (r->a2_rows != nullptr) ? r->a2_maskbit_rows : nullptr;
  p.x.heads_rows = r->heads_rows; p.x.heads_rows_ld = (long)r->heads_rows_ld;
  p.x.fstore = r->frame_store; p.x.fs_slot_stride = (long)r->frame_store_slot_stride;
  p.x.nvalid = r->nvalid_rows; p.x.nvalid_carry = r->nvalid_carry;
  p.x.states_lazy = (r->states_lazy && r->frame_store) ? 1 : 0;
  {
    static const bool no_tag = getenv("A2C_NO_TAGGED") != nullptr && getenv("A2C_NO_TAGGED")[0] == '1';
    const bool ok = r->tagged && r->frame_bits && !no_tag && r->tagged_chunks >= 2 && r->tagged_chunks <= 64 &&
                    r->tagged_stride >= 16 * (int64_t)r->tagged_chunks && r->tagged_stride % 16 == 0 && ((uintptr_t)r->tagged % 16) == 0 &&
                    (r->tagged_chunks - 1) * 112 >= r->H * r->W;
    p.x.tagged = ok ? r->tagged : nullptr;
    p.x.tagged_stride = ok ? (long)r->tagged_stride : 0;
    p.x.tagged_nd = ok ? r->tagged_chunks - 1 : 0;
  }
  p.x.frame_bits = r->frame_bits ? 1 : 0;
  p.x.dbg = g_ring_dbg;
  {
    static const int gap = getenv("A2C_RING_POLL") ? atoi(getenv("A2C_RING_POLL")) : 4;
    p.x.poll_gap = gap > 0 ? gap : 1;
    static const int early = getenv("A2C_RING_EARLY") ? atoi(getenv("A2C_RING_EARLY")) : 1;
    p.x.early_poll = early;
  }
  if (r->frame_store && (!r->nvalid_rows || !r->nvalid_carry || r->T < 4 || ((uintptr_t)r->frame_store % 16) ||
                         r->frame_store_slot_stride % 16 || r->frame_store_slot_stride < (r->T + 4) * (int64_t)r->H * r->W))
    return A2C_ERR_ARG;
  if (r->heads_rows && r->heads_rows_ld < r->n_actions + 1) return A2C_ERR_ARG;
  if (!step_shapes(a.C, a.H, a.W, a.n_actions, p) || step_lds(p) > 160 * 1024 || !u8_shapes(a.H, a.W)) return A2C_ERR_ARG;
  if (!r->states || !r->bookmark || !r->u || !r->actions || !r->cmd || !r->rec || !r->frames || !r->err) return A2C_ERR_ARG;
  if (!a.wfrag1 || !a.bias1 || !a.wfrag2 || !a.bias2 || !a.Wc || !a.bc || !a.heads) return A2C_ERR_ARG;
  if (!a.val_prev || !a.rewards || !a.dones || !a.deltas || a.T < 1 || a.ldh < a.n_actions + 1) return A2C_ERR_ARG;
  if (r->frame_stride % 16 || r->frame_stride < (r->frame_bits ? (a.H * a.W + 7) / 8 : a.H * a.W) || r->timeout_ticks < 1 ||
      r->env0 < 0)
    return A2C_ERR_ARG;
  if ((r->a1_rows || r->a2_rows) && ((p.OH1 * p.OW1) % 4 || (((uintptr_t)r->a1_rows | (uintptr_t)r->a2_rows) % 16))) return A2C_ERR_ARG;
  if (r->a1_lanemask_rows && (!r->a1_rows || ((uintptr_t)r->a1_lanemask_rows % 8) || (16 * p.OH1 * p.OW1) % 256)) return A2C_ERR_ARG;
  if (r->a2_maskbit_rows && (!r->a2_rows || p.F % 8)) return A2C_ERR_ARG;
  if ((((uintptr_t)r->states | (uintptr_t)r->bookmark | (uintptr_t)r->frames | (uintptr_t)a.wfrag2 | (uintptr_t)a.Wc) % 16) ||
      (((uintptr_t)r->cmd | (uintptr_t)r->rec) % 8))
    return A2C_ERR_ARG;
  if (!set_lds_attr()) return A2C_ERR_LAUNCH;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
  // every env has a CU to itself: the state stays in LDS for the whole slot (a3c_ring_kernel)
  const int hnt = a.n_actions + 1 <= 4 ? 4 : 8;
  // More envs than CUs: ceil(B / CUs) launches one after the other, launch k playing the envs b = k (mod nblk) -- every env
  // worker thread owns a contiguous range of envs, so an interleaved block keeps ALL of them busy.  (The one-launch
  // alternative below takes the envs of a CU in turns and re-reads / re-writes the fp32 state row every step: 23.8 us per
  // env step at 2048 envs against the ring's 16-17.)  A2C_RING_BLOCKS=0: ring kernel only when B <= CUs.
  p.x.bstride = 1; p.x.boff = 0;
  if (ring_applies(p, a.B, cus, hnt, r->conv1_weight)) {
    const size_t rl = ring_lds(p, hnt);
    const bool bf = ring_bf16();
    const int nblk = (a.B + cus - 1) / cus;
    p.x.bstride = nblk;
    for (int k = 0; k < nblk; ++k) {
      p.x.boff = k;
      const int cnt = (a.B - k + nblk - 1) / nblk;
      if (bf) {
        if (hnt == 4) hipLaunchKernelGGL((a3c_ring_kernel<4, true>), dim3(cnt), dim3(NT), rl, a2c_s(stream), p, r->conv1_weight);
        else hipLaunchKernelGGL((a3c_ring_kernel<8, true>), dim3(cnt), dim3(NT), rl, a2c_s(stream), p, r->conv1_weight);
      } else {
        if (hnt == 4) hipLaunchKernelGGL((a3c_ring_kernel<4, false>), dim3(cnt), dim3(NT), rl, a2c_s(stream), p, r->conv1_weight);
        else hipLaunchKernelGGL((a3c_ring_kernel<8, false>), dim3(cnt), dim3(NT), rl, a2c_s(stream), p, r->conv1_weight);
      }
      A2C_CHECK_LAUNCH();
    }
    return A2C_OK;
  }
  if (p.x.states_lazy || p.x.a1_lm || p.x.a2_mb) return A2C_ERR_ARG;      // only the ring kernel can leave the fp32 rows out / writes mask bits
  const size_t lds = step_lds(p);
  // one workgroup per CU at most (157 KB of LDS each): all of them resident, envs beyond that take turns
  const int grid = a.B < cus ? a.B : cus;
  if (a.n_actions + 1 <= 4) hipLaunchKernelGGL((a3c_step_kernel<true, true, true, 4>), dim3(grid), dim3(NT), lds, a2c_s(stream), p);
  else hipLaunchKernelGGL((a3c_step_kernel<true, true, true, 8>), dim3(grid), dim3(NT), lds, a2c_s(stream), p);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
