// Reverse segmented discounted scans (utils.discount, utils.py:63-79) and the
// normalisation helpers of Updater.update_model (updater.py:70-98).
//
// Layout: n_seg rows of T contiguous fp32 (rollout-major buffers, training.py:88-101).
// One 64-lane workgroup owns S consecutive rows (S*ld <= TILE floats per array): the rows
// are one contiguous span in HBM, so the loads/stores are fully coalesced; they go through
// LDS with row stride ld = T|1 (odd => the per-lane walk is bank-conflict free), and lane j
// then walks row j backwards with the reference's exact operation order
//   run = x[i] + g*run   (one fp32 multiply, one fp32 add; no FMA contraction)
// so the result is bit-identical to the reference's Python loop.
// Algorithmic traffic: 12 B/element (x, dones in; y out); fused advs+returns: 20 B/element.
#include "a2c_common.h"

namespace {
constexpr int TILE = 2048;  // floats per array per workgroup -> 24 KB LDS (fused) => 6 WG/CU

template <int NARR>
__global__ __launch_bounds__(64) void scan_rows_kernel(const float* __restrict__ x0,
                                                       const float* __restrict__ x1,
                                                       const float* __restrict__ dn,
                                                       float* __restrict__ y0, float* __restrict__ y1,
                                                       long n_seg, int T, int S, int ld, float g0,
                                                       float g1, int* err) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s0 = smem;
  float* sd = smem + S * ld;
  float* s1 = smem + 2 * S * ld;
  const int lane = threadIdx.x;
  const long n_tasks = (n_seg + S - 1) / S;
  for (long task = blockIdx.x; task < n_tasks; task += gridDim.x) {
    const long row0 = task * S;
    const int rows = (int)min((long)S, n_seg - row0);
    const long base = row0 * (long)T;
    const int cnt = rows * T;
    for (int e = lane; e < cnt; e += 64) {
      const int j = e / T, t = e - j * T, a = j * ld + t;
      s0[a] = x0[base + e];
      sd[a] = dn[base + e];
      if (NARR == 2) s1[a] = x1[base + e];
    }
    __syncthreads();
    if (lane < rows) {
      int a = lane * ld + T - 1;
      if (err != nullptr && n_seg > 1 && sd[a] != 1.0f) *err = 1;
      float r0 = 0.f, r1 = 0.f;
      for (int t = T - 1; t >= 0; --t, --a) {
        if (sd[a] == 1.0f) { r0 = 0.f; r1 = 0.f; }
        r0 = __fadd_rn(s0[a], __fmul_rn(g0, r0));
        s0[a] = r0;
        if (NARR == 2) {
          r1 = __fadd_rn(s1[a], __fmul_rn(g1, r1));
          s1[a] = r1;
        }
      }
    }
    __syncthreads();
    for (int e = lane; e < cnt; e += 64) {
      const int j = e / T, t = e - j * T, a = j * ld + t;
      y0[base + e] = s0[a];
      if (NARR == 2) y1[base + e] = s1[a];
    }
    __syncthreads();
  }
}

// Bandwidth path (T % 4 == 0, 16 B aligned): one wave owns 64 rows and ALL 64 lanes walk (lane j =
// row j).  The rows are consumed in time chunks of TC = 32 steps from the end; a chunk is 64 row
// segments of 128 B, loaded 16 B per lane (8 lanes per segment, fully used 128 B lines), transposed
// through LDS (row stride 33 -> conflict-free column walk) and written back 16 B per lane.  The
// next chunk's loads are issued before the current chunk is walked, so HBM latency hides under
// the serial recurrence.  Same operation order as the reference -> bit-identical.
constexpr int TC = 32;
constexpr int TCP = TC + 1;

template <int NARR>
__global__ __launch_bounds__(64) void scan_chunk_kernel(const float* __restrict__ x0, const float* __restrict__ x1,
                                                        const float* __restrict__ dn, float* __restrict__ y0,
                                                        float* __restrict__ y1, long n_seg, int T, float g0, float g1,
                                                        int* err) {
  __shared__ __attribute__((aligned(16))) float s0[64 * TCP], sd[64 * TCP], s1[NARR == 2 ? 64 * TCP : 4];
  const int lane = threadIdx.x;
  const int lr = lane >> 3, lc = (lane & 7) << 2;        // this lane's (row within 8, first column) of a load
  const long n_grp = (n_seg + 63) >> 6;
  const int nchunk = (T + TC - 1) / TC;
  for (long grp = blockIdx.x; grp < n_grp; grp += gridDim.x) {
    const long row0 = grp << 6;
    const int rows = (int)min(64L, n_seg - row0);
    float4 p0[8], pd[8], p1[8];
    auto issue = [&](int ck) {                            // chunk ck covers t in [lo, lo + TC) clipped to [0, T)
      const int lo = T - (nchunk - ck) * TC;              // may be negative for the first (partial) chunk
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int r = i * 8 + lr, t = lo + lc;
        p0[i] = pd[i] = p1[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < rows && t >= 0) {                         // T % 4 == 0 and lo % 4 == 0: a float4 is all in or all out
          const long a = (row0 + r) * (long)T + t;
          p0[i] = *reinterpret_cast<const float4*>(x0 + a);
          pd[i] = *reinterpret_cast<const float4*>(dn + a);
          if (NARR == 2) p1[i] = *reinterpret_cast<const float4*>(x1 + a);
        }
      }
    };
    float r0 = 0.f, r1 = 0.f;
    issue(nchunk - 1);
    for (int ck = nchunk - 1; ck >= 0; --ck) {
      const int lo = T - (nchunk - ck) * TC;
#pragma unroll
      for (int i = 0; i < 8; ++i) {                       // registers -> LDS [row][TCP]
        const int a = (i * 8 + lr) * TCP + lc;
        s0[a] = p0[i].x; s0[a + 1] = p0[i].y; s0[a + 2] = p0[i].z; s0[a + 3] = p0[i].w;
        sd[a] = pd[i].x; sd[a + 1] = pd[i].y; sd[a + 2] = pd[i].z; sd[a + 3] = pd[i].w;
        if (NARR == 2) { s1[a] = p1[i].x; s1[a + 1] = p1[i].y; s1[a + 2] = p1[i].z; s1[a + 3] = p1[i].w; }
      }
      __syncthreads();
      if (ck > 0) issue(ck - 1);                          // in flight during the walk
      const int tlo = lo < 0 ? -lo : 0;                   // first valid column of this chunk
      if (lane < rows) {
        if (ck == nchunk - 1 && err != nullptr && n_seg > 1 && sd[lane * TCP + TC - 1] != 1.0f) *err = 1;
#pragma unroll 8
        for (int c = TC - 1; c >= tlo; --c) {
          const int a = lane * TCP + c;
          if (sd[a] == 1.0f) { r0 = 0.f; r1 = 0.f; }
          r0 = s0[a] + g0 * r0;                           // contraction is off: mul then add, like the reference
          s0[a] = r0;
          if (NARR == 2) { r1 = s1[a] + g1 * r1; s1[a] = r1; }
        }
      }
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 8; ++i) {                       // LDS -> HBM, 16 B per lane
        const int r = i * 8 + lr, t = lo + lc;
        if (r < rows && t >= 0) {
          const int a = r * TCP + lc;
          const long o = (row0 + r) * (long)T + t;
          *reinterpret_cast<float4*>(y0 + o) = make_float4(s0[a], s0[a + 1], s0[a + 2], s0[a + 3]);
          if (NARR == 2) *reinterpret_cast<float4*>(y1 + o) = make_float4(s1[a], s1[a + 1], s1[a + 2], s1[a + 3]);
        }
      }
      __syncthreads();
    }
  }
}

// rows longer than the LDS tile: one workgroup per row, time-chunks walked from the end with
// the running sums carried in registers of lane 0 (exact; used for the flat n_seg==1 form).
template <int NARR>
__global__ __launch_bounds__(64) void scan_long_kernel(const float* __restrict__ x0,
                                                       const float* __restrict__ x1,
                                                       const float* __restrict__ dn,
                                                       float* __restrict__ y0, float* __restrict__ y1,
                                                       long n_seg, long T, float g0, float g1, int* err,
                                                       const int* __restrict__ run_if) {
  __shared__ float s0[TILE], sd[TILE], s1[NARR == 2 ? TILE : 1];
  const int lane = threadIdx.x;
  if (run_if != nullptr && *run_if == 0) return;      // flat fall-back pass: only when the row pass flagged a violation
  for (long row = blockIdx.x; row < n_seg; row += gridDim.x) {
    const long base = row * T;
    float r0 = 0.f, r1 = 0.f;
    for (long hi = T; hi > 0; hi -= TILE) {
      const long lo = hi > TILE ? hi - TILE : 0;
      const int cnt = (int)(hi - lo);
      for (int e = lane; e < cnt; e += 64) {
        s0[e] = x0[base + lo + e];
        sd[e] = dn[base + lo + e];
        if (NARR == 2) s1[e] = x1[base + lo + e];
      }
      __syncthreads();
      if (lane == 0) {
        if (hi == T && err != nullptr && n_seg > 1 && sd[cnt - 1] != 1.0f) *err = 1;
        for (int a = cnt - 1; a >= 0; --a) {
          if (sd[a] == 1.0f) { r0 = 0.f; r1 = 0.f; }
          r0 = __fadd_rn(s0[a], __fmul_rn(g0, r0));
          s0[a] = r0;
          if (NARR == 2) {
            r1 = __fadd_rn(s1[a], __fmul_rn(g1, r1));
            s1[a] = r1;
          }
        }
      }
      __syncthreads();
      for (int e = lane; e < cnt; e += 64) {
        y0[base + lo + e] = s0[e];
        if (NARR == 2) y1[base + lo + e] = s1[e];
      }
      __syncthreads();
    }
  }
}

template <int NARR>
int launch_scan(const float* x0, const float* x1, const float* dn, float* y0, float* y1, int64_t n_seg,
                int64_t T, float g0, float g1, int* err, hipStream_t st) {
  if (n_seg < 0 || T < 0) return A2C_ERR_ARG;
  if (n_seg == 0 || T == 0) return A2C_OK;
  if (!x0 || !dn || !y0 || (NARR == 2 && (!x1 || !y1))) return A2C_ERR_ARG;
  if (err) a2c_zero_async(err, sizeof(int), st);
  const int64_t ld = T | 1;
  const bool al16 = (((uintptr_t)x0 | (uintptr_t)dn | (uintptr_t)y0 | (uintptr_t)(NARR == 2 ? x1 : x0) |
                      (uintptr_t)(NARR == 2 ? y1 : y0)) % 16) == 0;
  if (T % 4 == 0 && T >= TC && n_seg >= 64 && al16) {
    const int64_t n_grp = (n_seg + 63) / 64;
    const int grid = (int)(n_grp < 256 * 6 ? n_grp : 256 * 6);
    hipLaunchKernelGGL(scan_chunk_kernel<NARR>, dim3(grid), dim3(64), 0, st, x0, x1, dn, y0, y1, (long)n_seg, (int)T, g0,
                       g1, err);
  } else if (ld <= TILE) {
    int S = (int)(TILE / ld);
    if (S > 64) S = 64;
    if (S > n_seg) S = (int)n_seg;
    const int64_t n_tasks = (n_seg + S - 1) / S;
    const int grid = (int)(n_tasks < 256 * 6 ? n_tasks : 256 * 6);
    const size_t lds = (size_t)(NARR + 1) * S * ld * sizeof(float);
    hipLaunchKernelGGL(scan_rows_kernel<NARR>, dim3(grid), dim3(64), lds, st, x0, x1, dn, y0, y1,
                       (long)n_seg, (int)T, S, (int)ld, g0, g1, err);
  } else {
    const int grid = (int)(n_seg < 1024 ? n_seg : 1024);
    hipLaunchKernelGGL(scan_long_kernel<NARR>, dim3(grid), dim3(64), 0, st, x0, x1, dn, y0, y1,
                       (long)n_seg, (long)T, g0, g1, err, (const int*)nullptr);
  }
  A2C_CHECK_LAUNCH();
  if (err && n_seg > 1) {
    // rows that do not end with done == 1 are not independent: the reference's flat loop (utils.py:73-79) carries
    // the running sum across them.  The row pass flagged it; redo the array as ONE row, in order, on the device
    // (a few ms for 32k elements -- the Runner never produces such data, arbitrary callers may).
    hipLaunchKernelGGL(scan_long_kernel<NARR>, dim3(1), dim3(64), 0, st, x0, x1, dn, y0, y1, 1L, (long)(n_seg * T), g0, g1,
                       (int*)nullptr, (const int*)err);
    A2C_CHECK_LAUNCH();
  }
  return A2C_OK;
}

__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, long n, double* sums, double* scratch) {
  __shared__ double sm[4];
  double a = 0.0, b = 0.0;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
    const double v = (double)x[i];
    a += v;
    b += v * v;
  }
  a = block_sum_256(a, sm);
  b = block_sum_256(b, sm);
  const double v2[2] = {a, b};
  grid_sum_ordered<2>(v2, sums, scratch, sm);       // fixed-order second stage: no fp64 atomics
}

__device__ __forceinline__ void mean_std(const double* sums, long n, float& mean, float& stdv) {
  const double m = sums[0] / (double)n;
  double var = (sums[1] - (double)n * m * m) / (double)(n - 1);  // unbiased (Tensor.std)
  if (var < 0.0) var = 0.0;
  mean = (float)m;
  stdv = (float)sqrt(var);
}

__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                        long n, const double* sums, long n_global, float eps) {
  float mean, stdv;
  mean_std(sums, n_global, mean, stdv);
  const float den = stdv + eps;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) y[i] = (x[i] - mean) / den;
}

__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ y, long n) {
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) y[i] = a[i] + b[i];
}
}  // namespace

extern "C" {
int a2c_version(void) { return 2; }

const char* a2c_error_string(int code) {
  switch (code) {
    case A2C_OK: return "ok";
    case A2C_ERR_ARG: return "invalid argument";
    case A2C_ERR_LAUNCH: return "kernel launch failed";
    case A2C_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

int a2c_discount_scan(const float* x, const float* dones, float* y, int64_t n_seg, int64_t T, float g,
                      int* err_flag, a2c_stream_t stream) {
  return launch_scan<1>(x, nullptr, dones, y, nullptr, n_seg, T, g, 0.f, err_flag, a2c_s(stream));
}

int a2c_gae_returns_fused(const float* deltas, const float* rewards, const float* dones, float* advs,
                          float* rets, int64_t n_seg, int64_t T, float g_adv, float g_ret, int* err_flag,
                          a2c_stream_t stream) {
  return launch_scan<2>(deltas, rewards, dones, advs, rets, n_seg, T, g_adv, g_ret, err_flag, a2c_s(stream));
}

int a2c_moments(const float* x, int64_t n, double* sums, double* scratch, a2c_stream_t stream) {
  if (!sums || !scratch || n < 0 || (n > 0 && !x)) return A2C_ERR_ARG;
  if (n == 0) {
    a2c_zero_async(sums, 2 * sizeof(double), a2c_s(stream));
    return A2C_OK;
  }
  hipLaunchKernelGGL(moments_kernel, dim3(a2c_grid_1d(n, 256, A2C_REDUCE_MAX_BLOCKS)), dim3(256), 0, a2c_s(stream), x, (long)n,
                     sums, scratch);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_normalize(const float* x, float* y, int64_t n, const double* sums, int64_t n_global, float eps,
                  a2c_stream_t stream) {
  if (n < 0 || n_global < 2 || !sums || (n > 0 && (!x || !y))) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL(normalize_kernel, dim3(a2c_grid_1d(n, 256)), dim3(256), 0, a2c_s(stream), x, y, (long)n, sums,
                     (long)n_global, eps);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}

int a2c_add(const float* a, const float* b, float* y, int64_t n, a2c_stream_t stream) {
  if (n < 0 || (n > 0 && (!a || !b || !y))) return A2C_ERR_ARG;
  if (n == 0) return A2C_OK;
  hipLaunchKernelGGL(add_kernel, dim3(a2c_grid_1d(n, 256)), dim3(256), 0, a2c_s(stream), a, b, y, (long)n);
  A2C_CHECK_LAUNCH();
  return A2C_OK;
}
}
