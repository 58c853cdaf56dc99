// Shared helpers for the gfx950 kernels of liba2c_mi355x.so (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/a2c_mi355x.h"

#define A2C_CHECK_LAUNCH()                                   \
  do {                                                       \
    if (hipGetLastError() != hipSuccess) return A2C_ERR_LAUNCH; \
  } while (0)

static inline hipStream_t a2c_s(a2c_stream_t s) { return (hipStream_t)s; }

// memory-bound elementwise kernels: cap the grid and grid-stride (guide G11)
static inline int a2c_grid_1d(int64_t n, int block, int max_blocks = 2048) {
  int64_t g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}

// wave64 sum (all lanes get the result)
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block sum for blockDim.x == 256 (4 waves); result valid in thread 0
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T* sm /* >= 4 */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) sm[w] = v;
  __syncthreads();
  T r = sm[0] + sm[1] + sm[2] + sm[3];
  __syncthreads();
  return r;
}

// Deterministic grid-wide sum of NV doubles per workgroup (256 threads): every workgroup leaves its partials in its own slot
// of `scratch`, takes a ticket, and the LAST one to arrive adds the slots in workgroup order (fixed tree) and writes out[0..NV).
// No fp64 atomics: the result does not depend on the order in which the workgroups finish.  scratch: A2C_REDUCE_SCRATCH_DOUBLES
// doubles, word 0 = the ticket counter, zero before the first use (the last workgroup resets it); at most A2C_REDUCE_MAX_BLOCKS
// workgroups; kernels that share a scratch must be ordered (same stream).
#define A2C_REDUCE_MAX_BLOCKS 1024
static_assert(A2C_REDUCE_SCRATCH_DOUBLES >= 8 + 3 * A2C_REDUCE_MAX_BLOCKS, "include/a2c_mi355x.h: A2C_REDUCE_SCRATCH_DOUBLES");
template <int NV>
__device__ __forceinline__ void grid_sum_ordered(const double (&v)[NV], double* __restrict__ out, double* __restrict__ scratch,
                                                 double* sm /* >= 4 */) {
  __shared__ unsigned int s_last;
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < NV; ++i) scratch[8 + (long)blockIdx.x * NV + i] = v[i];
    __threadfence();
    s_last = atomicAdd(reinterpret_cast<unsigned int*>(scratch), 1u) == gridDim.x - 1 ? 1u : 0u;
  }
  __syncthreads();
  if (s_last == 0u) return;
  __threadfence();
  const volatile double* part = scratch + 8;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double a = 0.0;
    for (unsigned int b = threadIdx.x; b < gridDim.x; b += 256) a += part[(long)b * NV + i];
    a = block_sum_256(a, sm);
    if (threadIdx.x == 0) out[i] = a;
  }
  if (threadIdx.x == 0) *reinterpret_cast<unsigned int*>(scratch) = 0u;
}

// Small device buffers (error flags, reduction sums; <= 256 B, 4-byte multiples) are cleared by a one-wave kernel,
// not by hipMemsetAsync: as a NODE of a captured hipGraph the memset was observed to leave 0x01010101 in a
// 4-byte flag on some replays (ROCm 7.2, gfx950), which a kernel node never does.
namespace {
__global__ void a2c_zero_words_kernel(unsigned int* p, int n) {
  for (int i = threadIdx.x; i < n; i += 64) p[i] = 0u;
}
}  // namespace
static inline void a2c_zero_async(void* p, size_t bytes, hipStream_t st) {
  hipLaunchKernelGGL(a2c_zero_words_kernel, dim3(1), dim3(64), 0, st, (unsigned int*)p, (int)(bytes / 4));
}
