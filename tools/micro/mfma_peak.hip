// Sustained fp32 MFMA rate on this part: chains of v_mfma_f32_16x16x4_f32 / 32x32x2, by waves per SIMD
// and number of independent accumulators per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(1024) void k16(float* out, int iters) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(1024) void k32(float* out, int iters) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
  if (s == 123.456f) out[0] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < 3; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flop_per_launch / ms / 1e9);
}
int main() {
  float* out; hipMalloc(&out, 4);
  const int iters = 4000, grid = 256;
  for (int wps : {1, 2, 4}) {           // waves per SIMD = block/256 with one block per CU
    const int block = 256 * wps;
    char nm[128];
    const double f16 = 2.0 * 16 * 16 * 4, f32 = 2.0 * 32 * 32 * 2;
    const double waves = (double)grid * block / 64;
    snprintf(nm, 128, "16x16x4 f32, %d waves/SIMD, 1 acc chain", wps);
    run(nm, [&]{ hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(block), 0, 0, out, iters); }, waves * iters * 8 * 1 * f16);
    snprintf(nm, 128, "16x16x4 f32, %d waves/SIMD, 2 acc chains", wps);
    run(nm, [&]{ hipLaunchKernelGGL(k16<2>, dim3(grid), dim3(block), 0, 0, out, iters); }, waves * iters * 8 * 2 * f16);
    snprintf(nm, 128, "16x16x4 f32, %d waves/SIMD, 4 acc chains", wps);
    run(nm, [&]{ hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(block), 0, 0, out, iters); }, waves * iters * 8 * 4 * f16);
    snprintf(nm, 128, "32x32x2 f32, %d waves/SIMD, 1 acc chain", wps);
    run(nm, [&]{ hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(block), 0, 0, out, iters); }, waves * iters * 8 * 1 * f32);
    snprintf(nm, 128, "32x32x2 f32, %d waves/SIMD, 2 acc chains", wps);
    run(nm, [&]{ hipLaunchKernelGGL(k32<2>, dim3(grid), dim3(block), 0, 0, out, iters); }, waves * iters * 8 * 2 * f32);
  }
  return 0;
}
