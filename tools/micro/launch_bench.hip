// Launch-overhead microbenchmark: empty kernels, back to back, by grid / block / dynamic LDS / kernarg size.
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { char pad[320]; };
__global__ void k_empty(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_empty_big(Big b, int* p) { if (p && threadIdx.x == 9999) *p = b.pad[0]; }
__global__ void k_lds(int* p) { extern __shared__ float l[]; if (p && threadIdx.x == 9999) *p = (int)l[0]; }
static float timeit(void (*launch)(hipStream_t), hipStream_t st, int n) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch(st);
  hipEventRecord(e0, st);
  for (int i = 0; i < n; ++i) launch(st);
  hipEventRecord(e1, st); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / n * 1e3f;
}
static int G, Bk; static size_t L;
int main() {
  hipStream_t st; hipStreamCreate(&st);
  hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  int grids[] = {1, 256, 1024}; int blocks[] = {64, 256, 512, 1024};
  for (int g : grids) for (int b : blocks) { G = g; Bk = b;
    printf("empty grid=%4d block=%4d: %6.2f us\n", g, b, timeit([](hipStream_t s){ hipLaunchKernelGGL(k_empty, dim3(G), dim3(Bk), 0, s, nullptr); }, st, 200)); }
  G = 256; Bk = 512;
  printf("empty+320B kernarg grid=256 block=512: %6.2f us\n", timeit([](hipStream_t s){ Big b{}; hipLaunchKernelGGL(k_empty_big, dim3(G), dim3(Bk), 0, s, b, nullptr); }, st, 200));
  size_t ls[] = {0, 16 << 10, 64 << 10, 100 << 10, 157 << 10};
  for (size_t l : ls) for (int b : {256, 512}) { L = l; Bk = b;
    printf("lds=%3zu KB grid=256 block=%d: %6.2f us\n", l >> 10, b, timeit([](hipStream_t s){ hipLaunchKernelGGL(k_lds, dim3(G), dim3(Bk), L, s, nullptr); }, st, 200)); }
  for (int g : {128, 256, 512, 2048}) { G = g; L = 157 << 10; Bk = 512;
    printf("lds=157 KB grid=%d block=512: %6.2f us\n", g, timeit([](hipStream_t s){ hipLaunchKernelGGL(k_lds, dim3(G), dim3(Bk), L, s, nullptr); }, st, 200)); }
  return 0;
}
