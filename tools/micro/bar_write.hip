// Can the host CPU write straight into device memory (large BAR)?  hipcc -O2 tools/micro/bar_write.hip -o tools/micro/bin/bar_write
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <cstdint>
static sigjmp_buf jb;
static void on_segv(int) { siglongjmp(jb, 1); }
__global__ void spin_kernel(volatile unsigned long long* flag, volatile unsigned long long* ack, int rounds, long long* ticks) {
  long long t0 = wall_clock64();
  for (int r = 1; r <= rounds; ++r) {
    // tell the host (ack is host memory), wait for the host to answer in DEVICE memory
    __hip_atomic_store((unsigned long long*)ack, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const long long w0 = wall_clock64();
    while (__hip_atomic_load((unsigned long long*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (unsigned long long)r) {
      if (wall_clock64() - w0 > 50000000LL) { *ticks = -r; return; }        // 0.5 s: give up
    }
  }
  *ticks = wall_clock64() - t0;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  signal(SIGSEGV, on_segv);
  signal(SIGBUS, on_segv);
  for (int mode = 1; mode < 2; ++mode) {
    void* d = nullptr;
    hipError_t e = mode == 0 ? hipMalloc(&d, 1 << 16) : hipExtMallocWithFlags(&d, 1 << 16, hipDeviceMallocFinegrained);
    printf("mode %d (%s): alloc %s\n", mode, mode ? "finegrained" : "hipMalloc", hipGetErrorString(e));
    if (e != hipSuccess) continue;
    hipMemset(d, 0, 1 << 16);
    hipDeviceSynchronize();
    if (sigsetjmp(jb, 1) == 0) {
      volatile uint64_t* h = (volatile uint64_t*)d;
      h[1] = 0x1234567890abcdefULL;
      __sync_synchronize();
      uint64_t back = 0;
      hipMemcpy(&back, (char*)d + 8, 8, hipMemcpyDeviceToHost);
      printf("  host store -> device read back: %llx (%s)\n", (unsigned long long)back, back == 0x1234567890abcdefULL ? "OK" : "MISMATCH");
      uint64_t rd = h[1];
      printf("  host load: %llx\n", (unsigned long long)rd);
      // ping-pong: GPU -> host ack (host memory), host -> GPU flag (device memory)
      unsigned long long* ack = nullptr;
      hipHostMalloc(&ack, 64, hipHostMallocMapped);
      *ack = 0;
      long long* ticks = nullptr;
      hipHostMalloc(&ticks, 8, hipHostMallocMapped);
      const int R = 2000;
      hipMemset(d, 0, 64);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, 0, (volatile unsigned long long*)d, (volatile unsigned long long*)ack, R, ticks);
      auto t0 = std::chrono::steady_clock::now();
      for (int r = 1; r <= R; ++r) {
        auto w0 = std::chrono::steady_clock::now();
        bool dead = false;
        while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != (unsigned long long)r) {
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 1.0) { dead = true; break; }
        }
        if (dead) { printf("  host gave up at round %d\n", r); break; }
        __atomic_store_n((unsigned long long*)d, (unsigned long long)r, __ATOMIC_RELEASE);
        _mm_sfence();                       // flush the CPU's write-combining buffer (the BAR is mapped WC)
      }
      hipDeviceSynchronize();
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("  ping-pong (GPU store to host memory, host store to DEVICE memory, GPU polls local): %.2f us per round trip (GPU clock: %.2f)\n",
             us / R, *ticks / 100.0 / R);
    } else {
      printf("  host access FAULTED\n");
    }
  }
  // reference: both directions through host memory (GPU polls over PCIe)
  {
    unsigned long long *ack = nullptr, *flag = nullptr;
    hipHostMalloc(&ack, 64, hipHostMallocMapped);
    hipHostMalloc(&flag, 64, hipHostMallocMapped);
    *ack = 0; *flag = 0;
    long long* ticks = nullptr;
    hipHostMalloc(&ticks, 8, hipHostMallocMapped);
    const int R = 2000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(1), 0, 0, (volatile unsigned long long*)flag, (volatile unsigned long long*)ack, R, ticks);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 1; r <= R; ++r) {
      auto w0 = std::chrono::steady_clock::now();
      bool dead = false;
      while (__atomic_load_n(ack, __ATOMIC_ACQUIRE) != (unsigned long long)r) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() > 1.0) { dead = true; break; }
      }
      if (dead) { printf("  host gave up at round %d\n", r); break; }
      __atomic_store_n(flag, (unsigned long long)r, __ATOMIC_RELEASE);
    }
    hipDeviceSynchronize();
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("reference (flag in HOST memory, GPU polls over PCIe): %.2f us per round trip (GPU clock: %.2f)\n", us / R, *ticks / 100.0 / R);
  }
  return 0;
}
