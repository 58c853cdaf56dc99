// Host <-> device hand-shake through pinned host memory, as the persistent rollout kernel uses it
// (one workgroup per env; host worker threads play the envs):
//   device: poll rec[b] (8-byte {payload, tag} granule in pinned host memory, system-scope loads) until tag == k,
//           read the env's 7056-byte uint8 frame straight from pinned host memory (sc0 sc1 16-byte loads),
//           "compute" for `busy_us`, publish cmd[b] = (k << 32) | checksum with one 8-byte system-scope store;
//   host:   worker thread spins on cmd[b], checks the checksum of the frame it wrote, writes the next frame and
//           then the tagged granule.
// Measures the per-step round trip and verifies every frame byte reached the device fresh (no stale lines).
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/micro/pingpong.hip -o /tmp/pp && /tmp/pp [envs] [steps] [busy_us] [threads] [shm]
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int FRAME = 7056, FSTRIDE = 7168, NT = 512;

__global__ __launch_bounds__(NT) void pingpong_kernel(const unsigned char* frames, const unsigned long long* rec,
                                                      unsigned long long* cmd, int steps, int busy_ticks, int* err,
                                                      unsigned long long* dev_sink) {
  __shared__ unsigned int red[NT / 64];
  __shared__ unsigned long long tagv;
  const int b = blockIdx.x, tid = threadIdx.x;
  __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc((void*)(frames + (size_t)b * FSTRIDE), 0, FSTRIDE, 0x00020000);
  for (int k = 0; k < steps; ++k) {
    if (tid == 0) {
      const unsigned long long t0 = wall_clock64();
      unsigned long long v;
      for (;;) {
        v = __hip_atomic_load(rec + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(v >> 32) == (unsigned)k) break;
        if (wall_clock64() - t0 > 200000000ULL) { *err = 1; break; }   // 2 s
        __builtin_amdgcn_s_sleep(8);
      }
      tagv = v;
    }
    __syncthreads();
    if (*(volatile int*)err) return;
    unsigned int s = 0;
    if (tid * 16 < FRAME) {
      const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(fr, tid * 16, 0, 1 | 16);   // sc0 sc1: system scope
#pragma unroll
      for (int i = 0; i < 4; ++i) s += (x[i] & 0xff) + ((x[i] >> 8) & 0xff) + ((x[i] >> 16) & 0xff) + (x[i] >> 24);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
      unsigned int tot = 0;
      for (int i = 0; i < NT / 64; ++i) tot += red[i];
      const unsigned long long t1 = wall_clock64();
      while ((long long)(wall_clock64() - t1) < busy_ticks) __builtin_amdgcn_s_sleep(4);
      __hip_atomic_store(cmd + b, ((unsigned long long)k << 32) | tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      dev_sink[b] = tagv;
    }
    __syncthreads();
  }
}

static inline unsigned frame_byte(int b, int k, int i) { return (unsigned)((b * 131 + k * 31 + i * 7 + (i >> 8)) & 0xff); }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 256, steps = argc > 2 ? atoi(argv[2]) : 400;
  const int busy_us = argc > 3 ? atoi(argv[3]) : 0, nthr = argc > 4 ? atoi(argv[4]) : 32, use_shm = argc > 5 ? atoi(argv[5]) : 1;
  const size_t bytes = (size_t)B * FSTRIDE + (size_t)B * 64 * 2;
  unsigned char* host = nullptr;
  if (use_shm) {   // what the Python env workers map: POSIX shared memory, registered with HIP by the main process
    char name[64];
    snprintf(name, sizeof name, "/a2c_pp_%d", (int)getpid());
    int fd = shm_open(name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, bytes) != 0) { perror("shm"); return 2; }
    host = (unsigned char*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    shm_unlink(name);
    if (host == MAP_FAILED) { perror("mmap"); return 2; }
    memset(host, 0, bytes);
    CK(hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable));
  } else {
    CK(hipHostMalloc((void**)&host, bytes, hipHostMallocMapped | hipHostMallocCoherent));
    memset(host, 0, bytes);
  }
  unsigned char* dptr;
  CK(hipHostGetDevicePointer((void**)&dptr, host, 0));
  unsigned char* frames = host;
  volatile unsigned long long* rec = (volatile unsigned long long*)(host + (size_t)B * FSTRIDE);   // one 64-B line per env
  volatile unsigned long long* cmd = rec + (size_t)B * 8;
  // device views with a 64-byte stride are awkward for the kernel's indexing: use dense arrays instead
  // (rec[b], cmd[b] 8 bytes apart; false sharing between host workers is part of what is measured)
  rec = (volatile unsigned long long*)(host + (size_t)B * FSTRIDE);
  cmd = rec + B;
  int* d_err;
  unsigned long long* d_sink;
  CK(hipMalloc(&d_err, 4));
  CK(hipMemset(d_err, 0, 4));
  CK(hipMalloc(&d_sink, 8 * B));
  for (int b = 0; b < B; ++b) cmd[b] = ~0ULL;
  std::atomic<int> bad{0};
  std::atomic<bool> go{false};
  std::vector<std::thread> th;
  for (int w = 0; w < nthr; ++w)
    th.emplace_back([&, w] {
      std::vector<int> mine;
      for (int b = w; b < B; b += nthr) mine.push_back(b);
      std::vector<int> k(mine.size(), 0);
      // frame 0 + tag 0
      for (size_t q = 0; q < mine.size(); ++q) {
        const int b = mine[q];
        for (int i = 0; i < FRAME; ++i) frames[(size_t)b * FSTRIDE + i] = (unsigned char)frame_byte(b, 0, i);
        __atomic_store_n((unsigned long long*)&rec[b], 0ULL << 32, __ATOMIC_RELEASE);
      }
      while (!go.load()) std::this_thread::yield();
      size_t left = mine.size();
      const auto t0 = std::chrono::steady_clock::now();
      while (left) {
        for (size_t q = 0; q < mine.size(); ++q) {
          if (k[q] >= steps) continue;
          const int b = mine[q];
          const unsigned long long c = __atomic_load_n((unsigned long long*)&cmd[b], __ATOMIC_ACQUIRE);
          if ((unsigned)(c >> 32) != (unsigned)k[q]) continue;
          unsigned want = 0;
          for (int i = 0; i < FRAME; ++i) want += frame_byte(b, k[q], i);
          if ((unsigned)c != want) bad++;
          ++k[q];
          if (k[q] >= steps) { --left; continue; }
          unsigned char* f = frames + (size_t)b * FSTRIDE;
          for (int i = 0; i < FRAME; ++i) f[i] = (unsigned char)frame_byte(b, k[q], i);
          __atomic_store_n((unsigned long long*)&rec[b], (unsigned long long)k[q] << 32, __ATOMIC_RELEASE);
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) { bad += 1000000; break; }
      }
    });
  std::this_thread::sleep_for(std::chrono::milliseconds(200));
  unsigned long long* d_rec = (unsigned long long*)(dptr + (size_t)B * FSTRIDE);
  unsigned long long* d_cmd = d_rec + B;
  go = true;
  const auto t0 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(pingpong_kernel, dim3(B), dim3(NT), 0, 0, dptr, d_rec, d_cmd, steps, busy_us * 100, d_err, d_sink);
  CK(hipDeviceSynchronize());
  const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  for (auto& t : th) t.join();
  int err = 0;
  CK(hipMemcpy(&err, d_err, 4, hipMemcpyDeviceToHost));
  printf("pingpong: envs=%d steps=%d busy=%dus threads=%d mem=%s : %.2f us per step (%.2f ms total), bad checksums=%d, device timeout=%d\n",
         B, steps, busy_us, nthr, use_shm ? "shm+hipHostRegister" : "hipHostMalloc", us / steps, us / 1e3, bad.load(), err);

  // reference numbers: hipMemcpyAsync H2D of one step's frames and a D2H of the actions + event sync
  unsigned char* dbuf;
  CK(hipMalloc(&dbuf, (size_t)B * FSTRIDE));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  for (int rep = 0; rep < 2; ++rep) {
    const auto a0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) {
      CK(hipMemcpyAsync(dbuf, host, (size_t)B * FSTRIDE, hipMemcpyHostToDevice, st));
      CK(hipStreamSynchronize(st));
    }
    const double u = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a0).count() / 200;
    if (rep) printf("hipMemcpyAsync H2D %zu bytes + sync: %.1f us (%.1f GB/s)\n", (size_t)B * FSTRIDE, u, B * FSTRIDE / u / 1e3);
  }
  for (int rep = 0; rep < 2; ++rep) {
    const auto a0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 200; ++i) {
      CK(hipMemcpyAsync(host, dbuf, (size_t)B * 8, hipMemcpyDeviceToHost, st));
      CK(hipStreamSynchronize(st));
    }
    const double u = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a0).count() / 200;
    if (rep) printf("hipMemcpyAsync D2H %d bytes + sync: %.1f us\n", B * 8, u);
  }
  return (bad.load() || err) ? 1 : 0;
}
