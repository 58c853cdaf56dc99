// Host-side entry points of the C ABI that touch the HIP runtime but launch no kernel: the pinned,
// device-mapped pool region the env workers and the rollout kernels share (include/a2c_hostpool.h) and
// the async copies of the memcpy ingest.  SURVEY.md section 8(b): a2c_rollout_buffer_{create,destroy}.
#include "a2c_common.h"

#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

extern "C" {
int a2c_pinned_register(void* host, size_t bytes, void** dev_out) {
  if (!host || !bytes || !dev_out) return A2C_ERR_ARG;
  if (hipHostRegister(host, bytes, hipHostRegisterMapped | hipHostRegisterPortable) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  if (hipHostGetDevicePointer(dev_out, host, 0) != hipSuccess) {
    (void)hipGetLastError();
    (void)hipHostUnregister(host);
    return A2C_ERR_LAUNCH;
  }
  return A2C_OK;
}

int a2c_pinned_unregister(void* host) {
  if (!host) return A2C_ERR_ARG;
  if (hipHostUnregister(host) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  return A2C_OK;
}

int a2c_rollout_buffer_create(const char* shm_name, size_t bytes, void** host_out, void** dev_out) {
  if (!shm_name || !bytes || !host_out || !dev_out) return A2C_ERR_ARG;
  const int fd = shm_open(shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return A2C_ERR_ARG;
  if (ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    shm_unlink(shm_name);
    return A2C_ERR_WORKSPACE;
  }
  void* h = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (h == MAP_FAILED) {
    shm_unlink(shm_name);
    return A2C_ERR_WORKSPACE;
  }
  memset(h, 0, bytes);
  const int rc = a2c_pinned_register(h, bytes, dev_out);
  if (rc != A2C_OK) {
    munmap(h, bytes);
    shm_unlink(shm_name);
    return rc;
  }
  *host_out = h;
  return A2C_OK;
}

// Device memory the HOST writes into (large-BAR systems: the pointer is valid on both sides): fine-grained, so the
// device's loads are not served from a stale L2 line.  The env worker threads push their answers (packed frame, then the
// rec granule, each behind an sfence) straight into HBM; the rollout kernels then poll and fetch LOCALLY instead of
// reading host memory over PCIe (two dependent round trips per env step).  include/a2c_mi355x.h
int a2c_push_buffer_alloc(size_t bytes, void** ptr_out) {
  if (!bytes || !ptr_out) return A2C_ERR_ARG;
  void* d = nullptr;
  if (hipExtMallocWithFlags(&d, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  if (hipMemset(d, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(d);
    return A2C_ERR_LAUNCH;
  }
  // hipExtMallocWithFlags succeeds without a large BAR too; the env worker threads would then fault on their first
  // _mm_stream store.  Guarded probe: the KERNEL stores a pattern through the pointer (read() from a pipe = copy_to_user:
  // EFAULT, not SIGSEGV, when the address is not mapped into this process) and the device must read that pattern back.
  // One word per 4 KB page AND the last word of the buffer: a partially mapped BAR would otherwise only show up as a fault in
  // a worker thread pushing the rec / frame rows of a high env index.
  bool ok = false;
  int pfd[2];
  if (bytes >= sizeof(unsigned long long) && pipe(pfd) == 0) {
    const unsigned long long pat = 0xA2C0BA5EC0FFEE01ull;
    const size_t W = sizeof pat;
    std::vector<size_t> offs;
    for (size_t off = 0; off + W <= bytes; off += 4096) offs.push_back(off);
    const size_t last = (bytes - W) & ~(size_t)7;
    if (offs.back() != last) offs.push_back(last);
    ok = true;
    for (size_t off : offs) {
      const unsigned long long v = pat ^ (unsigned long long)off;
      if (write(pfd[1], &v, W) != (ssize_t)W || read(pfd[0], (char*)d + off, W) != (ssize_t)W) { ok = false; break; }
    }
    if (ok) {
      std::vector<char> back(bytes);
      ok = hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost) == hipSuccess;
      for (size_t i = 0; ok && i < offs.size(); ++i) {
        unsigned long long v;
        memcpy(&v, back.data() + offs[i], W);
        ok = v == (pat ^ (unsigned long long)offs[i]);
      }
    }
    close(pfd[0]);
    close(pfd[1]);
  }
  (void)hipGetLastError();
  if (!ok || hipMemset(d, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    (void)hipFree(d);
    return A2C_ERR_LAUNCH;         // not host-writable: the caller keeps the pinned host region (ThreadEnvPool: push_ptr = 0)
  }
  *ptr_out = d;
  return A2C_OK;
}
int a2c_push_buffer_free(void* ptr) {
  if (!ptr) return A2C_ERR_ARG;
  return hipFree(ptr) == hipSuccess ? A2C_OK : A2C_ERR_LAUNCH;
}

int a2c_rollout_buffer_destroy(const char* shm_name, void* host, size_t bytes) {
  if (!host || !bytes) return A2C_ERR_ARG;
  int rc = a2c_pinned_unregister(host);
  munmap(host, bytes);
  if (shm_name) shm_unlink(shm_name);
  return rc;
}

int a2c_set_blocking_sync(int on) {
  // must run before the first kernel / allocation of the process on the device
  if (hipSetDeviceFlags(on ? hipDeviceScheduleBlockingSync : hipDeviceScheduleAuto) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  return A2C_OK;
}

int a2c_device_pci_bus_id(char* out, int len) {
  if (!out || len < 16) return A2C_ERR_ARG;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetPCIBusId(out, len, dev) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  return A2C_OK;
}

int a2c_memcpy_async(void* dst, const void* src, size_t bytes, int kind, a2c_stream_t stream) {
  if (!bytes) return A2C_OK;
  if (!dst || !src || kind < 1 || kind > 3) return A2C_ERR_ARG;
  const hipMemcpyKind k = kind == 1 ? hipMemcpyHostToDevice : kind == 2 ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  if (hipMemcpyAsync(dst, src, bytes, k, a2c_s(stream)) != hipSuccess) {
    (void)hipGetLastError();
    return A2C_ERR_LAUNCH;
  }
  return A2C_OK;
}
}
