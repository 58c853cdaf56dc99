// liba2c_hostpool.so: host side of the rollout ingest (see include/a2c_hostpool.h).  Plain C,
// no HIP: env worker processes load it without initialising the GPU runtime.
#define _GNU_SOURCE
#include "../../include/a2c_hostpool.h"

#include <pthread.h>
#include <sys/prctl.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#if defined(__x86_64__) || defined(__i386__)
#include <immintrin.h>
#include <emmintrin.h>
#define cpu_relax() _mm_pause()
#define A2C_HAVE_SSE2 1
#else
#define cpu_relax() __asm__ __volatile__("" ::: "memory")
#endif

static inline int64_t now_ns(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000000LL + ts.tv_nsec;
}
static inline void sleep_ns(long ns) {
  struct timespec ts = {0, ns};
  nanosleep(&ts, NULL);
}
static inline a2c_pool_header *hdr(void *base) { return (a2c_pool_header *)base; }
static inline const a2c_pool_header *chdr(const void *base) { return (const a2c_pool_header *)base; }
static inline uint64_t *cmd_of(void *base) { return (uint64_t *)((char *)base + hdr(base)->off_cmd); }
static inline uint64_t *rec_of(void *base) { return (uint64_t *)((char *)base + hdr(base)->off_rec); }

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

/* frame slots are 16 B apart at least (one 16-byte load per lane on the device); fp32 frames whose size is a
 * multiple of 16 B are therefore dense, which is what the fp32 frame-stack kernels expect */
static size_t frame_stride_of(int frame_bytes) { return align_up((size_t)frame_bytes, 16); }

size_t a2c_pool_bytes(int n_envs, int frame_bytes) {
  if (n_envs < 1 || frame_bytes < 1) return 0;
  const size_t stride = frame_stride_of(frame_bytes);
  return align_up(4096 + 2 * align_up((size_t)n_envs * 8, 4096) + (size_t)n_envs * stride, 4096);
}

int a2c_pool_init(void *base, size_t bytes, int n_envs, int frame_bytes, int frame_dtype, int n_workers,
                  double rew_ema0) {
  if (!base || n_envs < 1 || frame_bytes < 1 || bytes < a2c_pool_bytes(n_envs, frame_bytes)) return -1;
  if (frame_dtype != A2C_FRAME_U8 && frame_dtype != A2C_FRAME_F32 && frame_dtype != A2C_FRAME_BITS) return -1;
  a2c_pool_header *h = hdr(base);
  memset(h, 0, 4096);
  h->version = A2C_POOL_VERSION;
  h->n_envs = (uint32_t)n_envs;
  h->frame_bytes = (uint32_t)frame_bytes;
  h->frame_stride = (uint32_t)frame_stride_of(frame_bytes);
  h->frame_dtype = (uint32_t)frame_dtype;
  h->n_workers = (uint32_t)n_workers;
  h->off_cmd = 4096;
  h->off_rec = h->off_cmd + align_up((size_t)n_envs * 8, 4096);
  h->off_frames = h->off_rec + align_up((size_t)n_envs * 8, 4096);
  h->total_bytes = a2c_pool_bytes(n_envs, frame_bytes);
  h->rew_ema = rew_ema0;
  h->frame_elems = frame_dtype == A2C_FRAME_F32 ? (uint32_t)frame_bytes / 4u
                   : frame_dtype == A2C_FRAME_BITS ? (uint32_t)frame_bytes * 8u : (uint32_t)frame_bytes;
  h->seq_start = 0;
  uint64_t *c = cmd_of(base), *r = rec_of(base);
  for (int j = 0; j < n_envs; ++j) {
    c[j] = ~0ULL;      /* seq 0xffffffff: nothing requested yet */
    r[j] = ~0ULL;      /* no frame yet                          */
  }
  __atomic_store_n(&h->magic, A2C_POOL_MAGIC, __ATOMIC_RELEASE);
  return 0;
}

static uint32_t tagged_chunks_of(uint32_t frame_elems) { return (frame_elems + 111u) / 112u + 1u; }

size_t a2c_pool_bytes_tagged(int n_envs, int frame_bytes, uint32_t frame_elems) {
  const size_t base = a2c_pool_bytes(n_envs, frame_bytes);
  if (!base) return 0;
  return base + align_up((size_t)n_envs * 16u * tagged_chunks_of(frame_elems), 4096);
}

int a2c_pool_enable_tagged(void *base, size_t region_bytes) {
  if (a2c_pool_check(base)) return -1;
  a2c_pool_header *h = hdr(base);
  if (h->frame_dtype != A2C_FRAME_BITS || h->off_tagged) return -1;
  const uint32_t chunks = tagged_chunks_of(h->frame_elems);
  const size_t off = align_up(h->total_bytes, 4096), need = off + align_up((size_t)h->n_envs * 16u * chunks, 4096);
  if (region_bytes < need) return -1;
  h->tagged_chunks = chunks;
  h->tagged_stride = 16u * chunks;
  h->total_bytes = need;
  memset((char *)base + off, 0xff, (size_t)h->n_envs * h->tagged_stride);      /* tag 0xffff: no frame yet */
  __atomic_store_n(&h->off_tagged, (uint64_t)off, __ATOMIC_RELEASE);
  return 0;
}

/* the mirror of env's packed frame (see a2c_pool_header): every chunk leaves as ONE aligned 16-byte store */
static void write_tagged(void *base, int env, const uint8_t *packed, uint32_t seq, float rew, int done) {
  const a2c_pool_header *h = chdr(base);
  if (!h->off_tagged) return;
  uint8_t *dst = (uint8_t *)base + h->off_tagged + (size_t)env * h->tagged_stride;
  const uint32_t nd = h->tagged_chunks - 1u, fb = h->frame_bytes;
  const uint16_t tag = (uint16_t)(seq & 0xffffu);
  for (uint32_t c = 0; c < nd; ++c) {
    uint8_t tmp[16] __attribute__((aligned(16)));
    const uint32_t o = 14u * c, n = o + 14u <= fb ? 14u : (o < fb ? fb - o : 0u);
    memset(tmp, 0, 14);
    memcpy(tmp, packed + o, n);
    memcpy(tmp + 14, &tag, 2);
#ifdef A2C_HAVE_SSE2
    _mm_store_si128((__m128i *)(dst + 16u * c), _mm_load_si128((const __m128i *)tmp));
#else
    memcpy(dst + 16u * c, tmp, 16);
#endif
  }
  uint8_t meta[16] __attribute__((aligned(16)));
  const uint32_t d32 = done ? 1u : 0u;
  const uint16_t z16 = 0;
  memcpy(meta, &rew, 4); memcpy(meta + 4, &d32, 4); memcpy(meta + 8, &seq, 4); memcpy(meta + 12, &z16, 2); memcpy(meta + 14, &tag, 2);
#ifdef A2C_HAVE_SSE2
  _mm_store_si128((__m128i *)(dst + 16u * nd), _mm_load_si128((const __m128i *)meta));
#else
  memcpy(dst + 16u * nd, meta, 16);
#endif
}

void a2c_pool_set_frame_elems(void *base, uint32_t frame_elems) { hdr(base)->frame_elems = frame_elems; }
void a2c_pool_set_seq_start(void *base, uint32_t seq_start) { hdr(base)->seq_start = seq_start; }

int a2c_pool_check(const void *base) {
  if (!base) return -1;
  const a2c_pool_header *h = chdr(base);
  return (__atomic_load_n(&h->magic, __ATOMIC_ACQUIRE) == A2C_POOL_MAGIC && h->version == A2C_POOL_VERSION) ? 0 : -1;
}

void a2c_pool_set_phase(void *base, uint32_t phase) { __atomic_store_n(&hdr(base)->phase, phase, __ATOMIC_RELEASE); }
uint32_t a2c_pool_phase(const void *base) { return __atomic_load_n(&chdr(base)->phase, __ATOMIC_ACQUIRE); }

/* the scan starts at env `start` (the one behind the env served last: every env of a worker waits its turn once per
 * sweep instead of the low indices being served first every time) */
static int pool_poll_from(void *base, int env0, int n, const uint32_t *next_seq, int64_t spin_ns, int start) {
  const uint64_t *c = cmd_of(base) + env0;
  const int64_t t0 = now_ns();
  if (start < 0 || start >= n) start = 0;
  for (unsigned sweep = 0;; ++sweep) {
    for (int i = start; i < n; ++i)
      if ((uint32_t)(__atomic_load_n(c + i, __ATOMIC_ACQUIRE) >> 32) == next_seq[i]) return i;
    for (int i = 0; i < start; ++i)
      if ((uint32_t)(__atomic_load_n(c + i, __ATOMIC_ACQUIRE) >> 32) == next_seq[i]) return i;
    if ((sweep & 15) == 15) {
      const uint32_t ph = a2c_pool_phase(base);
      if (ph == A2C_POOL_SHUTDOWN) return -2;
      if (ph == A2C_POOL_IDLE) sleep_ns(20000);   /* no rollout running: do not burn the core */
      if (now_ns() - t0 > spin_ns) return -1;
    }
    cpu_relax();
  }
}

int a2c_pool_poll(void *base, int env0, int n, const uint32_t *next_seq, int64_t spin_ns) {
  const uint64_t *c = cmd_of(base) + env0;
  const int64_t t0 = now_ns();
  for (unsigned sweep = 0;; ++sweep) {
    for (int i = 0; i < n; ++i)
      if ((uint32_t)(__atomic_load_n(c + i, __ATOMIC_ACQUIRE) >> 32) == next_seq[i]) return i;
    if ((sweep & 15) == 15) {
      const uint32_t ph = a2c_pool_phase(base);
      if (ph == A2C_POOL_SHUTDOWN) return -2;
      if (ph == A2C_POOL_IDLE) sleep_ns(20000);   /* no rollout running: do not burn the core */
      if (now_ns() - t0 > spin_ns) return -1;
    }
    cpu_relax();
  }
}

int32_t a2c_pool_action(const void *base, int env) {
  const uint64_t *c = (const uint64_t *)((const char *)base + chdr(base)->off_cmd);
  return (int32_t)(uint32_t)__atomic_load_n(c + env, __ATOMIC_ACQUIRE);
}

int a2c_pool_take(void *base, int env0, int n, const uint32_t *next_seq, int64_t spin_ns, int32_t *action) {
  const int i = a2c_pool_poll(base, env0, n, next_seq, spin_ns);
  if (i >= 0) *action = a2c_pool_action(base, env0 + i);
  return i;
}

void a2c_pool_publish(void *base, int env, const void *frame, uint32_t seq, float rew, int done) {
  a2c_pool_header *h = hdr(base);
  memcpy((char *)base + h->off_frames + (size_t)env * h->frame_stride, frame, h->frame_bytes);
  uint32_t rb;
  memcpy(&rb, &rew, 4);
  const uint64_t g = ((uint64_t)((seq << 1) | (done ? 1u : 0u)) << 32) | rb;
  __atomic_store_n(rec_of(base) + env, g, __ATOMIC_RELEASE);   /* frame bytes are visible before the tag */
}

#ifdef A2C_HAVE_SSE2
__attribute__((target("avx2"))) static unsigned pack_bits_avx2(const uint8_t *src, uint8_t *dst, size_t n, size_t *done) {
  __m256i acc = _mm256_setzero_si256();
  size_t p = 0;
  for (; p + 32 <= n; p += 32) {
    const __m256i x = _mm256_loadu_si256((const __m256i *)(src + p));
    acc = _mm256_or_si256(acc, x);
    const uint32_t m = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(x, 7));
    memcpy(dst + (p >> 3), &m, 4);
  }
  const __m128i a = _mm_or_si128(_mm256_castsi256_si128(acc), _mm256_extracti128_si256(acc, 1));
  uint64_t lo = (uint64_t)_mm_cvtsi128_si64(a) | (uint64_t)_mm_cvtsi128_si64(_mm_srli_si128(a, 8));
  lo |= lo >> 32; lo |= lo >> 16; lo |= lo >> 8;
  *done = p;
  return (unsigned)(lo & 0xffu);
}
#endif

#ifdef A2C_HAVE_SSE2
/* 64 pixels per step: vptestmb gives the 64 packed bits at once (Zen 4/5, Sapphire Rapids: full-rate AVX-512) */
__attribute__((target("avx512f,avx512bw"))) static unsigned pack_bits_avx512(const uint8_t *src, uint8_t *dst, size_t n, size_t *done) {
  __m512i acc = _mm512_setzero_si512();
  const __m512i one = _mm512_set1_epi8(1);
  size_t p = 0;
  for (; p + 64 <= n; p += 64) {
    const __m512i x = _mm512_loadu_si512((const void *)(src + p));
    acc = _mm512_or_si512(acc, x);
    const uint64_t m = (uint64_t)_mm512_test_epi8_mask(x, one);
    memcpy(dst + (p >> 3), &m, 8);
  }
  *done = p;
  return _mm512_test_epi8_mask(acc, _mm512_set1_epi8((char)0xfe)) ? 2u : (_mm512_test_epi8_mask(acc, one) ? 1u : 0u);
}
#endif

/* n uint8 pixels (each 0 / 1) -> ceil(n/8) bytes, pixel p = bit p%8 of byte p/8; returns the OR of all pixels (anything above 1
 * means "not a binary frame"; the wide paths report 2 for that) */
static unsigned pack_bits(const uint8_t *src, uint8_t *dst, size_t n) {
  size_t p = 0;
  unsigned any = 0;
#ifdef A2C_HAVE_SSE2
  /* ONE state word, computed into a local and published last: every worker thread packs its reset frame at start-up, and a
   * thread that sees a half-written pair of flags must not take the AVX-512 path on a host without it.  0 = not probed yet,
   * else 4 | (avx512 << 1) | avx2; racing threads compute the same value. */
  static int isa_state = 0;
  int isa = __atomic_load_n(&isa_state, __ATOMIC_ACQUIRE);
  if (isa == 0) {
    const char *no512 = getenv("A2C_NO_AVX512");
    const int a2 = __builtin_cpu_supports("avx2") ? 1 : 0;
    const int a512 = (__builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512f") && !(no512 && no512[0] == '1')) ? 1 : 0;
    isa = 4 | (a512 << 1) | a2;
    __atomic_store_n(&isa_state, isa, __ATOMIC_RELEASE);
  }
  if ((isa & 2) != 0) any = pack_bits_avx512(src, dst, n, &p);
  else if ((isa & 1) != 0) any = pack_bits_avx2(src, dst, n, &p);
#endif
#ifdef A2C_HAVE_SSE2
  __m128i acc = _mm_setzero_si128();
  const unsigned any_wide = any;
  for (; p + 16 <= n; p += 16) {
    const __m128i x = _mm_loadu_si128((const __m128i *)(src + p));
    acc = _mm_or_si128(acc, x);
    const unsigned m = (unsigned)_mm_movemask_epi8(_mm_slli_epi16(x, 7));   /* bit 0 of every byte -> its sign bit */
    dst[p >> 3] = (uint8_t)m;
    dst[(p >> 3) + 1] = (uint8_t)(m >> 8);
  }
  acc = _mm_or_si128(acc, _mm_srli_si128(acc, 8));
  acc = _mm_or_si128(acc, _mm_srli_si128(acc, 4));
  acc = _mm_or_si128(acc, _mm_srli_si128(acc, 2));
  acc = _mm_or_si128(acc, _mm_srli_si128(acc, 1));
  any = any_wide | ((unsigned)_mm_cvtsi128_si32(acc) & 0xffu);
#endif
  for (; p < n; p += 8) {
    unsigned b = 0;
    for (size_t q = 0; q < 8 && p + q < n; ++q) {
      any |= src[p + q];
      b |= (unsigned)(src[p + q] & 1u) << q;
    }
    dst[p >> 3] = (uint8_t)b;
  }
  return any;
}

int a2c_pool_publish_bits(void *base, int env, const uint8_t *frame_u8, uint32_t seq, float rew, int done) {
  a2c_pool_header *h = hdr(base);
  if (h->frame_dtype != A2C_FRAME_BITS) return -1;
  uint8_t *slot = (uint8_t *)base + h->off_frames + (size_t)env * h->frame_stride;
  if (pack_bits(frame_u8, slot, h->frame_elems) > 1u) return -1;       /* not a binary frame: refuse, do not mangle */
  write_tagged(base, env, slot, seq, rew, done);
  uint32_t rb;
  memcpy(&rb, &rew, 4);
  const uint64_t g = ((uint64_t)((seq << 1) | (done ? 1u : 0u)) << 32) | rb;
  __atomic_store_n(rec_of(base) + env, g, __ATOMIC_RELEASE);
  return 0;
}

void a2c_pool_episode(void *base, double ep_rew) {
  a2c_pool_header *h = hdr(base);
  while (__atomic_exchange_n(&h->ema_lock, 1u, __ATOMIC_ACQUIRE)) cpu_relax();
  h->rew_ema = .99 * h->rew_ema + .01 * ep_rew;
  h->episodes += 1;
  __atomic_store_n(&h->ema_lock, 0u, __ATOMIC_RELEASE);
}

void a2c_pool_worker_ready(void *base) { __atomic_fetch_add(&hdr(base)->workers_ready, 1u, __ATOMIC_ACQ_REL); }
void a2c_pool_worker_failed(void *base, int worker_id) {
  __atomic_store_n(&hdr(base)->worker_error, (uint32_t)worker_id + 1u, __ATOMIC_RELEASE);
}

void a2c_pool_post_actions(void *base, int env0, int n, const int64_t *actions, int64_t stride, uint32_t seq) {
  uint64_t *c = cmd_of(base) + env0;
  for (int i = 0; i < n; ++i)
    __atomic_store_n(c + i, ((uint64_t)seq << 32) | (uint32_t)(int32_t)actions[(int64_t)i * stride], __ATOMIC_RELEASE);
}

int a2c_pool_wait_frames(void *base, int env0, int n, uint32_t seq, int64_t timeout_ns) {
  const uint64_t *r = rec_of(base) + env0;
  const int64_t t0 = now_ns();
  int i = 0;
  for (unsigned sweep = 0; i < n; ++sweep) {
    while (i < n && (uint32_t)(__atomic_load_n(r + i, __ATOMIC_ACQUIRE) >> 33) == (seq & 0x7fffffffu)) ++i;
    if (i == n) break;
    if ((sweep & 63) == 63) {
      if (__atomic_load_n(&hdr(base)->worker_error, __ATOMIC_ACQUIRE)) return -3;
      if (now_ns() - t0 > timeout_ns) return -1;
    }
    cpu_relax();
  }
  return 0;
}

void a2c_pool_unpack(const void *base, int env0, int n, float *rew, float *done) {
  const uint64_t *r = (const uint64_t *)((const char *)base + chdr(base)->off_rec) + env0;
  for (int i = 0; i < n; ++i) {
    const uint64_t g = __atomic_load_n(r + i, __ATOMIC_ACQUIRE);
    const uint32_t rb = (uint32_t)g;
    memcpy(rew + i, &rb, 4);
    done[i] = (float)((g >> 32) & 1u);
  }
}

double a2c_pool_rew_ema(const void *base) {
  a2c_pool_header *h = (a2c_pool_header *)base;
  while (__atomic_exchange_n(&h->ema_lock, 1u, __ATOMIC_ACQUIRE)) cpu_relax();
  const double v = h->rew_ema;
  __atomic_store_n(&h->ema_lock, 0u, __ATOMIC_RELEASE);
  return v;
}

/* ------------------------------------------------------------------ native env worker threads */
typedef struct {
  void *base;
  const a2c_env_vtable *vt;
  void *const *envs;
  int env0, n, shift, pong;
  uint64_t *push_rec;       /* optional mirror in device memory (a2c_pool_threads_start_push) */
  uint8_t *push_frames;
} worker_arg;

struct a2c_pool_threads {
  int n_threads;
  pthread_t *threads;
  worker_arg *args;
};

static void publish_inplace(void *base, int env, uint32_t seq, float rew, int done) {
  uint32_t rb;
  memcpy(&rb, &rew, 4);
  __atomic_store_n(rec_of(base) + env, ((uint64_t)((seq << 1) | (done ? 1u : 0u)) << 32) | rb, __ATOMIC_RELEASE);
}

static inline void wc_fence(void) {
#ifdef A2C_HAVE_SSE2
  _mm_sfence();               /* drains this core's write-combining buffers: the stores before it are on their way */
#else
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
/* the mirror in device memory: the frame (whole 16-byte pieces: the slots are 16-byte multiples), fence, the granule, fence */
static void push_frame(const worker_arg *w, const a2c_pool_header *h, int env, const void *frame) {
  uint8_t *dst = w->push_frames + (size_t)env * h->frame_stride;
  const size_t nb = ((size_t)h->frame_bytes + 15u) & ~(size_t)15u;
#ifdef A2C_HAVE_SSE2
  for (size_t q = 0; q < nb; q += 16) _mm_stream_si128((__m128i *)(dst + q), _mm_loadu_si128((const __m128i *)((const uint8_t *)frame + q)));
#else
  memcpy(dst, frame, nb);
#endif
}
static void push_rec(const worker_arg *w, int env, uint32_t seq, float rew, int done) {
  uint32_t rb;
  memcpy(&rb, &rew, 4);
  __atomic_store_n(w->push_rec + env, ((uint64_t)((seq << 1) | (done ? 1u : 0u)) << 32) | rb, __ATOMIC_RELEASE);
}
static void push_answer(const worker_arg *w, const a2c_pool_header *h, int env, const void *frame, uint32_t seq, float rew,
                        int done, int lazy) {
  if (!w->push_rec) return;
  push_frame(w, h, env, frame);
  wc_fence();
  push_rec(w, env, seq, rew, done);
  if (!lazy) wc_fence();     /* lazy: the caller has more answers to write -- the next one's fence (or its own, when it runs out of
                              * work) pushes this granule out; the kernel waits for the LAST env of the step either way */
}

/* the native tape env (below): the worker warms the frame the NEXT env of its block is about to return -- an emulator has the
 * frame it draws in its cache, a tape frame last touched `n envs` steps ago sits in L3 / DRAM (one thread serving 256 envs:
 * 1.8 MB of frames between two visits).  A2C_TAPE_WARM=0 switches it off (A/B runs). */
static const a2c_env_vtable tape_vtable;
static void tape_warm(void *env);
static int tape_warm_on(void) {
  static int on = -1;
  if (on < 0) { const char *v = getenv("A2C_TAPE_WARM"); on = !(v && v[0] == '0'); }
  return on;
}

static void *worker_main(void *p) {
  worker_arg *w = (worker_arg *)p;
  a2c_pool_header *h = hdr(w->base);
  char *frames = (char *)w->base + h->off_frames;
  prctl(PR_SET_TIMERSLACK, 1000UL, 0, 0, 0);   /* the idle-phase sleeps of a2c_pool_poll really are ~20 us, not 20 + 50 */
  uint32_t *next_seq = (uint32_t *)calloc((size_t)w->n, sizeof(uint32_t));
  double *ep_rew = (double *)calloc((size_t)w->n, sizeof(double));
  /* packed transport: the env writes its uint8 observation into this thread's scratch (cache resident), the
   * pixels are packed 8 to a byte into the pinned slot */
  const int bits = h->frame_dtype == A2C_FRAME_BITS;
  uint8_t *scratch = bits ? (uint8_t *)calloc((size_t)h->frame_elems + 64, 1) : NULL;
  int bad = 0;
  const int peek = bits && w->vt->peek != NULL;     /* pack straight from the env's own observation buffer */
  for (int i = 0; i < w->n; ++i) {          /* frame 0 = reset observation, done = 1 */
    void *slot0 = frames + (size_t)(w->env0 + i) * h->frame_stride;
    w->vt->reset(w->envs[w->env0 + i], bits ? (peek ? NULL : (void *)scratch) : slot0);
    const uint8_t *obs0 = peek ? (const uint8_t *)w->vt->peek(w->envs[w->env0 + i]) : scratch;
    if (bits && pack_bits(obs0, (uint8_t *)slot0, h->frame_elems) > 1u) bad = 1;
    if (bits) write_tagged(w->base, w->env0 + i, (const uint8_t *)slot0, h->seq_start, 0.f, 1);
    next_seq[i] = h->seq_start;
    push_answer(w, h, w->env0 + i, slot0, h->seq_start, 0.f, 1, 0);
    publish_inplace(w->base, w->env0 + i, h->seq_start, 0.f, 1);
  }
  if (bad) a2c_pool_worker_failed(w->base, w->env0);
  a2c_pool_worker_ready(w->base);
  const int warm = w->vt == &tape_vtable && tape_warm_on();
  const char *rr_env = getenv("A2C_POLL_RR");
  const int rr = !(rr_env && rr_env[0] == '0');
  int start = 0;
  /* (threads with few envs -- the headline: 17 -- push every granule out at once: their kernel proceeds env by env, and a granule
   * that waits for the next answer's fence cost the ring step 1.5 %; A2C_PUSH_LAZY_FENCE=1 / 0 forces either) */
  const char *lf_env = getenv("A2C_PUSH_LAZY_FENCE");
  const int lazy = w->push_rec != NULL && (lf_env ? lf_env[0] != '0' : w->n >= 32);
  int wc_dirty = 0;
  /* A thread that serves many envs (one thread per rank on a small host: 256 envs) is throughput-bound and the kernel waits for
   * its LAST answer: the frames of up to FB answers share ONE fence, their granules follow it.  A thread with few envs (the
   * headline: 17) answers one at a time -- its kernel proceeds env by env.  A2C_PUSH_BATCH overrides (1 = never batch). */
  enum { FBMAX = 8 };
  const char *fb_env = getenv("A2C_PUSH_BATCH");
  int FB = fb_env ? atoi(fb_env) : (w->n >= 32 ? 4 : 1);
  if (FB < 1 || !lazy) FB = 1;
  if (FB > FBMAX) FB = FBMAX;
  struct { int env; uint32_t seq; float rew; int done; } pend[FBMAX];
  int npend = 0;
  for (;;) {
    int32_t action = 0;
    int i = -1;
    if (wc_dirty || npend) {   /* one look at the env that is due next: if its action is not there yet, push what is pending out */
      const int nx = (rr && start < w->n) ? start : 0;
      if ((uint32_t)(__atomic_load_n(cmd_of(w->base) + w->env0 + nx, __ATOMIC_ACQUIRE) >> 32) == next_seq[nx]) i = nx;
      else {
        wc_fence();
        if (npend) {
          for (int q = 0; q < npend; ++q) push_rec(w, pend[q].env, pend[q].seq, pend[q].rew, pend[q].done);
          npend = 0;
          wc_fence();
        }
        wc_dirty = 0;
      }
    }
    if (i < 0) i = pool_poll_from(w->base, w->env0, w->n, next_seq, 200000000LL, rr ? start : 0);
    if (i == -2) break;
    if (i < 0) continue;
    action = a2c_pool_action(w->base, w->env0 + i);
    start = i + 1;
    const int j = w->env0 + i;
    void *pinned = frames + (size_t)j * h->frame_stride;
    void *slot = bits ? (peek ? NULL : (void *)scratch) : pinned;
    float rew = 0.f;
    int done = 0;
    w->vt->step(w->envs[j], action + w->shift, slot, &rew, &done);      /* runner.py:208 */
    ep_rew[i] += rew;
    const int reset = done != 0;
    if (w->pong && rew != 0.f) done = 1;                                 /* runner.py:212-214 */
    if (done) {                                                          /* runner.py:215-217 */
      a2c_pool_episode(w->base, ep_rew[i]);
      ep_rew[i] = 0.0;
    }
    if (reset) w->vt->reset(w->envs[j], slot);                           /* utils.py:36-38 */
    if (bits && pack_bits(peek ? (const uint8_t *)w->vt->peek(w->envs[j]) : scratch, (uint8_t *)pinned, h->frame_elems) > 1u) {
      a2c_pool_worker_failed(w->base, j);                                /* not a binary frame */
      break;
    }
    next_seq[i] += 1;
    if (bits) write_tagged(w->base, j, (const uint8_t *)pinned, next_seq[i], rew, reset);
    if (FB > 1) {                /* the device's copy first: it is the one a kernel waits for */
      push_frame(w, h, j, pinned);
      pend[npend].env = j; pend[npend].seq = next_seq[i]; pend[npend].rew = rew; pend[npend].done = reset;
      if (++npend == FB) {
        wc_fence();
        for (int q = 0; q < npend; ++q) push_rec(w, pend[q].env, pend[q].seq, pend[q].rew, pend[q].done);
        npend = 0;
        wc_dirty = 1;
      }
    } else {
      push_answer(w, h, j, pinned, next_seq[i], rew, reset, lazy);
      wc_dirty = lazy;
    }
    publish_inplace(w->base, j, next_seq[i], rew, reset);
    if (warm) tape_warm(w->envs[w->env0 + (i + 1 < w->n ? i + 1 : 0)]);
  }
  free(next_seq);
  free(ep_rew);
  free(scratch);
  return NULL;
}

a2c_pool_threads *a2c_pool_threads_start(void *base, int n_threads, const a2c_env_vtable *vt, void *const *envs,
                                         int action_shift, int pong) {
  return a2c_pool_threads_start_push(base, n_threads, vt, envs, action_shift, pong, NULL, NULL);
}

a2c_pool_threads *a2c_pool_threads_start_push(void *base, int n_threads, const a2c_env_vtable *vt, void *const *envs,
                                              int action_shift, int pong, void *push_rec, void *push_frames) {
  if (a2c_pool_check(base) || n_threads < 1 || !vt || !vt->reset || !vt->step || !envs) return NULL;
  if ((push_rec == NULL) != (push_frames == NULL) || ((uintptr_t)push_rec % 8) || ((uintptr_t)push_frames % 16)) return NULL;
  if (push_rec && (hdr(base)->frame_dtype == A2C_FRAME_F32 || hdr(base)->frame_stride % 16)) return NULL;
  a2c_pool_header *h = hdr(base);
  const int n_envs = (int)h->n_envs;
  if (n_threads > n_envs) n_threads = n_envs;
  a2c_pool_threads *t = (a2c_pool_threads *)calloc(1, sizeof(*t));
  t->threads = (pthread_t *)calloc((size_t)n_threads, sizeof(pthread_t));
  t->args = (worker_arg *)calloc((size_t)n_threads, sizeof(worker_arg));
  const int per = (n_envs + n_threads - 1) / n_threads;
  int started = 0;
  for (int w = 0; w < n_threads && w * per < n_envs; ++w) {
    worker_arg *a = &t->args[w];
    a->base = base; a->vt = vt; a->envs = envs; a->env0 = w * per;
    a->n = per < n_envs - w * per ? per : n_envs - w * per;
    a->shift = action_shift; a->pong = pong;
    a->push_rec = (uint64_t *)push_rec; a->push_frames = (uint8_t *)push_frames;
    if (pthread_create(&t->threads[w], NULL, worker_main, a) != 0) break;
    ++started;
  }
  t->n_threads = started;
  h->n_workers = (uint32_t)started;
  const int want = (n_envs + per - 1) / per;
  if (started != want) {
    a2c_pool_threads_stop(t);
    return NULL;
  }
  while (__atomic_load_n(&h->workers_ready, __ATOMIC_ACQUIRE) < (uint32_t)started) sleep_ns(100000);
  return t;
}

void a2c_pool_threads_stop(a2c_pool_threads *t) {
  if (!t) return;
  if (t->n_threads > 0) a2c_pool_set_phase(t->args[0].base, A2C_POOL_SHUTDOWN);
  for (int w = 0; w < t->n_threads; ++w) pthread_join(t->threads[w], NULL);
  free(t->threads);
  free(t->args);
  free(t);
}

/* the synthetic tape env */
typedef struct {
  unsigned char *frames;
  double *rews;
  uint8_t *dones;
  int length, frame_bytes;
  long t;
} tape_env;

void *a2c_tape_env_create(const void *frames, const double *rews, const uint8_t *dones, int length, int frame_bytes) {
  if (!frames || !rews || !dones || length < 1 || frame_bytes < 1) return NULL;
  tape_env *e = (tape_env *)calloc(1, sizeof(*e));
  e->frames = (unsigned char *)malloc((size_t)length * frame_bytes);
  e->rews = (double *)malloc((size_t)length * sizeof(double));
  e->dones = (uint8_t *)malloc((size_t)length);
  memcpy(e->frames, frames, (size_t)length * frame_bytes);
  memcpy(e->rews, rews, (size_t)length * sizeof(double));
  memcpy(e->dones, dones, (size_t)length);
  e->length = length;
  e->frame_bytes = frame_bytes;
  return e;
}

void a2c_tape_env_destroy(void *env) {
  tape_env *e = (tape_env *)env;
  if (!e) return;
  free(e->frames);
  free(e->rews);
  free(e->dones);
  free(e);
}

static void tape_reset(void *env, void *frame_out) {
  tape_env *e = (tape_env *)env;
  if (frame_out) memcpy(frame_out, e->frames + (size_t)(e->t % e->length) * e->frame_bytes, (size_t)e->frame_bytes);
}

static const void *tape_peek(void *env) {
  tape_env *e = (tape_env *)env;
  return e->frames + (size_t)(e->t % e->length) * e->frame_bytes;
}

static int tape_prefetch_on(void) {
  static int on = -1;
  if (on < 0) { const char *v = getenv("A2C_TAPE_PREFETCH"); on = !(v && v[0] == '0'); }
  return on;
}
static void tape_step(void *env, int32_t action, void *frame_out, float *rew, int *done) {
  (void)action;
  tape_env *e = (tape_env *)env;
  const long k = e->t % e->length;
  e->t += 1;
  if (frame_out) memcpy(frame_out, e->frames + (size_t)(e->t % e->length) * e->frame_bytes, (size_t)e->frame_bytes);
  *rew = (float)e->rews[k];
  *done = e->dones[k] != 0;
  /* An emulator leaves the frame it has just drawn in the cache; a tape read for the first time in `length` steps comes
   * from DRAM (0.3-0.7 us of the worker's ~1 us per env step).  Ask for the NEXT step's frame now: it is one env step
   * (>= 15 us) away.  A2C_TAPE_PREFETCH=0 switches it off (A/B runs). */
  if (tape_prefetch_on()) {
    const uint8_t *nx = e->frames + (size_t)((e->t + 1) % e->length) * e->frame_bytes;
    for (int q = 0; q < e->frame_bytes; q += 64) __builtin_prefetch(nx + q, 0, 3);
  }
}

static void tape_warm(void *env) {
  tape_env *e = (tape_env *)env;
  const uint8_t *nx = e->frames + (size_t)((e->t + 1) % e->length) * e->frame_bytes;
  for (int q = 0; q < e->frame_bytes; q += 64) __builtin_prefetch(nx + q, 0, 3);
}

static const a2c_env_vtable tape_vtable = {tape_reset, tape_step, tape_peek};
const a2c_env_vtable *a2c_tape_env_vtable(void) { return &tape_vtable; }
