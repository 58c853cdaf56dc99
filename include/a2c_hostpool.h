/*
 * a2c_hostpool.h -- C ABI of liba2c_hostpool.so: the HOST side of the rollout ingest
 * (plain C, no HIP, no torch; env worker processes load it without touching the GPU).
 *
 * Reference structure it replaces: n_envs OS processes each stepping one gym env and writing its
 * state / reward / done straight into shared tensors, one element at a time, across the
 * host-device boundary (/root/reference/a2c/training.py:109-121, runner.py:199,208,222-226).
 * Here env stepping stays on host CPUs (worker processes or threads), but every env talks to the
 * MI355X through ONE region of pinned host memory (POSIX shared memory that the GPU process
 * registers with hipHostRegister), laid out as
 *
 *     header | cmd[n_envs] | rec[n_envs] | frames[n_envs][frame_stride]
 *
 *   cmd[j]  8-byte granule written by the DEVICE (persistent rollout kernel, system-scope store) or
 *           by the GPU process after a D2H copy:   (seq << 32) | (uint32) action
 *           "perform env step number `seq` of env j with this action"  (runner.py:207-208)
 *   rec[j]  8-byte granule written by the env WORKER after the step (release store, after the
 *           frame bytes):   ((seq << 1 | done) << 32) | float_bits(reward)     [seq modulo 2^31 here: readers
 *           compare (rec >> 33) with (seq & 0x7fffffff); cmd carries seq modulo 2^32]
 *           "frame number `seq` is in frames[j]";  seq counts the env steps taken so far, frame 0 is
 *           the frame of the initial env.reset() (done = 1: the frame stack restarts, utils.py:37-42).
 *           `done` is the env's real done (the worker has already called env.reset() and stored the
 *           reset observation, utils.py:36-38); the Pong override done = (rew != 0) of
 *           runner.py:212-214 is applied on the device.
 *   frames  the prepped observation (preprocessing.py), uint8 or fp32, one slot per env; a slot is
 *           rewritten only after the device has published cmd for the next step, i.e. after it consumed it.
 * One naturally aligned 8-byte store per hand-off, data and tag in the same granule: no flag can
 * overtake its payload.  The device reads frames with system-scope (sc0 sc1) loads either directly
 * from this memory (zero-copy ingest inside a2c_a3c_rollout) or after a hipMemcpyAsync of the
 * frames block into HBM (memcpy ingest, a2c_frame_stack_push_u8 / a2c_a3c_step).
 */
#ifndef A2C_HOSTPOOL_H
#define A2C_HOSTPOOL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A2C_POOL_MAGIC 0x4132435F504F4F4CULL /* "A2C_POOL" */
#define A2C_POOL_VERSION 3
#define A2C_POOL_IDLE 0     /* between rollouts: workers sleep-poll                    */
#define A2C_POOL_ROLLOUT 1  /* a rollout is running: workers spin on their cmd granules */
#define A2C_POOL_SHUTDOWN 2
#define A2C_FRAME_U8 0
#define A2C_FRAME_F32 1
/* packed transport for BINARY preprocessors (pong_prep yields {0,1}, preprocessing.py:15-16): the worker publishes
 * one bit per pixel (pixel p = bit p%8 of byte p/8; frame_bytes = ceil(frame_elems / 8)) and the device expands the
 * bits in registers -- 8x fewer bytes over the host link, results bit-identical to the uint8 transport.  A worker
 * that meets a pixel other than 0/1 fails loudly (a2c_pool_publish_bits returns -1).                              */
#define A2C_FRAME_BITS 2

typedef struct {
  uint64_t magic;
  uint32_t version, n_envs, frame_bytes, frame_stride, frame_dtype, n_workers;
  uint64_t off_cmd, off_rec, off_frames, total_bytes;
  volatile uint32_t phase;          /* A2C_POOL_*                                            */
  volatile uint32_t workers_ready;  /* workers that have published frame 0 of all their envs  */
  volatile uint32_t worker_error;   /* 1 + id of a worker that died with an exception         */
  volatile uint32_t ema_lock;
  volatile double rew_ema;          /* runner.py:216: .99*ema + .01*episode reward            */
  volatile uint64_t episodes;       /* finished episodes (all envs)                           */
  uint32_t frame_elems;             /* pixels / values per frame (A2C_FRAME_BITS: frame_bytes = ceil(frame_elems/8)) */
  uint32_t seq_start;               /* step number of frame 0 (0 unless a2c_pool_set_seq_start was called)     */
  /* optional SELF-VALIDATING mirror of the packed frames (a2c_pool_enable_tagged; 0 = none), tagged_stride bytes per env:
   * tagged_chunks chunks of 16 bytes, each written with ONE aligned 16-byte store.  Chunk c < tagged_chunks - 1 =
   * 14 bytes of packed pixels [112c, 112c + 112) + uint16 tag; the last chunk = {float reward, uint32 done, uint32 seq,
   * uint16 0, uint16 tag}; tag = seq & 0xffff of the env step that produced the frame.  A reader that loads the chunks
   * in ANY order and finds every tag equal to the step it waits for holds a complete frame and its record: the poll of
   * rec[j] and the fetch of frames[j] (two dependent PCIe round trips) become ONE 16-byte load per lane.            */
  uint64_t off_tagged;
  uint32_t tagged_stride, tagged_chunks;
} a2c_pool_header;

/* bytes of the region for n_envs envs with frame_bytes per frame (page aligned)             */
size_t a2c_pool_bytes(int n_envs, int frame_bytes);
/* the same plus room for the self-validating mirror (A2C_FRAME_BITS pools)                   */
size_t a2c_pool_bytes_tagged(int n_envs, int frame_bytes, uint32_t frame_elems);
/* after a2c_pool_init + a2c_pool_set_frame_elems on a region of region_bytes >= a2c_pool_bytes_tagged(...): lay the
 * mirror out behind the frames; the workers then keep it up to date (a2c_pool_publish_bits and the native worker
 * threads write it before they publish rec).  0, or -1 (not a bits pool / region too small)                      */
int a2c_pool_enable_tagged(void *base, size_t region_bytes);
/* format a zero-filled region (GPU process, before the workers attach); 0 or -1             */
int a2c_pool_init(void *base, size_t bytes, int n_envs, int frame_bytes, int frame_dtype,
                  int n_workers, double rew_ema0);
/* optional, after a2c_pool_init and before the workers attach: pixels per frame (needed for A2C_FRAME_BITS; the
 * default is frame_bytes / sizeof(element), 8 * frame_bytes for bits) and the step number the envs start counting
 * from (tests of the 31-bit wrap of the rec granule's step counter)                                            */
void a2c_pool_set_frame_elems(void *base, uint32_t frame_elems);
void a2c_pool_set_seq_start(void *base, uint32_t seq_start);
/* 0 when base holds a formatted region                                                      */
int a2c_pool_check(const void *base);
void a2c_pool_set_phase(void *base, uint32_t phase);
uint32_t a2c_pool_phase(const void *base);

/* ---- worker side ---------------------------------------------------------------------- */
/* Spin until one of the envs env0..env0+n-1 has cmd.seq == next_seq[i]; returns i, or -1 after
 * spin_ns without one, or -2 when the phase is SHUTDOWN.  While the phase is IDLE the call sleeps
 * between polls instead of spinning.                                                         */
int a2c_pool_poll(void *base, int env0, int n, const uint32_t *next_seq, int64_t spin_ns);
int32_t a2c_pool_action(const void *base, int env);
/* a2c_pool_poll + a2c_pool_action in one call: *action = the action of the env returned              */
int a2c_pool_take(void *base, int env0, int n, const uint32_t *next_seq, int64_t spin_ns,
                  int32_t *action);
/* frames[env] = frame (frame_bytes), then rec[env] = {seq, done, rew} with release order     */
void a2c_pool_publish(void *base, int env, const void *frame, uint32_t seq, float rew, int done);
/* A2C_FRAME_BITS pools: pack the frame_elems uint8 pixels of `frame_u8` (each 0 or 1) into the env's slot, then
 * publish rec like a2c_pool_publish.  Returns 0, or -1 (nothing published) when a pixel is neither 0 nor 1.   */
int a2c_pool_publish_bits(void *base, int env, const uint8_t *frame_u8, uint32_t seq, float rew, int done);
/* episode finished with total reward ep_rew: ema = .99*ema + .01*ep_rew under a spin lock    */
void a2c_pool_episode(void *base, double ep_rew);
void a2c_pool_worker_ready(void *base);
void a2c_pool_worker_failed(void *base, int worker_id);

/* ---- GPU-process side (memcpy ingest: the host relays actions and records) -------------- */
/* cmd[env0+i] = (seq << 32) | actions[i*stride]                                              */
void a2c_pool_post_actions(void *base, int env0, int n, const int64_t *actions, int64_t stride,
                           uint32_t seq);
/* wait until rec.seq == seq for all envs env0..env0+n-1: 0, or -1 on timeout, -3 on worker error */
int a2c_pool_wait_frames(void *base, int env0, int n, uint32_t seq, int64_t timeout_ns);
/* unpack rec[env0..] into rew[i], done[i] (floats, e.g. a pinned staging buffer)             */
void a2c_pool_unpack(const void *base, int env0, int n, float *rew, float *done);
double a2c_pool_rew_ema(const void *base);

/* ---- native env workers (threads of the GPU process) ------------------------------------
 * For envs implemented in C: n_threads pthreads inside this process play the worker loop above
 * (a2c_pool_take -> step -> reset on done -> publish) without Python.  The env writes its observation
 * STRAIGHT into its pinned frame slot (no staging copy).  Python gym envs use worker processes
 * instead (a2c_amd.hostpool_worker); both speak the same protocol to the device.               */
typedef struct {
  /* write the observation of a fresh episode into frame_out */
  void (*reset)(void *env, void *frame_out);
  /* one env step: observation into frame_out, reward and real done; on done the pool calls reset() next */
  void (*step)(void *env, int32_t action, void *frame_out, float *rew, int *done);
  /* optional (may be NULL): pointer to the observation the last reset()/step() produced, valid until the next call.
   * A pool with the packed transport then calls reset()/step() with frame_out == NULL and packs straight from this
   * pointer into the pinned slot (no staging copy).                                                        */
  const void *(*peek)(void *env);
} a2c_env_vtable;
typedef struct a2c_pool_threads a2c_pool_threads;
/* starts the threads; they publish frame 0 of every env before the call returns.  NULL on error.  */
a2c_pool_threads *a2c_pool_threads_start(void *base, int n_threads, const a2c_env_vtable *vt,
                                         void *const *envs, int action_shift, int pong);
/* the same, and every answer is ALSO written to a second place the caller names (a2c_push_buffer_alloc: device memory
 * mapped into this process): frame j at push_frames + j * frame_stride (the pool's stride), then -- behind an sfence,
 * the mapping is write-combining -- its rec granule at push_rec[j] and a second sfence.  A thread that serves 32 envs
 * or more (throughput-bound: the kernel waits for its LAST answer) drops the second fence -- the granule is pushed out
 * by the next answer's fence, or by a fence of the worker's own as soon as it finds no action waiting
 * (A2C_PUSH_LAZY_FENCE=0 / 1 forces either) -- and writes the frames of up to four answers before one fence, their
 * granules behind it (A2C_PUSH_BATCH=1: one at a time).  A frame is never visible after its granule.
 * NULL pointers = no mirror. */
a2c_pool_threads *a2c_pool_threads_start_push(void *base, int n_threads, const a2c_env_vtable *vt,
                                              void *const *envs, int action_shift, int pong,
                                              void *push_rec, void *push_frames);
/* sets the phase to SHUTDOWN, joins the threads, frees the handle                                */
void a2c_pool_threads_stop(a2c_pool_threads *h);

/* The synthetic benchmark env as a native env: replays a tape of `length` frames /
 * rewards / dones (copied from the caller's arrays; a2c_amd.synthetic.TapeEnv generates them), ignoring
 * the action -- step k returns frame (k+1) % length, reward[k % length], done[k % length].       */
void *a2c_tape_env_create(const void *frames, const double *rews, const uint8_t *dones, int length,
                          int frame_bytes);
void a2c_tape_env_destroy(void *env);
const a2c_env_vtable *a2c_tape_env_vtable(void);

#ifdef __cplusplus
}
#endif
#endif /* A2C_HOSTPOOL_H */
