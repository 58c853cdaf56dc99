/*
 * a2c_mi355x.h -- C ABI of liba2c_mi355x.so: the MI355X (gfx950 / CDNA4) kernels of
 * the A2C rollout+update hot path of grantsrb/PyTorch-A2C.
 *
 * The reference has no FFI layer (it is pure Python on torch); its boundary is the
 * Python API of a2c/runner.py, a2c/updater.py, a2c/models.py and a2c/utils.py.
 * Each entry point below replaces the arithmetic of one reference call site
 * (cited as file:line into /root/reference/a2c/).  The Python package
 * pytorch-a2c_amd/a2c_amd binds these with ctypes and mirrors the reference API.
 *
 * Conventions
 *   - plain C, no torch types: raw DEVICE pointers into caller-owned memory, sizes,
 *     and the HIP stream (hipStream_t passed as void*; NULL = default stream);
 *   - no allocation, no ownership transfer, no host synchronisation, no global
 *     state: every call only enqueues kernels on `stream` (graph-capturable);
 *     scratch memory is passed in as `ws` / `ws_bytes` and sized by the
 *     matching *_ws_bytes() query;
 *   - all floating-point data is IEEE fp32, row-major, contiguous unless a
 *     leading dimension / stride argument says otherwise; actions are int64;
 *   - return value: 0 = A2C_OK, negative = error (a2c_error_string()).
 */
#ifndef A2C_MI355X_H
#define A2C_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define A2C_OK 0
#define A2C_ERR_ARG (-1)      /* invalid argument (NULL pointer, bad size, unsupported shape) */
#define A2C_ERR_LAUNCH (-2)   /* hipGetLastError() after the launch was not hipSuccess          */
#define A2C_ERR_WORKSPACE (-3)/* ws_bytes smaller than the matching *_ws_bytes() query          */

typedef void *a2c_stream_t;   /* hipStream_t */

int a2c_version(void);                 /* ABI version, currently 2 */
const char *a2c_error_string(int code);

/* ------------------------------------------------------------------ a5 / a7: scans
 * utils.discount (utils.py:63-79): y[i] = x[i] + g * (dones[i]==1 ? 0 : y[i+1]),
 * evaluated strictly sequentially per row (fp32 multiply then fp32 add, no FMA), so the
 * result is bit-identical to the reference.  The input is n_seg independent rows of
 * length T (row r = elements [r*T, (r+1)*T)); the running sum starts at 0 at each row end.
 * With n_seg == 1 this is exactly utils.discount on a flat array of T elements.
 * When n_seg > 1 the rows are only independent if every row ends with dones == 1
 * (Runner guarantees it: runner.py:223,244).  With err_flag (device int) given, a violation
 * sets it to 1 AND the call recomputes the array as one flat row on the device, so the
 * result is the reference's for any data; with err_flag == NULL the rows are taken as
 * independent without a check.                                                         */
int a2c_discount_scan(const float *x, const float *dones, float *y, int64_t n_seg,
                      int64_t T, float g, int *err_flag, a2c_stream_t stream);
/* updater.py:70-71 + 86-88 fused: advs = discount(deltas, dones, g_adv) and
 * rets = discount(rewards, dones, g_ret) in one pass over the three inputs.            */
int a2c_gae_returns_fused(const float *deltas, const float *rewards, const float *dones,
                          float *advs, float *rets, int64_t n_seg, int64_t T, float g_adv,
                          float g_ret, int *err_flag, a2c_stream_t stream);
/* The three scalar reductions of the update (a2c_moments, a2c_loss_fwd_bwd's loss sums, a2c_gradnorm_sq) are DETERMINISTIC:
 * per-workgroup fp64 partials go to `reduce_scratch` and the last workgroup to finish adds them in workgroup order (no
 * atomics on the sums).  reduce_scratch: A2C_REDUCE_SCRATCH_DOUBLES doubles of device memory, ZERO before its first use
 * (word 0 is a ticket counter the kernels reset); calls that share one scratch must be ordered (same stream).           */
#define A2C_REDUCE_SCRATCH_DOUBLES (8 + 3 * 1024)
/* sums[0] = sum x, sums[1] = sum x*x in double; feeds the
 * mean / unbiased-std normalisations of updater.py:89-98.                              */
int a2c_moments(const float *x, int64_t n, double *sums, double *reduce_scratch, a2c_stream_t stream);
/* y = (x - mean) / (std + eps), mean/std (unbiased) derived from sums (a2c_moments,
 * possibly all-reduced across ranks) over n_global elements: updater.py:98.            */
int a2c_normalize(const float *x, float *y, int64_t n, const double *sums, int64_t n_global,
                  float eps, a2c_stream_t stream);
/* y = a + b (returns = advs + vals.data, updater.py:84)                                */
int a2c_add(const float *a, const float *b, float *y, int64_t n, a2c_stream_t stream);

/* ------------------------------------------------------------------ a1: frame stack
 * utils.next_state (utils.py:26-43) + the state write of runner.py:199, batched over B
 * envs.  out[b] = reset_mask[b] ? [0,..,0, frame_new[b]] : [prev[b][1:], frame_new[b]];
 * C = n_frame_stack planes of HW floats.  prev / out are addressed with a per-env stride
 * in floats so they can point into the rollout buffer `states[idx*T + t]`.              */
int a2c_frame_stack_push(const float *frame_new, const float *reset_mask, const float *prev,
                         int64_t prev_stride, float *out, int64_t out_stride, int B, int C,
                         int HW, a2c_stream_t stream);
/* The same with the new frame as UINT8 pixels (the host pool's transport format: pong_prep,
 * preprocessing.py:11-17, yields uint8; runner.py:199 `FloatTensor(state)` is the expansion):
 * frame_u8[b*frame_stride + i], i < HW, frame_stride in bytes (% 4 == 0); HW % 4 == 0.     */
int a2c_frame_stack_push_u8(const uint8_t *frame_u8, int64_t frame_stride, const float *reset_mask,
                            const float *prev, int64_t prev_stride, float *out, int64_t out_stride,
                            int B, int C, int HW, a2c_stream_t stream);

/* ------------------------------------------------------------------ g1: device relay of the host env pool
 * (a2c_hostpool.h) for the per-step rollout of the models without a persistent kernel.  Replaces, per env step, the
 * host round trip of the memcpy ingest -- D2H of the actions, a2c_pool_post_actions, a2c_pool_wait_frames,
 * a2c_pool_unpack, H2D of the frames block (runner.py:199,207-226 of the reference did the same hand-off one
 * element at a time) -- by two launches that stay on the stream:
 *   a2c_pool_publish_actions   cmd[i] = ((seq_base[0] + seq_off) << 32) | actions[i*act_stride], i < n
 *                              (system-scope 8-byte stores into the device-mapped pinned region)
 *   a2c_pool_ingest            waits until rec[i].seq == seq_base[0] + seq_off (every env, bounded by
 *                              timeout_ticks of the 100 MHz wall clock; on expiry *err = 1 and this and all later
 *                              calls return without waiting), then rew[i], done[i] <- rec[i] and
 *                              frames_out[i*out_stride ..] <- frames[i*frame_stride .. +frame_bytes] (16-byte
 *                              system-scope loads over PCIe).  frame_bytes, strides: multiples of 16.
 * cmd / rec / frames are the DEVICE addresses of the pool's granules of the first env of the block
 * (hipHostGetDevicePointer of the registered region); seq_base lives in device memory so that a captured hipGraph
 * of a segment can be replayed by later rollouts.                                               */
int a2c_pool_publish_actions(uint64_t *cmd, const int64_t *actions, int64_t act_stride, int n,
                             const uint32_t *seq_base, uint32_t seq_off, a2c_stream_t stream);
int a2c_pool_ingest(const uint64_t *rec, const uint8_t *frames, int64_t frame_stride, int frame_bytes, int n,
                    const uint32_t *seq_base, uint32_t seq_off, int64_t timeout_ticks, int *err, float *rew,
                    float *done, uint8_t *frames_out, int64_t out_stride, a2c_stream_t stream);
/* The same for a pool with the PACKED transport (A2C_FRAME_BITS in a2c_hostpool.h: binary preprocessors such as
 * pong_prep, preprocessing.py:15-16, publish one bit per pixel): n_pixels / 8 bytes per env cross the host link and
 * frames_out receives the n_pixels uint8 {0,1} pixels the uint8 kernels expect (n_pixels % 16 == 0).
 * a2c_unpack_bits is the expansion alone, for packed frames a hipMemcpyAsync staged in HBM (memcpy ingest).    */
int a2c_pool_ingest_bits(const uint64_t *rec, const uint8_t *frames, int64_t frame_stride, int n_pixels, int n,
                         const uint32_t *seq_base, uint32_t seq_off, int64_t timeout_ticks, int *err, float *rew,
                         float *done, uint8_t *frames_out, int64_t out_stride, a2c_stream_t stream);
int a2c_unpack_bits(const uint8_t *src, int64_t src_stride, uint8_t *dst, int64_t dst_stride, int n,
                    int n_pixels, a2c_stream_t stream);
/* a2c_pool_ingest / a2c_pool_ingest_bits (packed_bits != 0; frame_elems = uint8 pixels per frame) whose workgroup for env b
 * also does env b's share of a2c_rollout_post_frames for the env step the answer belongs to (runner.py:208-232: rewards /
 * dones / the TD residual of the step before, the valid-plane count of the next state, and for recurrent nets the hidden row
 * of the next step): the same values as the two calls in sequence, one launch.  Arguments after out_stride as in
 * a2c_rollout_post_frames (val = the value head's output of the state the action was sampled from).            */
int a2c_pool_ingest_post(int packed_bits, const uint64_t *rec, const uint8_t *frames, int64_t frame_stride,
                         int frame_elems, int n, const uint32_t *seq_base, uint32_t seq_off,
                         int64_t timeout_ticks, int *err, float *rew, float *done, uint8_t *frames_out,
                         int64_t out_stride, const float *val, int64_t val_stride, float *val_prev,
                         float *rewards, float *dones, float *deltas, int64_t T, int64_t t, int64_t slot0,
                         float gamma, int pong, float *done_eff, float *h, int hdim, float *h_rows,
                         int64_t h_rows_stride, const float *h_src, int *nvalid_rows, int *nvalid_carry,
                         a2c_stream_t stream);

/* ------------------------------------------------------------------ f4: preprocessing on the device, single-frame store
 * a2c_frame_prep_u8 = pong_prep / breakout_prep (preprocessing.py:11-23) on n raw (H, W, C) uint8 frames, raw_stride
 * bytes apart: out[oy][ox] = raw[y0 + oy*step][x0 + ox*step][0] for the crop y0:y1, x0:x1; binarise != 0 applies
 * pong_prep's "144 and 109 -> 0, everything else that is not 0 -> 1" (preprocessing.py:14-16); binarise == 0 keeps the
 * byte (breakout_prep: skimage <= 0.18's rgb2grey returns a 2-D slice unchanged).  pong_prep = (35, 195, 0, W, 2, 1),
 * breakout_prep = (35, 195, 8, W - 8, 2, 0).  Output rows out_stride bytes apart, OW % 4 == 0.                      */
int a2c_frame_prep_u8(const uint8_t *raw, int64_t raw_stride, int H, int W, int C, int y0, int y1, int x0, int x1,
                      int step, int binarise, uint8_t *out, int64_t out_stride, int n, a2c_stream_t stream);
/* Single-frame uint8 store (utils.py:26-43, runner.py:199 behind the boundary): slot r keeps T + C frames of HW bytes,
 * slot_stride bytes apart; state (r, t) = the window frames[r][t .. t+C-1], planes c < C - nvalid are zero (frames from
 * before the env's last reset).  a2c_frame_store_begin: start of a slot -- frames[r][0..C-1] = frames[r][T..T+C-1]
 * (the state the previous slot ended in), nvalid_rows[r*T] = nvalid_carry[r].  a2c_rollout_post_frames: the bookkeeping
 * of a2c_rollout_post[_rec] for env step t WITHOUT a frame-stack copy (the ingest wrote the new frame straight into
 * frames[r][t+C]): rewards / dones / deltas / val_prev [/ hidden-state reset and h_states row], and the valid-plane
 * count of state t+1 (1 after a real done, else min(count + 1, C)) into nvalid_rows[(slot0+b)*T + t+1] / nvalid_carry.
 * a2c_frames_to_states: the reference's fp32 `states` rows from the store, on demand: state k (< nt) of slot r (< R)
 * -> out + r*out_slot_stride + k*C*HW floats, valid-plane counts nvalid[r*nvalid_slot_stride + k] (NULL: all valid). */
int a2c_frame_store_begin(uint8_t *frame_store, int64_t slot_stride, int64_t T, int C, int HW, int32_t *nvalid_rows,
                          const int32_t *nvalid_carry, int B, a2c_stream_t stream);
int a2c_rollout_post_frames(const float *rew, const float *done, const float *val, int64_t val_stride, float *val_prev,
                            float *rewards, float *dones, float *deltas, int64_t T, int64_t t, int64_t slot0,
                            float gamma, int pong, int B, float *done_eff, float *h, int hdim, float *h_rows,
                            int64_t h_rows_stride, const float *h_src, int32_t *nvalid_rows, int32_t *nvalid_carry,
                            a2c_stream_t stream);
int a2c_frames_to_states(const uint8_t *frame_store, int64_t slot_stride, const int32_t *nvalid,
                         int64_t nvalid_slot_stride, float *out, int64_t out_slot_stride, int R, int nt, int C, int HW,
                         a2c_stream_t stream);

/* ------------------------------------------------------------------ a2: sampler
 * SequentialEnvironment.get_action discrete branch (runner.py:94-97) + utils.sample_action
 * (utils.py:45-60): p = softmax(logits); running fp32 cumsum in index order; first a with
 * cumsum >= u[b]; -1 if none.  actions int64 with element stride act_stride;
 * probs (B,A) optional (may be NULL).                                                   */
int a2c_softmax_sample(const float *logits, int64_t ld_logits, const float *u, int64_t *actions,
                       int64_t act_stride, float *probs, int B, int A, a2c_stream_t stream);
/* utils.sample_action on given probabilities (no softmax); actions as fp32 like the
 * reference's return value.                                                             */
int a2c_sample_probs(const float *probs, const float *u, float *actions, int64_t B, int A,
                     a2c_stream_t stream);

/* ------------------------------------------------------------------ a3: rollout records
 * One env-step of Runner.rollout bookkeeping for B slots (runner.py:212-232): with
 * e = slot*T + t, writes rewards[e] = rew, dones[e] = done_eff where
 * done_eff = done || (pong && rew != 0) (runner.py:213-214), and, when t > 0,
 * deltas[e-1] = rewards[e-1] + (gamma*val[b])*(1-dones[e-1]) - val_prev[b] (runner.py:231);
 * then val_prev[b] = val[b].  If h != NULL (recurrent, hdim columns) rows of h whose
 * done_eff is set are zeroed (runner.py:219-220).                                       */
int a2c_rollout_record(const float *rew, const float *done, const float *val, int64_t val_stride,
                       float *val_prev,
                       float *rewards, float *dones, float *deltas, float *done_eff_out,
                       float *h, int hdim, int B, int64_t T, int64_t t, int64_t slot0,
                       float gamma, int pong, a2c_stream_t stream);
/* a2c_rollout_record (feed-forward nets: no hidden state) and a2c_frame_stack_push of the same
 * env step in ONE launch (one hipGraph node fewer per rollout step).                     */
int a2c_rollout_post(const float *rew, const float *done, const float *val, int64_t val_stride,
                     float *val_prev, float *rewards, float *dones, float *deltas, int64_t T,
                     int64_t t, int64_t slot0, float gamma, int pong, const float *frame_new,
                     const float *reset_mask, const float *prev, int64_t prev_stride, float *out,
                     int64_t out_stride, int B, int C, int HW, a2c_stream_t stream);
int a2c_rollout_post_u8(const float *rew, const float *done, const float *val, int64_t val_stride,
                        float *val_prev, float *rewards, float *dones, float *deltas, int64_t T,
                        int64_t t, int64_t slot0, float gamma, int pong, const uint8_t *frame_u8,
                        int64_t frame_stride, const float *reset_mask, const float *prev,
                        int64_t prev_stride, float *out, int64_t out_stride, int B, int C, int HW,
                        a2c_stream_t stream);
/* The same for recurrent nets, with the hidden-state handling of the step folded into the launch (runner.py:201,
 * 219-221): h[b] = 0 where the env step ended an episode (done, or the Pong override), done_eff_out[b] = that flag,
 * and, when h_rows != NULL, h_rows[b*h_rows_stride ..] = h[b] (the h_states row of the step that follows).
 * h_src (NULL = h): where the previous step's cell left its new hidden rows (B, hdim); h = masked h_src, so a cell
 * that writes h_new somewhere else (e.g. into the update's time-major buffers) needs no copy back.
 * Exactly one of frame_new (fp32) / frame_u8 is given.                                          */
int a2c_rollout_post_rec(const float *rew, const float *done, const float *val, int64_t val_stride,
                         float *val_prev, float *rewards, float *dones, float *deltas, int64_t T,
                         int64_t t, int64_t slot0, float gamma, int pong, const float *frame_new,
                         const uint8_t *frame_u8, int64_t frame_stride, const float *reset_mask,
                         const float *prev, int64_t prev_stride, float *out, int64_t out_stride, int B,
                         int C, int HW, float *done_eff_out, float *h, int hdim, float *h_rows,
                         int64_t h_rows_stride, const float *h_src, a2c_stream_t stream);
/* End of slot (runner.py:236-245): e = slot*T + T-1; if dones[e] == 0:
 * rewards[e] += gamma*val_boot[b], dones[e] = 1; then deltas[e] = rewards[e] - val_prev[b]. */
int a2c_rollout_bootstrap(const float *val_boot, int64_t val_stride, const float *val_prev, float *rewards,
                          float *dones, float *deltas, int B, int64_t T, int64_t slot0,
                          float gamma, a2c_stream_t stream);
/* One whole rollout step of the A3CModel-shaped policy (models.py:60-90: conv 8x8/s4 -> 16,
 * conv 4x4/s2 -> 32, proj_matrx WITHOUT activation, pi/value heads) in ONE launch, one workgroup
 * per env; replaces, per step of Runner.rollout (runner.py:190-232):
 *   1. (rew != NULL) a2c_rollout_record of the env step that produced this state: step index
 *      t_rec, value read from heads[b*ldh + n_actions] as left by the previous call;
 *   2. the state: frame_new != NULL: a2c_frame_stack_push(prev, frame_new, reset_mask);
 *      frame_new == NULL: the rows of prev as they are.  If out != NULL the state is also written
 *      to out[b*out_stride ..] (the states row of the rollout buffer / the bookmark);
 *   3. the forward pass on that state with the weights prepared for inference:
 *      wfrag1/wfrag2 = a2c_conv2d_prep_weights(kind 0) of the two conv layers,
 *      Wc ((n_actions+1) x F) = [pi.weight; value.weight] . proj_matrx.weight, bc likewise;
 *      heads[b*ldh + 0..n_actions] = [logits | value];
 *   4. (u != NULL) a2c_softmax_sample of the logits into actions[b*act_stride];
 *   5. (bootstrap != 0, t_rec == T-1) a2c_rollout_bootstrap with the value just computed
 *      (runner.py:236-245).
 * Requires C == 4, W % 4 == 0 and the state + activations to fit one CU's LDS
 * (a2c_a3c_step_supported); other shapes use the per-layer entry points.               */
typedef struct {
  int B, C, H, W, n_actions;
  const float *prev; int64_t prev_stride;
  const float *frame_new;          /* (B, H*W) or NULL                                   */
  const float *reset_mask;         /* (B,) or NULL                                       */
  float *out; int64_t out_stride;  /* or NULL                                            */
  const float *wfrag1, *bias1, *wfrag2, *bias2, *Wc, *bc;
  float *heads; int64_t ldh;
  const float *u;                  /* (B,) uniforms or NULL (no sampling)                */
  int64_t *actions; int64_t act_stride;
  const float *rew, *done;         /* (B,) of env step t_rec, or NULL (no bookkeeping)   */
  float *val_prev, *rewards, *dones, *deltas;
  int64_t T, t_rec, slot0;
  float gamma;
  int pong, bootstrap;
  const uint8_t *frame_u8;         /* instead of frame_new: uint8 frame (B, frame_stride bytes), */
  int64_t frame_stride;            /* e.g. the pool's frames block after its H2D copy            */
  /* optional stash of the conv activations of this state (post-ReLU, NCHW): a1 (16, OH1, OW1) and a2
   * (32, OH2, OW2) of env b go to a1_out + b*a1_stride / a2_out + b*a2_stride.  The weights do not change
   * between a rollout and the update that consumes it (training.py:150-165), so Updater.update_model's
   * forward (updater.py:80) would recompute exactly these tensors: it reads the stash instead.           */
  float *a1_out; int64_t a1_stride;
  float *a2_out; int64_t a2_stride;
  float *heads_out; int64_t heads_out_stride;   /* optional second copy of [logits | value] (same stash) */
} a2c_a3c_step_args;
int a2c_a3c_step_supported(int C, int H, int W, int n_actions);
int a2c_a3c_step(const a2c_a3c_step_args *args, a2c_stream_t stream);

/* A WHOLE rollout slot of the A3CModel-shaped policy -- all n_tsteps iterations of the loop of
 * Runner.rollout (runner.py:198-232) plus the bootstrap (runner.py:236-245) for B envs -- in ONE
 * persistent launch that talks to the host env workers directly through the pinned pool region
 * (include/a2c_hostpool.h), with no host code and no kernel boundary between steps:
 *   iteration t of env b (one workgroup per env, workgroups of several envs take turns):
 *     t > 0: wait for rec[env0+b].seq == seq0+t, read reward/done from the granule and the uint8
 *            frame straight from the pinned frames slot (system-scope loads over PCIe), record env
 *            step t-1 exactly like a2c_a3c_step;  t == 0: the state is bookmark[b];
 *     state -> states[(slot0+b)*T + t] (t == T: -> bookmark[b]); forward; 
 *     t < T: sample with u[t*u_stride + b] -> actions[(slot0+b)*T + t], and publish
 *            cmd[env0+b] = (seq0+t) << 32 | action with one 8-byte system-scope store: the worker
 *            that owns the env steps it as soon as it sees the granule;  t == T: bootstrap.
 * Results are identical to T+1 calls of a2c_a3c_step (see conv1_weight below for the one exception).  Every wait on the host is bounded by
 * timeout_ticks (100 MHz ticks); on a timeout *err is set to 1 and the workgroup stops.
 * Shapes as a2c_a3c_step_supported, uint8 frames, (H*W) % 16 == 0.                          */
typedef struct {
  int B, C, H, W, n_actions;
  float *states;                   /* rollout buffer, rows of S = C*H*W floats                */
  float *bookmark;                 /* (B, S)                                                  */
  const float *wfrag1, *bias1, *wfrag2, *bias2, *Wc, *bc;
  float *heads; int64_t ldh;
  const float *u; int64_t u_stride;
  int64_t *actions;
  float *val_prev, *rewards, *dones, *deltas;
  int64_t T, slot0;
  float gamma;
  int pong;
  uint64_t *cmd;                   /* device-mapped pool arrays (a2c_rollout_buffer_create)   */
  const uint64_t *rec;
  const uint8_t *frames; int64_t frame_stride;
  uint32_t seq0;                   /* env steps every env has taken before this slot          */
  int env0;                        /* pool index of env b is env0 + b                         */
  int *err;                        /* device int                                              */
  int64_t timeout_ticks;
  float *a1_rows, *a2_rows;        /* optional activation stash, row e = (slot0+b)*T + t (see a2c_a3c_step_args) */
  float *heads_rows; int64_t heads_rows_ld;   /* optional [logits | value] of state e, row stride heads_rows_ld */
  /* optional single-frame uint8 store (row f4 of SURVEY.md section 8): slot r keeps T+4 frames of H*W bytes,
   * frame_store_slot_stride bytes apart; the 4 planes of state (r, t) are the CONTIGUOUS window
   * frames[r][t .. t+3] (the newest frame of state t is frames[r][t+3]; frames[r][0..3] are copied from
   * frames[r][T..T+3], the state the previous slot ended in).  nvalid_rows[e] = how many of the 4 planes of state
   * e are real (planes older than the env's last reset are zero, utils.py:37-42); nvalid_carry[b] carries that
   * count of the bookmark state from slot to slot.  Consumer: a2c_conv2d_bwd_weight_frames.              */
  uint8_t *frame_store; int64_t frame_store_slot_stride;
  int32_t *nvalid_rows, *nvalid_carry;
  int frame_bits;                  /* 1: `frames` holds the packed transport (one bit per pixel, frame_stride >= H*W/8) */
  /* optional: the first conv layer's weight tensor (16, 4, 8, 8) as the model stores it.  When given,
   * every env gets a workgroup that keeps its state in LDS for the whole slot (the "ring" kernel) and
   * overlaps the env worker's turn-around with the part of the next forward that does not depend on the new frame
   * (conv1's sum is then ordered plane-major: results equal a2c_a3c_step's up to fp32 re-association).  More envs than
   * CUs: ceil(B / CUs) such launches one after the other, launch k playing the envs b = k (mod that count); with
   * A2C_RING_BLOCKS=0 in the environment ONE launch of the per-step body instead, whose workgroups take several envs in
   * turns (results identical to a2c_a3c_step).                                                                   */
  const float *conv1_weight;
  /* with frame_store, ring kernel only: 1 = do NOT write the fp32 `states` rows (a2c_frames_to_states expands them from
   * the store on demand; the update reads the store: a2c_conv2d_bwd_weight_frames); the bookmark is always written.
   * A2C_ERR_ARG when the launch cannot run as the ring kernel (no conv1_weight, A2C_NO_RING=1, or B > CU count with
   * A2C_RING_BLOCKS=0).                                                                                            */
  int states_lazy;
  /* optional, ring kernel with frame_bits: the pool's SELF-VALIDATING mirror of the packed frames (a2c_hostpool.h:
   * tagged_chunks chunks of 16 bytes per env, tagged_stride bytes apart, every chunk carries the step number's low 16
   * bits).  One wave then fetches poll + frame + reward/done of an env step with ONE 16-byte load per lane -- one PCIe
   * round trip instead of two dependent ones (rec granule, then frame) -- and re-tries until every tag matches.    */
  const uint8_t *tagged; int64_t tagged_stride; int tagged_chunks;
  /* optional, ring kernel only, with a1_rows: the MASK BITS ("lane masks") of the a1 stash rows -- (16*OH1*OW1)/64 64-bit
   * words per state, row e = (slot0+b)*T + t like a1_rows: bit (i & 7) of byte (i >> 3) of the row = (a1[i] > 0), i the flat
   * (c, y, x) index.  Consumer: a2c_conv2d_bwd_data_lanemask (the update's conv2 backward-data reads 800 B per sample instead
   * of the 25.6 KB activation row as its ReLU mask, updater.py:128's autograd).  A2C_ERR_ARG when the launch cannot run as
   * the ring kernel (ask a2c_a3c_ring_supported first) or 16*OH1*OW1 is not a multiple of 256.                         */
  uint64_t *a1_lanemask_rows;
  /* ... and, with a2_rows, the mask bits of the a2 stash rows: F/8 bytes per state (F = 32*OH2*OW2, a multiple of 8), bit
   * (i & 7) of byte (i >> 3) = (a2[i] > 0).  Consumer: a2c_small_n_bwd_data_bits.  Ring kernel only, like the field above. */
  uint8_t *a2_maskbit_rows;
} a2c_a3c_rollout_args;
int a2c_a3c_rollout(const a2c_a3c_rollout_args *args, a2c_stream_t stream);
/* 1 when a2c_a3c_rollout with these shapes, this many envs and this conv1_weight pointer runs the ring kernel (the only
 * body that honours states_lazy: runner.py:199's fp32 row left out), 0 when it would fall back to the per-step persistent
 * body -- same predicate as the launcher's (LDS budget, conv1 output <= 512 pixels, weight alignment, A2C_NO_RING /
 * A2C_RING_BLOCKS read per call).  A caller asks this BEFORE it decides to leave the rows out; a2c_a3c_rollout with
 * states_lazy set on a shape that is not ring-capable returns A2C_ERR_ARG.                                            */
int a2c_a3c_ring_supported(int B, int C, int H, int W, int n_actions, const float *conv1_weight);

/* ------------------------------------------------------------------ b: pinned staging
 * The pool region lives in POSIX shared memory (shm_open name `shm_name`, created by this call)
 * so that env worker processes can map it; it is pinned and mapped into the device's address
 * space with hipHostRegister (no copy): *host_out = the mapping in this process, *dev_out = the
 * address kernels and hipMemcpyAsync use.  Replaces the reference's share_memory_() tensors
 * (training.py:93-101) for the data that crosses the host-device boundary every env step.     */
int a2c_rollout_buffer_create(const char *shm_name, size_t bytes, void **host_out, void **dev_out);
int a2c_rollout_buffer_destroy(const char *shm_name, void *host, size_t bytes);
/* pin + map an existing host range (page aligned) / undo it                                  */
int a2c_pinned_register(void *host, size_t bytes, void **dev_out);
int a2c_pinned_unregister(void *host);
/* Push buffer: fine-grained DEVICE memory that the host writes into directly (large-BAR: *ptr_out is valid on both
 * sides; zero-filled).  The native env worker threads of a pool started with a2c_pool_threads_start_push mirror every
 * answer there (packed / uint8 frame, sfence, rec granule, sfence), and a2c_a3c_rollout / a2c_pool_ingest* given the
 * push addresses poll and fetch from HBM instead of reading host memory over PCIe (training.py:93-101's shared tensors,
 * for the host -> device direction).  A2C_ERR_LAUNCH when the platform refuses the allocation OR the memory is not
 * writable from the host (no large BAR; checked with a guarded probe write that the device reads back).           */
int a2c_push_buffer_alloc(size_t bytes, void **ptr_out);
int a2c_push_buffer_free(void *ptr);
/* host threads waiting in hipStreamSynchronize sleep instead of spinning (hipDeviceScheduleBlockingSync): for
 * nodes where the ranks of a multi-GPU job have fewer CPUs than busy threads.  Call before any other HIP work. */
int a2c_set_blocking_sync(int on);
/* PCI bus id ("0000:c1:00.0") of the current HIP device: the host side places the pinned region and the env
 * workers on the NUMA node the GPU hangs off (/sys/bus/pci/devices/<id>/numa_node)                */
int a2c_device_pci_bus_id(char *out, int len);
/* *dev_ptr = value with one system-scope 4-byte store from the stream (e.g. the phase word of the host pool header,
 * a2c_hostpool.h: the stream itself tells the env workers when a rollout starts and ends, so that they spin only
 * while one is running)                                                                                        */
int a2c_store_u32_system(uint32_t *dev_ptr, uint32_t value, a2c_stream_t stream);
/* hipMemcpyAsync on `stream`: kind 1 = host->device, 2 = device->host, 3 = device->device      */
int a2c_memcpy_async(void *dst, const void *src, size_t bytes, int kind, a2c_stream_t stream);
/* dst[b*dst_stride + j] = src[b*src_stride + j], j < n  (h_states[e] = h, runner.py:201;
 * gathers/scatters of per-step rows of the rollout-major buffers)                       */
int a2c_copy_rows(const float *src, int64_t src_stride, float *dst, int64_t dst_stride, int B,
                  int64_t n, a2c_stream_t stream);
/* x[b, :] *= (1 - dones[b*done_stride])   (hs = hs*(1-dones[:,i]), updater.py:164)      */
int a2c_mask_rows(float *x, int64_t ld, const float *dones, int64_t done_stride, int B, int n,
                  a2c_stream_t stream);
/* dst[(t*R + r)*n + j] = src[(r*T + t)*n + j]: swaps the two leading axes of a (R,T,n) array
 * (rollout-major <-> time-major views of the buffers for the BPTT unroll, updater.py:155-168) */
int a2c_permute_rows(const float *src, float *dst, int64_t R, int64_t T, int64_t n,
                     a2c_stream_t stream);

/* ------------------------------------------------------------------ a8: loss
 * updater.py:100-106,124-127 forward AND backward in one pass over N_local rows:
 *   lsm = log_softmax(logits); log_p = lsm[n, actions[n]];
 *   Entropy = -entr_coef * mean_n sum_a lsm*softmax;  Pi_Loss = -pi_coef * mean(log_p*adv);
 *   ValLoss = val_coef * mean((V-R)^2);  Loss = Pi_Loss + ValLoss - Entropy,
 * means over n_global rows (n_global > n_local when the batch is sharded over GPUs).
 * If adv_sums != NULL the advantages are normalised on the fly as in a2c_normalize
 * (norm_advs, updater.py:97-98) using eps 1e-6.
 * Outputs: dlogits (N, ldd) = dLoss/dlogits, dvals (N) = dLoss/dV,
 * loss_sums[0..2] (double) = sum log_p*adv, sum (V-R)^2, sum_n sum_a p*lsm
 * over the local rows.                                                                  */
int a2c_loss_fwd_bwd(const float *logits, int64_t ld_logits, const float *vals, int64_t val_stride,
                     const int64_t *actions, const float *advs, const float *returns,
                     const double *adv_sums, int64_t n_local, int64_t n_global, int A,
                     float pi_coef, float val_coef, float entr_coef, float *dlogits, int64_t ldd,
                     float *dvals, int64_t dval_stride, double *loss_sums, double *reduce_scratch,
                     a2c_stream_t stream);

/* ------------------------------------------------------------------ a6: dense layers
 * C[M,N] = (accumulate ? C : 0) + opA(A)[M,K] * opB(B)[K,N] (+ bias[N]) ; then optional ReLU ; then optional
 * multiply by (mask[m*ldmask+n] > 0) (the ReLU derivative of the layer below, fused into
 * the producer).  fp32 in / fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32:
 * exact fp32 FMA chain in k order).
 *   transA = 0: A stored [M][lda] (k contiguous);  transA = 1: A stored [K][lda] (m contiguous)
 *   transB = 0: B stored [K][ldb] (n contiguous);  transB = 1: B stored [N][ldb] (k contiguous)
 * splitk > 1 splits K over gridDim.z into partial slabs in ws that a second kernel sums in
 * fixed order (deterministic); ws_bytes >= a2c_gemm_ws_bytes(M,N,splitk).
 * Linear forward  y = x W^T + b  (torch.nn.Linear, models.py:73,84,85 ...): transA=0, transB=1.
 * Linear backward dx = dy W: transA=0, transB=0;  dW = dy^T x: transA=1, transB=0.       */
size_t a2c_gemm_ws_bytes(int64_t M, int64_t N, int splitk);
/* Large products (M, N, K >= 1024 and M N K >= 2e10: ConvModel's 28224 x 2000 layers at update batch, models.py:246-264) run
 * on the BF16 matrix pipe with fp32 results when the workspace also holds a2c_gemm_x9_ws_bytes(M, N, K) bytes BEHIND the
 * a2c_gemm_ws_bytes(M, N, splitk) ones (rounded up to 256): each operand is split, in one pass, into three bf16 images
 * a = a0 + a1 + a2 (exact: 3 x 8 significant bits); the SIX piece products with qa + qb <= 2 are issued (each exact; the
 * three dropped ones are below 2^-24 of |a b|, under the rounding of the fp32 product itself) and every sum is the MFMA's
 * fp32 accumulator's -- nn.Linear's fp32 sum, re-associated, at 6/16 of the fp32 matrix time on paper and 1.7-1.9 x the
 * fp32 kernels measured (DESIGN.md section 4 "bf16 x 6").  A2C_GEMM_X9=0: fp32 MFMA kernels only; =1: all nine products
 * (gemm_x9_kernel; ties the fp32 kernels); =2: the six-product kernel from M, N, K >= 256 and M N K >= 2.5e8 on (tests).
 * 0: the product does not take that path.  Without the extra bytes a2c_gemm_f32 runs the fp32 MFMA kernels.            */
size_t a2c_gemm_x9_ws_bytes(int64_t M, int64_t N, int64_t K);
int a2c_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A,
                 int64_t lda, const float *B, int64_t ldb, float *C, int64_t ldc,
                 const float *bias, int relu, const float *mask, int64_t ldmask, int accumulate,
                 int splitk, void *ws, size_t ws_bytes, a2c_stream_t stream);
/* The same six-product kernel from PREBUILT images (a weight matrix split once per update and used by every step of the
 * rollout: ConvModel's resize_emb at 256 envs, models.py:246): C[M][N] = A[M][K] B[N][K]^T with the epilogue of a2c_gemm_f32.
 *   a2c_gemm_x6_image_bytes  bytes of the image of a rows x K operand (rows padded to 256, K to 16; 16-byte aligned)
 *   a2c_gemm_x6_split        writes it: k_contiguous = 1: src[r * ld + k], 0: src[k * ld + r]; layout
 *                            image[piece][row / 256][k / 16][(k / 8) & 1][row % 256][8] bf16
 *   a2c_gemm_x6_images       the product; splitk = the largest number of K splits allowed (ws_bytes >= splitk * M * N * 4
 *                            when > 1; the kernel takes fewer when the tiles alone fill the chip), slabs summed in fixed order */
size_t a2c_gemm_x6_image_bytes(int64_t rows, int64_t K);
int a2c_gemm_x6_split(const float *src, int64_t ld, int64_t rows, int64_t K, int k_contiguous, void *image,
                      a2c_stream_t stream);
int a2c_gemm_x6_images(int64_t M, int64_t N, int64_t K, const void *image_a, const void *image_b, float *C, int64_t ldc,
                       const float *bias, int relu, const float *mask, int64_t ldmask, int accumulate, int splitk, void *ws,
                       size_t ws_bytes, a2c_stream_t stream);
/* Split-K phase only: writes `a2c_gemm_splits(K, splitk)` partial slabs [split][M][N] (dense,
 * no epilogue) into ws and leaves the fixed-order sum to the consumer (a2c_heads_fused).   */
int a2c_gemm_splits(int64_t K, int splitk);
/* dx[m, k] = (sum_{n < N} dy[m*ldy + n] * W[n*K + k]) * bit(m, k), N <= 8: the skinny input-gradient product of
 * a2c_gemm_f32 (transA = transB = 0, K <= 8 there) with the ReLU mask of the layer below given as ONE BIT per activation
 * -- bit (k & 7) of byte maskbits[m*mask_row_bytes + (k >> 3)] = (act[m][k] > 0) -- instead of the fp32 activation itself
 * (A3CModel's da2 = (dl . Wc[:A]) * (a2 > 0), models.py:73-85 through autograd: 340 MB of mask reads per update become
 * 10 MB).  The same FMA order as a2c_gemm_f32's path: bit-identical results.  K % 8 == 0, ldx % 4 == 0, W and dx 16-byte
 * aligned.  Producer of the bits: a2c_a3c_rollout's ring kernel (a2_maskbit_rows).                                  */
int a2c_small_n_bwd_data_bits(const float *dy, int64_t ldy, const float *W, float *dx, int64_t ldx,
                              const uint8_t *maskbits, int64_t mask_row_bytes, int64_t M, int N, int64_t K,
                              a2c_stream_t stream);
int a2c_gemm_f32_partial(int transA, int transB, int64_t M, int64_t N, int64_t K, const float *A,
                         int64_t lda, const float *B, int64_t ldb, int splitk, void *ws,
                         size_t ws_bytes, a2c_stream_t stream);
/* Fused tail of a forward pass (A3CModel: proj_matrx epilogue + pi/value heads + action sampling,
 * models.py:73,84-85 + runner.py:94-97), one wave per row m:
 *   x[m,:]  = sum_z xs[z*slab_stride + m*ldx + :]  (z < nslab, fixed order) + bias_in  [ReLU]
 *   (optionally stored to emb_out[m*ld_emb + :])
 *   heads[m*ldh + n] = x[m,:] . W[n,:] + b[n]          n < N <= 8, W stored [N][K]
 *   if u != NULL: actions[m*act_stride] = inverse-CDF sample of softmax(heads[m, :n_logits])
 * K % 4 == 0, 16 B aligned rows.                                                          */
int a2c_heads_fused(const float *xs, int nslab, int64_t slab_stride, int64_t ldx,
                    const float *bias_in, int relu_in, float *emb_out, int64_t ld_emb,
                    const float *W, const float *b, float *heads, int64_t ldh, int64_t M, int N,
                    int K, const float *u, int n_logits, int64_t *actions, int64_t act_stride,
                    a2c_stream_t stream);
/* a2c_heads_fused whose sampling thread also hands the action to the env worker (the device relay's
 * a2c_pool_publish_actions, runner.py:129-131 "pipe.send(action)"): with cmd != NULL (needs u)
 *   cmd[m] = ((seq_base[0] + seq_off) << 32) | action[m]       one system-scope 8-byte store per row
 * cmd == NULL is a2c_heads_fused.                                                            */
int a2c_heads_fused_publish(const float *xs, int nslab, int64_t slab_stride, int64_t ldx,
                            const float *bias_in, int relu_in, float *emb_out, int64_t ld_emb,
                            const float *W, const float *b, float *heads, int64_t ldh, int64_t M,
                            int N, int K, const float *u, int n_logits, int64_t *actions,
                            int64_t act_stride, uint64_t *cmd, const uint32_t *seq_base,
                            uint32_t seq_off, a2c_stream_t stream);
int a2c_gemm_f32_nt(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
                    int64_t ldb, float *C, int64_t ldc, const float *bias, int relu,
                    a2c_stream_t stream);
int a2c_gemm_f32_nn(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
                    int64_t ldb, float *C, int64_t ldc, const float *mask, int64_t ldmask,
                    a2c_stream_t stream);
int a2c_gemm_f32_tn(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda, const float *B,
                    int64_t ldb, float *C, int64_t ldc, int splitk, void *ws, size_t ws_bytes,
                    a2c_stream_t stream);
/* Inference-only composition of two stacked nn.Linear layers with NO activation in between
 * (A3CModel: emb = proj_matrx(flat), then pi(emb) / value(emb); models.py:73, 84-85):
 *   Wc (N x F) = Wh (N x H) . Wp (H x F)        bc (N) = Wh . bp + bh
 * so that [logits | value] = flat . Wc^T + bc in one skinny layer (a2c_heads_fused / a2c_a3c_step).
 * N <= 8.  Rebuilt after every optimiser step; the update keeps the exact two-layer path.        */
int a2c_compose_heads(const float *Wh, const float *bh, const float *Wp, const float *bp, float *Wc,
                      float *bc, int N, int H, int F, a2c_stream_t stream);
/* out[n] = sum_m x[m*ld + n]   (bias gradients), deterministic two-stage reduction;
 * ws_bytes >= a2c_colsum_ws_bytes(N)                                                    */
size_t a2c_colsum_ws_bytes(int64_t N);
int a2c_colsum(const float *x, int64_t ld, int64_t M, int64_t N, float *out, void *ws,
               size_t ws_bytes, a2c_stream_t stream);

/* ------------------------------------------------------------------ a6: convolutions
 * torch.nn.Conv2d (+ReLU) of the model classes (models.py:98,312,685), NCHW fp32,
 * square kernel ks, stride, zero padding pad.  LDS-staged input tiles, implicit-GEMM inner
 * product on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
 * `in` samples are addressed with a per-sample stride in floats (so a forward can read
 * states[idx*T+t] straight out of the rollout buffer); planes are contiguous H*W.
 * Constraints: Cin % 4 == 0, Cout % 4 == 0 (true for every reference layer).
 * wprep holds the weights re-laid as matrix-core A fragments; build it with
 * a2c_conv2d_prep_weights after every optimiser step (sizes from a2c_conv2d_prep_floats).
 *   kind 0: forward fragments      kind 1: backward-data fragments                      */
typedef struct {
  int Cin, H, W;      /* input planes and size              */
  int Cout, ks, stride, pad;
  int OH, OW;         /* output size = (H - ks + 2 pad)/stride + 1 */
} a2c_conv_desc;
size_t a2c_conv2d_prep_floats(const a2c_conv_desc *d, int kind);
int a2c_conv2d_prep_weights(const a2c_conv_desc *d, int kind, const float *weight, float *wprep,
                            a2c_stream_t stream);
/* out = relu?(conv(in, W) + bias)                                                       */
int a2c_conv2d_fwd(const a2c_conv_desc *d, const float *in, int64_t in_bstride,
                   const float *wprep_fwd, const float *bias, int relu, float *out,
                   int64_t out_bstride, int B, a2c_stream_t stream);
/* din = conv_transpose(dout, W), then optionally din *= (mask > 0) where mask is the
 * (B,Cin,H,W) activation that fed this conv (the ReLU below it)                          */
int a2c_conv2d_bwd_data(const a2c_conv_desc *d, const float *dout, const float *wprep_bwd,
                        const float *mask, float *din, int B, a2c_stream_t stream);
/* The ReLU mask as SIGN WORDS.  For the 3x3 layers of the conv-stack models the forward can leave, next to `out`, one
 * bit per activation: bit (x & 31) of word signs[b*signs_bstride + (c*OH + y)*ceil(OW/32) + (x >> 5)] = (out[b][c][y][x] > 0).
 * a2c_conv2d_bwd_data_signs of the layer ABOVE takes them in place of the float mask (din *= bit): 1/32 of the mask's
 * HBM reads, same result bit for bit.  a2c_conv2d_sign_words(d) = words per sample of d's output, 0 when d's forward
 * cannot write them (then use a2c_conv2d_fwd and the float mask); a2c_conv2d_bwd_data_signs_supported(d) = 1 when d's
 * backward-data can read the sign words of ITS INPUT ([Cin][H][ceil(W/32)] words per sample); signs == NULL: no mask. */
int64_t a2c_conv2d_sign_words(const a2c_conv_desc *d);
int a2c_conv2d_fwd_signs(const a2c_conv_desc *d, const float *in, int64_t in_bstride, const float *wprep_fwd,
                         const float *bias, int relu, float *out, int64_t out_bstride, uint32_t *signs,
                         int64_t signs_bstride, int B, a2c_stream_t stream);
int a2c_conv2d_bwd_data_signs_supported(const a2c_conv_desc *d);
/* The ReLU mask as one bit per activation, in the layout a wavefront writes without packing work ("lane masks"; the streaming
 * backward-data kernel of the 4x4 / stride-2 layer of A3CModel, models.py:35-37).  lanemask = (B, Cin*H*W/64) 64-bit words =
 * Cin*H*W/8 bytes per sample: bit (i & 7) of byte (i >> 3) = (act[b][i] > 0), i the flat (c, y, x) index -- the four bits of
 * a lane's float4 are one nibble, two neighbouring lanes make a byte.  Producers: a2c_a3c_rollout's ring kernel
 * (a1_lanemask_rows) or a2c_lanemask_from_act over any tensor whose float count is a multiple of 256.
 * din = conv_transpose(dout, W) * bit: the same values as a2c_conv2d_bwd_data with the float mask, bit for bit, for 1/32 of
 * the mask's HBM reads.  _supported(d, B) = 1 when a2c_conv2d_bwd_data_lanemask can run (layer shape, Cin*H*W % 256 == 0,
 * B large enough for the streaming kernel); otherwise use a2c_conv2d_bwd_data with the float mask.                     */
int a2c_lanemask_from_act(const float *act, uint64_t *lanemask, int64_t n_floats, a2c_stream_t stream);
int a2c_conv2d_bwd_data_lanemask_supported(const a2c_conv_desc *d, int B);
int a2c_conv2d_bwd_data_lanemask(const a2c_conv_desc *d, const float *dout, const float *wprep_bwd, const uint64_t *lanemask,
                                 float *din, int B, a2c_stream_t stream);
/* The same layer below a rank-n_logits head (A3CModel's update: updater.py:128's autograd through pi(emb), proj_matrx and the
 * ReLU of conv2, models.py:35-37, 73, 84): the layer's dOut is never materialised.  With Wc = pi.weight proj_matrx.weight
 * (n_logits x Cout*OH*OW, a2c_compose_heads) and the ReLU mask of the layer's OUTPUT as one bit per activation (maskbits:
 * bit e & 7 of byte e >> 3 of a row of mask_row_bytes bytes, e the flat (co, oy, ox) index; the ring kernel's a2_maskbit_rows),
 *     dOut[b][e] = (act[b][e] > 0) ? sum over n < n_logits of dl[b * ld_dl + n] * Wc[n][e] : 0
 * is formed while a sample is staged -- the sums of a2c_small_n_bwd_data_bits (n ascending, separate multiply and add), so both
 * passes return bit for bit what they return on that entry point's output -- by
 *   a2c_conv2d_bwd_data_lanemask_rank   din = conv_transpose(dOut, W) * bit(lanemask)            (bwd_x6_kernel)
 *   a2c_conv2d_bwd_weight_rank          dW = sum dOut (x) patches(in), db = sum dOut             (wgrad_x6_kernel)
 * both on the bf16 matrix pipe as six exact piece products with fp32 sums.  _supported = 1 for A3CModel's conv2
 * (16 x 20 x 20 -> 32 x 9 x 9, 4 x 4, stride 2) at streaming batch with n_logits <= 4; otherwise form dOut with
 * a2c_small_n_bwd_data_bits and call the plain entry points.                                                            */
int a2c_conv2d_bwd_rank_supported(const a2c_conv_desc *d, int n_logits, int B);
int a2c_conv2d_bwd_data_lanemask_rank(const a2c_conv_desc *d, const float *dl, int64_t ld_dl, int n_logits, const float *Wc,
                                      const uint8_t *maskbits, int64_t mask_row_bytes, const float *wprep_bwd,
                                      const uint64_t *lanemask, float *din, int B, a2c_stream_t stream);
int a2c_conv2d_bwd_weight_rank(const a2c_conv_desc *d, const float *in, int64_t in_bstride, const float *dl, int64_t ld_dl,
                               int n_logits, const float *Wc, const uint8_t *maskbits, int64_t mask_row_bytes, float *dW,
                               float *db, int B, void *ws, size_t ws_bytes, a2c_stream_t stream);
int a2c_conv2d_bwd_data_signs(const a2c_conv_desc *d, const float *dout, const float *wprep_bwd, const uint32_t *signs,
                              int64_t signs_bstride, float *din, int B, a2c_stream_t stream);
/* a2c_conv2d_bwd_data_signs of layer d2 FUSED with a2c_conv2d_bwd_weight_frames of the FIRST layer d1 below it (round 6;
 * GRUModel's conv2 over conv1, models.py:570-590: 16 <- 24 channels at stride 2 on 84 x 84 over 4 -> 16, 3 x 3, stride 1,
 * pad 1).  The first layer's input needs no gradient, so layer 2's masked input gradient (the `din` of the unfused call:
 * 16 x 84 x 84 floats per sample, 14.8 GB at N = 32,768) feeds nothing but d1's weight gradient: the kernel keeps each finished
 * band of it in LDS, splits it into three exact bf16 pieces in registers and multiplies it against the band's rows of the
 * uint8 frames (exact in bf16) on the bf16 matrix pipe -- it is never written to HBM and never read back (the two unfused
 * launches move 29.6 GB for it).  dW1 (16, 4, 3, 3), db1 (16): the sums of a2c_conv2d_bwd_weight_frames over
 * a2c_conv2d_bwd_data_signs's output, re-associated; per-workgroup partials in ws + a fixed-order reduction (deterministic).
 * signs = the sign words of d1's OUTPUT (a2c_conv2d_fwd_signs / a2c_conv2d_fwd_frames), frame_store / nvalid as in
 * a2c_conv2d_bwd_weight_frames.  _ws_bytes returns 0 when the pair of layers has no fused kernel (then use the two calls). */
size_t a2c_conv2d_bwd_data_w1_frames_ws_bytes(const a2c_conv_desc *d2, const a2c_conv_desc *d1, int B);
int a2c_conv2d_bwd_data_w1_frames(const a2c_conv_desc *d2, const float *dout, const float *wprep_bwd, const uint32_t *signs,
                                  int64_t signs_bstride, const a2c_conv_desc *d1, const uint8_t *frame_store,
                                  int64_t slot_stride, int64_t T, const int32_t *nvalid, float *dW1, float *db1, int B,
                                  void *ws, size_t ws_bytes, a2c_stream_t stream);
/* a2c_conv2d_bwd_data of layer d2 FUSED with a2c_conv2d_bwd_weight of the layer d1 below it, for the case where
 * nothing but d1's weight gradient reads d2's input gradient (d1 = first conv of the stack: models.py:196-215, its
 * input needs no gradient): the masked input gradient (the `din` of the call above) is assembled band by band
 * in LDS, multiplied against the band's rows of d1's input x on the matrix cores and dropped -- it is never
 * written to HBM.  dW1 (d1.Cout, d1.Cin, 3, 3), db1 (d1.Cout): the same sums as the two separate calls, per-workgroup
 * partials in ws + fixed-order reduction.  Applies when d1 is 3x3 / stride 1 / pad 1 with d1.Cin*9 <= 48,
 * d1.Cout == d2.Cin <= 16, d2.W % 4 == 0: _ws_bytes returns 0 when it does not (then use the two calls).         */
size_t a2c_conv2d_bwd_data_w1_ws_bytes(const a2c_conv_desc *d2, const a2c_conv_desc *d1, int B);
int a2c_conv2d_bwd_data_w1(const a2c_conv_desc *d2, const float *dout, const float *wprep_bwd,
                           const float *mask, const a2c_conv_desc *d1, const float *x, int64_t x_bstride,
                           float *dW1, float *db1, int B, void *ws, size_t ws_bytes, a2c_stream_t stream);
/* dW (Cout,Cin,ks,ks) and db (Cout) summed over the batch: per-workgroup partial slabs in
 * ws + fixed-order reduction (deterministic).                                            */
size_t a2c_conv2d_bwd_weight_ws_bytes(const a2c_conv_desc *d, int B);
int a2c_conv2d_bwd_weight(const a2c_conv_desc *d, const float *in, int64_t in_bstride,
                          const float *dout, float *dW, float *db, int B, void *ws,
                          size_t ws_bytes, a2c_stream_t stream);
/* The same for the first layer of the A3CModel-shaped nets (8x8 / stride 4 on 4 stacked frames) with the
 * input taken from the single-frame uint8 store a2c_a3c_rollout leaves (stack-on-load: 28 KB of uint8 per
 * sample instead of the 113 KB fp32 state; identical fp32 values, identical summation order): sample
 * n = r*T + t reads frames[r][t .. t+3], planes c < 4 - nvalid[n] are zero.                              */
int a2c_conv2d_bwd_weight_frames(const a2c_conv_desc *d, const uint8_t *frame_store, int64_t slot_stride,
                                 int64_t T, const int32_t *nvalid, const float *dout, float *dW,
                                 float *db, int B, void *ws, size_t ws_bytes, a2c_stream_t stream);
/* A CHAIN of conv + ReLU layers of one rollout step in ONE launch (models.py:648-652: `self.features(x)` of GRUModel,
 * i.e. nn.Sequential over the conv blocks of models.py:570-636): layer i reads layer i-1's output (in / in_bstride for
 * layer 0), writes out[i] (+ b * out_bstride[i] per sample; the rollout passes rows of the update's activation stash) and
 * results are BIT-IDENTICAL to n_layers calls of a2c_conv2d_fwd[_signs] -- same tiles, same summation order; one workgroup
 * walks one sample through all layers, so no launch boundary (fill + tail per launch, 16-58 us each at 256 envs) sits
 * between them.  signs (may be NULL) / signs[i] (may be NULL): sign words of layer i's output (rows of signs_bstride[i]
 * words), as a2c_conv2d_fwd_signs writes them; supported for layer 0, or layers 0 and 1.
 * a2c_conv2d_fwd_chain_supported: 1 for the layer runs that have such a kernel (GRUModel conv2 .. conv5: 16 -> 24 @84,
 * 24 -> 32 @42, 32 -> 48 @21, 48 -> 64 @11, 3x3 / stride 2 / pad 1; A2C_NO_CHAIN=1 switches it off), else the caller
 * launches layer by layer.  wprep_fwd[i]: a2c_conv2d_prep_weights(kind 0) of layer i.                                */
#define A2C_CONV_CHAIN_MAX 8
int a2c_conv2d_fwd_chain_supported(const a2c_conv_desc *d, int n_layers);
int a2c_conv2d_fwd_chain(const a2c_conv_desc *d, int n_layers, const float *in, int64_t in_bstride,
                         const float *const *wprep_fwd, const float *const *bias, int relu, float *const *out,
                         const int64_t *out_bstride, uint32_t *const *signs, const int64_t *signs_bstride, int B,
                         a2c_stream_t stream);
/* ... and, with the same signature, for the first layer of the 3x3 stacks (ConvModel / GRUModel: 4 -> 16 channels at
 * 84 x 84, models.py:201-207, 570-576).  Their FORWARD with the input stacked on load (rollout step: sample b reads the
 * window frame_store + b * sample_stride, T = 1, nvalid[b * nvalid_stride]; in general sample n reads
 * frame_store + (n / T) * sample_stride + (n % T) * H * W): the loader waves of the streaming kernel expand the bytes to
 * the fp32 image, planes c < 4 - nvalid are zero; everything else is a2c_conv2d_fwd[_signs] (signs may be NULL).
 * a2c_conv2d_fwd_frames_supported: 1 when the layer has such a kernel (else: a2c_frames_to_states + a2c_conv2d_fwd).  */
int a2c_conv2d_fwd_frames_supported(const a2c_conv_desc *d);
int a2c_conv2d_fwd_frames(const a2c_conv_desc *d, const uint8_t *frame_store, int64_t sample_stride, int64_t T,
                          const int32_t *nvalid, int64_t nvalid_stride, const float *wprep_fwd, const float *bias,
                          int relu, float *out, int64_t out_bstride, uint32_t *signs, int64_t signs_bstride, int B,
                          a2c_stream_t stream);

/* ------------------------------------------------------------------ a6: GRU cell, LayerNorm
 * models.GRU.forward (models.py:465-476) given the six pre-activation products:
 *   gx = x [Wx0|Wx1|Wx2] (B,3h),  gh = h [Wh0|Wh1] (B,2h),  and, after r is known,
 *   rh_u = (r*h) Wh2 (B,h).  Two elementwise stages around the (r*h) GEMM:
 *   stage1: z = sig(gx0+gh0+b0), r = sig(gx1+gh1+b1), rh = r*h
 *   stage2: c = tanh(gx2 + rh_u + b2), h_new = z*h + (1-z)*c                             */
int a2c_gru_gates(const float *gx, const float *gh, const float *b, const float *h, float *z,
                  float *r, float *rh, int B, int hdim, a2c_stream_t stream);
int a2c_gru_out(const float *gx, const float *rh_u, const float *b, const float *h,
                const float *z, float *c, float *h_new, int B, int hdim, a2c_stream_t stream);
/* The whole cell forward of one rollout step (models.py:465-476 at batch n_envs) in two launches instead of five:
 * x (B, xs; row stride ldx) and h (B, hdim) -> z, r, rh = r*h, c (may be NULL), h_new (may alias h: the in-place rollout
 * step), with WxC = [W_x[0] | W_x[1] | W_x[2]] (xs, 3 hdim), WhC = [W_h[0] | W_h[1]] (hdim, 2 hdim), Wh2 = W_h[2] and
 * b (3 hdim).  gx (B, 3 hdim) is scratch (its candidate columns carry x W_x[2] between the launches).  Bit-identical to
 * a2c_gemm_f32 (x WxC, h WhC) + a2c_gru_gates + a2c_gemm_f32 (rh Wh2) + a2c_gru_out.  hdim % 32 == 0, xs % 8 == 0.    */
int a2c_gru_cell_fwd(const float *x, int64_t ldx, const float *h, const float *WxC, const float *WhC,
                     const float *Wh2, const float *b, float *gx, float *z, float *r, float *rh, float *c,
                     float *h_new, int B, int xs, int hdim, a2c_stream_t stream);
/* backward of stage2: given dh_new (B,h): dc_pre = dh_new*(1-z)*(1-c^2) (B,h) ;
 * dz = dh_new*(h - c) ; dh_direct = dh_new*z                                            */
int a2c_gru_out_bwd(const float *dh_new, const float *h, const float *z, const float *c,
                    float *dc_pre, float *dz, float *dh, int B, int hdim, a2c_stream_t stream);
/* the same inside the BPTT unroll (updater.py:161-166 walked backwards): dh_new is replaced by dh_new + carry * (1 - done[b]),
 * carry (B,h) = the gradient that reached the next step's h_in = h_new * (1 - done) (done[b] = dones[b * done_stride]);
 * one launch instead of mask_rows + add + gru_out_bwd, same values.  carry may be dh (in place).                   */
int a2c_gru_out_bwd_carry(const float *dh_new, const float *carry, const float *dones, int64_t done_stride,
                          const float *h, const float *z, const float *c, float *dc_pre, float *dz, float *dh,
                          int B, int hdim, a2c_stream_t stream);
/* backward of stage1: d_rh (B,h) from the Wh2 GEMM; dz from above:
 * dz_pre = dz*z*(1-z) ; dr = d_rh*h ; dr_pre = dr*r*(1-r) ; dh += d_rh*r               */
int a2c_gru_gates_bwd(const float *d_rh, const float *dz, const float *h, const float *z,
                      const float *r, float *dz_pre, float *dr_pre, float *dh, int B, int hdim,
                      a2c_stream_t stream);
/* One step of the BPTT unroll's backward (updater.py:139-169 differentiated) in two launches instead of five; bit-identical
 * to a2c_gru_out_bwd[_carry] + a2c_gemm_f32 (dc_pre Wh[2]^T) + a2c_gru_gates_bwd + two accumulating a2c_gemm_f32
 * (dz_pre Wh[0]^T, dr_pre Wh[1]^T):  g = dh_new + carry * (1 - dones[b * done_stride]) (carry may be NULL);
 * dc_pre = g (1-z)(1-c^2), dz = g (h - c), d_rh = dc_pre Wh[2]^T, dz_pre = dz z (1-z), dr_pre = d_rh h r (1-r),
 * dh = g z + d_rh r + dz_pre Wh[0]^T + dr_pre Wh[1]^T.  Wh = gru.W_h (3, hdim, hdim); carry must not alias dh.        */
int a2c_gru_cell_bwd(const float *dh_new, const float *carry, const float *dones, int64_t done_stride,
                     const float *h, const float *z, const float *r, const float *c, const float *Wh,
                     float *dc_pre, float *dz, float *dz_pre, float *dr_pre, float *dh, int B, int hdim,
                     a2c_stream_t stream);
/* torch.nn.LayerNorm over the last dim n (eps 1e-5), FCModel value head (models.py:392) */
int a2c_layernorm_fwd(const float *x, const float *w, const float *b, float *y, float *mean,
                      float *rstd, int64_t rows, int n, a2c_stream_t stream);
int a2c_layernorm_bwd(const float *dy, const float *x, const float *w, const float *mean,
                      const float *rstd, float *dx, float *dw_rows, int64_t rows, int n,
                      int accumulate_dx, a2c_stream_t stream);

/* ------------------------------------------------------------------ a9: clip + optimiser
 * sumsq[0] (double, zeroed by the call) = sum g^2 over the flat gradient arena
 * (nn.utils.clip_grad_norm_, updater.py:129).                                           */
int a2c_gradnorm_sq(const float *grads, int64_t n, double *sumsq, double *reduce_scratch, a2c_stream_t stream);
/* clip (coef = min(1, max_norm/(norm+1e-6)), grads scaled in place like torch) and
 * torch.optim.RMSprop step (alpha, eps; momentum 0, not centered): updater.py:131,227-228 */
int a2c_clip_rmsprop(float *params, float *grads, float *square_avg, int64_t n,
                     const double *sumsq, double max_norm, double lr, double alpha, double eps,
                     float *norm_out, a2c_stream_t stream);
/* same with torch.optim.Adam (beta1, beta2, eps, no amsgrad); step = 1-based step count */
int a2c_clip_adam(float *params, float *grads, float *exp_avg, float *exp_avg_sq, int64_t n,
                  const double *sumsq, double max_norm, double lr, double beta1, double beta2,
                  double eps, int64_t step, float *norm_out, a2c_stream_t stream);

/* out5 = [loss_sums[0..2], grad_norm, err]: the five scalars update_model reads back (updater.py:134-136)
 * gathered into ONE device buffer for a single D2H copy                                  */
int a2c_pack_update_scalars(const double *loss_sums, const float *grad_norm, const int *err,
                            double *out5, a2c_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* A2C_MI355X_H */
