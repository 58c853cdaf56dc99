/* CPU ORACLE (C part) -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the scalar loops of the reference's hot path, used as a fast checker
 * at sizes the Python oracle (oracle/a2c_oracle.py) cannot walk in seconds, and by
 * bench.py's cpu_baseline leg.  Built by oracle/Makefile into oracle/_ref/liba2c_oracle.so.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may load it.
 * Pinned by tests/test_oracle_golden.py::test_c_oracle_matches_golden (the same recorded
 * reference outputs as the Python oracle).  Citations are file:line into /root/reference/a2c/.
 */
#include <math.h>
#include <stdint.h>

/* utils.discount, utils.py:63-79: strictly sequential, fp32 multiply then fp32 add.
 * volatile stops the compiler from contracting the two roundings into an FMA. */
void oracle_discount(const float *x, const float *dones, float *y, int64_t n, float g) {
  volatile float run = 0.0f;
  for (int64_t i = n - 1; i >= 0; --i) {
    if (dones[i] == 1.0f) run = 0.0f;
    volatile float prod = g * run;
    run = x[i] + prod;
    y[i] = run;
  }
}

/* utils.sample_action, utils.py:45-60, on B rows of A probabilities with explicit uniforms */
void oracle_sample_action(const float *pi, const float *u, float *actions, int64_t B, int A) {
  for (int64_t b = 0; b < B; ++b) {
    volatile float cs = 0.0f;
    float act = -1.0f;
    for (int a = 0; a < A; ++a) {
      cs = cs + pi[b * A + a];
      if (cs >= u[b] && act < 0.0f) act = (float)a;
    }
    actions[b] = act;
  }
}

/* TD delta of runner.py:231: prev_rew + gamma*val*(1-prev_done) - prev_val, left to right */
float oracle_td_delta(float prev_rew, float gamma, float val, float prev_done, float prev_val) {
  volatile float gv = gamma * val;
  volatile float m = gv * (1.0f - prev_done);
  volatile float s = prev_rew + m;
  return s - prev_val;
}
