/* refsim.c -- C restatement of the C2 hot path (TEST INFRASTRUCTURE ONLY).
 * Used by tests (second checker, bit-identical to the numpy oracle in float32)
 * and by bench.py's cpu_baseline leg ("port").  Build: oracle/cbuild.py.
 * Compile with -ffp-contract=off so the operation order below is what runs. */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define NAME(x) CAT(x, _f32)
#define SQRT sqrtf
#define FABS fabsf
#define POW powf
#include "refsim_body.h"
#undef REAL
#undef NAME
#undef SQRT
#undef FABS
#undef POW

#define REAL double
#define NAME(x) CAT(x, _f64)
#define SQRT sqrt
#define FABS fabs
#define POW pow
#include "refsim_body.h"

/* all replicas, optionally over OpenMP threads (replicas are independent) */
void refsim_ring_idm_f32_all(int R, int N, int steps, const float* ring_len, float jlen, float dt, float ramp,
                             const float* p, float veh_len, float max_speed, float target_v, float max_cost,
                             float crash_gap, int step_limit, float* x, float* v, int32_t* tc, float* obs,
                             float* rew, uint8_t* done, int obs_every_step, int threads) {
  if (threads <= 1) {
    refsim_ring_idm_f32(R, 0, R, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                        crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step);
    return;
  }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    int r0 = (int)((long long)R * t / threads), r1 = (int)((long long)R * (t + 1) / threads);
    refsim_ring_idm_f32(R, r0, r1, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                        crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step);
  }
}

void refsim_ring_idm_f64_all(int R, int N, int steps, const double* ring_len, double jlen, double dt, double ramp,
                             const double* p, double veh_len, double max_speed, double target_v, double max_cost,
                             double crash_gap, int step_limit, double* x, double* v, int32_t* tc, float* obs,
                             float* rew, uint8_t* done, int obs_every_step, int threads) {
  if (threads <= 1) {
    refsim_ring_idm_f64(R, 0, R, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                        crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step);
    return;
  }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    int r0 = (int)((long long)R * t / threads), r1 = (int)((long long)R * (t + 1) / threads);
    refsim_ring_idm_f64(R, r0, r1, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                        crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step);
  }
}
