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

/* ---- FS_MIXED twin (flow_amd/csrc/flowsim_pair.h, T = double) -------------------------------------------
 * Positions and speeds are float64 and integrated in float64; IDMController.get_accel (same reference lines as
 * refsim_body.h) is evaluated in float32 on their rounded images; observation = state * RN64(1/normaliser)
 * rounded to float32; reward in float32 on the rounded speeds (tree order of refsim_body.h).  Every operation
 * below is one IEEE operation of the kernel, in the kernel's order (-ffp-contract=off). */
void refsim_ring_idm_mixed(int R, int r0, int r1, int N, int steps, const double* ring_len, double jlen, double dt,
                           double ramp, const double* p /* [6][N]: v0,T,a,b,delta,s0 per slot */,
                           const double* veh_len /* [N] */, double max_speed, double target_v, double max_cost,
                           double crash_gap, int step_limit, double* x, double* v, int32_t* time_counter,
                           float* obs, float* rew, uint8_t* done, int obs_every_step,
                           const double* sm /* NULL, or [6][N]: speed_mode bits, sumo tau, minGap, maxSpeed,
                                               max_accel, max_decel per slot (S7/S8 speed-mode clamps) */) {
  double xn[64], vn[64];
  float hh[64], dd[64];
  const double rc_ms = 1.0 / max_speed;
  const float gap32 = (float)crash_gap, tv32 = (float)target_v, mc32 = (float)max_cost;
  for (int r = r0; r < r1; ++r) {
    double* xr = x + (size_t)r * N;
    double* vr = v + (size_t)r * N;
    const double L = ring_len[r] + 4.0 * jlen;
    const double rc_L = 1.0 / L;
    int tc = time_counter[r];
    /* headways of the current snapshot, rounded to float32 */
    for (int i = 0; i < N; ++i) {
      const int j = (i + 1 >= N) ? 0 : i + 1;
      double d = xr[j] - xr[i];
      if (d < 0) d = d + L;
      hh[i] = (float)(d - veh_len[j]);
    }
    for (int s = 0; s < steps; ++s) {
      for (int i = 0; i < N; ++i) {
        const int j = (i + 1 >= N) ? 0 : i + 1;
        const float v0 = (float)p[0 * N + i], Tt = (float)p[1 * N + i], a = (float)p[2 * N + i],
                    b = (float)p[3 * N + i], delta = (float)p[4 * N + i], s0 = (float)p[5 * N + i];
        const float two_sqrt_ab = 2.0f * sqrtf(a * b);
        const float vi = (float)vr[i], vl = (float)vr[j];
        const float h = fabsf(hh[i]) < 1e-3f ? 1e-3f : hh[i];
        const float dyn = vi * Tt + vi * (vi - vl) / two_sqrt_ab;
        const float s_star = s0 + (dyn > 0 ? dyn : 0.0f);
        const float q = s_star / h;
        const float ratio = vi / v0;
        float pw;
        if (delta == 4.0f) { float r2 = ratio * ratio; pw = r2 * r2; }
        else if (delta == 2.0f) pw = ratio * ratio;
        else if (delta == 1.0f) pw = ratio;
        else if (delta == 3.0f) pw = (ratio * ratio) * ratio;
        else if (delta == 8.0f) { float r2 = ratio * ratio, r4 = r2 * r2; pw = r4 * r4; }
        else pw = powf(ratio, delta);
        const float acc = a * (1.0f - pw - q * q);
        double nv = vr[i] + (double)acc * dt;
        if (!(nv > 0)) nv = 0;
        double v_new = vr[i] + (nv - vr[i]) * ramp;
        if (sm) {
          /* speed-mode clamps: the acceleration of sumo_idm_speed (oracle/controllers.py) in float32 on the rounded
           * images, applied in float64: bit 0 min(vc, max(0, v + acc_s dt)), bit 1 min(vc, v + max_accel dt),
           * bit 2 max(vc, v - max_decel dt) */
          const int mode = (int)sm[0 * N + i];
          const float tau = (float)sm[1 * N + i], min_gap = (float)sm[2 * N + i], smax = (float)sm[3 * N + i],
                      ma = (float)sm[4 * N + i], md = (float)sm[5 * N + i];
          const float gap = hh[i] > 1e-3f ? hh[i] : 1e-3f;
          const float ts = 2.0f * sqrtf(ma * md);
          const float dyn_s = vi * tau + vi * (vi - vl) / ts;
          const float ss = min_gap + (0.0f > dyn_s ? 0.0f : dyn_s);
          const float qs = ss / gap;
          const float rs = vi / smax;
          const float rs2 = rs * rs;
          const float acc_s = ma * (1.0f - rs2 * rs2 - qs * qs);
          if (mode & 1) {
            double vs = vr[i] + (double)acc_s * dt;
            if (!(vs > 0)) vs = 0;
            if (!(v_new < vs)) v_new = vs;
          }
          if (mode & 2) { const double cap = vr[i] + sm[4 * N + i] * dt; if (!(v_new < cap)) v_new = cap; }
          if (mode & 4) { const double flo = vr[i] - sm[5 * N + i] * dt; if (!(v_new > flo)) v_new = flo; }
        }
        double x_new = xr[i] + v_new * dt;
        if (x_new >= L) x_new = x_new - L;
        xn[i] = x_new;
        vn[i] = v_new;
      }
      for (int i = 0; i < N; ++i) { xr[i] = xn[i]; vr[i] = vn[i]; }
      int crashed = 0, bad = 0;
      for (int i = 0; i < N; ++i) {
        const int j = (i + 1 >= N) ? 0 : i + 1;
        double d = xr[j] - xr[i];
        if (d < 0) d = d + L;
        hh[i] = (float)(d - veh_len[j]);
        if (hh[i] - gap32 < 0) crashed = 1;
      }
      tc += 1;
      if (obs_every_step || s == steps - 1) {
        const size_t so = obs_every_step ? (size_t)s : 0;
        float* o = obs + (so * R + r) * (size_t)(2 * N);
        for (int i = 0; i < N; ++i) {
          o[i] = (float)(vr[i] * rc_ms);
          o[N + i] = (float)(xr[i] * rc_L);
          const float v32 = (float)vr[i];
          const float dv = v32 - tv32;
          dd[i] = dv * dv;
          if (v32 < -100.0f) bad = 1;
        }
        float cost = sqrtf(tree_sum_f32(dd, N));
        float rw = mc32 - cost;
        if (!(rw > 0)) rw = 0;
        rw = rw / (mc32 + 1.1920928955078125e-07f);
        if (bad || crashed) rw = 0;
        rew[so * R + r] = rw;
        done[so * R + r] = (uint8_t)((tc >= step_limit) || crashed);
      }
    }
    time_counter[r] = tc;
  }
}

void refsim_ring_idm_mixed_all(int R, int N, int steps, const double* ring_len, double jlen, double dt, double ramp,
                               const double* p, const double* veh_len, double max_speed, double target_v,
                               double max_cost, double crash_gap, int step_limit, double* x, double* v, int32_t* tc,
                               float* obs, float* rew, uint8_t* done, int obs_every_step, int threads,
                               const double* sm) {
  if (threads <= 1) {
    refsim_ring_idm_mixed(R, 0, R, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                          crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step, sm);
    return;
  }
#pragma omp parallel for num_threads(threads) schedule(static)
  for (int t = 0; t < threads; ++t) {
    int r0 = (int)((long long)R * t / threads), r1 = (int)((long long)R * (t + 1) / threads);
    refsim_ring_idm_mixed(R, r0, r1, N, steps, ring_len, jlen, dt, ramp, p, veh_len, max_speed, target_v, max_cost,
                          crash_gap, step_limit, x, v, tc, obs, rew, done, obs_every_step, sm);
  }
}

/* x / c through float64 (flowsim_pair.h div_via_f64): exhaustive check helper, returns the number of float x in
 * [bits_lo, bits_hi) whose result differs from the IEEE float32 quotient */
long long refsim_div_via_f64_mismatches(float c, uint32_t bits_lo, uint32_t bits_hi, int threads) {
  const double cd = (double)c, rc = 1.0 / cd;
  long long bad = 0;
#pragma omp parallel for num_threads(threads) reduction(+ : bad) schedule(static)
  for (long long bb = bits_lo; bb < (long long)bits_hi; ++bb) {
    union { uint32_t u; float f; } in, a, g;
    in.u = (uint32_t)bb;
    const double xd = (double)in.f;
    double q = xd * rc;
    const double rr = fma(-q, cd, xd);
    q = fma(rr, rc, q);
    g.f = (float)q;
    a.f = in.f / c;
    if (a.u != g.u) ++bad;
  }
  return bad;
}

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
