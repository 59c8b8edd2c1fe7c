/* refsim_body.h -- included twice by refsim.c with REAL = float / double.
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the headline configuration C2 (RingNetwork, N IDM
 * vehicles, AccelEnv observation, desired_velocity reward, speed_mode
 * "aggressive"), following the same reference lines as oracle/refsim.py:
 *   IDMController.get_accel          flow/controllers/car_following_models.py:464-482
 *   apply_acceleration               flow/core/kernel/vehicle/traci.py:952-963
 *   AccelEnv.get_state               flow/envs/ring/accel.py:116-123
 *   rewards.desired_velocity         flow/core/rewards.py:6-59
 *   done                             flow/envs/base.py:398-400
 * Operation order is identical to the numpy oracle, so float results are
 * bit-identical to oracle/refsim.py with dtype float32 (tests/test_oracle_c.py).
 */
static REAL NAME(tree_sum)(const REAL* a, int n) {
  REAL buf[64];
  int seg = 1;
  while (seg < n) seg *= 2;
  for (int i = 0; i < seg; ++i) buf[i] = i < n ? a[i] : (REAL)0;
  while (seg > 1) {
    for (int i = 0; i < seg / 2; ++i) buf[i] = buf[2 * i] + buf[2 * i + 1];
    seg /= 2;
  }
  return buf[0];
}

/* Advance replicas [r0, r1) by `steps` env steps.  obs/rew/done hold the LAST step
 * (obs_every_step = 0) or every step ([steps, R, ...]). */
void NAME(refsim_ring_idm)(int R, int r0, int r1, int N, int steps, const REAL* ring_len, REAL jlen, REAL dt,
                           REAL ramp, const REAL* p /* v0,T,a,b,delta,s0 */, REAL veh_len, REAL max_speed,
                           REAL target_v, REAL max_cost, REAL crash_gap, int step_limit, REAL* x, REAL* v,
                           int32_t* time_counter, float* obs, float* rew, uint8_t* done, int obs_every_step) {
  const REAL v0 = p[0], Tt = p[1], a = p[2], b = p[3], delta = p[4], s0 = p[5];
  const REAL two_sqrt_ab = (REAL)2 * SQRT(a * b);
  REAL xn[64], vn[64], dd[64];
  for (int r = r0; r < r1; ++r) {
    REAL* xr = x + (size_t)r * N;
    REAL* vr = v + (size_t)r * N;
    const REAL L = ring_len[r] + (REAL)4 * jlen;
    int tc = time_counter[r];
    for (int s = 0; s < steps; ++s) {
      for (int i = 0; i < N; ++i) {
        const int j = (i + 1 >= N) ? 0 : i + 1;
        REAL h, vl = vr[j];
        const int has = N > 1;
        if (has) {
          REAL d = xr[j] - xr[i];
          if (d < 0) d = d + L;
          h = d - veh_len;
        } else {
          h = (REAL)1000;
        }
        const REAL vi = vr[i];
        REAL hh = FABS(h) < (REAL)1e-3 ? (REAL)1e-3 : h;
        REAL dyn = vi * Tt + vi * (vi - vl) / two_sqrt_ab;
        REAL s_star = has ? s0 + (dyn > 0 ? dyn : (REAL)0) : (REAL)0;
        REAL q = s_star / hh;
        REAL ratio = vi / v0, pw;
        if (delta == (REAL)4) { REAL r2 = ratio * ratio; pw = r2 * r2; }
        else if (delta == (REAL)2) pw = ratio * ratio;
        else pw = POW(ratio, delta);
        REAL acc = a * ((REAL)1 - pw - q * q);
        REAL next_vel = vi + acc * dt;
        if (!(next_vel > 0)) next_vel = 0;
        REAL v_new = vi + (next_vel - vi) * ramp;
        REAL x_new = xr[i] + v_new * dt;
        if (x_new >= L) x_new = x_new - L;
        xn[i] = x_new;
        vn[i] = v_new;
      }
      int crashed = 0, bad = 0;
      for (int i = 0; i < N; ++i) { xr[i] = xn[i]; vr[i] = vn[i]; }
      for (int i = 0; i < N && N > 1; ++i) {
        const int j = (i + 1 >= N) ? 0 : i + 1;
        REAL d = xr[j] - xr[i];
        if (d < 0) d = d + L;
        if (d - veh_len < crash_gap) crashed = 1;
      }
      tc += 1;
      if (obs_every_step || s == steps - 1) {
        const size_t so = obs_every_step ? (size_t)s : 0;
        float* o = obs + (so * R + r) * (size_t)(2 * N);
        for (int i = 0; i < N; ++i) {
          o[i] = (float)(vr[i] / max_speed);
          o[N + i] = (float)(xr[i] / L);
          REAL dv = vr[i] - target_v;
          dd[i] = dv * dv;
          if (vr[i] < (REAL)-100) bad = 1;
        }
        REAL cost = SQRT(NAME(tree_sum)(dd, N));
        REAL rw = max_cost - cost;
        if (!(rw > 0)) rw = 0;
        rw = rw / (max_cost + (REAL)1.1920928955078125e-07);
        if (bad || crashed) rw = 0;
        rew[so * R + r] = (float)rw;
        done[so * R + r] = (uint8_t)((tc >= step_limit) || crashed);
      }
    }
    time_counter[r] = tc;
  }
}
