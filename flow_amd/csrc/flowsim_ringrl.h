// flowsim_ringrl.h -- k_ring_pair: the step kernel of single-lane rings whose vehicles are IDMControllers and
// RLControllers (the reference's RL ring experiments: examples/exp_configs/rl/singleagent/singleagent_ring.py:17-65 --
// 21 x IDMController(noise = 0.2) + 1 x RLController, WaveAttenuationPOEnv, ring length drawn per episode), in the
// two-vehicles-per-lane mapping of flowsim_pair.h.
//
// k_rollout_pair (flowsim_pair.h) is the hand-written rollout of the all-IDM AccelEnv ring; this kernel is the same
// arithmetic (idm_pair / sumo_acc_pair, the same operation order, the same Philox draws: bit-identical to the generic
// k_steps on every configuration both run) with what the RL experiments add:
//   * RL slots: the acceleration is the (clipped) action of the slot's column, read from an action tape [K, R, num_rl]
//     one step ahead; without actions (warm-up steps, envs/base.py:554-555) an RL vehicle is uncommanded and follows
//     SUMO's car-following model (S5/S7), like in k_steps;
//   * the heads: HEAD 0 AccelEnv (accel.py:109-123), HEAD 1 WaveAttenuationPOEnv (wave_attenuation.py:113-139, 248-269),
//     HEAD 2 MultiAgentWaveAttenuationPOEnv (multiagent/ring/wave_attenuation.py:128-252: the reference's
//     examples/exp_configs/rl/multiagent/multiagent_ring.py), HEAD 3 MultiAgentAccelPOEnv (multiagent/ring/accel.py:84-227);
//     the multi-agent heads (float32 only) see no crash (multiagent/base.py:188-190);
//   * MC: several action columns in the 16-step group form (every lane reads the columns of its own slots and of its
//     places in the reward's reduction four steps ahead, into the register the step four steps earlier has consumed);
//   * everything a reset needs: a replica mask (masked replicas alone advance), zero-step launches (observation of the
//     current state), observation at the last step only -- so a FS_MIXED handle can run the warm-up steps of
//     Env.reset in ITS arithmetic and the whole closed-loop path holds the 1e-4 trajectory bar.
// T = float: the float32 bit-twin of oracle/refsim.py.  T = double: FS_MIXED -- state and integration in float64,
// controllers in float32 on the rounded images (oracle/csim/refsim_rl.c is its bit-twin); no noise form.
// FAST: exponent 4 and the exact reciprocal divisions (every divisor proven on the host: Sim::ringrl_fast_ok).
#pragma once
#include "flowsim_pair.h"

namespace fs {

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): an unrolled loop whose index is a constant expression
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<N, I + 1>(f);
  }
}

template <typename T, int ROW, int HEAD, bool NOISE, bool FAST, bool MC = false>
__global__ __launch_bounds__(256) void k_ring_pair(DevView<T> s, int num_steps_arg, const uint8_t* __restrict__ mask,
                                                   const float* __restrict__ actions, size_t act_stride,
                                                   float* __restrict__ obs, float* __restrict__ rew,
                                                   uint8_t* __restrict__ done, int obs_every_step) {
  constexpr bool MIXED = sizeof(T) == 8;
  constexpr bool WA = HEAD == 1 || HEAD == 2;       // WaveAttenuationEnv's reward (the single- and the multi-agent head)
  constexpr bool MA = HEAD >= 2;                    // per-agent observation blocks, crash = 0
  static_assert(!(MA && MIXED), "the multi-agent heads exist in float32 only");
  // (FS_MIXED with noise: the float32 controller output + sigma g in float32, as in the float32 kernel; the C twin cannot
  // reproduce the hardware's log / cos, so this form is held against the float64 kernel with the same Philox streams at
  // 1e-4 instead of against a bit-twin -- tests/test_ringrl_gpu.py)
  constexpr int RPW = 64 / ROW;
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = lane / ROW;
  const int k = lane % ROW;
  const int r = wave * RPW + row;
  const int N = s.N;
  const int LP = N >> 1;                            // occupied lanes of a row (N is even: host-checked)
  const bool rvalid = r < s.R;
  const bool valid = rvalid && k < LP;
  // idle lanes (k >= LP, or a replica index past R) are clones of lane LP-1 / replica R-1 (flowsim_pair.h): they carry
  // the same state through the same arithmetic; only `valid` lanes enter reductions and write state
  const int rr = rvalid ? r : s.R - 1;
  const int kk = k < LP ? k : LP - 1;
  const bool last = (kk == LP - 1);                 // B's leader is slot 0 (lane 0 of the row)
  const int iA = 2 * kk, iB = iA + 1;
  const size_t idx = size_t(rr) * N + iA;
  const bool live_replica = mask == nullptr || mask[rr] != 0;
  // a masked launch advances nothing in a wave none of whose replicas is selected: the zero-step form (k_steps)
  const int num_steps = (mask != nullptr && __ballot(live_replica) == 0ull) ? 0 : num_steps_arg;

  // ---- slots -------------------------------------------------------------------------------------------------
  const bool rlA = s.ctrl[iA] == FS_CTRL_RL, rlB = s.ctrl[iB] == FS_CTRL_RL;
  const int colA = rlA ? s.rl_index[iA] : 0, colB = rlB ? s.rl_index[iB] : 0;
  f2 p[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) p[q] = f2{float(s.p[q * N + iA]), float(s.p[q * N + iB])};
  {   // an RL slot holds no IDM parameters: benign ones keep its (discarded) half of the packed arithmetic finite
    const float dflt[6] = {30.0f, 1.0f, 1.0f, 1.5f, 4.0f, 2.0f};
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      p[q].x = rlA ? dflt[q] : p[q].x;
      p[q].y = rlB ? dflt[q] : p[q].y;
    }
  }
  const T lenB = s.length[iB];
  const T len_nextA = next_a<ROW>(T(s.length[iA]), last, lane);
  const T L = s.ring_len[rr] + T(4) * s.jlen;
  int tcount = s.time[rr];
  uint32_t nctr = NOISE ? s.noise_ctr[rr] : 0u;
  const f2 sigma = NOISE ? f2{float(s.noise[iA]), float(s.noise[iB])} : f2{0.0f, 0.0f};
  const bool noisyA = NOISE && sigma.x > 0.0f && !rlA, noisyB = NOISE && sigma.y > 0.0f && !rlB;
  float gA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (NOISE) {         // a launch that starts inside a block of four draws evaluates it and rotates up to there
    if ((nctr & 3u) != 0u) {
      gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
      gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
      for (uint32_t q = 0; q < (nctr & 3u); ++q) {
        gA[0] = gA[1]; gA[1] = gA[2]; gA[2] = gA[3];
        gB[0] = gB[1]; gB[1] = gB[2]; gB[2] = gB[3];
      }
    }
  }

  const float dt = float(s.dt), ramp = float(s.ramp);
  const f2 two_sqrt_ab = {2.0f * tsqrt(p[2].x * p[3].x), 2.0f * tsqrt(p[2].y * p[3].y)};
  const f2 rc_v0 = {1.0f / p[0].x, 1.0f / p[0].y}, rc_ab = {1.0f / two_sqrt_ab.x, 1.0f / two_sqrt_ab.y};
  const float Lf = float(L);
  const f2 L2 = splat(Lf), rc_L2 = splat(1.0f / Lf), one = splat(1.0f), dt2 = splat(dt), ramp2 = splat(ramp);
  const f2 gap2 = splat(float(s.crash_gap));
  const double L64 = double(L), rc_L64 = 1.0 / L64, dt64 = double(s.dt), ramp64 = double(s.ramp);
  const double ms64 = double(s.max_speed), rc_ms64 = 1.0 / ms64;
  const f2 len_lead = {float(lenB), float(len_nextA)};

  // S7 / S8: SUMO's car-following acceleration and the speed-mode clamps (flowsim_pair.h SumoPair)
  SumoPair sc;
  double floor0A, floor0B, adtA, adtB, ddtA, ddtB;
  {
    const int mA = s.speed_mode[iA], mB = s.speed_mode[iB];
    const float maA = float(s.max_accel[iA]), maB = float(s.max_accel[iB]);
    const float mdA = float(s.max_decel[iA]), mdB = float(s.max_decel[iB]);
    sc.tau = f2{float(s.sumo_tau[iA]), float(s.sumo_tau[iB])};
    sc.min_gap = f2{float(s.sumo_min_gap[iA]), float(s.sumo_min_gap[iB])};
    sc.maxa = f2{maA, maB};
    sc.smax = f2{float(s.sumo_max_speed[iA]), float(s.sumo_max_speed[iB])};
    sc.rc_smax = f2{1.0f / sc.smax.x, 1.0f / sc.smax.y};
    sc.ts = f2{2.0f * tsqrt(maA * mdA), 2.0f * tsqrt(maB * mdB)};
    sc.rc_ts = f2{1.0f / sc.ts.x, 1.0f / sc.ts.y};
    const float BIG = 3.0e38f;
    sc.floor0 = f2{(mA & 1) ? 0.0f : BIG, (mB & 1) ? 0.0f : BIG};
    sc.adt = f2{(mA & 2) ? maA * dt : BIG, (mB & 2) ? maB * dt : BIG};
    sc.ddt = f2{(mA & 4) ? mdA * dt : BIG, (mB & 4) ? mdB * dt : BIG};
    floor0A = double(sc.floor0.x); floor0B = double(sc.floor0.y);
    adtA = (mA & 2) ? double(s.max_accel[iA]) * dt64 : double(BIG);
    adtB = (mB & 2) ? double(s.max_accel[iB]) * dt64 : double(BIG);
    ddtA = (mA & 4) ? double(s.max_decel[iA]) * dt64 : double(BIG);
    ddtB = (mB & 4) ? double(s.max_decel[iB]) * dt64 : double(BIG);
  }
  const bool have_act = actions != nullptr;

  // ---- state -------------------------------------------------------------------------------------------------
  f2 x, v;                       // float32 images (float: THE state)
  double xdA = 0, xdB = 0, vdA = 0, vdB = 0;
  if (MIXED) {
    xdA = double(s.pos[idx]); xdB = double(s.pos[idx + 1]);
    vdA = double(s.vel[idx]); vdB = double(s.vel[idx + 1]);
    v = f2{float(vdA), float(vdB)};
    x = f2{0.0f, 0.0f};
  } else {
    x = f2{float(s.pos[idx]), float(s.pos[idx + 1])};
    v = f2{float(s.vel[idx]), float(s.vel[idx + 1])};
  }
  // headways of the current snapshot (S10); `dgap` = the distances before the leader's length is subtracted
  f2 dgap = {0.0f, 0.0f};
  double dgA = 0, dgB = 0;
  auto headway = [&]() -> f2 {
    if (MIXED) {
      const double xn = next_a<ROW>(xdA, last, lane);
      double dA = xdB - xdA, dB = xn - xdB;
      dA = dA < 0.0 ? dA + L64 : dA;
      dB = dB < 0.0 ? dB + L64 : dB;
      dgA = dA;
      dgB = dB;
      return f2{float(dA - double(lenB)), float(dB - double(len_nextA))};
    } else {
      const f2 xl = {x.y, next_a<ROW>(x.x, last, lane)};
      f2 d = pk_sub(xl, x);
      const f2 dw = pk_add(d, L2);                 // d < 0 ? d + L : d  (d in (-L, L), never -0; d + L > d)
      d.x = nonneg_else(d.x, dw.x);
      d.y = nonneg_else(d.y, dw.y);
      dgap = d;
      return pk_sub(d, len_lead);
    }
  };
  f2 h = headway();
  f2 vl = {v.y, next_a<ROW>(v.x, last, lane)};

  // ---- actions: this lane's commands (its RL slots' columns) and, for the WaveAttenuation reward, the columns that
  // sit at its vehicles' places in the reduction (column c is summed where vehicle c stands: wave_attenuation.py:131);
  // read one step ahead so that no step waits for HBM.  One RL vehicle (the reference's ring experiments): the
  // replica's single action is one broadcast load per lane and step.
  const int num_rl = s.num_rl;
  const bool single_rl = num_rl == 1;                                            // wave-uniform
  const bool redA = WA && valid && iA < num_rl, redB = WA && valid && iB < num_rl;
  float ownA_n = 0.0f, ownB_n = 0.0f, redA_n = 0.0f, redB_n = 0.0f;
  auto load_actions = [&](int step) {
    const float* a0 = actions + size_t(step) * act_stride + size_t(rr) * num_rl;
    if (single_rl) {
      ownA_n = a0[0];
    } else {
      if (rlA) ownA_n = a0[colA];
      if (rlB) ownB_n = a0[colB];
      if (redA) redA_n = a0[iA];
      if (redB) redB_n = a0[iB];
    }
  };
  if (have_act && num_steps > 0) load_actions(0);
  // clipping as an unconditional clamp (two instructions, no branch, no select): without clip_actions the bounds are +-3e38
  const bool clip_on = s.clip_actions != 0;
  const float act_lo = clip_on ? float(s.act_lo) : -3.0e38f, act_hi = clip_on ? float(s.act_hi) : 3.0e38f;
  auto clip = [&](float a) -> float { return hmin(hmax(a, act_lo), act_hi); };

  // sigma * g of this step for the lane's two vehicles (-0.0 for a slot without noise: x + (-0) keeps every bit of x);
  // a replica that does not advance keeps its draws
  auto noise_term = [&](bool live) -> f2 {
    f2 nz = {-0.0f, -0.0f};
    if constexpr (NOISE) {
      const bool fresh = live && (nctr & 3u) == 0u;              // a new block of four draws starts with this step
      if (__ballot(fresh) != 0ull) {
        if (fresh) {
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
        }
      }
      const float tA = sigma.x * gA[0], tB = sigma.y * gB[0];
      nz.x = noisyA ? tA : -0.0f;
      nz.y = noisyB ? tB : -0.0f;
      if (live) {
        gA[0] = gA[1]; gA[1] = gA[2]; gA[2] = gA[3];
        gB[0] = gB[1]; gB[1] = gB[2]; gB[2] = gB[3];
        nctr += 1u;
      }
    }
    return nz;
  };

  // ---- heads -------------------------------------------------------------------------------------------------
  const int obs_dim = HEAD == 1 ? 3 : (HEAD == 2 ? 3 * num_rl : (HEAD == 3 ? 6 * num_rl : 2 * N));
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;
  const bool poA = HEAD == 1 && valid && rlA && colA == 0, poB = HEAD == 1 && valid && rlB && colB == 0;
  const double rc15 = 1.0 / 15.0, pml64 = double(s.po_max_length), rc_pml64 = 1.0 / pml64;
  // multi-agent heads: the slots of this lane that are RL vehicles, and -- MultiAgentAccelPOEnv's follower terms are
  // written by the FOLLOWER's lane, which holds them -- the RL vehicles this lane's slots follow (A follows B, B follows
  // the next lane's A): no lane reads backwards
  const bool maA = MA && valid && rlA, maB = MA && valid && rlB;
  const int col_nextA = MA ? __builtin_bit_cast(int, next_a<ROW>(__builtin_bit_cast(float, rlA ? colA : -1), last, lane)) : -1;      // column of B's leader, -1: not an RL vehicle
  const bool fwB = HEAD == 3 && valid && col_nextA >= 0;
  const double Lq64 = double(Lf), rc_Lq64 = 1.0 / Lq64;
  const float len_meA = float(s.length[iA]), len_meB = float(lenB);
  auto write_obs = [&]() {
    if constexpr (HEAD == 2) {
      // MultiAgentWaveAttenuationPOEnv.get_state (multiagent/ring/wave_attenuation.py:188-208): per RL vehicle, at column
      // rl_index, [v / 15, (v_lead - v) / 15, headway / max_length] (get_headway: bumper to bumper).  The lane's RL
      // half is selected first; a lane with two RL vehicles writes A's block in a second region
      if (maA || maB) {
        const float v_me = maB ? v.y : v.x, v_ld = maB ? vl.y : vl.x, h_me = maB ? h.y : h.x;
        float* o = orow + 3 * (maB ? colB : colA);
        o[0] = div_via_f64(v_me, 15.0, rc15);
        o[1] = div_via_f64(v_ld - v_me, 15.0, rc15);
        o[2] = div_via_f64(h_me, pml64, rc_pml64);
      }
      if (maA && maB) {
        float* o = orow + 3 * colA;
        o[0] = div_via_f64(v.x, 15.0, rc15);
        o[1] = div_via_f64(vl.x - v.x, 15.0, rc15);
        o[2] = div_via_f64(h.x, pml64, rc_pml64);
      }
    } else if constexpr (HEAD == 3) {
      // MultiAgentAccelPOEnv.get_state (multiagent/ring/accel.py:163-208): per RL vehicle [x / L, v / v_max,
      // (v_lead - v) / v_max, (x_lead - x - len_ego) / L (no wrap-around), (v - v_follow) / v_max, headway(follower) / L]
      const float xn = next_a<ROW>(x.x, last, lane);                  // position of B's leader
      if (maA) {
        float* o = orow + 6 * colA;
        o[0] = div_via_f64(x.x, Lq64, rc_Lq64);
        o[1] = div_via_f64(v.x, ms64, rc_ms64);
        o[2] = div_via_f64(vl.x - v.x, ms64, rc_ms64);
        o[3] = div_via_f64((x.y - x.x) - len_meA, Lq64, rc_Lq64);
      }
      if (maB) {
        float* o = orow + 6 * colB;
        o[0] = div_via_f64(x.y, Lq64, rc_Lq64);
        o[1] = div_via_f64(v.y, ms64, rc_ms64);
        o[2] = div_via_f64(vl.y - v.y, ms64, rc_ms64);
        o[3] = div_via_f64((xn - x.y) - len_meB, Lq64, rc_Lq64);
        o[4] = div_via_f64(v.y - v.x, ms64, rc_ms64);                 // B's follower is A
        o[5] = div_via_f64(h.x, Lq64, rc_Lq64);
      }
      if (fwB) {                                                      // B follows the RL vehicle in the next lane's slot A
        float* o = orow + 6 * col_nextA;
        o[4] = div_via_f64(vl.y - v.y, ms64, rc_ms64);
        o[5] = div_via_f64(h.y, Lq64, rc_Lq64);
      }
    } else if (HEAD == 1) {
      // WaveAttenuationPOEnv.get_state (wave_attenuation.py:248-269), written by the lane of the RL vehicle of column 0
      // (its half selected first: one exec-mask region, three quotients)
      if (MIXED) {
        const double vdn = next_a<ROW>(vdA, last, lane);          // (speed of B's leader)
        const double v_me = poB ? vdB : vdA, v_ld = poB ? vdn : vdB, d_me = poB ? dgB : dgA;
        if (poA || poB) {
          orow[0] = float(v_me * rc15);
          orow[1] = float((v_ld - v_me) * rc15);
          orow[2] = float(d_me * rc_pml64);
        }
      } else {
        const float v_me = poB ? v.y : v.x, v_ld = poB ? vl.y : vl.x, d_me = poB ? dgap.y : dgap.x;
        if (poA || poB) {
          orow[0] = div_via_f64(v_me, 15.0, rc15);
          orow[1] = div_via_f64(v_ld - v_me, 15.0, rc15);
          orow[2] = div_via_f64(d_me, pml64, rc_pml64);
        }
      }
    } else if (valid) {
      // AccelEnv.get_state (accel.py:116-123)
      f2 ov, ox;
      if (MIXED) {
        ov = f2{float(vdA * rc_ms64), float(vdB * rc_ms64)};
        ox = f2{float(xdA * rc_L64), float(xdB * rc_L64)};
      } else {
        ov = f2{div_via_f64(v.x, ms64, rc_ms64), div_via_f64(v.y, ms64, rc_ms64)};
        ox = FAST ? div_const2<true>(x, L2, rc_L2) : f2{x.x / L2.x, x.y / L2.y};
      }
      *reinterpret_cast<f2*>(orow + iA) = ov;
      *reinterpret_cast<f2*>(orow + N + iA) = ox;
    }
  };

  // one step without its head: controllers on the snapshot (S1), integration (S4-S9), the new snapshot (S10).
  // LIVE_ALL: no replica mask (a rollout): every replica advances and nothing is selected per lane
  // (LIVE_ALL: the caller hands in the step's noise terms `nz_in`; otherwise they come from the rotating draws)
  auto advance = [&](auto live_all, float ownA, float ownB, f2 nz_in = f2{-0.0f, -0.0f}) {
    constexpr bool LIVE_ALL = decltype(live_all)::value;
    const bool live = LIVE_ALL ? true : live_replica;
    f2 acc = idm_pair<FAST, FAST>(v, vl, h, p, two_sqrt_ab, rc_ab, rc_v0, one);
    if constexpr (NOISE) acc = pk_add(acc, LIVE_ALL ? nz_in : noise_term(live));
    const float clA = clip(ownA), clB = clip(ownB);
    acc.x = rlA ? clA : acc.x;
    acc.y = rlB ? clB : acc.y;
    const bool cmdA = !rlA || have_act, cmdB = !rlB || have_act;       // rl_actions = None: no command (S5)
    // SUMO's model is evaluated unconditionally: without a speed-mode bit its caps are 3e38 (the identity), and every
    // population this kernel is chosen for has an RL slot (FLAG_NEED_SUMO) anyway -- no wave-uniform branch in the step
    const f2 acc_s = sumo_acc_pair<FAST>(v, vl, h, sc, one);
    if (MIXED) {
      const double aA = double(acc.x), aB = double(acc.y);
      const double nA = tmax(vdA + aA * dt64, 0.0), nB = tmax(vdB + aB * dt64, 0.0);
      double cA = vdA + (nA - vdA) * ramp64, cB = vdB + (nB - vdB) * ramp64;
      {
        const double vsA = vdA + double(acc_s.x) * dt64, vsB = vdB + double(acc_s.y) * dt64;
        cA = tmin(cA, tmax(vsA, floor0A)); cB = tmin(cB, tmax(vsB, floor0B));
        cA = tmin(cA, vdA + adtA); cB = tmin(cB, vdB + adtB);
        cA = tmax(cA, vdA - ddtA); cB = tmax(cB, vdB - ddtB);
        cA = cmdA ? cA : tmax(vsA, 0.0);
        cB = cmdB ? cB : tmax(vsB, 0.0);
      }
      const double xA = xdA + cA * dt64, xB = xdB + cB * dt64;
      const double wA = xA >= L64 ? xA - L64 : xA, wB = xB >= L64 ? xB - L64 : xB;
      if (live) {
        vdA = cA; vdB = cB;
        xdA = wA; xdB = wB;
      }
      v = f2{float(vdA), float(vdB)};
    } else {
      f2 nv = pk_add(v, pk_mul(acc, dt2));
      nv.x = hmax(nv.x, 0.0f);
      nv.y = hmax(nv.y, 0.0f);
      f2 vc = pk_add(v, pk_mul(pk_sub(nv, v), ramp2));
      {                          // S7/S8: min(vc, v_sumo), min(vc, v + max_accel dt), max(vc, v - max_decel dt)
        const f2 vs = pk_add(v, pk_mul(acc_s, dt2));
        const f2 cap1 = pk_add(v, sc.adt), flo = pk_sub(v, sc.ddt);
        vc.x = hmin(vc.x, hmax(sc.floor0.x, vs.x));
        vc.y = hmin(vc.y, hmax(sc.floor0.y, vs.y));
        vc.x = hmax(hmin(vc.x, cap1.x), flo.x);
        vc.y = hmax(hmin(vc.y, cap1.y), flo.y);
        const float zA = hmax(0.0f, vs.x), zB = hmax(0.0f, vs.y);
        vc.x = cmdA ? vc.x : zA;
        vc.y = cmdB ? vc.y : zB;
      }
      const f2 xn = pk_add(x, pk_mul(vc, dt2));
      const f2 xw = pk_sub(xn, L2);                  // x_new >= L ? x_new - L : x_new  (0 <= x_new - L < x_new)
      f2 xq;
      xq.x = nonneg_else(xw.x, xn.x);
      xq.y = nonneg_else(xw.y, xn.y);
      if (live) {
        v = vc;
        x = xq;
      }
    }
    if (live) tcount += 1;
    h = headway();
    vl = f2{v.y, next_a<ROW>(v.x, last, lane)};
  };
  // the lane's terms of the replica's reductions after a step: flags (bit 0 a gap below crash_gap, bit 1 a speed below
  // -100, rewards.py:46), first sum (PO: speeds; Accel: squared deviations), second sum (PO: |actions|)
  auto terms = [&](float aredA, float aredB, unsigned& fl, float& t0, float& t1) {
    // h < gap <=> h - gap negative (no -0 from a non-zero difference, denormals are kept), v < -100 likewise; the smaller
    // of the lane's two differences carries the sign (flowsim_pair.h one_step)
    const f2 hc = pk_sub(h, gap2), vb = pk_sub(v, f2{-100.0f, -100.0f});
    fl = ((__builtin_bit_cast(unsigned, hmin(hc.x, hc.y)) >> 31) | ((__builtin_bit_cast(unsigned, hmin(vb.x, vb.y)) >> 31) << 1)) &
         (valid ? 3u : 0u);
    if (WA) {
      t0 = valid ? v.x + v.y : 0.0f;
      const float caA = tabs(clip(aredA)), caB = tabs(clip(aredB));
      t1 = (redA ? caA : 0.0f) + (redB ? caB : 0.0f);
    } else {
      const float tv = float(s.target_velocity);
      const f2 dv = {valid ? v.x - tv : 0.0f, valid ? v.y - tv : 0.0f};
      t0 = dv.x * dv.x + dv.y * dv.y;
      t1 = 0.0f;
    }
  };
  // reward and done flag of one step from its reduced terms
  auto finish = [&](unsigned fany, float s0, float s1, bool live, int t_after, float& reward, uint8_t& dflag) {
    const bool crashed = !MA && live && (fany & 1u) != 0u;          // (multiagent/base.py:188-190: crash = 0)
    const bool bad = (fany & 2u) != 0u || crashed;
    if (WA) {                                           // wave_attenuation.py:113-139
      const float mean_v = div_via_f64(s0, double(N), 1.0 / double(N));
      const float mean_a = div_via_f64(s1, double(num_rl), 1.0 / double(num_rl));
      reward = div_via_f64(4.0f * mean_v, 20.0, 1.0 / 20.0);
      if (mean_a > 0.0f) reward = reward + 4.0f * (0.0f - mean_a);
      reward = bad ? 0.0f : reward;
      reward = have_act ? reward : 0.0f;
    } else {                                                   // rewards.desired_velocity (rewards.py:6-59)
      const float cost = tsqrt(s0);
      const float max_cost = float(s.max_cost);
      reward = tmax(max_cost - cost, 0.0f) / (max_cost + 1.1920928955078125e-07f);
      reward = bad ? 0.0f : reward;
    }
    dflag = done_flag(t_after >= s.step_limit, crashed);       // envs/base.py:398-400
  };

  // one step of the "every other form" loop below (also used to align the rollout form with the noise blocks)
  auto single_step = [&](int step) {
    const float ownA = ownA_n, ownB = single_rl ? ownA_n : ownB_n;
    const float aredA = single_rl ? ownA_n : redA_n, aredB = redB_n;
    if (have_act && step + 1 < num_steps) load_actions(step + 1);
    advance(std::false_type{}, ownA, ownB);
    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      unsigned fl;
      float t0, t1;
      terms(aredA, aredB, fl, t0, t1);
      const unsigned fany = seg_or<ROW>(fl);
      write_obs();
      const float s0 = seg_sum<ROW>(t0), s1 = WA ? seg_sum<ROW>(t1) : 0.0f;
      float reward;
      uint8_t dflag;
      finish(fany, s0, s1, live_replica, tcount, reward, dflag);
      if (rvalid && k == 0) {
        *rrow = reward;
        *drow = dflag;
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
  };

  int step = 0;
  // ---- rollout form (no mask, observation every step, at most one action column): groups of 16 steps = 4 blocks of 4.
  //  * a block is straight-line code: its four reward tails are finished together (the per-lane terms are summed over
  //    the lanes at once, transposed_sum: the same xor-tree, hence the same bits; lane j of a row finishes step j);
  //  * a block is a Philox block: the launch is first walked, step by step, to a draw counter that is a multiple of
  //    four, and every block then starts with ONE evaluation of the four draws of its vehicles (no per-step test);
  //  * lane j of a row reads the action of step j of the NEXT group -- one load per lane and 16 steps, a whole group
  //    ahead of its use (an HBM round trip is ~1 us, three steps) -- and a step takes its value by a row broadcast
  //    (the row is rotated by four lanes after every block, so the broadcast lanes are compile-time constants).
  // Kept small on purpose: sixteen unrolled steps with their Philox evaluations were 11 000 instructions, past the
  // instruction cache, and ran at the speed of the single-step loop.
  constexpr bool GROUPS = ROW == 16;
  if (GROUPS && obs_every_step && mask == nullptr && (single_rl || !have_act || MC)) {
    if constexpr (NOISE) {
#pragma unroll 1
      while (step < num_steps && (nctr & 3u) != 0u) {
        single_step(step);
        step += 1;
      }
    }
    const bool deep = !MC && single_rl && have_act;
    // MC: the action columns of the lane's roles (its slots' commands, its places in the reward's sum), four steps ahead:
    // slot q of a block holds step (block start + q), refilled with step + 4 as soon as the step has taken its values.
    // Every lane loads (a role it does not hold reads column 0: no exec-mask region); only the roles' values are used.
    float qA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float qrA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, qrB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rcA = redA ? iA : 0, rcB = redB ? iB : 0;
    const bool mc_act = MC && have_act;
    const float* mc_p = actions + size_t(rr) * num_rl;
    if (mc_act) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (step + q < num_steps) {
          const float* a0 = mc_p + size_t(step + q) * act_stride;
          qA[q] = a0[colA];
          qB[q] = a0[colB];
          if (WA) { qrA[q] = a0[rcA]; qrB[q] = a0[rcB]; }
        }
      }
      mc_p += size_t(step + 4) * act_stride;
    }
    auto group_actions = [&](int first) -> float {
      const int st = first + k < num_steps ? first + k : num_steps - 1;
      return actions[size_t(st) * act_stride + size_t(rr)];
    };
    float a_cur = 0.0f, a_nxt = 0.0f;
    if (deep && step + 16 <= num_steps) a_nxt = group_actions(step);
    // WaveAttenuationPOEnv rows of a block, finished TOGETHER: the three quotients of a step are exact divisions through
    // float64 (div_via_f64: five instructions, three of them float64, all of them issued for the ONE lane that holds the
    // RL vehicle).  Instead the lane pushes its twelve numerators of a block into a row-wide shift register (rotate the
    // row by one lane, insert at the RL vehicle's lane: two instructions per value) and after the block twelve lanes of
    // the row divide one value each and store it -- one division sequence per block instead of twelve.
    // Lane offset o = (k - k_po) mod 16 ends up with numerator 11 - o (step (11 - o) / 3 of the block, value (11 - o) % 3).
    constexpr bool PO_ROWS = HEAD == 1 && !MIXED;
    const bool po_lane = poA || poB;
    const unsigned long long po_m = __ballot(po_lane);
    const int k_po = po_m ? (__builtin_ctzll(po_m) & 15) : 0;
    const int po_idx = 11 - ((k - k_po) & 15);                        // < 0: the row's four spare lanes
    const int po_st = po_idx < 0 ? 0 : po_idx / 3, po_c = po_idx < 0 ? 0 : po_idx % 3;
    const double po_div = po_c == 2 ? pml64 : 15.0, po_rc = po_c == 2 ? rc_pml64 : rc15;
    const size_t po_off = size_t(po_st) * step_rows * obs_dim + po_c;
    const bool po_store = rvalid && po_idx >= 0;
    float oacc = 0.0f;
    auto po_push = [&](float n) {
      oacc = dpp<0x120 + 1>(oacc);                                    // row_ror:1 (lane i <- lane i - 1)
      oacc = po_lane ? n : oacc;
    };
#pragma unroll 1
    for (; step + 16 <= num_steps; step += 16) {
      a_cur = a_nxt;
      if (deep && step + 32 <= num_steps) a_nxt = group_actions(step + 16);
#pragma unroll 1
      for (int blk = 0; blk < 4; ++blk) {
        if constexpr (NOISE) {
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
          nctr += 4u;
        }
        float t0[4], t1[4];
        unsigned crash_bits = 0u, bad_bits = 0u;
        static_for<4>([&](auto slot_c) {
          constexpr int slot = decltype(slot_c)::value;
          const float a = deep ? dpp<DPP_ROW_NEWBCAST0 + slot>(a_cur) : 0.0f;
          f2 nz = {-0.0f, -0.0f};
          if constexpr (NOISE) {
            const float tA = sigma.x * gA[slot], tB = sigma.y * gB[slot];
            nz.x = noisyA ? tA : -0.0f;
            nz.y = noisyB ? tB : -0.0f;
          }
          const float oA = MC ? qA[slot] : a, oB = MC ? qB[slot] : a;
          const float rA = MC ? qrA[slot] : a, rB = MC ? qrB[slot] : 0.0f;
          if constexpr (MC) {
            if (mc_act && step + 4 * blk + slot + 4 < num_steps) {     // wave-uniform
              qA[slot] = mc_p[colA];
              qB[slot] = mc_p[colB];
              if (WA) { qrA[slot] = mc_p[rcA]; qrB[slot] = mc_p[rcB]; }
            }
            mc_p += act_stride;
          }
          advance(std::true_type{}, oA, oB, nz);
          if constexpr (PO_ROWS) {
            const float v_me = poB ? v.y : v.x, v_ld = poB ? vl.y : vl.x, d_me = poB ? dgap.y : dgap.x;
            po_push(v_me);
            po_push(v_ld - v_me);
            po_push(d_me);
          } else {
            write_obs();
            orow += step_rows * obs_dim;
          }
          unsigned fl;
          terms(rA, rB, fl, t0[slot], t1[slot]);
          crash_bits = (crash_bits << 1) | (fl & 1u);
          bad_bits = (bad_bits << 1) | (fl >> 1);
        });
        if constexpr (PO_ROWS) {
          const float q = div_via_f64(oacc, po_div, po_rc);
          if (po_store) orow[po_off] = q;
          orow += size_t(4) * step_rows * obs_dim;
        }
        const float s0 = transposed_sum<ROW, 4>(t0, lane);
        const float s1 = WA ? transposed_sum<ROW, 4>(t1, lane) : 0.0f;
        const unsigned crash_any = seg_or<ROW>(crash_bits), bad_any = seg_or<ROW>(bad_bits);
        if (k < 4) {                                           // lane k finishes step k of the block
          const unsigned fany = ((crash_any >> (3 - k)) & 1u) | (((bad_any >> (3 - k)) & 1u) << 1);
          float reward;
          uint8_t dflag;
          finish(fany, s0, s1, true, tcount - (3 - k), reward, dflag);
          if (rvalid) {
            rrow[size_t(k) * s.R] = reward;
            drow[size_t(k) * s.R] = dflag;
          }
        }
        rrow += size_t(4) * s.R;
        drow += size_t(4) * s.R;
        a_cur = dpp<0x120 + 12>(a_cur);                        // row_ror:12 (lane i <- lane i + 4): the next block's actions into lanes 0..3
      }
    }
    if constexpr (NOISE) {       // the single-step loop keeps its draws rotated: none are held at a block boundary
      gA[0] = gA[1] = gA[2] = gA[3] = 0.0f;
      gB[0] = gB[1] = gB[2] = gB[3] = 0.0f;
    }
  }
  // ---- every other form (a replica mask, the observation of the last step only, several action columns, the steps
  // left over by the groups)
  if (have_act && step > 0 && step < num_steps) load_actions(step);
#pragma unroll 1
  for (; step < num_steps; ++step) single_step(step);

  if (num_steps == 0) {          // observation of the current state only (Env.reset, envs/base.py:544-551)
    write_obs();
    return;
  }
  if (valid && live_replica) {
    if (MIXED) {
      s.pos[idx] = T(xdA); s.pos[idx + 1] = T(xdB);
      s.vel[idx] = T(vdA); s.vel[idx + 1] = T(vdB);
    } else {
      s.pos[idx] = T(x.x); s.pos[idx + 1] = T(x.y);
      s.vel[idx] = T(v.x); s.vel[idx + 1] = T(v.y);
    }
    if (kk == 0) s.time[rr] = tcount;
    if (NOISE && kk == 0) s.noise_ctr[rr] = nctr;
  }
}

}  // namespace fs
