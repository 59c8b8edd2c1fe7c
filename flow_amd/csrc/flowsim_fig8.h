// flowsim_fig8.h -- k_rollout_loop: rollout kernel of closed single-lane loops WITH a segment table, a crossing and
// mixed IDM / RL / SUMO-driven populations (BASELINE configs[2]: FigureEightNetwork, 13 noisy IDM + 1 RL vehicle).
//
// Same arithmetic as the generic k_steps<float, SEG, 0, 1> (the helper functions and their operation order are
// shared; tests/test_parity_gpu.py::test_loop_rollout_kernel_equals_generic_kernel compares the two bit for bit,
// noise included), restructured for a wave that is ALONE on its SIMD (4096 replicas x 14 vehicles = 1024 waves):
// every instruction costs an issue slot and nothing hides a wait, so
//   * the segment a vehicle is on is an index advanced by compares; its table row (start of the next segment,
//     Flow-table start and slope) is re-read from an LDS copy of the table with plain ds_read -- issued right after
//     the move, consumed at the end of the step;
//   * the per-replica facts of a step (stream a busy, stream b in the box, both streams on the crossing point, a
//     headway below the crash gap, a speed below -100) are bits of ONE word reduced with one OR-butterfly (DPP);
//   * a vehicle is on at most one approach of the crossing: ONE evaluation of the SUMO car-following speed towards
//     the stop line serves both yield rules;
//   * one Philox call yields the noise of four steps (gauss4);
//   * the reward tail (sqrt, divide) of PERIOD steps is finished at once by PERIOD lanes (transposed_sum), as in
//     k_rollout_idm; launch constants live in registers, actions are read one step ahead.
// Chosen by the host when: float, segment table present, every slot IDM / RL / Sim without fail-safe, Euler,
// sims_per_step 1, no reset mask, no sorting / shuffling, AccelEnv (not evaluate) or
// WaveAttenuationPOEnv head, observation every step, N <= 16.
#pragma once
#include <type_traits>
#include "flowsim_kernels.h"

namespace fs {

// a launch constant kept in a VGPR (the loop has far more uniform values than SGPRs: hipcc spilled 45 of them to
// VGPR lanes and read them back with v_readlane inside the loop)
__device__ __forceinline__ float in_vgpr(float x) { asm volatile("" : "+v"(x)); return x; }
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }
// x / c for a launch constant c: the float64 route of div_via_f64 (exact for every float x)
struct DivC { double c, rc; float cf, rcf; };
__device__ __forceinline__ DivC make_divc(float c) {
  return DivC{in_vgpr(double(c)), in_vgpr(1.0 / double(c)), in_vgpr(c), in_vgpr(1.0f / c)};
}
__device__ __forceinline__ float divc(float x, const DivC& d) { return div_via_f64(x, d.c, d.rc); }
// FASTC: the controller's constant divisions as div_const (multiply by RN(1/c), one residual correction: 3 float32
// instructions instead of 5 through float64) -- host-verified per divisor, and for dividends below 2^-70 without effect
// on the results they feed (the argument of flowsim_kernels.h div_const: pw = (v/v0)^4 underflows either way, a tiny
// v (v - vl) / c vanishes next to s0 / minGap >= 1e-3).  Observations are OUTPUTS: they keep divc.
template <bool FASTC>
__device__ __forceinline__ float divk(float x, const DivC& d) {
  return FASTC ? div_const<true>(x, d.cf, d.rcf) : div_via_f64(x, d.c, d.rc);
}
// value of the NEXT slot of the 16-lane row (slot 0's for the last occupied slot and the idle lanes)
__device__ __forceinline__ float lead16(float v, bool wrap) {
  const float t = dpp<DPP_ROW_SHL1>(v), w = dpp<0x150>(v);     // row_shl:1, row_newbcast:0
  return wrap ? w : t;
}

// x >= y ? x - y : x for 0 <= x < 2 y (a position past the end of the loop), and d < 0 ? d + y : d for -y <= d < y (an
// arc): non-negative floats order like their bit patterns and a negative one is a huge unsigned, so both are one v_min_u32
__device__ __forceinline__ float wrap_down(float x, float y) {
  return __builtin_bit_cast(float, min(__builtin_bit_cast(unsigned, x), __builtin_bit_cast(unsigned, x - y)));
}
__device__ __forceinline__ float wrap_up(float d, float y) {
  return __builtin_bit_cast(float, min(__builtin_bit_cast(unsigned, d), __builtin_bit_cast(unsigned, d + y)));
}

// ---- FS_MIXED instantiation (MX): positions, speeds, the loop geometry and every decision taken on a position are float64
// (the reference's arithmetic type), the car-following models are evaluated in float32 on the rounded speeds and gaps
__device__ __forceinline__ double lead16(double v, bool wrap) {
  const double t = dpp<DPP_ROW_SHL1>(v), w = dpp<0x150>(v);
  return wrap ? w : t;
}
__device__ __forceinline__ double wrap_down(double x, double y) { return x >= y ? x - y : x; }
__device__ __forceinline__ double wrap_up(double d, double y) { return d < 0.0 ? d + y : d; }
__device__ __forceinline__ float xmax(float a, float b) { return hmax(a, b); }
__device__ __forceinline__ float xmin(float a, float b) { return hmin(a, b); }
__device__ __forceinline__ double xmax(double a, double b) { return a > b ? a : b; }
__device__ __forceinline__ double xmin(double a, double b) { return a < b ? a : b; }

// ctrl_idm / sumo_idm_speed (flowsim_kernels.h) with their divisions made cheap WITHOUT changing a bit: divisors
// that are launch constants go through divc, the others (|h| >= 1e-3, gap >= 1e-3; dividends s* >= s0 >= 1e-3 and
// ss >= minGap >= 1e-3, host-checked) through div_core
struct IdmC { float p1, p2, p4, p5; DivC v0, two_sqrt; };
template <bool DELTA4, bool FASTC = false>
__device__ __forceinline__ float idm_fast(float v, float vl, float h, bool has, const IdmC& c) {
  const float hh = tabs(h) < 1e-3f ? 1e-3f : h;
  const float dyn = v * c.p1 + divk<FASTC>(v * (v - vl), c.two_sqrt);
  const float s_star = has ? c.p5 + hmax(0.0f, dyn) : 0.0f;
  const float q = div_core(s_star, hh);
  const float ratio = divk<FASTC>(v, c.v0);
  float pw;
  if (DELTA4) { const float r2 = ratio * ratio; pw = r2 * r2; } else pw = pow_delta(ratio, c.p4);
  return c.p2 * (1.0f - pw - q * q);
}
struct SumoC { float min_gap, tau, max_accel; DivC two_sqrt, max_speed; };
template <bool FASTC = false>
__device__ __forceinline__ float sumo_acc_fast(float v, float vl, float h, bool has, const SumoC& c) {
  const float gap = hmax(h, 1e-3f);
  const float ss = c.min_gap + hmax(0.0f, v * c.tau + divk<FASTC>(v * (v - vl), c.two_sqrt));
  const float q = has ? div_core(ss, gap) : 0.0f;
  const float r = divk<FASTC>(v, c.max_speed);
  const float r2 = r * r;
  return c.max_accel * (1.0f - r2 * r2 - q * q);
}
template <bool FASTC = false>
__device__ __forceinline__ float sumo_fast(float v, float vl, float h, bool has, float dt, const SumoC& c) {
  return hmax(0.0f, v + sumo_acc_fast<FASTC>(v, vl, h, has, c) * dt);
}
// (MX: the model's acceleration in float32 on the rounded state, its speed in float64)
template <bool FASTC = false>
__device__ __forceinline__ double sumo_fast(double v, double vl, float h, bool has, double dt, const SumoC& c) {
  return xmax(0.0, v + double(sumo_acc_fast<FASTC>(float(v), float(vl), h, has, c)) * dt);
}

// FULL: the launch is known to have noisy slots, speed-mode clamps / uncommanded slots, the crossing and an action
// tensor (BASELINE's C3 and the reference's figure-eight experiments): the four launch-constant tests become
// compile-time facts instead of taken branches (a wave alone on its SIMD pays an instruction-fetch bubble for each);
// and the host has proven the controller's constant divisors for div_const (FASTC, Sim::loop_fastc_ok).
// Predicates as VALU integers.  hipcc turns every `a & b` of two float compares into v_cmp, v_cmp, s_and_b64 -- and a
// scalar operation on a VALU-written mask waits ~14 cycles for it (scripts/ubench), ~30 times per step of this kernel.
// A float difference carries the same fact in its sign bit (a - b is never -0 for a != b, +0 for a == b; denormals
// are kept): sign masks are combined with v_and / v_or / v_bfi and become a bool (one v_cmp) only where a select
// needs it.  Bit 31 of the result is the predicate, the other bits are junk.
__device__ __forceinline__ unsigned fbits(float a) { return __builtin_bit_cast(unsigned, a); }
__device__ __forceinline__ unsigned sm_lt(float a, float b) { return fbits(a - b); }                      // a < b
__device__ __forceinline__ unsigned sm_in(float x, float lo, float hi) { return ~fbits(x - lo) & fbits(x - hi); }  // lo <= x < hi
__device__ __forceinline__ unsigned fbits(double a) { return unsigned(__builtin_bit_cast(unsigned long long, a) >> 32); }
__device__ __forceinline__ unsigned sm_lt(double a, double b) { return fbits(a - b); }
__device__ __forceinline__ unsigned sm_in(double x, double lo, double hi) { return ~fbits(x - lo) & fbits(x - hi); }
__device__ __forceinline__ bool sm_true(unsigned m) { return int(m) < 0; }

// A wave-wide test whose result steers a branch, evaluated HERE: a scalar branch that reads a VALU-written mask in
// the next instruction stands still for ~35 cycles (scripts/ubench: v_cmp + s_cbranch_vccz 44.6 cycles against 13 for
// the two alone) -- a wave alone on its SIMD has nothing to fill that with, so every such test of the step is
// evaluated as early as its inputs exist and consumed later.
__device__ __forceinline__ unsigned long long ballot_here(bool p) {
  unsigned long long m = __ballot(p);
  asm volatile("" : "+s"(m));
  return m;
}

template <int HEAD /* 0: AccelEnv, 1: WaveAttenuationPOEnv, 2: MultiAgentAccelPOEnv (float32 only) */, bool DELTA4 /* every IDM slot has delta = 4 */,
          bool FULL = false, bool MX = false /* FS_MIXED: float64 state and geometry (X), float32 car-following models (T) */>
__global__ __launch_bounds__(256) void k_rollout_loop(DevView<typename std::conditional<MX, double, float>::type> s,
                                                      int num_steps, const float* __restrict__ actions, size_t act_stride,
                                                      float* __restrict__ obs, float* __restrict__ rew,
                                                      uint8_t* __restrict__ done) {
  typedef float T;
  typedef typename std::conditional<MX, double, float>::type X;
  constexpr int SEG = 16, RPW = 4, PERIOD = 4;
  __shared__ X tab_start[FS_MAX_SEGMENTS + 2], tab_fs[FS_MAX_SEGMENTS + 2], tab_sl[FS_MAX_SEGMENTS + 2];
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int seg = lane / SEG;
  const int i = lane % SEG;
  const int r = wave * RPW + seg;
  const int N = s.N;
  const bool rvalid = r < s.R;
  const bool valid = rvalid && i < N;
  const int rr = rvalid ? r : s.R - 1;
  const int ii = i < N ? i : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const bool wrap_lead = (i + 1 >= N);
  constexpr bool has = true;                          // N > 1 (host-checked)
  const int flags = s.flags;

  if (threadIdx.x < FS_MAX_SEGMENTS + 2) {
    const int q = threadIdx.x, qq = q < FS_MAX_SEGMENTS ? q : 0;
    tab_start[q] = q < s.nseg ? s.seg_start[qq] : X(3.0e38);
    tab_fs[q] = q < s.nseg ? s.seg_flow_start[qq] : X(0);
    tab_sl[q] = q < s.nseg ? s.seg_flow_slope[qq] : X(0);
  }
  __syncthreads();

  Slot<T> sl;
  sl.ctrl = s.ctrl[ii];
  sl.failsafe = 0;
  sl.speed_mode = s.speed_mode[ii];
  sl.rl_index = s.rl_index[ii];
  sl.pis_index = -1;
#pragma unroll
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) sl.p[k] = T(s.p[k * N + ii]);
  sl.noise = T(s.noise[ii]);
  sl.delay = T(0);
  sl.max_accel = T(s.max_accel[ii]);
  sl.max_decel = T(s.max_decel[ii]);
  sl.length = T(s.length[ii]);
  sl.sumo_tau = T(s.sumo_tau[ii]);
  sl.sumo_min_gap = T(s.sumo_min_gap[ii]);
  sl.sumo_max_speed = T(s.sumo_max_speed[ii]);
  const X len_me = s.length[ii];
  const X len_lead = lead16(len_me, wrap_lead);

  const X base_len = s.ring_len[rr];
  const X L = base_len + X(4) * s.jlen;
  int tcount = s.time[rr];
  const bool any_noise = FULL || (flags & FLAG_HAS_NOISE) != 0;
  uint32_t nctr = any_noise ? s.noise_ctr[rr] : 0u;
  const bool noisy = any_noise && sl.noise > T(0) && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM;

  X x = s.pos[idx];
  X v = s.vel[idx];
  // segment cursor: index + the row values read from LDS
  int k = 0;
  for (int q = 1; q < s.nseg; ++q) k = (x >= tab_start[q]) ? q : k;
  X c_st = tab_start[k], c_next = tab_start[k + 1], c_next2 = tab_start[k + 2], c_fs = tab_fs[k], c_sl = tab_sl[k];
  X xl = lead16(x, wrap_lead);
  X vl = lead16(v, wrap_lead);
  X d = xl - x;
  d = d < X(0) ? d + L : d;
  T h = has ? T(d - len_lead) : T(1000);

  // launch constants in VGPRs
  const X dt = in_vgpr(X(s.dt)), ramp = in_vgpr(X(s.ramp));
  const T crash_gap = in_vgpr(T(s.crash_gap)), target_v = in_vgpr(T(s.target_velocity));
  const X ja_in = in_vgpr(X(s.ja_in)), ja_out = in_vgpr(X(s.ja_out)), jb_in = in_vgpr(X(s.jb_in)), jb_out = in_vgpr(X(s.jb_out));
  const X look = in_vgpr(X(s.j_lookahead)), tgap = in_vgpr(X(s.j_time_gap));
  const X za_lo = in_vgpr(X(s.za_lo)), za_hi = in_vgpr(X(s.za_hi)), zb_lo = in_vgpr(X(s.zb_lo)), zb_hi = in_vgpr(X(s.zb_hi));
  const T max_cost = in_vgpr(T(s.max_cost));
  const X Lv = in_vgpr(L);
  const DivC d_ms = make_divc(T(s.max_speed)), d_L = make_divc(T(L)), d_15 = make_divc(15.0f), d_po = make_divc(T(s.po_max_length));
  // (MX: observations as ring FS_MIXED writes them: float(value * RN64(1 / c)))
  const double rc_ms64 = 1.0 / double(s.max_speed), rc_L64 = 1.0 / double(L), rc_15_64 = 1.0 / 15.0, rc_po64 = 1.0 / double(s.po_max_length);
  IdmC ic;
  ic.p1 = sl.p[1]; ic.p2 = sl.p[2]; ic.p4 = sl.p[4]; ic.p5 = sl.p[5];
  ic.v0 = make_divc(sl.p[0]);
  ic.two_sqrt = make_divc(T(2) * tsqrt(sl.p[2] * sl.p[3]));
  SumoC sc;
  sc.min_gap = sl.sumo_min_gap; sc.tau = sl.sumo_tau; sc.max_accel = sl.max_accel;
  sc.two_sqrt = make_divc(T(2) * tsqrt(sl.max_accel * sl.max_decel));
  sc.max_speed = make_divc(sl.sumo_max_speed);
  const unsigned seg_internal = s.seg_internal;
  const bool junction_on = FULL || s.junction_on != 0, need_sumo = FULL || (flags & FLAG_NEED_SUMO) != 0;
  const bool gated = s.junction_mode && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM;
  // (clipping as an unconditional clamp: without clip_actions the bounds are +-3e38)
  const T clip_lo = in_vgpr(s.clip_actions != 0 ? T(s.act_lo) : T(-3.0e38)), clip_hi = in_vgpr(s.clip_actions != 0 ? T(s.act_hi) : T(3.0e38));
  const bool rl_lane = sl.ctrl == FS_CTRL_RL, sim_lane = sl.ctrl == FS_CTRL_SIM;
  const bool use_act = FULL || actions != nullptr;
  const int num_rl = s.num_rl;
  const int own_col = sl.rl_index < 0 ? 0 : sl.rl_index;
  const bool red_lane = ii < num_rl && i < N;
  static_assert(!(HEAD == 2 && MX), "the multi-agent head exists in float32 only");
  const int obs_dim = HEAD == 1 ? 3 : (HEAD == 2 ? 6 * num_rl : 2 * N);
  const bool obs_lane = HEAD == 1 ? (valid && rl_lane && sl.rl_index == 0) : (HEAD == 2 ? (valid && rl_lane) : valid);
  // MultiAgentAccelPOEnv (multiagent/ring/accel.py:163-208; the reference's multiagent_figure_eight.py): an RL vehicle's
  // block holds two terms of its FOLLOWER -- written by the follower's lane, which holds them (no lane reads backwards)
  const int lead_col = HEAD == 2 ? __builtin_bit_cast(int, lead16(__builtin_bit_cast(float, rl_lane ? own_col : -1), wrap_lead)) : -1;
  const bool fw_lane = HEAD == 2 && valid && lead_col >= 0;
  const unsigned valid_bits = valid ? 0x3Fu : 0u;         // flag words of idle lanes are empty
  const unsigned gate_u = gated ? 1u : 0u, cmd_rl = (rl_lane && use_act) ? 1u : 0u,
                 cmd_other = (!rl_lane && !sim_lane) ? 1u : 0u, sm1_u = unsigned(sl.speed_mode) & 1u;
  const bool sm1_lane = (sl.speed_mode & 1) != 0;
  const X adt_c = (sl.speed_mode & 2) ? X(s.max_accel[ii]) * dt : X(3.0e38), ddt_c = (sl.speed_mode & 4) ? X(s.max_decel[ii]) * dt : X(3.0e38);

  // RL actions are read PERIOD steps ahead, into the register the step PERIOD steps earlier has just consumed (slot s
  // of a block <-> element s: static after unrolling).  The action tensor is streamed once -- every step's row is a
  // cold line from HBM, several thousand cycles away under load, more than one step of this kernel: with the load
  // issued ONE step ahead (the first version) the wave stood waiting for it a third of its life.
  float a_own_q[PERIOD] = {0.0f, 0.0f, 0.0f, 0.0f}, a_red_q[PERIOD] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int red_col = red_lane ? ii : 0;                 // every lane loads (no exec-mask branch); only its role's value is used
  if (use_act) {
#pragma unroll
    for (int q = 0; q < PERIOD; ++q) {
      if (q < num_steps) {
        const float* a0 = actions + size_t(q) * act_stride + size_t(rr) * num_rl;
        a_own_q[q] = a0[own_col];
        a_red_q[q] = a0[red_col];
      }
    }
  }
  const float* a_own_p = actions + size_t(PERIOD) * act_stride + size_t(rr) * num_rl + own_col;
  const float* a_red_p = actions + size_t(PERIOD) * act_stride + size_t(rr) * num_rl + red_col;
  // the four draws of the current noise block, ROTATED so that g4[0] is always the draw of the next step (no
  // per-step index select); a launch that starts in the middle of a block evaluates it and rotates up to there
  T g4[4] = {T(0), T(0), T(0), T(0)};
  if (any_noise && (nctr & 3u) != 0u && noisy) {
    gauss4<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr >> 2, g4, s.noise_exact != 0);
    for (uint32_t q = 0; q < (nctr & 3u); ++q) { g4[0] = g4[1]; g4[1] = g4[2]; g4[2] = g4[3]; }
  }

  float* orow = obs + size_t(rr) * obs_dim;
  const size_t obs_step = size_t(s.R) * obs_dim;
  unsigned crash_bits = 0u, bad_bits = 0u;
  // Stores are issued ONE STEP LATE (the values wait in registers): the action of the next step is a vector load,
  // vector loads and stores share vmcnt and complete in order, so waiting for that load at the top of a step also
  // waits for every store issued before the wait -- with the stores at the end of the previous step the wave stood
  // still until they had landed (38 % of its life in the first version).  Issued right AFTER the wait, they have a
  // whole step to complete before the next one.
  float po0 = 0.0f, po1 = 0.0f, po2 = 0.0f, po3 = 0.0f, po5 = 0.0f;   // pending observation values of the previous step
  bool pend_obs = false;
  float prew = 0.0f;                                      // pending reward / done of the previous block
  uint8_t pdone = 0;
  size_t prew_at = 0;
  bool pend_rew = false;
  auto flush_obs = [&]() {
    if (pend_obs) {                                       // wave-uniform
#ifndef FS_DIAG_LOOP_NOSTORE                                // timing experiment only
      if (obs_lane) {
        if (HEAD == 1) { orow[0] = po0; orow[1] = po1; orow[2] = po2; }
        else if (HEAD == 2) { float* o = orow + 6 * own_col; o[0] = po0; o[1] = po1; o[2] = po2; o[3] = po3; }
        else { orow[ii] = po0; orow[N + ii] = po1; }
      }
      if (HEAD == 2 && fw_lane) {                          // (v_rl - v_me) / v_max is this lane's own third value
        float* o = orow + 6 * lead_col;
        o[4] = po2;
        o[5] = po5;
      }
#else
      asm volatile("" :: "v"(po0), "v"(po1), "v"(po2), "v"(po3), "v"(po5));
#endif
      orow += obs_step;
      pend_obs = false;
    }
  };
  auto flush_rew = [&]() {
    if (pend_rew) {                                       // per lane
      rew[prew_at] = prew;
      done[prew_at] = pdone;
      pend_rew = false;
    }
  };
  X prev_v = v;                                           // track_aux: get_previous_speed / get_accel of the scalar Env
  T last_acc = T(0);

  // stream a busy (bit 0: a vehicle of stream a inside the box, not yet clear of it with its tail, or within time_gap
  // of it) / stream b in the box (bit 1), of ONE lane for the state (xx, vv)
  auto junction_flags = [&](X xx, X vv) -> unsigned {
    const unsigned busy_a = sm_in(xx, ja_in - tgap * vv, ja_out + len_me);
    const unsigned in_b = sm_in(xx, jb_in, jb_out + len_me);
    return (busy_a >> 31) | ((in_b >> 31) << 1);
  };
  unsigned jf = junction_on ? seg_or<SEG>(junction_flags(x, v) & valid_bits) : 0u;    // of the launch's first snapshot

  // one block of PERIOD steps; FB (a full block): the per-step "is this step inside the launch" test is a compile-time
  // fact -- the tail of a launch (num_steps % PERIOD steps) runs the same body with the test
  auto run_block = [&](const int base, auto full_block) {
    constexpr bool FB = decltype(full_block)::value;
    T red[PERIOD];                // per-step reward ingredients: AccelEnv (v - target)^2, PO: speeds (mean) in red, |a| in red2
    T red2[PERIOD];
    const int nsteps = FB ? PERIOD : num_steps - base;
#pragma unroll
    for (int slot = 0; slot < PERIOD; ++slot) {
      red[slot] = T(0);
      red2[slot] = T(0);
      if (FB || slot < nsteps) {                             // wave-uniform
        const int step = base + slot;
        // the step's wave-wide tests whose inputs are the snapshot: evaluated first, consumed where they steer
        const unsigned long long draw_m = any_noise ? ballot_here(noisy && (nctr & 3u) == 0u) : 0ull;
        bool on_a = false, on_b = false, on_any = false, on_both = false;
        unsigned long long cap_m = 0ull, cap2_m = 0ull;
        if (junction_on) {
          // per-replica facts of the snapshot (bit 0 stream a busy, bit 1 stream b in the box): `jf`, reduced over the
          // replica together with the collision flags of the state it describes -- at the end of the previous step
          const unsigned on_b_m = sm_in(x, jb_in - look, jb_in) & (jf << 31);
          const unsigned on_a_m = sm_in(x, ja_in - look, ja_in) & (jf << 30);
          on_b = sm_true(on_b_m);
          on_a = sm_true(on_a_m);
          on_any = sm_true(on_a_m | on_b_m);
          on_both = sm_true(on_a_m & on_b_m);
          cap_m = ballot_here(on_any);
          cap2_m = ballot_here(on_both);
        }
        const float a_own = a_own_q[slot], a_red = a_red_q[slot];
        asm volatile("" :: "v"(a_own), "v"(a_red));             // the wait for the prefetched action happens HERE
        flush_obs();
        if (slot == 0) flush_rew();
        if (use_act && step + PERIOD < num_steps) {
          a_own_q[slot] = *a_own_p;                              // the row of step + PERIOD (running pointers: one
          a_red_q[slot] = *a_red_p;                              // 64-bit add per step instead of the index arithmetic)
          a_own_p += act_stride;
          a_red_p += act_stride;
        }
        // ---- controllers on the snapshot (control_accel_on, CSET = 1) ----------------------------------
        // commanded (base_controller.py:93-106): an RL lane when there are actions, never a SimCarFollowing lane, any
        // other lane unless it is gated off an internal edge -- as an integer (constants: cmd_rl, cmd_other, gate_u)
        const unsigned on_edge_u = 1u ^ (gate_u & (seg_internal >> k));
        const unsigned commanded_u = (on_edge_u & cmd_other) | cmd_rl;
        const bool commanded = commanded_u != 0u;
        T acc = T(0);
        if (any_noise) {
          if (draw_m != 0ull) {
            if (noisy && (nctr & 3u) == 0u)
              gauss4<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr >> 2, g4, s.noise_exact != 0);
          }
        }
        {
          T a = idm_fast<DELTA4, FULL>(T(v), T(vl), h, has, ic);
          // (every lane evaluates it: left to itself hipcc wraps the evaluation into an exec-mask region of the lanes that
          // use it -- s_and_saveexec, a branch, s_or -- which costs a wave alone on its SIMD more than the masked lanes save)
          if (FULL) asm volatile("" : "+v"(a));
          if (any_noise) {
            const T an = a + sl.noise * g4[0];
            a = noisy ? an : a;
            g4[0] = g4[1]; g4[1] = g4[2]; g4[2] = g4[3];
          }
          const T arl = hmin(hmax(T(a_own), clip_lo), clip_hi);
          acc = rl_lane ? (use_act ? arl : T(0)) : (sim_lane ? T(0) : a);
        }
        // ---- apply_acceleration + SUMO integration (S4-S9) ---------------------------------------------
        X next_vel = xmax(v + X(acc) * dt, X(0));
        X vc = v + (next_vel - v) * ramp;
        X v_new = vc;
        if (need_sumo) {
          X v_sumo = sumo_fast<FULL>(v, vl, h, has, dt, sc);
          // S7/S8 without a per-slot test: a slot whose bit is clear holds 3e38 in the clamp's place (k_rollout_pair's form)
          vc = xmin(vc, sm1_lane ? v_sumo : X(3.0e38));
          vc = xmin(vc, v + adt_c);
          vc = xmax(vc, v - ddt_c);
          v_new = commanded ? vc : v_sumo;
        }
        if (junction_on) {
          // (jf, on_a, on_b and the two tests: top of the step)
          // a vehicle is on at most one approach (the two lines are different places of the loop); should both
          // hold for a degenerate table, stream b's line is evaluated first and stream a's overrides as min would
          if (cap_m != 0ull) {
            const X line = on_b ? jb_in - x : ja_in - x;
            X cap = sumo_fast<FULL>(v, X(0), T(line), true, dt, sc);
            cap = on_any ? cap : X(3.0e38);
            if (cap2_m != 0ull) {                                // degenerate table: both lines ahead of one vehicle
              const X cap_a = sumo_fast<FULL>(v, X(0), T(ja_in - x), true, dt, sc);
              cap = xmin(cap, on_both ? cap_a : X(3.0e38));
            }
            const bool cap_applies = (sm1_u | (commanded_u ^ 1u)) != 0u;          // (speed_mode & 1) || !commanded
            v_new = xmin(v_new, cap_applies ? cap : X(3.0e38));
          }
        }
        X x_new = x + v_new * dt;
        x_new = wrap_down(x_new, Lv);
        prev_v = v;
        last_acc = acc;
        x = x_new;
        v = v_new;
        tcount += 1;
        nctr += 1u;
        // ---- segment cursor: advance by compares, re-read the row from LDS --------------------------------
        // (up to two starts passed per step without a table access; the row of the new segment is read from LDS
        // now and consumed at the end of the step, the rare third advance / advance after a wrap is caught there)
        k = (x < c_st) ? 0 : k + ((x >= c_next) ? 1 : 0) + ((x >= c_next2) ? 1 : 0);
        const X n_st = tab_start[k], n_next = tab_start[k + 1], n_next2 = tab_start[k + 2];
        const X n_fs = tab_fs[k], n_sl = tab_sl[k];
        // the reads must be ISSUED here: without the fence hipcc sinks them into the block of their first use (the
        // `while` below) and the wave waits out two LDS latencies there, every step
        asm volatile("" ::: "memory");
        // ---- new neighbour snapshot (S10) + per-replica facts of the new state -----------------------------
        xl = lead16(x, wrap_lead);
        vl = lead16(v, wrap_lead);
        d = wrap_up(xl - x, Lv);
        h = has ? T(d - len_lead) : T(1000);
        // bit 0 a gap below crash_gap, bits 1 / 2 a body on the crossing point of stream a / b, bit 3 v < -100 (sign masks)
        unsigned f2 = sm_lt(h, crash_gap) >> 31;
        if (junction_on) {
          f2 |= (sm_in(x, za_lo, za_hi) >> 31) << 1;
          f2 |= (sm_in(x, zb_lo, zb_hi) >> 31) << 2;
        }
        f2 |= (sm_lt(v, X(-100)) >> 31) << 3;
        if (junction_on) f2 |= junction_flags(x, v) << 4;        // the next step's jf rides in bits 4 / 5 of the same butterfly
        f2 &= valid_bits;
        unsigned long long adv_m = ballot_here(x >= n_next);     // a third start passed / one passed after a wrap (rare)
        f2 = seg_or<SEG>(f2);
        jf = (f2 >> 4) & 3u;
        // (f2 & 1) || ((f2 & 6) == 6); the multi-agent head sees no crash (multiagent/base.py:188-190)
        const unsigned crashed = HEAD == 2 ? 0u : (f2 | ((f2 >> 1) & (f2 >> 2))) & 1u;
        const unsigned bad = ((f2 >> 3) | crashed) & 1u;
        crash_bits |= crashed << slot;
        bad_bits |= bad << slot;
        // ---- the segment row read above
        c_st = n_st; c_next = n_next; c_next2 = n_next2; c_fs = n_fs; c_sl = n_sl;
        while (adv_m != 0ull) {                                  // rare
          asm volatile("" ::: "memory");                         // (keeps the body's reads from being hoisted above the test)
          k += (x >= c_next) ? 1 : 0;
          c_st = tab_start[k];
          c_next = tab_start[k + 1];
          c_next2 = tab_start[k + 2];
          c_fs = tab_fs[k];
          c_sl = tab_sl[k];
          adv_m = __ballot(x >= c_next);
        }
        // ---- observation ---------------------------------------------------------------------------------
        if (HEAD == 1) {
          if constexpr (MX) {
            po0 = float(double(v) * rc_15_64);
            po1 = float(double(vl - v) * rc_15_64);
            po2 = float(double(d) * rc_po64);
          } else {
            po0 = divc(v, d_15);                                 // wave_attenuation.py:248-269
            po1 = divc(vl - v, d_15);
            po2 = divc(d, d_po);
          }
          red[slot] = valid ? T(v) : T(0);
          T a = T(0);
          if (red_lane && use_act) {
            a = tabs(hmin(hmax(T(a_red), clip_lo), clip_hi));
          }
          red2[slot] = a;
        } else if constexpr (HEAD == 2) {
          const T xo = c_fs + c_sl * (x - c_st);
          const T xol = lead16(xo, wrap_lead);
          po0 = divc(xo, d_L);
          po1 = divc(v, d_ms);
          po2 = divc(vl - v, d_ms);
          po3 = divc((xol - xo) - len_me, d_L);                  // (:186-188: no wrap-around, the ego's length)
          po5 = divc(h, d_L);                                    // get_headway(follower): for the leader's block
          const T dv = valid ? T(v) - target_v : T(0);
          red[slot] = dv * dv;
        } else {
          const X xo = c_fs + c_sl * (x - c_st);
          if constexpr (MX) {
            po0 = float(double(v) * rc_ms64);
            po1 = float(double(xo) * rc_L64);
          } else {
            po0 = divc(v, d_ms);                                 // accel.py:116-123
            po1 = divc(xo, d_L);
          }
          const T dv = valid ? T(v) - target_v : T(0);
          red[slot] = dv * dv;
        }
        pend_obs = true;
      }
    }
    // ---- rewards + done of the block: lane j finishes step j ------------------------------------------------
    const T racc = transposed_sum<SEG, PERIOD>(red, lane);
    T racc2 = T(0);
    if (HEAD == 1) racc2 = transposed_sum<SEG, PERIOD>(red2, lane);
    const unsigned crash_any = crash_bits, bad_any = bad_bits;       // already per-replica (seg_or above)
    crash_bits = 0u;
    bad_bits = 0u;
    const int j = i;
    if (rvalid && j < nsteps) {
      const bool my_crash = (crash_any >> j) & 1u, my_bad = (bad_any >> j) & 1u;
      const int t_j = tcount - (nsteps - 1 - j);
      T reward;
      if (HEAD == 1) {                                               // wave_attenuation.py:113-139
        if (!use_act) {
          reward = T(0);
        } else {
          const T mean_v = racc / T(N);
          const T mean_a = racc2 / T(num_rl);
          reward = T(4.0) * mean_v / T(20);
          if (mean_a > T(0)) reward = reward + T(4) * (T(0) - mean_a);
          reward = my_bad ? T(0) : reward;
        }
      } else {                                                       // rewards.py:6-59
        const T cost = tsqrt(racc);
        reward = tmax(max_cost - cost, T(0)) / (max_cost + T(1.1920928955078125e-07));
        reward = my_bad ? T(0) : reward;
      }
      const size_t o = size_t(base + j) * s.R + rr;
      prew = float(reward);
      pdone = done_flag(t_j >= s.step_limit, my_crash);
      prew_at = o;
      pend_rew = true;
    }
  };
  int base0 = 0;
  for (; base0 + PERIOD <= num_steps; base0 += PERIOD) run_block(base0, std::true_type{});
  if (base0 < num_steps) run_block(base0, std::false_type{});
  flush_obs();
  flush_rew();
  if (valid) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    if (s.track_aux && num_steps > 0) {
      s.prev_vel[idx] = prev_v;
      s.accel[idx] = last_acc;
    }
    if (ii == 0) {
      s.time[rr] = tcount;
      if (any_noise) s.noise_ctr[rr] = nctr;
    }
  }
}

}  // namespace fs
