// flowsim_pair.h -- k_rollout_pair: the rollout kernel of the headline configuration, TWO vehicles per lane.
//
// Same configuration class as k_rollout_idm (every slot a plain IDMController, speed mode "aggressive", Euler,
// sims_per_step 1, AccelEnv head, observation every step) for an EVEN number of vehicles.  What changes is the
// mapping, chosen from the issue costs measured on gfx950 (scripts/ubench, profiles/r02_valu_issue_ubench.txt):
// a wave that is alone on its SIMD issues ONE instruction per ~4.4 cycles whatever the instruction is, and at
// BASELINE's 4096 replicas there is about one wave per SIMD -- so the lever is work per instruction:
//   * lane k of a ROW-lane row holds vehicles 2k (A) and 2k+1 (B) of one replica; the mul / add / fma chains of
//     the two run as ONE packed instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on a register pair);
//   * the leader of A is B of the same lane (no cross-lane read); the leader of B is A of the next lane: one
//     row shift (DPP row_shl:1) plus the wrap lane's broadcast of lane 0 (DPP row_newbcast) per quantity;
//   * the first level of the reward tree (oracle/rewards.py tree_sum pairs (2k, 2k+1)) is an in-lane add;
//   * a lane stores its two speeds and its two positions as one dwordx2 each, through a buffer descriptor whose
//     scalar offset selects the step (no per-step address arithmetic in VALU or SALU).
// Arithmetic per vehicle is the operation sequence of k_rollout_idm / oracle/refsim.py (float: bit-twin of the
// float32 oracle), with one exception that is value-preserving: the speed observation v / max_speed is evaluated
// through float64 (cvt, mul by RN(1/c), two fma, cvt) -- correctly rounded for EVERY float v including the
// denormal range (exhaustively checked for the divisors of tests/test_oracle_c.py::test_div_via_f64), which
// replaces the ten-instruction IEEE sequence.
//
// SM: slots with speed-mode clamps (the reference's default "right_of_way"): sumo_acc_pair + three branch-free clamps.
// NOISE (float32): IDMController(noise = sigma) slots: acc + sigma * g with the generic kernel's draws.  Both keep the
// hand-written step in the hot instantiation (part A: flowsim_pair_step_a.inc / _a_sm.inc).
//
// MIXED (state type double): positions and speeds are kept and integrated in float64, the controller (the IDM
// acceleration) runs in float32 on their rounded images -- "fp32 physics on f64 accumulators".  This is the
// precision that meets BASELINE's 1e-4 trajectory bar at fp32 cost: float32 state drifts ~5e-3 m from the
// reference's float64 arithmetic over 1500 steps of the (string-unstable) sugiyama ring, the mixed form 4e-5 m
// (oracle/csim refsim_ring_idm_mixed is its bit-twin; tests/test_pair_gpu.py).
#pragma once
#include "flowsim_kernels.h"

namespace fs {

typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2v __attribute__((__vector_size__(8)));

enum : int { DPP_ROW_NEWBCAST0 = 0x150, DPP_ROW_NEWBCAST8 = 0x158 };

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float a) { return f2{a, a}; }

// packed float32 arithmetic of the C++ step (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32)
__device__ __forceinline__ f2 pk_mul(f2 a, f2 b) { return a * b; }
__device__ __forceinline__ f2 pk_add(f2 a, f2 b) { return a + b; }
__device__ __forceinline__ f2 pk_sub(f2 a, f2 b) { return a - b; }
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return fma2(a, b, c); }
__device__ __forceinline__ f2 pk_fnma(f2 a, f2 b, f2 c) { return fma2(-a, b, c); }
// x >= 0 ? x : y for y >= 0 (and x never -0): negative floats compare above every non-negative one as unsigned
// integers, so this is one v_min_u32 on the bit patterns when additionally x <= y whenever x >= 0
__device__ __forceinline__ float nonneg_else(float x, float y) {
  const unsigned a = __builtin_bit_cast(unsigned, x), b = __builtin_bit_cast(unsigned, y);
  return __builtin_bit_cast(float, min(a, b));             // (a < b ? a : b compiles to v_cmp + s_nop + v_cndmask)
}

// value of slot A of the NEXT lane of the row (lane 0's for the last occupied lane and its idle clones)
template <int ROW>
__device__ __forceinline__ float next_a(float v, bool last, int lane) {
  if (ROW == 16) {
    const float t = dpp<DPP_ROW_SHL1>(v), w = dpp<DPP_ROW_NEWBCAST0>(v);
    return last ? w : t;
  } else if (ROW == 8) {
    const float t = dpp<DPP_ROW_SHL1>(v), w0 = dpp<DPP_ROW_NEWBCAST0>(v), w8 = dpp<DPP_ROW_NEWBCAST8>(v);
    return last ? ((lane & 8) ? w8 : w0) : t;
  } else if (ROW == 32) {
    const float t = dpp<DPP_WAVE_SHL1>(v), w0 = read_lane(v, 0), w1 = read_lane(v, 32);
    return last ? ((lane & 32) ? w1 : w0) : t;
  } else {
    const float t = dpp<DPP_WAVE_SHL1>(v), w = read_lane(v, 0);
    return last ? w : t;
  }
}
template <int ROW>
__device__ __forceinline__ double next_a(double v, bool last, int lane) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const float lo = next_a<ROW>(__builtin_bit_cast(float, unsigned(b)), last, lane);
  const float hi = next_a<ROW>(__builtin_bit_cast(float, unsigned(b >> 32)), last, lane);
  return __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, hi) << 32) |
                                        __builtin_bit_cast(unsigned, lo));
}

template <bool FASTDIV>
__device__ __forceinline__ f2 div_const2(f2 x, f2 c, f2 rc) {
  if (FASTDIV) {
    const f2 q0 = pk_mul(x, rc);
    const f2 r = pk_fnma(q0, c, x);
    return pk_fma(r, rc, q0);
  }
  return f2{x.x / c.x, x.y / c.y};
}
// div_core (flowsim_kernels.h) on a pair: two v_rcp_f32, the refinement packed
__device__ __forceinline__ f2 div_core2(f2 n, f2 d, f2 one) {
  const f2 y0 = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f2 e = pk_fnma(d, y0, one);
  const f2 y = pk_fma(e, y0, y0);
  const f2 q0 = pk_mul(n, y);
  const f2 r0 = pk_fnma(d, q0, n);
  const f2 q1 = pk_fma(r0, y, q0);
  const f2 r1 = pk_fnma(d, q1, n);
  return pk_fma(r1, y, q1);
}

// IDMController.get_accel for the two vehicles of a lane (ctrl_idm's operation order; car_following_models.py:464-482)
template <bool DELTA4, bool FASTDIV>
__device__ __forceinline__ f2 idm_pair(f2 v, f2 vl, f2 h, const f2* p, f2 two_sqrt_ab, f2 rc_ab, f2 rc_v0, f2 one) {
  f2 hh;
  hh.x = tabs(h.x) < 1e-3f ? 1e-3f : h.x;
  hh.y = tabs(h.y) < 1e-3f ? 1e-3f : h.y;
  const f2 num = pk_mul(v, pk_sub(v, vl));
  const f2 dq = div_const2<FASTDIV>(num, two_sqrt_ab, rc_ab);
  const f2 ratio = div_const2<FASTDIV>(v, p[0], rc_v0);
  const f2 dyn = pk_add(pk_mul(v, p[1]), dq);
  f2 s_star;
  s_star.x = tmax(dyn.x, 0.0f);
  s_star.y = tmax(dyn.y, 0.0f);
  s_star = pk_add(p[5], s_star);
  const f2 q = FASTDIV ? div_core2(s_star, hh, one) : f2{s_star.x / hh.x, s_star.y / hh.y};
  f2 pw;
  if (DELTA4) {
    const f2 r2 = pk_mul(ratio, ratio);
    pw = pk_mul(r2, r2);
  } else {
    pw = f2{pow_delta(ratio.x, p[4].x), pow_delta(ratio.y, p[4].y)};
  }
  return pk_mul(p[2], pk_sub(pk_sub(one, pw), pk_mul(q, q)));
}

// Speed-mode clamps (S7/S8: SumoCarFollowingParams.speed_mode bits 0-2, the reference's default "right_of_way" = 25
// sets bit 0) for the two vehicles of a lane.  sumo_acc_pair = the acceleration inside sumo_idm_speed
// (flowsim_kernels.h; oracle/controllers.py sumo_idm_speed), same operation order; v * (v - vl) is the product the
// IDM controller forms too.  Divisions as in idm_pair: by the launch constants through div_const2 (host-verified per
// constant), ss / gap through div_core2 (gap >= 1e-3, ss >= minGap >= 1e-3: host-checked).
struct SumoPair {
  f2 tau, min_gap, maxa, smax, rc_smax, ts, rc_ts;
  f2 floor0;     // bit 0 set: 0 (v_sumo = max(0, v + acc dt)), clear: 3e38 (the cap never binds)
  f2 adt, ddt;   // bit 1 / bit 2 set: max_accel dt / max_decel dt, clear: 3e38
};
template <bool FASTDIV>
__device__ __forceinline__ f2 sumo_acc_pair(f2 v, f2 vl, f2 h, const SumoPair& c, f2 one) {
  f2 gap;
  gap.x = tmax(h.x, 1e-3f);
  gap.y = tmax(h.y, 1e-3f);
  const f2 num = pk_mul(v, pk_sub(v, vl));
  const f2 dq = div_const2<FASTDIV>(num, c.ts, c.rc_ts);
  const f2 dyn = pk_add(pk_mul(v, c.tau), dq);
  f2 ss;
  ss.x = tmax(0.0f, dyn.x);
  ss.y = tmax(0.0f, dyn.y);
  ss = pk_add(c.min_gap, ss);
  const f2 q = FASTDIV ? div_core2(ss, gap, one) : f2{ss.x / gap.x, ss.y / gap.y};
  const f2 r = div_const2<FASTDIV>(v, c.smax, c.rc_smax);
  const f2 r2 = pk_mul(r, r);
  return pk_mul(c.maxa, pk_sub(pk_sub(one, pk_mul(r2, r2)), pk_mul(q, q)));
}

// ---------------------------------------------------------------------------------------------------------
// One step of the float32 twin written out instruction by instruction (ROW = 16, delta = 4, exact reciprocal
// divisions, no v < -100 check): the hot instantiation.  Why not leave it to hipcc: (1) its hazard recogniser
// puts an `s_nop 0` between EVERY packed instruction and a consumer that follows directly (its "dst_sel
// forwarding" rule reads VOP3P's op_sel_hi bit of src0 as a destination select; packed f32 writes whole
// registers) -- 26 per step in the C++ form of this loop, and a wave that is alone on its SIMD pays a full
// issue slot (~4.4 cycles, scripts/ubench) for each; wrapping the packed operations one per asm statement makes
// it worse (every asm result is treated as such a hazard); (2) the leader differences fold their row shift into
// the subtraction (v_sub_f32_dpp) and the wrap-around selects are integer minima, which it does not find.
// The sequence is ctrl_idm / k_rollout_idm's operation order per vehicle; hazards kept by construction:
//   v_cmp -> v_cndmask on that SGPR pair: >= 2 instructions between;  v_rcp_f32 -> first use: >= 4;
//   VALU write -> DPP read of the register (v', x'): >= 8.
// Register map (pinned so that the halves of a pair can be named): V v[112:113], X v[114:115], H v[116:117],
// DVL (= v - v_leader) v[118:119], HH v[120:121], Y v[122:123], A v[124:125], B v[126:127], OV v[128:129],
// C v[130:131], D v[132:133], E v[134:135], F v[136:137].
struct PairConsts {
  f2 p0, p1, p2, p5, rc_v0, tsab, rc_ab, one, dt2, ramp2, L2, rc_L2, len_lead, gap2, vmask, tvm;
  float c1e3;
  double ms64, rc_ms64;
  unsigned long long last_mask;
};
// part A: controller, integration, position observation (stored by the caller between the two parts: the two
// stores of a step reach the CU's memory pipeline half a step apart)
#define FS_PAIR_A_NAME pair_step_asm_a
#define FS_PAIR_A_NZ 0
#define FS_PAIR_A_NZ_PARAM
#include "flowsim_pair_step_a.inc"
#undef FS_PAIR_A_NAME
#undef FS_PAIR_A_NZ
#undef FS_PAIR_A_NZ_PARAM
#define FS_PAIR_A_NAME pair_step_asm_a_nz
#define FS_PAIR_A_NZ 1
#define FS_PAIR_A_NZ_PARAM , f2 nz
#include "flowsim_pair_step_a.inc"
#undef FS_PAIR_A_NAME
#undef FS_PAIR_A_NZ
#undef FS_PAIR_A_NZ_PARAM
// part A with the speed-mode clamps (SM: sumo_acc_pair's operations interleaved with the controller's, same order per
// quantity; the clamps as max / min against constants that are 3e38 for a slot whose bit is clear).  Extra registers:
// G (gap) v[138:139], YS v[140:141], S v[142:143], Q v[144:145], R v[146:147]; E v[134:135] as a second residual.
#define FS_PAIR_A_NAME pair_step_asm_a_sm
#define FS_PAIR_A_NZ 0
#define FS_PAIR_A_NZ_PARAM
#include "flowsim_pair_step_a_sm.inc"
#undef FS_PAIR_A_NAME
#undef FS_PAIR_A_NZ
#undef FS_PAIR_A_NZ_PARAM
#define FS_PAIR_A_NAME pair_step_asm_a_sm_nz
#define FS_PAIR_A_NZ 1
#define FS_PAIR_A_NZ_PARAM , f2 nz
#include "flowsim_pair_step_a_sm.inc"
#undef FS_PAIR_A_NAME
#undef FS_PAIR_A_NZ
#undef FS_PAIR_A_NZ_PARAM
// part B: new neighbour snapshot, collision bit, reward term, speed observation
__device__ __forceinline__ void pair_step_asm_b(f2& V, f2& X, f2& H, f2& DVL, f2& OV, unsigned& crash_bits, float& sq,
                                                const PairConsts& c) {
  asm volatile(
      // ---- new snapshot (S10): v - v_leader and the gaps, leader of B = A of the next lane (lane 0 for the last)
      "v_sub_f32_e32 v118, v112, v113\n"
      "v_sub_f32_e32 v132, v115, v114\n"
      "v_subrev_f32_dpp v119, v112, v113 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_subrev_f32_dpp v134, v112, v113 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_sub_f32_dpp v133, v114, v115 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_sub_f32_dpp v135, v114, v115 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_e64 v119, v119, v134, %[last]\n"
      "v_cndmask_b32_e64 v133, v133, v135, %[last]\n"
      "v_pk_add_f32 v[126:127], v[132:133], %[L2]\n"                                        // d + L
      "v_min_u32_e32 v132, v132, v126\n"
      "v_min_u32_e32 v133, v133, v127\n"
      "v_pk_add_f32 v[116:117], v[132:133], %[lenlead] neg_lo:[0,1] neg_hi:[0,1]\n"         // H = new headways
      // ---- collision (S12): sign of min(h - crash_gap)
      "v_pk_add_f32 v[126:127], v[116:117], %[gap2] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_min_f32_e32 v126, v126, v127\n"
      "v_alignbit_b32 %[cb], %[cb], v126, 31\n"
      // ---- reward term (v - target)^2 of both vehicles, added
      "v_pk_fma_f32 v[124:125], v[112:113], %[vmask], %[tvm]\n"
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_add_f32_e32 %[sq], v124, v125\n"
      // ---- observation v'/max_speed through float64 (div_via_f64)
      "v_cvt_f64_f32_e32 v[130:131], v112\n"
      "v_cvt_f64_f32_e32 v[132:133], v113\n"
      "v_mul_f64 v[134:135], v[130:131], %[rcms]\n"
      "v_mul_f64 v[136:137], v[132:133], %[rcms]\n"
      "v_fma_f64 v[130:131], -v[134:135], %[ms], v[130:131]\n"
      "v_fma_f64 v[132:133], -v[136:137], %[ms], v[132:133]\n"
      "v_fma_f64 v[134:135], v[130:131], %[rcms], v[134:135]\n"
      "v_fma_f64 v[136:137], v[132:133], %[rcms], v[136:137]\n"
      "v_cvt_f32_f64_e32 v128, v[134:135]\n"
      "v_cvt_f32_f64_e32 v129, v[136:137]\n"
      : "+{v[112:113]}"(V), "+{v[114:115]}"(X), "+{v[116:117]}"(H), "+{v[118:119]}"(DVL), "={v[128:129]}"(OV),
        [cb] "+v"(crash_bits), [sq] "=&v"(sq)
      : [L2] "v"(c.L2), [lenlead] "v"(c.len_lead), [gap2] "v"(c.gap2), [vmask] "v"(c.vmask), [tvm] "v"(c.tvm),
        [ms] "v"(c.ms64), [rcms] "v"(c.rc_ms64), [last] "s"(c.last_mask)
      : "v124", "v125", "v126", "v127", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137");
}

// The FS_MIXED step (state float64, controller float32), same construction.  Register map: V32 (float images of
// the speeds) v[112:113], XA v[114:115], XB v[116:117] (positions, float64), VA v[118:119], VB v[138:139] (speeds,
// float64), H v[140:141], DVL v[142:143], HH v[120:121], Y v[122:123], A v[124:125], B v[126:127], OV v[128:129],
// C v[130:131], D v[132:133], E v[134:135], F v[136:137], OXr v[144:145].
struct MixedConsts {
  f2 p0, p1, p2, p5, rc_v0, tsab, rc_ab, one, gap2, vmask, tvm;
  float c1e3;
  double dt, ramp, L, rc_L, rc_ms, len_b, len_next, zero;
  unsigned long long last_mask;
};
__device__ __forceinline__ void mixed_step_asm_a(f2& V32, double& XA, double& XB, double& VA, double& VB, f2& H, f2& DVL,
                                                 f2& OV, const MixedConsts& c) {
  unsigned long long sa, sb;
  asm volatile(
      // ---- IDMController.get_accel in float32 on the rounded speeds / headways (as pair_step_asm)
      "v_cmp_nlt_f32_e64 %[sa], |v140|, %[c1e3]\n"
      "v_cmp_nlt_f32_e64 %[sb], |v141|, %[c1e3]\n"
      "v_pk_mul_f32 v[124:125], v[112:113], v[142:143]\n"
      "v_pk_mul_f32 v[126:127], v[124:125], %[rcab]\n"
      "v_pk_fma_f32 v[130:131], v[126:127], %[tsab], v[124:125] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[126:127], v[130:131], %[rcab], v[126:127]\n"
      "v_cndmask_b32_e64 v120, %[c1e3], v140, %[sa]\n"
      "v_cndmask_b32_e64 v121, %[c1e3], v141, %[sb]\n"
      "v_pk_mul_f32 v[124:125], v[112:113], %[rcv0]\n"
      "v_rcp_f32_e32 v122, v120\n"
      "v_rcp_f32_e32 v123, v121\n"
      "v_pk_fma_f32 v[130:131], v[124:125], %[p0], v[112:113] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[124:125], v[130:131], %[rcv0], v[124:125]\n"
      "v_pk_mul_f32 v[130:131], v[112:113], %[p1]\n"
      "v_pk_add_f32 v[130:131], v[130:131], v[126:127]\n"
      "v_pk_fma_f32 v[126:127], v[120:121], v[122:123], %[one] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_max_f32_e32 v130, 0, v130\n"
      "v_max_f32_e32 v131, 0, v131\n"
      "v_pk_fma_f32 v[122:123], v[126:127], v[122:123], v[122:123]\n"
      "v_pk_add_f32 v[130:131], %[p5], v[130:131]\n"
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_pk_mul_f32 v[126:127], v[130:131], v[122:123]\n"
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_pk_fma_f32 v[132:133], v[120:121], v[126:127], v[130:131] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_add_f32 v[124:125], %[one], v[124:125] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_fma_f32 v[126:127], v[132:133], v[122:123], v[126:127]\n"
      "v_pk_fma_f32 v[132:133], v[120:121], v[126:127], v[130:131] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[126:127], v[132:133], v[122:123], v[126:127]\n"
      "v_pk_mul_f32 v[126:127], v[126:127], v[126:127]\n"
      "v_pk_add_f32 v[124:125], v[124:125], v[126:127] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_mul_f32 v[124:125], %[p2], v[124:125]\n"                                        // A = acc (float32)
      // ---- apply_acceleration + integration (S4-S9) in float64
      "v_cvt_f64_f32_e32 v[130:131], v124\n"
      "v_cvt_f64_f32_e32 v[132:133], v125\n"
      "v_mul_f64 v[130:131], v[130:131], %[dt]\n"
      "v_mul_f64 v[132:133], v[132:133], %[dt]\n"
      "v_add_f64 v[130:131], v[118:119], v[130:131]\n"
      "v_add_f64 v[132:133], v[138:139], v[132:133]\n"
      "v_max_f64 v[130:131], v[130:131], %[zero]\n"                                        // next_vel
      "v_max_f64 v[132:133], v[132:133], %[zero]\n"
      "v_add_f64 v[130:131], v[130:131], -v[118:119]\n"
      "v_add_f64 v[132:133], v[132:133], -v[138:139]\n"
      "v_mul_f64 v[130:131], v[130:131], %[ramp]\n"
      "v_mul_f64 v[132:133], v[132:133], %[ramp]\n"
      "v_add_f64 v[118:119], v[118:119], v[130:131]\n"                                      // VA = v'
      "v_add_f64 v[138:139], v[138:139], v[132:133]\n"                                      // VB
      "v_mul_f64 v[130:131], v[118:119], %[dt]\n"
      "v_mul_f64 v[132:133], v[138:139], %[dt]\n"
      "v_add_f64 v[130:131], v[114:115], v[130:131]\n"                                      // x_new
      "v_add_f64 v[132:133], v[116:117], v[132:133]\n"
      "v_add_f64 v[134:135], v[130:131], -%[L]\n"                                           // x_new - L
      "v_add_f64 v[136:137], v[132:133], -%[L]\n"
      "v_cvt_f32_f64_e32 v112, v[118:119]\n"                                                // V32 = float images
      "v_cvt_f32_f64_e32 v113, v[138:139]\n"
      "v_cmp_gt_i32_e64 %[sa], 0, v135\n"                                                   // x_new - L < 0 (sign)
      "v_cmp_gt_i32_e64 %[sb], 0, v137\n"
      "v_mul_f64 v[120:121], v[118:119], %[rcms]\n"                                         // observation v' / max_speed
      "v_mul_f64 v[122:123], v[138:139], %[rcms]\n"
      "v_cndmask_b32_e64 v114, v134, v130, %[sa]\n"                                         // XA = x'
      "v_cndmask_b32_e64 v115, v135, v131, %[sa]\n"
      "v_cndmask_b32_e64 v116, v136, v132, %[sb]\n"                                         // XB
      "v_cndmask_b32_e64 v117, v137, v133, %[sb]\n"
      "v_cvt_f32_f64_e32 v128, v[120:121]\n"
      "v_cvt_f32_f64_e32 v129, v[122:123]\n"
      : "+{v[112:113]}"(V32), "+{v[114:115]}"(XA), "+{v[116:117]}"(XB), "+{v[118:119]}"(VA), "+{v[138:139]}"(VB),
        "+{v[140:141]}"(H), "+{v[142:143]}"(DVL), "={v[128:129]}"(OV), [sa] "=&s"(sa), [sb] "=&s"(sb)
      : [c1e3] "v"(c.c1e3), [rcab] "v"(c.rc_ab), [tsab] "v"(c.tsab), [rcv0] "v"(c.rc_v0), [p0] "v"(c.p0),
        [p1] "v"(c.p1), [p2] "v"(c.p2), [p5] "v"(c.p5), [one] "v"(c.one), [dt] "v"(c.dt), [ramp] "v"(c.ramp),
        [L] "v"(c.L), [rcms] "v"(c.rc_ms), [zero] "v"(c.zero)
      : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v130", "v131", "v132", "v133", "v134",
        "v135", "v136", "v137");
}
// mixed part A with the speed-mode clamps: SUMO's acceleration in float32 next to the controller's (temporaries G
// v[146:147], YS v[148:149], S v[150:151], Q v[152:153], R v[154:155], second residual v[134:135]), the clamps in
// float64 (refsim_ring_idm_mixed's order: min(vc, max(v + a_s dt, floor0)), min(.., v + adt), max(.., v - ddt)).
struct MixedSumoConsts { double fl0a, fl0b, adta, adtb, ddta, ddtb; };
__device__ __forceinline__ void mixed_step_asm_a_sm(f2& V32, double& XA, double& XB, double& VA, double& VB, f2& H,
                                                    f2& DVL, f2& OV, const MixedConsts& c, const SumoPair& m,
                                                    const MixedSumoConsts& d) {
  unsigned long long sa, sb;
  asm volatile(
      "v_cmp_nlt_f32_e64 %[sa], |v140|, %[c1e3]\n"
      "v_cmp_nlt_f32_e64 %[sb], |v141|, %[c1e3]\n"
      "v_pk_mul_f32 v[124:125], v[112:113], v[142:143]\n"
      "v_pk_mul_f32 v[126:127], v[124:125], %[rcab]\n"
      "v_pk_mul_f32 v[150:151], v[124:125], %[rcts]\n"
      "v_pk_fma_f32 v[130:131], v[126:127], %[tsab], v[124:125] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[152:153], v[150:151], %[ts], v[124:125] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[126:127], v[130:131], %[rcab], v[126:127]\n"
      "v_pk_fma_f32 v[150:151], v[152:153], %[rcts], v[150:151]\n"                          // S = num / ts
      "v_cndmask_b32_e64 v120, %[c1e3], v140, %[sa]\n"
      "v_cndmask_b32_e64 v121, %[c1e3], v141, %[sb]\n"
      "v_max_f32_e32 v146, v140, %[c1e3]\n"                                                 // G = max(h, 1e-3)
      "v_max_f32_e32 v147, v141, %[c1e3]\n"
      "v_pk_mul_f32 v[124:125], v[112:113], %[rcv0]\n"
      "v_rcp_f32_e32 v122, v120\n"
      "v_rcp_f32_e32 v123, v121\n"
      "v_rcp_f32_e32 v148, v146\n"
      "v_rcp_f32_e32 v149, v147\n"
      "v_pk_fma_f32 v[130:131], v[124:125], %[p0], v[112:113] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_mul_f32 v[154:155], v[112:113], %[rcsm]\n"
      "v_pk_fma_f32 v[124:125], v[130:131], %[rcv0], v[124:125]\n"
      "v_pk_fma_f32 v[152:153], v[154:155], %[smax], v[112:113] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_mul_f32 v[130:131], v[112:113], %[p1]\n"
      "v_pk_fma_f32 v[154:155], v[152:153], %[rcsm], v[154:155]\n"                          // R = v / maxSpeed
      "v_pk_add_f32 v[130:131], v[130:131], v[126:127]\n"
      "v_pk_mul_f32 v[152:153], v[112:113], %[tau]\n"
      "v_pk_fma_f32 v[126:127], v[120:121], v[122:123], %[one] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_add_f32 v[150:151], v[152:153], v[150:151]\n"                                   // dyn_s
      "v_pk_fma_f32 v[152:153], v[146:147], v[148:149], %[one] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_max_f32_e32 v130, 0, v130\n"
      "v_max_f32_e32 v131, 0, v131\n"
      "v_max_f32_e32 v150, 0, v150\n"
      "v_max_f32_e32 v151, 0, v151\n"
      "v_pk_fma_f32 v[122:123], v[126:127], v[122:123], v[122:123]\n"
      "v_pk_fma_f32 v[148:149], v[152:153], v[148:149], v[148:149]\n"
      "v_pk_add_f32 v[130:131], %[p5], v[130:131]\n"
      "v_pk_add_f32 v[150:151], %[mgap], v[150:151]\n"                                      // ss
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_pk_mul_f32 v[154:155], v[154:155], v[154:155]\n"
      "v_pk_mul_f32 v[126:127], v[130:131], v[122:123]\n"
      "v_pk_mul_f32 v[152:153], v[150:151], v[148:149]\n"                                   // q0_s
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_pk_mul_f32 v[154:155], v[154:155], v[154:155]\n"
      "v_pk_fma_f32 v[132:133], v[120:121], v[126:127], v[130:131] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[134:135], v[146:147], v[152:153], v[150:151] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_add_f32 v[124:125], %[one], v[124:125] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_add_f32 v[154:155], %[one], v[154:155] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_fma_f32 v[126:127], v[132:133], v[122:123], v[126:127]\n"
      "v_pk_fma_f32 v[152:153], v[134:135], v[148:149], v[152:153]\n"
      "v_pk_fma_f32 v[132:133], v[120:121], v[126:127], v[130:131] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[134:135], v[146:147], v[152:153], v[150:151] neg_lo:[1,0,0] neg_hi:[1,0,0]\n"
      "v_pk_fma_f32 v[126:127], v[132:133], v[122:123], v[126:127]\n"
      "v_pk_fma_f32 v[152:153], v[134:135], v[148:149], v[152:153]\n"                       // Q = ss / gap
      "v_pk_mul_f32 v[126:127], v[126:127], v[126:127]\n"
      "v_pk_mul_f32 v[152:153], v[152:153], v[152:153]\n"
      "v_pk_add_f32 v[124:125], v[124:125], v[126:127] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_add_f32 v[154:155], v[154:155], v[152:153] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_pk_mul_f32 v[124:125], %[p2], v[124:125]\n"                                        // A = acc (float32)
      "v_pk_mul_f32 v[154:155], %[maxa], v[154:155]\n"                                      // R = SUMO's acceleration (float32)
      // ---- apply_acceleration + clamps + integration in float64
      "v_cvt_f64_f32_e32 v[130:131], v124\n"
      "v_cvt_f64_f32_e32 v[132:133], v125\n"
      "v_cvt_f64_f32_e32 v[134:135], v154\n"
      "v_cvt_f64_f32_e32 v[136:137], v155\n"
      "v_mul_f64 v[130:131], v[130:131], %[dt]\n"
      "v_mul_f64 v[132:133], v[132:133], %[dt]\n"
      "v_mul_f64 v[134:135], v[134:135], %[dt]\n"
      "v_mul_f64 v[136:137], v[136:137], %[dt]\n"
      "v_add_f64 v[130:131], v[118:119], v[130:131]\n"
      "v_add_f64 v[132:133], v[138:139], v[132:133]\n"
      "v_add_f64 v[134:135], v[118:119], v[134:135]\n"                                      // v + a_s dt
      "v_add_f64 v[136:137], v[138:139], v[136:137]\n"
      "v_max_f64 v[130:131], v[130:131], %[zero]\n"                                        // next_vel
      "v_max_f64 v[132:133], v[132:133], %[zero]\n"
      "v_max_f64 v[134:135], v[134:135], %[fl0a]\n"                                        // v_sumo (3e38 when bit 0 is clear)
      "v_max_f64 v[136:137], v[136:137], %[fl0b]\n"
      "v_add_f64 v[130:131], v[130:131], -v[118:119]\n"
      "v_add_f64 v[132:133], v[132:133], -v[138:139]\n"
      "v_mul_f64 v[130:131], v[130:131], %[ramp]\n"
      "v_mul_f64 v[132:133], v[132:133], %[ramp]\n"
      "v_add_f64 v[130:131], v[118:119], v[130:131]\n"                                      // vc
      "v_add_f64 v[132:133], v[138:139], v[132:133]\n"
      "v_min_f64 v[130:131], v[130:131], v[134:135]\n"
      "v_min_f64 v[132:133], v[132:133], v[136:137]\n"
      "v_add_f64 v[134:135], v[118:119], %[adta]\n"
      "v_add_f64 v[136:137], v[138:139], %[adtb]\n"
      "v_min_f64 v[130:131], v[130:131], v[134:135]\n"
      "v_min_f64 v[132:133], v[132:133], v[136:137]\n"
      "v_add_f64 v[134:135], v[118:119], -%[ddta]\n"
      "v_add_f64 v[136:137], v[138:139], -%[ddtb]\n"
      "v_max_f64 v[118:119], v[130:131], v[134:135]\n"                                      // VA = v'
      "v_max_f64 v[138:139], v[132:133], v[136:137]\n"                                      // VB
      "v_mul_f64 v[130:131], v[118:119], %[dt]\n"
      "v_mul_f64 v[132:133], v[138:139], %[dt]\n"
      "v_add_f64 v[130:131], v[114:115], v[130:131]\n"                                      // x_new
      "v_add_f64 v[132:133], v[116:117], v[132:133]\n"
      "v_add_f64 v[134:135], v[130:131], -%[L]\n"                                           // x_new - L
      "v_add_f64 v[136:137], v[132:133], -%[L]\n"
      "v_cvt_f32_f64_e32 v112, v[118:119]\n"                                                // V32 = float images
      "v_cvt_f32_f64_e32 v113, v[138:139]\n"
      "v_cmp_gt_i32_e64 %[sa], 0, v135\n"                                                   // x_new - L < 0 (sign)
      "v_cmp_gt_i32_e64 %[sb], 0, v137\n"
      "v_mul_f64 v[120:121], v[118:119], %[rcms]\n"                                         // observation v' / max_speed
      "v_mul_f64 v[122:123], v[138:139], %[rcms]\n"
      "v_cndmask_b32_e64 v114, v134, v130, %[sa]\n"                                         // XA = x'
      "v_cndmask_b32_e64 v115, v135, v131, %[sa]\n"
      "v_cndmask_b32_e64 v116, v136, v132, %[sb]\n"                                         // XB
      "v_cndmask_b32_e64 v117, v137, v133, %[sb]\n"
      "v_cvt_f32_f64_e32 v128, v[120:121]\n"
      "v_cvt_f32_f64_e32 v129, v[122:123]\n"
      : "+{v[112:113]}"(V32), "+{v[114:115]}"(XA), "+{v[116:117]}"(XB), "+{v[118:119]}"(VA), "+{v[138:139]}"(VB),
        "+{v[140:141]}"(H), "+{v[142:143]}"(DVL), "={v[128:129]}"(OV), [sa] "=&s"(sa), [sb] "=&s"(sb)
      : [c1e3] "v"(c.c1e3), [rcab] "v"(c.rc_ab), [tsab] "v"(c.tsab), [rcv0] "v"(c.rc_v0), [p0] "v"(c.p0),
        [p1] "v"(c.p1), [p2] "v"(c.p2), [p5] "v"(c.p5), [one] "v"(c.one), [dt] "v"(c.dt), [ramp] "v"(c.ramp),
        [L] "v"(c.L), [rcms] "v"(c.rc_ms), [zero] "v"(c.zero), [rcts] "v"(m.rc_ts), [ts] "v"(m.ts),
        [rcsm] "v"(m.rc_smax), [smax] "v"(m.smax), [tau] "v"(m.tau), [mgap] "v"(m.min_gap), [maxa] "v"(m.maxa),
        [fl0a] "v"(d.fl0a), [fl0b] "v"(d.fl0b), [adta] "v"(d.adta), [adtb] "v"(d.adtb), [ddta] "v"(d.ddta),
        [ddtb] "v"(d.ddtb)
      : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v130", "v131", "v132", "v133", "v134",
        "v135", "v136", "v137", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155");
}
__device__ __forceinline__ void mixed_step_asm_b(f2& V32, double& XA, double& XB, f2& H, f2& DVL, f2& OX,
                                                 unsigned& crash_bits, float& sq, const MixedConsts& c) {
  unsigned long long sa, sb;
  asm volatile(
      // ---- new snapshot: v - v_leader (float32 images), gaps in float64
      "v_sub_f32_e32 v142, v112, v113\n"
      "v_subrev_f32_dpp v143, v112, v113 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_subrev_f32_dpp v124, v112, v113 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_mov_b32_dpp v130, v114 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"        // position of the next lane's A
      "v_mov_b32_dpp v131, v115 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
      "v_mov_b32_dpp v132, v114 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_mov_b32_dpp v133, v115 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
      "v_cndmask_b32_e64 v143, v143, v124, %[last]\n"
      "v_cndmask_b32_e64 v130, v130, v132, %[last]\n"
      "v_cndmask_b32_e64 v131, v131, v133, %[last]\n"
      "v_mul_f64 v[120:121], v[114:115], %[rcL]\n"                                          // observation x' / L
      "v_mul_f64 v[122:123], v[116:117], %[rcL]\n"
      "v_add_f64 v[134:135], v[116:117], -v[114:115]\n"                                     // dA = xB - xA
      "v_add_f64 v[136:137], v[130:131], -v[116:117]\n"                                     // dB = x_next - xB
      "v_cvt_f32_f64_e32 v144, v[120:121]\n"
      "v_cvt_f32_f64_e32 v145, v[122:123]\n"
      "v_add_f64 v[130:131], v[134:135], %[L]\n"
      "v_add_f64 v[132:133], v[136:137], %[L]\n"
      "v_cmp_gt_i32_e64 %[sa], 0, v135\n"                                                   // d < 0 (d is never -0)
      "v_cmp_gt_i32_e64 %[sb], 0, v137\n"
      "v_pk_fma_f32 v[124:125], v[112:113], %[vmask], %[tvm]\n"                             // reward term
      "v_pk_mul_f32 v[124:125], v[124:125], v[124:125]\n"
      "v_cndmask_b32_e64 v134, v134, v130, %[sa]\n"
      "v_cndmask_b32_e64 v135, v135, v131, %[sa]\n"
      "v_cndmask_b32_e64 v136, v136, v132, %[sb]\n"
      "v_cndmask_b32_e64 v137, v137, v133, %[sb]\n"
      "v_add_f32_e32 %[sq], v124, v125\n"
      "v_add_f64 v[134:135], v[134:135], -%[lenb]\n"
      "v_add_f64 v[136:137], v[136:137], -%[lennext]\n"
      "v_cvt_f32_f64_e32 v140, v[134:135]\n"                                                // H = new headways (float32)
      "v_cvt_f32_f64_e32 v141, v[136:137]\n"
      // ---- collision (S12)
      "v_pk_add_f32 v[126:127], v[140:141], %[gap2] neg_lo:[0,1] neg_hi:[0,1]\n"
      "v_min_f32_e32 v126, v126, v127\n"
      "v_alignbit_b32 %[cb], %[cb], v126, 31\n"
      : "+{v[112:113]}"(V32), "+{v[114:115]}"(XA), "+{v[116:117]}"(XB), "+{v[140:141]}"(H), "+{v[142:143]}"(DVL),
        "={v[144:145]}"(OX), [cb] "+v"(crash_bits), [sq] "=&v"(sq), [sa] "=&s"(sa), [sb] "=&s"(sb)
      : [L] "v"(c.L), [rcL] "v"(c.rc_L), [gap2] "v"(c.gap2), [vmask] "v"(c.vmask), [tvm] "v"(c.tvm),
        [lenb] "v"(c.len_b), [lennext] "v"(c.len_next), [last] "s"(c.last_mask)
      : "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v130", "v131", "v132", "v133", "v134",
        "v135", "v136", "v137");
}

// T = float: the float32 bit-twin.  T = double: MIXED (see the header).  ROW = lanes per replica (N <= 2 ROW).
// SM: some slot carries a speed-mode clamp (bits 0-2): the C++ step with sumo_acc_pair.
// NOISE (float32 only): IDMController(noise = sigma) slots -- acc + sigma * g with the generic kernel's draws (gauss4 keyed
// by seed, global replica, vehicle, draw counter / 4; base_controller.py:109-110), the C++ step.
template <typename T, int ROW, bool DELTA4, bool FASTDIV, bool BADCHK, bool SM = false, bool NOISE = false>
__global__ __launch_bounds__(256) void k_rollout_pair(DevView<T> s, int num_steps, float* __restrict__ obs,
                                                      float* __restrict__ rew, uint8_t* __restrict__ done) {
  constexpr bool MIXED = sizeof(T) == 8;
  static_assert(!(NOISE && MIXED), "FS_MIXED has no noise form (its C twin cannot reproduce the hardware's log / cos)");
  // the hand-written steps: pair_step_asm_a / _a_sm + _b (float32, without / with the speed-mode clamps), mixed_step_asm
  constexpr bool ASM = ROW == 16 && DELTA4 && FASTDIV && !BADCHK;
  constexpr int RPW = 64 / ROW;
  constexpr int PERIOD = ROW < 16 ? ROW : 16;       // steps whose reward tail is finished together
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int row = lane / ROW;
  const int k = lane % ROW;
  const int r = wave * RPW + row;
  const int N = s.N;
  const int LP = N >> 1;                            // occupied lanes of a row (N is even: host-checked)
  const bool rvalid = r < s.R;
  const bool valid = rvalid && k < LP;
  // idle lanes (k >= LP, or a replica index past R) are exact clones of lane LP-1 / replica R-1: same state, same
  // leader, same parameters -- they compute and store the SAME values to the SAME addresses, never masked off
  const int rr = rvalid ? r : s.R - 1;
  const int kk = k < LP ? k : LP - 1;
  const bool last = (kk == LP - 1);                 // B's leader is slot 0 (lane 0 of the row)
  const int iA = 2 * kk, iB = iA + 1;
  const size_t idx = size_t(rr) * N + iA;

  f2 p[6];
#pragma unroll
  for (int q = 0; q < 6; ++q) p[q] = f2{float(s.p[q * N + iA]), float(s.p[q * N + iB])};
  const T lenB = s.length[iB];
  const T len_nextA = next_a<ROW>(T(s.length[iA]), last, lane);
  const T base_len = s.ring_len[rr];
  const T L = base_len + T(4) * s.jlen;
  int tcount = s.time[rr];
  // acceleration noise: the two vehicles' blocks of four draws, rotated so that [0] is the next step's (k_rollout_loop's
  // scheme); a launch that starts inside a block evaluates it and rotates up to there
  uint32_t nctr = NOISE ? s.noise_ctr[rr] : 0u;
  const f2 sigma = NOISE ? f2{float(s.noise[iA]), float(s.noise[iB])} : f2{0.0f, 0.0f};
  const bool noisyA = NOISE && sigma.x > 0.0f, noisyB = NOISE && sigma.y > 0.0f;
  float gA[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gB[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  if constexpr (NOISE) {
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
  const f2 gap2 = splat(float(s.crash_gap)), m100 = splat(-100.0f);
  // reward terms of idle lanes are zero: (v - target)^2 is formed as fma(v, m, -target m) with m = 1 (valid: the
  // product v * 1 is exact, so the fma rounds v - target once, as the subtraction does) or m = 0
  const f2 vmask = splat(valid ? 1.0f : 0.0f), tvm = splat(valid ? -float(s.target_velocity) : 0.0f);
  const double ms64 = double(s.max_speed), rc_ms64 = 1.0 / ms64;
  const double L64 = double(L), rc_L64 = 1.0 / L64, dt64 = double(s.dt), ramp64 = double(s.ramp);

  // state: float pairs (twin) or two doubles + their float images (mixed)
  f2 x, v;                       // float32 images (twin: THE state)
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
  const f2 len_lead = {float(lenB), float(len_nextA)};

  // headways of the current snapshot (S10)
  auto headway = [&]() -> f2 {
    if (MIXED) {
      const double xn = next_a<ROW>(xdA, last, lane);
      double dA = xdB - xdA, dB = xn - xdB;
      dA = dA < 0.0 ? dA + L64 : dA;
      dB = dB < 0.0 ? dB + L64 : dB;
      return f2{float(dA - double(lenB)), float(dB - double(len_nextA))};
    } else {
      const f2 xl = {x.y, next_a<ROW>(x.x, last, lane)};
      f2 d = pk_sub(xl, x);
      const f2 dw = pk_add(d, L2);                 // d < 0 ? d + L : d  (d in (-L, L), never -0; d + L > d)
      d.x = nonneg_else(d.x, dw.x);
      d.y = nonneg_else(d.y, dw.y);
      return pk_sub(d, len_lead);
    }
  };
  f2 h = headway();
  f2 vl = {v.y, next_a<ROW>(v.x, last, lane)};
  f2 dvl = v - vl;                                          // ASM path carries v - v_leader instead of v_leader
  PairConsts pc;
  pc.p0 = p[0]; pc.p1 = p[1]; pc.p2 = p[2]; pc.p5 = p[5]; pc.rc_v0 = rc_v0; pc.tsab = two_sqrt_ab; pc.rc_ab = rc_ab;
  pc.one = one; pc.dt2 = dt2; pc.ramp2 = ramp2; pc.L2 = L2; pc.rc_L2 = rc_L2; pc.len_lead = len_lead;
  pc.gap2 = gap2; pc.vmask = vmask; pc.tvm = tvm; pc.c1e3 = 1e-3f; pc.ms64 = ms64; pc.rc_ms64 = rc_ms64;
  pc.last_mask = __ballot(last);
  MixedConsts mc;
  mc.p0 = p[0]; mc.p1 = p[1]; mc.p2 = p[2]; mc.p5 = p[5]; mc.rc_v0 = rc_v0; mc.tsab = two_sqrt_ab; mc.rc_ab = rc_ab;
  mc.one = one; mc.gap2 = gap2; mc.vmask = vmask; mc.tvm = tvm; mc.c1e3 = 1e-3f; mc.dt = dt64; mc.ramp = ramp64;
  mc.L = L64; mc.rc_L = rc_L64; mc.rc_ms = rc_ms64; mc.len_b = double(lenB); mc.len_next = double(len_nextA);
  mc.zero = 0.0; mc.last_mask = pc.last_mask;
  SumoPair sc{};
  double floor0A = 0, floor0B = 0, adtA = 0, adtB = 0, ddtA = 0, ddtB = 0;      // MIXED: the clamp constants in float64
  if constexpr (SM) {
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
  const MixedSumoConsts msc = {floor0A, floor0B, adtA, adtB, ddtA, ddtB};

  // observation stores: buffer descriptor over the PERIOD-step block being written + per-lane byte offsets; the
  // scalar offset of an unrolled step is a launch constant (slot * bytes per step)
  const unsigned rowb = 2u * unsigned(N) * 4u;                              // bytes of one replica's observation
  const unsigned off_v = unsigned(rr) * rowb + unsigned(iA) * 4u;           // BYTE offsets: host guarantees < 2^32
  const unsigned off_x = off_v + unsigned(N) * 4u;
  const size_t step_bytes = size_t(s.R) * rowb;
  const unsigned step_b32 = unsigned(step_bytes);                           // PERIOD * step_bytes < 2^32 (host)
  char* ob = reinterpret_cast<char*>(obs);
  unsigned crash_bits = 0u, bad_bits = 0u;

  // Transposed observation store of the hand-written step (ROW = 16): the row image [v_0..v_{N-1} | x_0..x_{N-1}]
  // of a replica (8 N bytes = N/2 pieces of 16 bytes) is assembled in LDS -- lane k writes its two speeds at byte
  // 8 k and its two positions at byte 4 N + 8 k -- and read back so that lane k holds piece k: ONE
  // buffer_store_dwordx4 per lane and step, 704 contiguous bytes per wave, instead of two dwordx2 with 88-byte
  // runs.  The piece read at the end of step t is stored during step t + 1 (its ds_read has long landed: the only
  // wait is an already satisfied lgkmcnt(0) in front of the store), the last one of a block after the block.
  // Measured in one gpurun call (scripts/ubench/pair_bench.hip, 4096 x 22 x 1500): mixed 0.415 -> 0.393 ms, f32
  // 0.366 -> 0.361 ms.  What the stores still cost against a launch that keeps its results (FS_DIAG_NOSTORE, 0.27
  // ms) is by the counters (scripts/dbg/pmc_variants.sh) mostly NOT wave time: ~50 cycles per step of LDS /
  // vector-memory issue for a wave alone on its SIMD, the rest a lower clock while 3 TB/s leave the chip (2.18 GHz
  // without stores, 1.94-2.02 GHz with).  Handing the stores to a second wave per SIMD through an LDS ring (tried,
  // removed) gained nothing: two ds_write_b64 per step cost the stepping wave what the stores did.
  typedef unsigned u4v __attribute__((__vector_size__(16)));
  constexpr bool XPOSE = ASM;
  __shared__ float row_img[XPOSE ? 4 : 1][XPOSE ? RPW * 64 : 1];          // [wave of the block][row * 2 N floats], N <= 32
  float* const my_img = row_img[XPOSE ? (threadIdx.x >> 6) & 3 : 0] + (XPOSE ? row * 2 * N : 0);
  const unsigned off_16 = unsigned(rr) * rowb + unsigned(kk) * 16u;         // byte offset of piece kk of my replica's row
  u4v piece = {0u, 0u, 0u, 0u};
  auto transpose_in = [&](f2 ov, f2 ox) {
    *reinterpret_cast<f2*>(my_img + 2 * kk) = ov;
    *reinterpret_cast<f2*>(my_img + N + 2 * kk) = ox;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");               // same wave: LDS operations execute in order
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    piece = *reinterpret_cast<const u4v*>(my_img + 4 * kk);
  };

  // sigma * g of this step for the lane's two vehicles (-0.0 for a slot without noise: x + (-0) keeps every bit of x)
  auto noise_term = [&]() -> f2 {
    f2 nz = {-0.0f, -0.0f};
    if constexpr (NOISE) {
      const bool fresh = (nctr & 3u) == 0u;                      // a new block of four draws starts with this step
      if (__ballot(fresh) != 0ull) {
        if (fresh) {
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iA), nctr >> 2, gA, s.noise_exact != 0);
          gauss4<float>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(iB), nctr >> 2, gB, s.noise_exact != 0);
        }
      }
      const float tA = sigma.x * gA[0], tB = sigma.y * gB[0];
      nz.x = noisyA ? tA : -0.0f;
      nz.y = noisyB ? tB : -0.0f;
      gA[0] = gA[1]; gA[1] = gA[2]; gA[2] = gA[3];
      gB[0] = gB[1]; gB[1] = gB[2]; gB[2] = gB[3];
      nctr += 1u;
    }
    return nz;
  };

  auto one_step = [&](int slot, __amdgpu_buffer_rsrc_t rs, float& sq_out) {
    if constexpr (ASM) {
      f2 ov, ox;
      const unsigned so = unsigned(slot) * step_b32;
      if constexpr (MIXED) {
        if constexpr (SM) mixed_step_asm_a_sm(v, xdA, xdB, vdA, vdB, h, dvl, ov, mc, sc, msc);
        else mixed_step_asm_a(v, xdA, xdB, vdA, vdB, h, dvl, ov, mc);
        if (slot > 0) __builtin_amdgcn_raw_buffer_store_b128(piece, rs, off_16, so - step_b32, 0);
        mixed_step_asm_b(v, xdA, xdB, h, dvl, ox, crash_bits, sq_out, mc);
        transpose_in(ov, ox);
      } else {
#if defined(FS_DIAG_NOSTORE)   // timing experiment (scripts/dbg/pmc_variants.sh): the results stay in registers
        pair_step_asm_a(v, x, h, dvl, ox, pc);
        pair_step_asm_b(v, x, h, dvl, ov, crash_bits, sq_out, pc);
        asm volatile("" :: "v"(ov), "v"(ox), "s"(so));
#elif defined(FS_DIAG_NOXPOSE)   // the two-dwordx2 form this replaced (timing comparisons)
        pair_step_asm_a(v, x, h, dvl, ox, pc);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, ox), rs, off_x, so, 0);
        pair_step_asm_b(v, x, h, dvl, ov, crash_bits, sq_out, pc);
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, ov), rs, off_v, so, 0);
#else
        if constexpr (NOISE) {
          const f2 nz = noise_term();
          if constexpr (SM) pair_step_asm_a_sm_nz(v, x, h, dvl, ox, pc, sc, nz);
          else pair_step_asm_a_nz(v, x, h, dvl, ox, pc, nz);
        } else {
          if constexpr (SM) pair_step_asm_a_sm(v, x, h, dvl, ox, pc, sc);
          else pair_step_asm_a(v, x, h, dvl, ox, pc);
        }
        if (slot > 0) __builtin_amdgcn_raw_buffer_store_b128(piece, rs, off_16, so - step_b32, 0);
        pair_step_asm_b(v, x, h, dvl, ov, crash_bits, sq_out, pc);
        transpose_in(ov, ox);
#endif
      }
      return;
    }
    // IDMController.get_accel on the snapshot
    f2 acc = idm_pair<DELTA4, FASTDIV>(v, vl, h, p, two_sqrt_ab, rc_ab, rc_v0, one);
    if constexpr (NOISE) acc = pk_add(acc, noise_term());
    f2 acc_s = {0.0f, 0.0f};
    if constexpr (SM) acc_s = sumo_acc_pair<FASTDIV>(v, vl, h, sc, one);
    f2 ox;
    if (MIXED) {
      // apply_acceleration + integration (S4-S9) in float64 on the float32 acceleration
      const double aA = double(acc.x), aB = double(acc.y);
      const double nA = tmax(vdA + aA * dt64, 0.0), nB = tmax(vdB + aB * dt64, 0.0);
      double cA = vdA + (nA - vdA) * ramp64, cB = vdB + (nB - vdB) * ramp64;
      if constexpr (SM) {        // S7/S8 in float64 on SUMO's float32 acceleration (floor0 = 3e38 switches the cap off)
        const double sA = tmax(vdA + double(acc_s.x) * dt64, floor0A), sB = tmax(vdB + double(acc_s.y) * dt64, floor0B);
        cA = tmin(cA, sA); cB = tmin(cB, sB);
        cA = tmin(cA, vdA + adtA); cB = tmin(cB, vdB + adtB);
        cA = tmax(cA, vdA - ddtA); cB = tmax(cB, vdB - ddtB);
      }
      vdA = cA;
      vdB = cB;
      const double xA = xdA + vdA * dt64, xB = xdB + vdB * dt64;
      xdA = xA >= L64 ? xA - L64 : xA;
      xdB = xB >= L64 ? xB - L64 : xB;
      v = f2{float(vdA), float(vdB)};
      ox = f2{float(xdA * rc_L64), float(xdB * rc_L64)};
    } else {
      f2 nv = pk_add(v, pk_mul(acc, dt2));
      nv.x = tmax(nv.x, 0.0f);
      nv.y = tmax(nv.y, 0.0f);
      f2 vc = pk_add(v, pk_mul(pk_sub(nv, v), ramp2));
      if constexpr (SM) {        // S7/S8: min(vc, v_sumo), min(vc, v + max_accel dt), max(vc, v - max_decel dt)
        const f2 vs = pk_add(v, pk_mul(acc_s, dt2));
        const f2 cap1 = pk_add(v, sc.adt), flo = pk_sub(v, sc.ddt);
        vc.x = tmin(vc.x, tmax(sc.floor0.x, vs.x));
        vc.y = tmin(vc.y, tmax(sc.floor0.y, vs.y));
        vc.x = tmax(tmin(vc.x, cap1.x), flo.x);
        vc.y = tmax(tmin(vc.y, cap1.y), flo.y);
      }
      v = vc;
      const f2 xn = pk_add(x, pk_mul(v, dt2));
      const f2 xw = pk_sub(xn, L2);                  // x_new >= L ? x_new - L : x_new  (0 <= x_new - L < x_new)
      x.x = nonneg_else(xw.x, xn.x);
      x.y = nonneg_else(xw.y, xn.y);
      ox = div_const2<FASTDIV>(x, L2, rc_L2);
    }
    // new neighbour snapshot (S10)
    h = headway();
    vl = f2{v.y, next_a<ROW>(v.x, last, lane)};
    // collision (S12): sign of h - crash_gap (h < gap  <=>  the difference is negative: no -0 from a non-zero
    // difference, denormals are kept); bit (PERIOD-1-slot) of crash_bits after the block
    // (the smaller of the two differences carries the sign: written as bits(a) | bits(b), hipcc 7.2 drops the
    // second operand -- "or of two bitcast vector elements, only bit 31 demanded" is miscompiled, found by
    // tests/test_pair_gpu.py::test_pair_f32_per_slot_parameters_lengths_and_crashes)
    const f2 hc = pk_sub(h, gap2);
    crash_bits = __builtin_amdgcn_alignbit(crash_bits, __builtin_bit_cast(unsigned, __builtin_fminf(hc.x, hc.y)), 31);
    if (BADCHK) {
      const f2 vb = pk_sub(v, m100);
      bad_bits = __builtin_amdgcn_alignbit(bad_bits, __builtin_bit_cast(unsigned, __builtin_fminf(vb.x, vb.y)), 31);
    }
    // AccelEnv.get_state (accel.py:116-123)
    f2 ov;
    if (MIXED) ov = f2{float(vdA * rc_ms64), float(vdB * rc_ms64)};
    else ov = f2{div_via_f64(v.x, ms64, rc_ms64), div_via_f64(v.y, ms64, rc_ms64)};
    const unsigned so = unsigned(slot) * step_b32;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, ov), rs, off_v, so, 0);
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2v, ox), rs, off_x, so, 0);
    // rewards.desired_velocity, first half: the lane's two terms, added (first level of the tree)
    const f2 dv = pk_fma(v, vmask, tvm);
    const f2 sq = pk_mul(dv, dv);
    sq_out = sq.x + sq.y;
  };

  // full blocks of PERIOD steps: straight-line code, rewards finished PERIOD at a time
  int base = 0;
  for (; base + PERIOD <= num_steps; base += PERIOD) {
    float sq[PERIOD];
    const size_t remain = size_t(num_steps - base) * step_bytes;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        ob, 0, remain > 0xFFFFFFFFull ? 0xFFFFFFFFu : unsigned(remain), 0x00020000);
#pragma unroll
    for (int slot = 0; slot < PERIOD; ++slot) one_step(slot, rs, sq[slot]);
#if !defined(FS_DIAG_NOSTORE) && !defined(FS_DIAG_NOXPOSE)
    if (XPOSE) __builtin_amdgcn_raw_buffer_store_b128(piece, rs, off_16, unsigned(PERIOD - 1) * step_b32, 0);
#endif
    ob += size_t(PERIOD) * step_bytes;
    tcount += PERIOD;
    const float racc = transposed_sum<ROW, PERIOD>(sq, lane);
    const unsigned crash_any = seg_or<ROW>(crash_bits);
    const unsigned bad_any = (BADCHK ? seg_or<ROW>(bad_bits) : 0u) | crash_any;    // v + 100 < 0  <=>  sign set
    crash_bits = 0u;
    bad_bits = 0u;
    if (rvalid && k < PERIOD) {                            // lane k finishes step k of the block
      const bool my_crash = (crash_any >> (PERIOD - 1 - k)) & 1u;
      const bool my_bad = (bad_any >> (PERIOD - 1 - k)) & 1u;
      const int t_k = tcount - (PERIOD - 1 - k);           // time counter after that step
      const float cost = tsqrt(racc);
      const float max_cost = float(s.max_cost);
      float reward = tmax(max_cost - cost, 0.0f) / (max_cost + 1.1920928955078125e-07f);   // rewards.py:59
      reward = my_bad ? 0.0f : reward;                                                     // rewards.py:46
      const size_t o = size_t(base + k) * s.R + rr;
      rew[o] = reward;
      done[o] = done_flag(t_k >= s.step_limit, my_crash);                                // envs/base.py:398-400
    }
  }
  // the remaining num_steps % PERIOD steps one at a time (same tree: seg_sum is transposed_sum's order)
#pragma unroll 1
  for (; base < num_steps; ++base) {
    float sq1;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ob, 0, unsigned(step_bytes), 0x00020000);
    one_step(0, rs, sq1);
#if !defined(FS_DIAG_NOSTORE) && !defined(FS_DIAG_NOXPOSE)
    if (XPOSE) __builtin_amdgcn_raw_buffer_store_b128(piece, rs, off_16, 0u, 0);
#endif
    ob += step_bytes;
    tcount += 1;
    const float racc = seg_sum<ROW>(sq1);
    const bool my_crash = (seg_or<ROW>(crash_bits) & 1u) != 0u;
    const bool my_bad = (BADCHK && (seg_or<ROW>(bad_bits) & 1u) != 0u) || my_crash;
    crash_bits = 0u;
    bad_bits = 0u;
    if (rvalid && k == 0) {
      const float cost = tsqrt(racc);
      const float max_cost = float(s.max_cost);
      float reward = tmax(max_cost - cost, 0.0f) / (max_cost + 1.1920928955078125e-07f);
      reward = my_bad ? 0.0f : reward;
      const size_t o = size_t(base) * s.R + rr;
      rew[o] = reward;
      done[o] = done_flag(tcount >= s.step_limit, my_crash);
    }
  }
  if (valid) {
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

// observation of the current state of a FS_MIXED handle (Env.reset): the mixed head's own form (reciprocal
// multiplication in float64, then the rounding to float32)
template <typename T>     // (T = float is never launched: it keeps launch_seg<float> well-formed)
__global__ void k_obs_mixed(DevView<T> s, float* __restrict__ obs) {
  if constexpr (std::is_same<T, double>::value) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= s.R * s.N) return;
    const int r = e / s.N, i = e % s.N;
    const double L = s.ring_len[r] + 4.0 * s.jlen;
    float* o = obs + size_t(r) * 2 * s.N;
    o[i] = float(s.vel[e] * (1.0 / double(s.max_speed)));
    o[s.N + i] = float(s.pos[e] * (1.0 / L));
  }
}

}  // namespace fs
