// flowsim_kernels.h -- device code of libflowsim: the fused batched step kernel
// for closed single-lane routes (RingNetwork), written for gfx950 (CDNA4).
//
// Mapping (docs/HISTORY.md "k_steps"): one 64-lane wavefront carries 64/SEG replicas,
// SEG = smallest power of two >= N (vehicles per replica); lane = seg*SEG + i
// holds vehicle i of its replica in registers for the whole launch.  The leader
// of slot i on a closed single-lane loop is slot i+1 (cyclic), so the
// leader/follower lookup is a fixed cross-lane rotation (ds_bpermute through the
// LDS crossbar, no LDS allocation) and every per-replica reduction (reward norm,
// crash/any flags) is an xor-butterfly inside the SEG-lane segment, built from DPP,
// v_readlane and the gfx950 v_permlane16/32_swap instructions (VALU only).
//
// Arithmetic contract: every floating-point expression below is evaluated in
// T with the SAME operation order as oracle/controllers.py / oracle/refsim.py
// (which restate the reference lines cited there); the library is built with
// -ffp-contract=off so that a*b+c is never fused.  float is the bit-twin of the
// float32 oracle; double restates the reference's Python-float arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "flowsim.h"

namespace fs {

enum : int {
  FLAG_NEED_FOLLOWER = 1,   // a BCM vehicle reads its follower (car_following_models.py:168-172)
  FLAG_NEED_MEAN = 2,       // NonLocalFollowerStopper reads the replica mean speed
  FLAG_HAS_NOISE = 4,       // some controller has noise > 0 (base_controller.py:109-110)
  FLAG_HAS_LAC = 8,         // per-vehicle controller state: LAC a (car_following_models.py:243), PISaturation v_cmd
  FLAG_NEED_SUMO = 16,      // some vehicle may be uncommanded or has speed_mode bit 0
  FLAG_HAS_FAILSAFE = 32,
  FLAG_ALL_IDM = 64,        // every slot is an IDM controller (fast path)
  FLAG_IDM_SET = 128,       // every slot is an IDMController, an RLController or a SimCarFollowingController
  FLAG_NO_FLOW_CTRL = 256,  // every slot is an RLController or a SimCarFollowingController: no Flow acceleration controller
  FLAG_DELTA4 = 512,        // every IDM slot has the exponent delta = 4
};

template <typename T>
struct DevView {
  // state, [R,N] unless noted
  T* pos;
  T* vel;
  T* prev_vel;
  T* accel;
  T* ctrl_state;
  int32_t* lane;        // [R,N]  (multi-lane only)
  int32_t* last_lc;     // [R,N]  (multi-lane only)
  const int32_t* init_lane;
  T* pis_hist;          // [R, n_pis, pis_H] PISaturation speed history (ring buffers)
  int32_t* pis_n;       // [R, n_pis] speeds appended so far
  const int32_t* pis_index;   // [N] index among the PISaturation slots, -1 otherwise
  int32_t* time;        // [R]
  T* sort_key;          // [R,N] AccelEnv.absolute_position of the last additional_command (sort_vehicles only)
  const int32_t* obs_perm;   // [N] place of the slot's vehicle in get_ids() (InitialConfig.shuffle), or NULL = identity
  uint32_t* noise_ctr;  // [R]
  const T* init_pos;
  const T* init_vel;
  const T* ring_len;    // [R]
  const T* init_ring_len;   // [R] the length a replica takes at its next reset (FS_FIELD_INIT_RING_LENGTH)
  // FS_F16S: the state of record between launches, IEEE halves: [0 .. RN) position hi, [RN .. 2RN) position lo
  // (x = hi + lo), [2RN .. 3RN) speed; NULL otherwise.  pos / vel (float32) are then staging for reset and host access.
  uint16_t* st16;
  // per-slot tables, [N]
  const int32_t* ctrl;
  const int32_t* failsafe;
  const int32_t* speed_mode;
  const int32_t* rl_index;
  const T* p;           // [FS_MAX_CTRL_PARAMS][N]
  const T* noise;
  const T* delay;
  const T* max_accel;
  const T* max_decel;
  const T* length;
  const T* sumo_tau;
  const T* sumo_min_gap;
  const T* sumo_max_speed;
  // scalars
  int R, N, num_rl, env, integrator, sims_per_step, junction_mode, clip_actions, evaluate, track_aux;
  int num_lanes, lane_change_mode, last_lc_quirk, n_pis, pis_H, sort_vehicles;
  int noise_exact;      // Box-Muller through bm_ln_exact / bm_cos_exact (fs_config.noise_exact)
  int step_limit;       // sims_per_step*(warmup+horizon), INT_MAX for horizon=inf
  int flags;
  uint32_t seed_lo, seed_hi;
  uint32_t rep0;        // global index of this handle's replica 0: noise streams are keyed by GLOBAL replica ids
  T dt, ramp, jlen, crash_gap, max_speed, target_velocity, max_cost, act_lo, act_hi, po_max_length;
  T lc_duration;
  // autonomous lane changing on multi-lane rings (ML7; the rule M11 of the lane-drop network)
  const int32_t* lc_auto;   // [N] 1: the slot's vehicle changes lane on its own
  int lc_enabled, lc_cooldown;
  T lc_min_gain;
  // non-ring closed loops (figure eight): edge table in route order + the crossing model (S-J)
  int nseg, junction_on;
  unsigned seg_internal;                 // bit k: segment k is a junction-internal edge
  T seg_start[FS_MAX_SEGMENTS], seg_flow_start[FS_MAX_SEGMENTS], seg_flow_slope[FS_MAX_SEGMENTS];
  T ja_in, ja_out, jb_in, jb_out, j_lookahead, j_time_gap, za_lo, za_hi, zb_lo, zb_hi;
};

// FS_F16S: a position as two halves (hi = RN16(x), lo = RN16(x - hi): 22 significant bits), a speed as one
__device__ __forceinline__ float half_bits_to_float(uint16_t b) { return float(__builtin_bit_cast(_Float16, b)); }
__device__ __forceinline__ uint16_t float_to_half_bits(float f) { return __builtin_bit_cast(uint16_t, _Float16(f)); }
template <typename T>
__device__ __forceinline__ void state16_load(const DevView<T>& s, size_t e, T& x, T& v) {
  const size_t RN = size_t(s.R) * s.N;
  x = T(half_bits_to_float(s.st16[e]) + half_bits_to_float(s.st16[RN + e]));
  v = T(half_bits_to_float(s.st16[2 * RN + e]));
}
template <typename T>
__device__ __forceinline__ void state16_store(const DevView<T>& s, size_t e, T x, T v) {
  const size_t RN = size_t(s.R) * s.N;
  const uint16_t hi = float_to_half_bits(float(x));
  s.st16[e] = hi;
  s.st16[RN + e] = float_to_half_bits(float(x) - half_bits_to_float(hi));
  s.st16[2 * RN + e] = float_to_half_bits(float(v));
}
// the float32 staging arrays <-> the half state (reset, fs_set_state / fs_get_state)
template <typename T>
__global__ void k_state16_pack(DevView<T> s, const uint8_t* __restrict__ mask) {
  const size_t n = size_t(s.R) * s.N;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x)
    if (mask == nullptr || mask[e / s.N] != 0) state16_store(s, e, s.pos[e], s.vel[e]);
}
template <typename T>
__global__ void k_state16_unpack(DevView<T> s) {
  const size_t n = size_t(s.R) * s.N;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x)
    state16_load(s, e, s.pos[e], s.vel[e]);
}

// done flag of a step (envs/base.py:398-400 `done = time_counter >= horizon limit or crash`): non-zero = done, and
// the two reasons stay readable -- bit 0: the horizon was reached, bit 1: a collision ended the episode (so a crash
// on the very last step of the horizon still reaches compute_reward(fail=True) in the scalar Env)
__device__ __forceinline__ uint8_t done_flag(bool horizon, bool crashed) {
  return uint8_t((horizon ? 1 : 0) | (crashed ? 2 : 0));
}

// ---------------------------------------------------------------------------
// small math helpers with a fixed operation order
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }   // np.maximum (no NaN)
template <typename T> __device__ __forceinline__ T tmin(T a, T b) { return a < b ? a : b; }   // np.minimum (no NaN)
// min / max as ONE instruction.  tmin / tmax (a < b ? a : b) compile to v_cmp + s_nop + v_cndmask -- hipcc may not
// assume the operands are numbers -- and a wave alone on its SIMD pays every one of those issue slots, a dozen times per
// step.  v_min_f32 / v_max_f32 return the same value for every pair of numbers; for a pair of zeros of opposite sign they
// may return the other zero, which no consumer in the rollout kernels that use them tells apart (sums with non-zero terms, products, compares
// -- the sign-mask predicates of flowsim_fig8.h take differences a - b, and 0 - 0 is +0 whatever the signs unless a is -0 and b
// is +0: every b of such a test is a launch constant or a position, never a -0).  k_rollout_pair's hand-written step
// uses the same instructions.
// (evaluate them BEFORE a select, `const float m = hmax(..); r = c ? m : r;` -- inside the arms of ?: they are conditional
// code, and hipcc builds an exec-mask region, s_and_saveexec + branch + s_or, around a one-cycle instruction)
__device__ __forceinline__ float hmax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float hmin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
// __builtin_sqrt* is correctly rounded under hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt;
// __fsqrt_rn is NOT (it maps to the native 1-ulp v_sqrt_f32 in this toolchain).
__device__ __forceinline__ float tsqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ double tsqrt(double x) { return __builtin_sqrt(x); }
__device__ __forceinline__ float tfloor(float x) { return floorf(x); }
__device__ __forceinline__ double tfloor(double x) { return floor(x); }
__device__ __forceinline__ float tabs(float x) { return fabsf(x); }
__device__ __forceinline__ double tabs(double x) { return fabs(x); }
__device__ __forceinline__ float tcos(float x) { return cosf(x); }
__device__ __forceinline__ double tcos(double x) { return cos(x); }
__device__ __forceinline__ float tlog(float x) { return logf(x); }
__device__ __forceinline__ double tlog(double x) { return log(x); }
__device__ __forceinline__ float tpow(float x, float y) { return powf(x, y); }
__device__ __forceinline__ double tpow(double x, double y) { return pow(x, y); }
__device__ __forceinline__ float tfmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ double tfmin(double a, double b) { return fmin(a, b); }

// oracle/controllers.py pow_delta
template <typename T>
__device__ __forceinline__ T pow_delta(T x, T d) {
  if (d == T(4)) { T x2 = x * x; return x2 * x2; }
  if (d == T(2)) return x * x;
  if (d == T(1)) return x;
  if (d == T(3)) return (x * x) * x;
  if (d == T(8)) { T x2 = x * x; T x4 = x2 * x2; return x4 * x4; }
  return tpow(x, d);
}

// ---------------------------------------------------------------------------
// cross-lane primitives: DPP / v_readlane / v_permlane*_swap (VALU, no LDS-crossbar latency)
// ---------------------------------------------------------------------------
// DPP controls (GFX9 encoding).
enum : int {
  DPP_QUAD_XOR1 = 0xB1,     // quad_perm:[1,0,3,2]
  DPP_QUAD_XOR2 = 0x4E,     // quad_perm:[2,3,0,1]
  DPP_WAVE_SHL1 = 0x130,    // lane i <- lane i+1
  DPP_WAVE_SHR1 = 0x138,    // lane i <- lane i-1
  DPP_ROW_MIRROR = 0x140,   // lane i <- lane 15-i   (within a row of 16)
  DPP_ROW_HALF_MIRROR = 0x141, // lane i <- lane 7-i (within a half row of 8)
  DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108,   // lane i <- lane i+n (row of 16)
  DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118    // lane i <- lane i-n
};

// bound_ctrl: a lane whose source does not exist reads 0 (only the last / first lane of the wave
// under wave_shl / wave_shr, which callers patch); with old = 0 the compiler folds the move into
// the consuming VALU instruction (v_add_f32_dpp), halving the cost of a reduction level.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
  return __builtin_bit_cast(float, dpp_i<CTRL>(__builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = dpp_i<CTRL>(int(b)), hi = dpp_i<CTRL>(int(b >> 32));
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
__device__ __forceinline__ float read_lane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = __builtin_amdgcn_readlane(int(b), lane), hi = __builtin_amdgcn_readlane(int(b >> 32), lane);
  return __builtin_bit_cast(double, (long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
}
// value that slot j (wave-uniform) of MY segment holds: v_readlane broadcasts through an SGPR, no LDS round trip
__device__ __forceinline__ int read_lane_i(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
template <int SEG>
__device__ __forceinline__ int seg_read_i(int v, int j, int seg) {
  int out = read_lane_i(v, j);
#pragma unroll
  for (int sg = 1; sg < 64 / SEG; ++sg) {
    const int f = read_lane_i(v, sg * SEG + j);
    out = (seg == sg) ? f : out;
  }
  return out;
}
template <int SEG, typename T>
__device__ __forceinline__ T seg_read(T v, int j, int seg) {
  T out = read_lane(v, j);
#pragma unroll
  for (int sg = 1; sg < 64 / SEG; ++sg) {
    const T f = read_lane(v, sg * SEG + j);
    out = (seg == sg) ? f : out;
  }
  return out;
}

// v + (value of the other 16-lane row of the pair): v_permlane16_swap_b32 (new on gfx950).
// The two operands are made opaque copies first: given the SAME SSA value twice, hipcc 7.2 folds
// the two results of the builtin into one register (it emitted v_add_f32 v,v,v after the swap).
__device__ __forceinline__ void swap_rows16(unsigned& a, unsigned& b) {
  asm volatile("" : "+v"(a), "+v"(b));
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ void swap_rows32(unsigned& a, unsigned& b) {
  asm volatile("" : "+v"(a), "+v"(b));
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
__device__ __forceinline__ float add_swap16(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  swap_rows16(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ float add_swap32(float v) {
  unsigned a = __builtin_bit_cast(unsigned, v), b = a;
  swap_rows32(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double add_swap16(double v) {
  unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  unsigned lo0 = unsigned(bits), hi0 = unsigned(bits >> 32), lo1 = lo0, hi1 = hi0;
  swap_rows16(lo0, lo1);
  swap_rows16(hi0, hi1);
  return __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0) +
         __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}
__device__ __forceinline__ double add_swap32(double v) {
  unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
  unsigned lo0 = unsigned(bits), hi0 = unsigned(bits >> 32), lo1 = lo0, hi1 = hi0;
  swap_rows32(lo0, lo1);
  swap_rows32(hi0, hi1);
  return __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0) +
         __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
}

// value held by the leader's lane (slot i+1, cyclic inside the SEG-lane segment): a one-lane
// wave shift plus a v_readlane fix-up of the wrap lane (slot N-1 reads slot 0 of its segment)
template <int SEG, typename T>
__device__ __forceinline__ T lead_read(T v, int seg, bool wrap) {
  T nxt = dpp<DPP_WAVE_SHL1>(v);
  T first = v;
#pragma unroll
  for (int sg = 0; sg < 64 / SEG; ++sg) {
    T f = read_lane(v, sg * SEG);
    first = (seg == sg) ? f : first;
  }
  return wrap ? first : nxt;
}
// value held by the follower's lane (slot i-1, cyclic): slot 0 reads slot N-1 of its segment
template <int SEG, typename T>
__device__ __forceinline__ T foll_read(T v, int seg, bool wrap, int N) {
  T prv = dpp<DPP_WAVE_SHR1>(v);
  T last = v;
#pragma unroll
  for (int sg = 0; sg < 64 / SEG; ++sg) {
    T f = read_lane(v, sg * SEG + N - 1);
    last = (seg == sg) ? f : last;
  }
  return wrap ? last : prv;
}

// sum over the SEG-lane segment in the oracle's tree order (oracle/rewards.py tree_sum):
// xor-1, xor-2 partners by quad_perm; once quads (then half rows, rows) are uniform the xor-4 /
// xor-8 / xor-16 / xor-32 partner sums are reached by half-mirror / mirror / permlane swaps.
template <int SEG, typename T>
__device__ __forceinline__ T seg_sum(T v) {
  v = v + dpp<DPP_QUAD_XOR1>(v);
  v = v + dpp<DPP_QUAD_XOR2>(v);
  v = v + dpp<DPP_ROW_HALF_MIRROR>(v);
  if (SEG >= 16) v = v + dpp<DPP_ROW_MIRROR>(v);
  if (SEG >= 32) v = add_swap16(v);
  if (SEG >= 64) v = add_swap32(v);
  return v;
}


// ---------------------------------------------------------------------------
// Transposed reduction: K = 2^k vectors (one value per lane each) are summed over the lanes of a segment
// TOGETHER -- after level b every lane keeps only the vectors whose index has bit b equal to its own lane bit b,
// so the register count halves per level and lane i ends up with the complete sum of vector i.  Each level
// adds the xor-2^b partner, i.e. exactly the tree of seg_sum / oracle tree_sum (a + b is commutative), in
// ~3 VALU per surviving vector instead of one full butterfly per vector.
//   lanes with bit b = 0 need lo_own + lo_partner(l + 2^b), lanes with bit b = 1 need hi_own + hi_partner(l - 2^b):
//   both sums are formed with a row shift (the out-of-row reads are never selected) and picked per lane;
//   the xor-16 / xor-32 levels use v_permlane{16,32}_swap(lo, hi), whose two results add up to exactly that.
// ---------------------------------------------------------------------------
template <int SHL, int SHR, typename T>
__device__ __forceinline__ T tr_combine(T lo, T hi, bool upper) {
  const T t = lo + dpp<SHL>(lo);
  const T u = hi + dpp<SHR>(hi);
  return upper ? u : t;
}
__device__ __forceinline__ float tr_combine16(float lo, float hi) {
  unsigned a = __builtin_bit_cast(unsigned, lo), b = __builtin_bit_cast(unsigned, hi);
  swap_rows16(a, b);
  return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
}
__device__ __forceinline__ double tr_combine16(double lo, double hi) {
  const unsigned long long bl = __builtin_bit_cast(unsigned long long, lo), bh = __builtin_bit_cast(unsigned long long, hi);
  unsigned l0 = unsigned(bl), l1 = unsigned(bl >> 32), h0 = unsigned(bh), h1 = unsigned(bh >> 32);
  swap_rows16(l0, h0);
  swap_rows16(l1, h1);
  return __builtin_bit_cast(double, ((unsigned long long)l1 << 32) | l0) +
         __builtin_bit_cast(double, ((unsigned long long)h1 << 32) | h0);
}
// K vectors in a[0..K-1] -> the sum of vector (lane % K) over the SEG-lane segment, in every lane
template <int SEG, int K, typename T>
__device__ __forceinline__ T transposed_sum(T* a, int lane) {
  static_assert(K <= 32 && K <= SEG, "at most min(SEG, 32) vectors");
  if (K >= 2) {
#pragma unroll
    for (int k = 0; k < K / 2; ++k) a[k] = tr_combine<DPP_ROW_SHL1, DPP_ROW_SHR1>(a[2 * k], a[2 * k + 1], (lane & 1) != 0);
  }
  if (K >= 4) {
#pragma unroll
    for (int k = 0; k < K / 4; ++k) a[k] = tr_combine<DPP_ROW_SHL2, DPP_ROW_SHR2>(a[2 * k], a[2 * k + 1], (lane & 2) != 0);
  }
  if (K >= 8) {
#pragma unroll
    for (int k = 0; k < K / 8; ++k) a[k] = tr_combine<DPP_ROW_SHL4, DPP_ROW_SHR4>(a[2 * k], a[2 * k + 1], (lane & 4) != 0);
  }
  if (K >= 16) {
#pragma unroll
    for (int k = 0; k < K / 16; ++k) a[k] = tr_combine<DPP_ROW_SHL8, DPP_ROW_SHR8>(a[2 * k], a[2 * k + 1], (lane & 8) != 0);
  }
  if (K >= 32) a[0] = tr_combine16(a[0], a[1]);
  T v = a[0];
  // lanes of the segment that hold the same vector index still have to be added up (SEG > K)
  // (the xor-4 / xor-8 partners are reached by row shifts picked per lane: the mirror forms of seg_sum need the
  // lanes of a quad / half row to hold the SAME value, which is only true for K = 1 here)
  if (K < 2 && SEG >= 2) v = v + dpp<DPP_QUAD_XOR1>(v);
  if (K < 4 && SEG >= 4) v = v + dpp<DPP_QUAD_XOR2>(v);
  if (K < 8 && SEG >= 8) {
    if (K == 1) v = v + dpp<DPP_ROW_HALF_MIRROR>(v);
    else v = tr_combine<DPP_ROW_SHL4, DPP_ROW_SHR4>(v, v, (lane & 4) != 0);
  }
  if (K < 16 && SEG >= 16) {
    if (K == 1) v = v + dpp<DPP_ROW_MIRROR>(v);
    else v = tr_combine<DPP_ROW_SHL8, DPP_ROW_SHR8>(v, v, (lane & 8) != 0);
  }
  if (K < 32 && SEG >= 32) v = add_swap16(v);
  if (SEG >= 64) v = add_swap32(v);
  return v;
}

// bitwise OR over the SEG-lane segment (same butterfly as seg_sum)
__device__ __forceinline__ unsigned or_swap16(unsigned v) {
  unsigned a = v, b = v;
  swap_rows16(a, b);
  return a | b;
}
__device__ __forceinline__ unsigned or_swap32(unsigned v) {
  unsigned a = v, b = v;
  swap_rows32(a, b);
  return a | b;
}
template <int SEG>
__device__ __forceinline__ unsigned seg_or(unsigned v) {
  v |= unsigned(dpp_i<DPP_QUAD_XOR1>(int(v)));
  v |= unsigned(dpp_i<DPP_QUAD_XOR2>(int(v)));
  v |= unsigned(dpp_i<DPP_ROW_HALF_MIRROR>(int(v)));
  if (SEG >= 16) v |= unsigned(dpp_i<DPP_ROW_MIRROR>(int(v)));
  if (SEG >= 32) v = or_swap16(v);
  if (SEG >= 64) v = or_swap32(v);
  return v;
}

template <int SEG>
__device__ __forceinline__ bool seg_any(bool pred, int seg) {
  unsigned long long b = __ballot(pred);
  if (SEG == 64) return b != 0ull;
  unsigned long long m = ((1ull << (SEG & 63)) - 1ull) << (seg * (SEG & 63));
  return (b & m) != 0ull;
}

template <int SEG>
__device__ __forceinline__ unsigned long long seg_ballot(bool pred, int seg) {
  unsigned long long b = __ballot(pred);
  if (SEG == 64) return b;
  return (b >> (seg * (SEG & 63))) & ((1ull << (SEG & 63)) - 1ull);
}

// minimum over the SEG-lane segment: the butterfly of seg_sum with min instead of +
__device__ __forceinline__ float min_swap16(float v) {
  unsigned a = __float_as_uint(v), b = a;
  swap_rows16(a, b);
  const float fa = __uint_as_float(a), fb = __uint_as_float(b);
  return fb < fa ? fb : fa;
}
__device__ __forceinline__ float min_swap32(float v) {
  unsigned a = __float_as_uint(v), b = a;
  swap_rows32(a, b);
  const float fa = __uint_as_float(a), fb = __uint_as_float(b);
  return fb < fa ? fb : fa;
}
__device__ __forceinline__ double min_swap16(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  unsigned lo0 = unsigned(u), lo1 = lo0, hi0 = unsigned(u >> 32), hi1 = hi0;
  swap_rows16(lo0, lo1);
  swap_rows16(hi0, hi1);
  const double a = __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0);
  const double b = __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
  return b < a ? b : a;
}
__device__ __forceinline__ double min_swap32(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  unsigned lo0 = unsigned(u), lo1 = lo0, hi0 = unsigned(u >> 32), hi1 = hi0;
  swap_rows32(lo0, lo1);
  swap_rows32(hi0, hi1);
  const double a = __builtin_bit_cast(double, ((unsigned long long)hi0 << 32) | lo0);
  const double b = __builtin_bit_cast(double, ((unsigned long long)hi1 << 32) | lo1);
  return b < a ? b : a;
}
template <int SEG, typename T>
__device__ __forceinline__ T seg_min(T v) {
  T w = dpp<DPP_QUAD_XOR1>(v);
  v = w < v ? w : v;
  w = dpp<DPP_QUAD_XOR2>(v);
  v = w < v ? w : v;
  w = dpp<DPP_ROW_HALF_MIRROR>(v);
  v = w < v ? w : v;
  if (SEG >= 16) {
    w = dpp<DPP_ROW_MIRROR>(v);
    v = w < v ? w : v;
  }
  if (SEG >= 32) v = min_swap16(v);
  if (SEG >= 64) v = min_swap32(v);
  return v;
}

template <int SEG, typename T>
__device__ __forceinline__ T seg_max(T v) { return -seg_min<SEG>(-v); }

// ---------------------------------------------------------------------------
// Philox-4x32-10 + Box-Muller: oracle/refsim.py philox4x32_10 / gaussian_noise
// ---------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3,
                                              uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 product each (v_mad_u64_u32 costs what ONE of v_mul_hi_u32 / v_mul_lo_u32 does: scripts/ubench)
    const uint64_t p0 = uint64_t(0xD2511F53u) * uint64_t(c0), p1 = uint64_t(0xCD9E8D57u) * uint64_t(c2);
    const uint32_t hi0 = uint32_t(p0 >> 32), lo0 = uint32_t(p0);
    const uint32_t hi1 = uint32_t(p1 >> 32), lo1 = uint32_t(p1);
    uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

// Four N(0,1) draws from ONE Philox call: the counter is (block = step / 4, vehicle, replica, 0); the words (c0, c1)
// and (c2, c3) feed two Box-Muller transforms that are each used twice, at the angle 2 pi u2 and a quarter turn back
// (i.e. the sine branch, written as a cosine so that ONE trigonometric function serves every draw): draw j = step % 4
// of the block is r_{j/2} * cos(2 pi (u2_{j/2} - (j & 1) / 4))  (oracle/refsim.py gaussian_noise, in lock-step).
// The generic kernels evaluate the draw of the current step; k_rollout_loop keeps the block's four for four steps.
// float: the hardware's log2 / cos (argument in revolutions) / sqrt -- noise is compared with the numpy oracle at
// libm tolerance anyway (DESIGN "Precision"); double: libm.
__device__ __forceinline__ float bm_radius(float u1) {
  return __builtin_amdgcn_sqrtf(-2.0f * (__builtin_amdgcn_logf(u1) * 0.6931471805599453f));
}
__device__ __forceinline__ double bm_radius(double u1) { return sqrt(-2.0 * log(u1)); }
__device__ __forceinline__ float bm_cos(float turns) { return __builtin_amdgcn_cosf(turns); }
__device__ __forceinline__ double bm_cos(double turns) { return cos(6.283185307179586 * turns); }
// EXACT forms (fs_config.noise_exact): ln and cos as fixed sequences of float32 operations -- every +, -, * is one IEEE
// rounding (the library is built with -ffp-contract=off), the one division and the square root are correctly rounded --
// so that oracle/refsim.py (exact_ln_f32 / exact_cos_turns_f32: the same sequences in numpy float32) reproduces a noisy
// float32 run bit for bit.  ln u = e ln 2 + 2 atanh((m - 1) / (m + 1)) for u = m 2^e, m in [sqrt(1/2), sqrt(2)), the odd
// series up to s^9 (|s| < 0.172: 3e-7 absolute); cos(2 pi t) from the quarter turn q = floor(4 t + 1/2) and the even / odd
// polynomials of the angle (4 t - q) pi / 2 in [-pi/4, pi/4] (9e-8).
__device__ __forceinline__ float bm_ln_exact(float u) {
  const unsigned bits = __builtin_bit_cast(unsigned, u);
  int e = int(bits >> 23) - 127;
  float m = __builtin_bit_cast(float, (bits & 0x7FFFFFu) | 0x3F800000u);
  const bool big = m > 1.4142135f;
  m = big ? m * 0.5f : m;
  e += big ? 1 : 0;
  const float t = m - 1.0f;
  const float s = t / (2.0f + t);
  const float z = s * s;
  float p = z * 0.11111111f + 0.14285715f;
  p = p * z + 0.2f;
  p = p * z + 0.33333334f;
  p = p * z + 1.0f;
  return float(e) * 0.6931472f + (2.0f * s) * p;
}
// cos(2 pi t) and, for a block's second draw of the same angle, cos(2 pi (t - 1/4)): the quarter turn of t - 1/4 is q - 1
// and its angle the SAME float (4 t, 4 t - 1 and the differences are exact), so ONE evaluation of the two polynomials
// serves both -- bit for bit what exact_cos_turns_f32 returns for t and for t - 0.25
__device__ __forceinline__ void bm_cos_exact2(float t, float& cos_t, float& cos_back) {
  const float a = t * 4.0f;
  const float q = __builtin_floorf(a + 0.5f);
  const float th = (a - q) * 1.5707964f;
  const float z = th * th;
  float c = z * 2.4801587e-05f + -1.3888889e-03f;
  c = c * z + 4.1666668e-02f;
  c = c * z + -0.5f;
  c = c * z + 1.0f;
  float sn = z * 2.7557319e-06f + -1.9841270e-04f;
  sn = sn * z + 8.3333338e-03f;
  sn = sn * z + -1.6666667e-01f;
  sn = sn * z + 1.0f;
  sn = sn * th;
  const int qi = int(q) & 3, qb = (qi + 3) & 3;
  const float even = (qi & 2) ? -c : c, odd = (qi & 2) ? sn : -sn;
  cos_t = (qi & 1) ? odd : even;
  const float even_b = (qb & 2) ? -c : c, odd_b = (qb & 2) ? sn : -sn;
  cos_back = (qb & 1) ? odd_b : even_b;
}
__device__ __forceinline__ float bm_cos_exact(float t) {
  float c0, c1;
  bm_cos_exact2(t, c0, c1);
  return c0;
}
__device__ __forceinline__ float bm_radius_x(float u1, bool exact) {
  return exact ? tsqrt(-2.0f * bm_ln_exact(u1)) : bm_radius(u1);
}
__device__ __forceinline__ double bm_radius_x(double u1, bool) { return bm_radius(u1); }     // (float64: libm either way)
__device__ __forceinline__ float bm_cos_x(float turns, bool exact) { return exact ? bm_cos_exact(turns) : bm_cos(turns); }
__device__ __forceinline__ double bm_cos_x(double turns, bool) { return bm_cos(turns); }
template <typename T>
__device__ __forceinline__ void gauss4(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t vehicle,
                                       uint32_t block, T* g, bool exact = false) {
  uint32_t c0 = block, c1 = vehicle, c2 = replica, c3 = 0u;
  philox4x32_10(c0, c1, c2, c3, seed_lo, seed_hi);
  const T k = T(1.0 / 16777216.0);                     // u1 in (0, 1], u2 in [0, 1): 24-bit integers, exact in float
  const T ra = bm_radius_x(T((c0 >> 8) + 1u) * k, exact), ua = T(c1 >> 8) * k;
  const T rb = bm_radius_x(T((c2 >> 8) + 1u) * k, exact), ub = T(c3 >> 8) * k;
  if constexpr (sizeof(T) == 4) {
    if (exact) {                                        // (wave-uniform)
      float ca, cab, cb, cbb;
      bm_cos_exact2(float(ua), ca, cab);
      bm_cos_exact2(float(ub), cb, cbb);
      g[0] = ra * ca; g[1] = ra * cab; g[2] = rb * cb; g[3] = rb * cbb;
      return;
    }
  }
  g[0] = ra * bm_cos_x(ua, exact);
  g[1] = ra * bm_cos_x(ua - T(0.25), exact);
  g[2] = rb * bm_cos_x(ub, exact);
  g[3] = rb * bm_cos_x(ub - T(0.25), exact);
}
template <typename T>
__device__ __forceinline__ T gauss(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t vehicle,
                                   uint32_t step, bool exact = false) {
  uint32_t c0 = step >> 2, c1 = vehicle, c2 = replica, c3 = 0u;
  philox4x32_10(c0, c1, c2, c3, seed_lo, seed_hi);
  const bool second = (step & 2u) != 0u;
  const uint32_t w1 = second ? c2 : c0, w2 = second ? c3 : c1;
  const T k = T(1.0 / 16777216.0);
  const T u2 = T(w2 >> 8) * k;
  return bm_radius_x(T((w1 >> 8) + 1u) * k, exact) * bm_cos_x((step & 1u) ? u2 - T(0.25) : u2, exact);
}

// The four draws of one Philox block, kept over the four steps they serve: draw(ctr) is gauss(.., ctr) bit for bit
// (gauss4 evaluates the same expressions), at a quarter of the Philox / log / sqrt work.  The block is re-evaluated by
// the WHOLE wave whenever some lane's counter has left the block it holds (replicas of one wave may be at different
// counters: a lane that is still inside its block recomputes the values it already has).
template <typename T>
struct NoiseBlock {
  T g[4];
  uint32_t block;
  bool loaded;
  __device__ __forceinline__ void init() { block = 0u; loaded = false; g[0] = g[1] = g[2] = g[3] = T(0); }
  __device__ __forceinline__ T draw(uint32_t seed_lo, uint32_t seed_hi, uint32_t replica, uint32_t vehicle, uint32_t ctr,
                                    bool exact = false) {
    if (__ballot(!loaded || block != (ctr >> 2)) != 0ull) {
      block = ctr >> 2;
      loaded = true;
      gauss4<T>(seed_lo, seed_hi, replica, vehicle, block, g, exact);
    }
    const uint32_t ph = ctr & 3u;
    const T lo = (ph & 1u) ? g[1] : g[0], hi = (ph & 1u) ? g[3] : g[2];
    return (ph & 2u) ? hi : lo;
  }
};

// ---------------------------------------------------------------------------
// controllers (one lane = one vehicle); see oracle/controllers.py for citations
// ---------------------------------------------------------------------------
template <typename T>
struct Slot {           // per-lane copy of the vehicle slot tables
  int ctrl, failsafe, speed_mode, rl_index, pis_index;
  T p[FS_MAX_CTRL_PARAMS];
  T noise, delay, max_accel, max_decel, length, sumo_tau, sumo_min_gap, sumo_max_speed;
};

template <typename T>
__device__ __forceinline__ T ctrl_idm(T v, T vl, T h, bool has, const T* p, T delta) {
  // p = {v0, T, a, b, delta, s0}; car_following_models.py:464-482
  T hh = tabs(h) < T(1e-3) ? T(1e-3) : h;
  T two_sqrt_ab = T(2) * tsqrt(p[2] * p[3]);
  T dyn = v * p[1] + v * (v - vl) / two_sqrt_ab;
  T s_star = has ? p[5] + tmax(T(0), dyn) : T(0);
  T q = s_star / hh;
  return p[2] * (T(1) - pow_delta(v / p[0], delta) - q * q);
}
template <typename T>
__device__ __forceinline__ T ctrl_idm(T v, T vl, T h, bool has, const T* p) { return ctrl_idm(v, vl, h, has, p, p[4]); }

template <typename T>
__device__ __forceinline__ T ctrl_cfm(T v, T vl, T h, bool has, T max_accel, const T* p) {
  // p = {k_d, k_v, k_c, d_des, v_des}; car_following_models.py:76-88
  T acc = p[0] * (h - p[3]) + p[1] * (vl - v) + p[2] * (p[4] - v);
  return has ? acc : max_accel;
}

template <typename T>
__device__ __forceinline__ T ctrl_bcm(T v, T vl, T h, bool has, T vf, T hf, T max_accel, const T* p) {
  // car_following_models.py:152-176
  T acc = p[0] * (h - hf) + p[1] * ((vl - v) - (v - vf)) + p[2] * (p[4] - v);
  return has ? acc : max_accel;
}

template <typename T>
__device__ __forceinline__ T ctrl_lac(T v, T vl, T h, T len, T a_prev, T dt, const T* p) {
  // p = {k_1, k_2, h, tau}; car_following_models.py:232-245
  T ex = h - len - p[2] * v;
  T ev = vl - v;
  T u = p[0] * ex + p[1] * ev;
  T a_dot = -(a_prev / p[3]) + (u / p[3]);
  return a_dot * dt + a_prev;
}

template <typename T>
__device__ __forceinline__ T ctrl_ovm(T v, T vl, T h, bool has, T max_accel, const T* p) {
  // p = {alpha, beta, h_st, h_go, v_max}; car_following_models.py:308-328
  T h_dot = vl - v;
  T mid = p[4] / T(2) * (T(1) - tcos(T(3.141592653589793) * (h - p[2]) / (p[3] - p[2])));
  T v_h = h <= p[2] ? T(0) : (h < p[3] ? mid : p[4]);
  T acc = p[0] * (v_h - v) + p[1] * h_dot;
  return has ? acc : max_accel;
}

template <typename T>
__device__ __forceinline__ T ctrl_linear_ovm(T v, T h, const T* p) {
  // p = {v_max, adaptation, h_st}; car_following_models.py:383-397
  const T alpha = T(1.689);
  T upper = p[2] + p[0] / alpha;
  T v_h = h < p[2] ? T(0) : (h <= upper ? alpha * (h - p[2]) : p[0]);
  return (v_h - v) / p[1];
}

template <typename T>
__device__ __forceinline__ T ctrl_gipps(T v, T vl, T h, T dt, const T* p) {
  // p = {v0, acc, b, b_l, s0, tau}; car_following_models.py:567-582
  T r = v / p[0];
  T v_acc = v + (T(2.5) * p[1] * p[5] * (T(1) - r) * tsqrt(T(0.025) + r));
  T tb = p[5] * p[2];
  T disc = (p[5] * p[5]) * (p[2] * p[2]) - (p[2] * ((T(2) * (h - p[4])) - (p[5] * v) - ((vl * vl) / p[3])));
  T v_safe = tb + tsqrt(disc);
  T v_next = tfmin(tfmin(v_acc, v_safe), p[0]);
  return (v_next - v) / dt;
}

template <typename T>
__device__ __forceinline__ T ctrl_follower_stopper(T v, T vl, T h, bool has, T dt, T v_des) {
  // velocity_controllers.py:75-116
  T dv_minus = tmin(vl - v, T(0));
  T dv2 = dv_minus * dv_minus;
  T dx_1 = T(4.5) + T(1.0 / (2 * 1.5)) * dv2;
  T dx_2 = T(5.25) + T(1.0 / (2 * 1.0)) * dv2;
  T dx_3 = T(6.0) + T(1.0 / (2 * 0.5)) * dv2;
  T vv = tmin(tmax(vl, T(0)), v_des);
  T c2 = vv * (h - dx_1) / (dx_2 - dx_1);
  T c3 = vv + (v_des - v) * (h - dx_2) / (dx_3 - dx_2);
  T v_cmd = h <= dx_1 ? T(0) : (h <= dx_2 ? c2 : (h <= dx_3 ? c3 : v_des));
  v_cmd = has ? v_cmd : v_des;
  return (v_cmd - v) / dt;
}

// PISaturation.get_accel (velocity_controllers.py:208-240); oracle/refsim.py pisaturation_step.
// hist: this vehicle's ring buffer of the last H speeds, n: appends so far, v_cmd: previous command.
template <typename T>
__device__ __forceinline__ T ctrl_pisaturation(T v, T vl, T h, T dt, T max_accel, T* hist, int32_t* n_ptr, int H,
                                               bool update, T& v_cmd) {
  const T dv = vl - v;
  const T dx_s = tmax(T(2) * dv, T(4));
  const int n = *n_ptr;
  const int pos = n % H;                           // append here (the oldest speed is overwritten once full)
  const int n_new = n + 1;
  const int cnt = n_new < H ? n_new : H;
  const int start = n_new > H ? n_new % H : 0;
  T total = T(0);
  for (int k = 0; k < cnt; ++k) {                  // oldest -> newest
    int idx = start + k;
    idx = idx >= H ? idx - H : idx;
    total = total + (idx == pos ? v : hist[idx]);
  }
  if (update) {                                    // only the vehicle's own lane, only when the step is committed
    hist[pos] = v;
    *n_ptr = n_new;
  }
  const T v_des = total / T(cnt);
  const T v_target = v_des + T(1) * tmin(tmax((h - T(7)) / (T(30) - T(7)), T(0)), T(1));
  const T alpha = tmin(tmax((h - dx_s) / T(2), T(0)), T(1));
  const T beta = T(1) - T(0.5) * alpha;
  const T new_cmd = beta * (alpha * v_target + (T(1) - alpha) * vl) + (T(1) - beta) * v_cmd;
  const T accel = tmin((new_cmd - v) / dt, max_accel);
  if (update) v_cmd = new_cmd;
  return accel;
}

template <typename T>
__device__ __forceinline__ T failsafe_instantaneous(T acc, T v, T h, bool has, T dt) {
  // base_controller.py:120-169 (num_vehicles > 1 checked by the caller)
  T next_vel = v + acc * dt;
  T thresh = dt * next_vel + v * T(1e-3) + T(0.5) * v * dt;
  bool stop = has && (next_vel > T(0)) && (h < thresh);
  return stop ? -v / dt : acc;
}

template <typename T>
__device__ __forceinline__ T failsafe_safe_velocity(T acc, T v, T vl, T h, T dt, T delay) {
  // base_controller.py:171-236
  T dv = vl - v;
  T v_safe = T(2) * h / dt + dv - v * (T(2) * delay);
  bool over = (v + acc * dt) > v_safe;
  T clipped = v_safe > T(0) ? (v_safe - v) / dt : -v / dt;
  return over ? clipped : acc;
}

template <typename T>
__device__ __forceinline__ T sumo_idm_speed(T v, T vl, T h, bool has, T dt, const Slot<T>& s) {
  // oracle/controllers.py sumo_idm_speed (SUMO-side, parity unpinned)
  T gap = tmax(h, T(1e-3));
  T two_sqrt = T(2) * tsqrt(s.max_accel * s.max_decel);
  T ss = s.sumo_min_gap + tmax(T(0), v * s.sumo_tau + v * (v - vl) / two_sqrt);
  T q = has ? ss / gap : T(0);
  T r = v / s.sumo_max_speed;
  T r2 = r * r;
  T acc = s.max_accel * (T(1) - r2 * r2 - q * q);
  return tmax(T(0), v + acc * dt);
}

// The closed-network segment table (<= FS_MAX_SEGMENTS rows) is copied to LDS once per launch (one wave per block)
// and read with plain, lane-indexed ds_read: a lookup inside the step loop makes no trip to the kernarg segment (a
// lane-varying index into the by-value DevView arrays compiles to global loads, whose latency one wave per SIMD
// cannot hide), and -- unlike the lane-held VGPR table + v_readlane / ds_bpermute form this replaced -- it is valid
// under ANY exec mask and holds no value the compiler could park in an AGPR and re-materialise under a partial
// mask in front of a v_readlane (the wide kernel's miscompile; tests/test_codegen.py).
template <typename T>
struct SegTab { const T* start; const T* flow_start; const T* flow_slope; };     // LDS rows, [FS_MAX_SEGMENTS + 1]

#define FS_SEGTAB_LDS(T, name)                                                                    \
  __shared__ T name##_st[FS_MAX_SEGMENTS + 1], name##_fs[FS_MAX_SEGMENTS + 1], name##_sl[FS_MAX_SEGMENTS + 1]

template <typename T>
__device__ __forceinline__ SegTab<T> load_segtab(const DevView<T>& s, int lane, T* st, T* fs0, T* sl) {
  if (lane <= FS_MAX_SEGMENTS) {
    const int q = lane < FS_MAX_SEGMENTS ? lane : 0;
    st[lane] = lane < s.nseg ? s.seg_start[q] : T(3.0e38);
    fs0[lane] = lane < s.nseg ? s.seg_flow_start[q] : T(0);
    sl[lane] = lane < s.nseg ? s.seg_flow_slope[q] : T(0);
  }
  __syncthreads();
  return SegTab<T>{st, fs0, sl};
}

// segment of loop coordinate x (the last one whose start is <= x): is it internal, and Flow's table
// coordinate of x (oracle/refsim.py _segment_lookup)
template <typename T>
__device__ __forceinline__ void segment_lookup(const DevView<T>& s, const SegTab<T>& tab, T x, bool& internal,
                                               T& flow_x) {
  int k = 0;
  for (int q = 1; q < s.nseg; ++q) k = (x >= tab.start[q]) ? q : k;
  const T st = tab.start[k], fs0 = tab.flow_start[k], sl = tab.flow_slope[k];
  internal = (s.seg_internal >> k) & 1u;
  flow_x = fs0 + sl * (x - st);
}

// the same lookup straight from the by-value tables, for code that runs once per launch or reset
template <typename T>
__device__ __forceinline__ void segment_lookup_args(const DevView<T>& s, T x, bool& internal, T& flow_x) {
  int k = 0;
  for (int q = 1; q < s.nseg; ++q) k = (x >= s.seg_start[q]) ? q : k;
  T st = s.seg_start[0], fs0 = s.seg_flow_start[0], sl = s.seg_flow_slope[0];
  for (int q = 1; q < s.nseg; ++q) {
    const bool hit = (q == k);
    st = hit ? s.seg_start[q] : st;
    fs0 = hit ? s.seg_flow_start[q] : fs0;
    sl = hit ? s.seg_flow_slope[q] : sl;
  }
  internal = (s.seg_internal >> k) & 1u;
  flow_x = fs0 + sl * (x - st);
}

// The segment a vehicle is on, FOLLOWED from step to step instead of searched: vehicles only move forward, so after a
// move the cached segment is still right unless x has passed the next start (advance) or wrapped around the loop
// (restart from segment 0).  A step then costs two compares and a ballot; the table rows of the current segment are
// re-gathered (ds_bpermute, whole wave) only when some lane changes segment.  One lookup per move serves both the
// observation of this step and the "on an edge" test of the next one (same x).  On C3 the two searches per step
// (an 11-iteration v_readlane loop each, ~135 cycles per iteration at one wave per SIMD) were 38 % of the step.
template <typename T>
struct SegCursor {
  int k;
  T st, fs0, sl, next;

  __device__ __forceinline__ void refresh(const SegTab<T>& tab, int nseg) {
    st = tab.start[k];
    fs0 = tab.flow_start[k];
    sl = tab.flow_slope[k];
    next = tab.start[k + 1];                           // row nseg holds 3.0e38
    (void)nseg;
  }
  __device__ __forceinline__ void init(const DevView<T>& s, const SegTab<T>& tab, T x) {
    k = 0;
    for (int q = 1; q < s.nseg; ++q) k = (x >= tab.start[q]) ? q : k;
    refresh(tab, s.nseg);
  }
  __device__ __forceinline__ void follow(const DevView<T>& s, const SegTab<T>& tab, T x) {
    if (__ballot(x < st) != 0ull) {
      k = (x < st) ? 0 : k;
      refresh(tab, s.nseg);
    }
    while (__ballot(x >= next) != 0ull) {
      k += (x >= next) ? 1 : 0;
      refresh(tab, s.nseg);
    }
  }
  __device__ __forceinline__ bool internal(const DevView<T>& s) const { return (s.seg_internal >> k) & 1u; }
  __device__ __forceinline__ T flow_x(T x) const { return fs0 + sl * (x - st); }
};

// ---------------------------------------------------------------------------
// FS_CTRL_USER: the get_accel of a user-defined controller (the reference's extension point: a BaseController subclass,
// base_controller.py:42-118), compiled into a copy of the library.  flow_amd.build.build_user writes the header
//   template <typename T> __device__ __forceinline__ T fs_user_accel(T v, T v_lead, T h, bool has_lead, T v_follow,
//                                                                    T h_follow, T dt, T max_accel, const T* p) { <body> }
// from the body flow_amd.controllers.CompiledController carries and compiles the parts with -DFS_USER_CONTROLLER_HEADER.
// Noise, fail-safes, the "on an edge" rule and the speed-mode clamps apply to it as to every other controller.
// ---------------------------------------------------------------------------
#ifdef FS_USER_CONTROLLER_HEADER
#include FS_USER_CONTROLLER_HEADER
constexpr bool kHasUserController = true;
#else
template <typename T>
__device__ __forceinline__ T fs_user_accel(T, T, T, bool, T, T, T, T, const T*) { return T(0); }
constexpr bool kHasUserController = false;
#endif

// ---------------------------------------------------------------------------
// BaseController.get_action for one vehicle (base_controller.py:70-118) + the RL command
// (envs/base.py:599-615): shared by the single-lane and the multi-lane step kernels.
// Returns the commanded acceleration; `commanded` = false means "no command this step" (S5).
// ---------------------------------------------------------------------------
// CSET = 1: the host guarantees every slot is IDM / RL / Sim (FLAG_IDM_SET), so the controller switch collapses.
// EXT_NOISE: the caller hands in this step's N(0,1) draw (`noise_g`: the kernels that keep the four draws of a Philox
// block over four steps, NoiseBlock below) instead of one Philox evaluation per call.
template <typename T, int CSET = 0, bool EXT_NOISE = false>
__device__ __forceinline__ T control_accel_on(const DevView<T>& s, const Slot<T>& sl, int flags, T v, T vl, T h, bool has,
                                              T vf, T hf, T mean_v, bool on_edge, bool have_rl, T a_rl, bool live,
                                              int rr, int ii, uint32_t nctr, T& cst, bool& commanded, T noise_g = T(0)) {
  if constexpr (CSET == 1 && EXT_NOISE) {
    // IDM / RL / Sim slots only, written without lane-divergent control flow: every lane evaluates the IDM law (on
    // whatever its parameter registers hold) and the slot's kind selects afterwards -- the same values as the
    // branches below, which cost a wave that is alone on its SIMD an exec-mask region each
    const bool is_rl = sl.ctrl == FS_CTRL_RL, is_sim = sl.ctrl == FS_CTRL_SIM;
    // (a slot that is not an IDM one holds no exponent: 4 keeps its lanes off pow()'s general path)
    T a = ctrl_idm(v, vl, h, has, sl.p, (is_rl || is_sim) ? T(4) : sl.p[4]);
    if (flags & FLAG_HAS_NOISE) a = (sl.noise > T(0)) ? a + sl.noise * noise_g : a;     // base_controller.py:109-110
    if (flags & FLAG_HAS_FAILSAFE) {                                                     // base_controller.py:113-116
      const T a1 = failsafe_instantaneous(a, v, h, has, s.dt), a2 = failsafe_safe_velocity(a, v, vl, h, s.dt, sl.delay);
      const T af = sl.failsafe == FS_FAILSAFE_INSTANTANEOUS ? a1 : (sl.failsafe == FS_FAILSAFE_SAFE_VELOCITY ? a2 : a);
      a = has ? af : a;
    }
    T ar = a_rl;
    if (s.clip_actions) ar = tmin(tmax(ar, s.act_lo), s.act_hi);
    commanded = is_rl ? have_rl : (is_sim ? false : on_edge);
    return is_rl ? (have_rl ? ar : T(0)) : (is_sim ? T(0) : a);
  }
  T acc = T(0);
  commanded = false;
  const int ct = sl.ctrl;
  if (ct == FS_CTRL_RL) {
    if (have_rl) {
      T a = a_rl;
      if (s.clip_actions) a = tmin(tmax(a, s.act_lo), s.act_hi);
      acc = a;
      commanded = true;
    }
  } else if (ct != FS_CTRL_SIM) {
    T a;
    if (CSET == 1) a = ctrl_idm(v, vl, h, has, sl.p);
    else switch (ct) {
      case FS_CTRL_PISATURATION: {
        const size_t slot = size_t(rr) * s.n_pis + (sl.pis_index < 0 ? 0 : sl.pis_index);
        a = ctrl_pisaturation(v, vl, h, s.dt, sl.max_accel, s.pis_hist + slot * s.pis_H, s.pis_n + slot, s.pis_H,
                              on_edge && live, cst);
        break;
      }
      case FS_CTRL_IDM: a = ctrl_idm(v, vl, h, has, sl.p); break;
      case FS_CTRL_CFM: a = ctrl_cfm(v, vl, h, has, sl.max_accel, sl.p); break;
      case FS_CTRL_BCM: a = ctrl_bcm(v, vl, h, has, vf, hf, sl.max_accel, sl.p); break;
      case FS_CTRL_LAC: a = ctrl_lac(v, vl, h, sl.length, cst, s.dt, sl.p); break;
      case FS_CTRL_OVM: a = ctrl_ovm(v, vl, h, has, sl.max_accel, sl.p); break;
      case FS_CTRL_LINEAR_OVM: a = ctrl_linear_ovm(v, h, sl.p); break;
      case FS_CTRL_GIPPS: a = ctrl_gipps(v, vl, h, s.dt, sl.p); break;
      case FS_CTRL_FOLLOWER_STOPPER: a = ctrl_follower_stopper(v, vl, h, has, s.dt, sl.p[0]); break;
      case FS_CTRL_USER: a = fs_user_accel<T>(v, vl, h, has, vf, hf, s.dt, sl.max_accel, sl.p); break;
      default: a = ctrl_follower_stopper(v, vl, h, has, s.dt, mean_v); break;
    }
    commanded = on_edge;
    if (CSET == 0 && ct == FS_CTRL_LAC && commanded && live) cst = a;
    if (flags & FLAG_HAS_NOISE) {                // base_controller.py:109-110
      if (sl.noise > T(0))
        a = a + sl.noise * (EXT_NOISE ? noise_g : gauss<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr, s.noise_exact != 0));
    }
    if (has) {                                   // base_controller.py:113-116, 141-142, 191-193
      if (sl.failsafe == FS_FAILSAFE_INSTANTANEOUS) a = failsafe_instantaneous(a, v, h, has, s.dt);
      else if (sl.failsafe == FS_FAILSAFE_SAFE_VELOCITY) a = failsafe_safe_velocity(a, v, vl, h, s.dt, sl.delay);
    }
    acc = a;
  }
  return acc;
}

// closed loops: "on an edge" (base_controller.py:98-99) from the loop coordinate; `internal` / `flow_x` are the
// segment lookup of x, made by every lane before the lane-divergent controller code
template <typename T, int CSET = 0>
__device__ __forceinline__ T control_accel(const DevView<T>& s, const SegTab<T>& tab, const Slot<T>& sl, int flags, T v,
                                           T vl, T h, bool has, T vf, T hf, T mean_v, T x, T quarter, T qj, bool have_rl,
                                           T a_rl, bool live, int rr, int ii, uint32_t nctr, T& cst, bool& commanded,
                                           T& flow_x) {
  bool on_edge = true;
  flow_x = x;
  if (s.nseg > 0) {
    bool internal;
    segment_lookup(s, tab, x, internal, flow_x);
    if (s.junction_mode && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM) on_edge = !internal;
  } else if (s.junction_mode && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM) {
    T u = x - tfloor(x / qj) * qj;
    on_edge = !(u >= quarter);
  }
  return control_accel_on<T, CSET>(s, sl, flags, v, vl, h, has, vf, hf, mean_v, on_edge, have_rl, a_rl, live, rr, ii,
                                   nctr, cst, commanded);
}

// ---------------------------------------------------------------------------
// the fused step kernel
// ---------------------------------------------------------------------------
// FAST = 1 is the specialisation the host selects for the headline configuration: every
// slot an IDMController without noise / fail-safe, speed_mode "aggressive", no junction
// mode, sims_per_step 1, AccelEnv head with the desired_velocity reward, no reset mask,
// no aux tracking.  It executes exactly the arithmetic of the generic path (same helper
// functions, same order), with everything it cannot need compiled out.
// CSET = 1 (generic path only) compiles the controller switch down to IDM / RL / Sim slots, the common
// "IDM humans + RL vehicles" population of the reference's experiments: every other feature stays.
template <typename T, int SEG, int FAST, int CSET = 0>
__global__ __launch_bounds__(64) void k_steps(DevView<T> s, int num_steps_arg, const uint8_t* __restrict__ mask,
                                              const float* __restrict__ actions, size_t act_stride,
                                              float* __restrict__ obs, float* __restrict__ rew,
                                              uint8_t* __restrict__ done, int obs_every_step) {
  constexpr int RPW = 64 / SEG;
  const int lane = threadIdx.x;
  const int seg = lane / SEG;
  const int i = lane % SEG;
  const int r = blockIdx.x * RPW + seg;
  const int N = s.N;
  const bool rvalid = r < s.R;
  const bool valid = rvalid && i < N;
  const int rr = rvalid ? r : s.R - 1;
  const int ii = i < N ? i : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const bool wrap_lead = (i + 1 >= N);            // slot N-1 (and the idle lanes): leader is slot 0
  const bool wrap_foll = (i == 0);
  const bool has = N > 1;
  const int flags = CSET == 1 ? (s.flags & ~(FLAG_NEED_FOLLOWER | FLAG_NEED_MEAN | FLAG_HAS_LAC)) : s.flags;

  // per-lane slot parameters (registers for the whole launch)
  Slot<T> sl;
  sl.ctrl = s.ctrl[ii];
  sl.failsafe = s.failsafe[ii];
  sl.speed_mode = s.speed_mode[ii];
  sl.rl_index = s.rl_index[ii];
  sl.pis_index = s.pis_index[ii];
#pragma unroll
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) sl.p[k] = s.p[k * N + ii];
  sl.noise = s.noise[ii];
  sl.delay = s.delay[ii];
  sl.max_accel = s.max_accel[ii];
  sl.max_decel = s.max_decel[ii];
  sl.length = s.length[ii];
  sl.sumo_tau = s.sumo_tau[ii];
  sl.sumo_min_gap = s.sumo_min_gap[ii];
  sl.sumo_max_speed = s.sumo_max_speed[ii];
  const T len_lead = lead_read<SEG>(sl.length, seg, wrap_lead);
  FS_SEGTAB_LDS(T, segrows);
  const SegTab<T> segtab = load_segtab(s, lane, segrows_st, segrows_fs, segrows_sl);
  const bool use_seg = !FAST && s.nseg > 0;            // figure eight: Flow's table coordinate / internal edges
  SegCursor<T> cur = {0, T(0), T(0), T(1), T(3.0e38)};

  // per-replica scalars
  const T base_len = s.ring_len[rr];
  const T L = base_len + T(4) * s.jlen;          // network.length(): edges + 4 junctions
  const T quarter = base_len / T(4);
  const T qj = quarter + s.jlen;
  const bool live_replica = rvalid && (FAST || mask == nullptr || mask[rr] != 0);
  // a masked launch (the warm-up steps of a reset) advances nothing in a wave none of whose replicas is selected: it
  // only has to report the observation of the unchanged state (the zero-step form) -- what makes a reset of the few
  // finished episodes cheap inside a captured closed-loop fragment (VecFlowEnv.capture with warm-up steps)
  const int num_steps = (!FAST && mask != nullptr && __ballot(live_replica) == 0ull) ? 0 : num_steps_arg;
  int tcount = s.time[rr];
  uint32_t nctr = (!FAST && (flags & FLAG_HAS_NOISE)) ? s.noise_ctr[rr] : 0u;

  // state
  T x = s.pos[idx];
  T v = s.vel[idx];
  if (use_seg) cur.init(s, segtab, x);
  T prev_v = v, last_acc = T(0);
  T cst = (!FAST && (flags & FLAG_HAS_LAC)) ? s.ctrl_state[idx] : T(0);
  if (!FAST && s.track_aux) { prev_v = s.prev_vel[idx]; last_acc = s.accel[idx]; }

  // time-t neighbour snapshot (S1/S10)
  T xl = lead_read<SEG>(x, seg, wrap_lead);
  T vl = lead_read<SEG>(v, seg, wrap_lead);
  T d = xl - x;
  d = d < T(0) ? d + L : d;
  T h = has ? d - len_lead : T(1000);

  // observation / action ORDER (accel.py:101-123, 150-169): get_ids() order = obs_perm (identity unless the start
  // positions were shuffled); with sort_vehicles the vehicles are ordered by the absolute position recorded at the
  // last additional_command (= the position before the last move), ties in id order (sorted() is stable)
  const bool sorted = !FAST && s.sort_vehicles != 0;
  const int perm_i = (!FAST && s.obs_perm != nullptr) ? s.obs_perm[ii] : ii;
  T xs = sorted ? s.sort_key[idx] : T(0);
  auto order_rank = [&](bool rl_only) -> int {
    int rk = 0;
    const int me_rl = (sl.ctrl == FS_CTRL_RL) ? 1 : 0;
    for (int j = 0; j < N; ++j) {
      const T xj = seg_read<SEG>(xs, j, seg);
      const int pj = seg_read_i<SEG>(perm_i, j, seg);
      const int rlj = seg_read_i<SEG>(me_rl, j, seg);
      const bool before = (xj < xs) || (xj == xs && pj < perm_i);
      if (before && (!rl_only || rlj != 0)) rk += 1;
    }
    return rk;
  };

  const T dt = s.dt;
  const int env = FAST ? int(FS_ENV_ACCEL) : s.env;
  const bool ma_wa = !FAST && env == FS_ENV_WAVE_ATTENUATION_PO_MA, ma_acc = !FAST && env == FS_ENV_ACCEL_PO_MA;
  const int obs_dim = (env == FS_ENV_WAVE_ATTENUATION_PO) ? 3 : (ma_wa ? 3 * s.num_rl : (ma_acc ? 6 * s.num_rl : 2 * N));
  const int sims_per_step = FAST ? 1 : s.sims_per_step;
  // the multi-agent ring heads (flow/envs/multiagent/ring/*): one block per RL vehicle at column rl_index; `xo` = Flow's
  // coordinate of x, `hh` the headway of the snapshot (written by the RL vehicle's lane; every lane makes the follower reads)
  auto write_ma = [&](float* orow_, T xo, T hh, T dd) {
    if (ma_wa) {                                             // multiagent/ring/wave_attenuation.py:188-208
      (void)dd;
      if (valid && sl.ctrl == FS_CTRL_RL) {
        float* o = orow_ + 3 * sl.rl_index;
        o[0] = float(v / T(15));
        o[1] = float((vl - v) / T(15));
        o[2] = float(hh / s.po_max_length);                  // get_headway: bumper to bumper
      }
    } else {                                                 // multiagent/ring/accel.py:163-208
      const T vf_ = foll_read<SEG>(v, seg, wrap_foll, N), hf_ = foll_read<SEG>(hh, seg, wrap_foll, N);
      const T xol = lead_read<SEG>(xo, seg, wrap_lead);
      if (valid && sl.ctrl == FS_CTRL_RL) {
        float* o = orow_ + 6 * sl.rl_index;
        const T lead_speed = has ? vl : s.max_speed, follow_speed = has ? vf_ : T(0);
        const T lead_head = has ? (xol - xo) - sl.length : L;        // (:186-188: no wrap-around, the ego's length)
        const T follow_head = has ? hf_ : L;                         // get_headway(follower)
        o[0] = float(xo / L);
        o[1] = float(v / s.max_speed);
        o[2] = float((lead_speed - v) / s.max_speed);
        o[3] = float(lead_head / L);
        o[4] = float((v - follow_speed) / s.max_speed);
        o[5] = float(follow_head / L);
      }
    }
  };
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;     // rows to advance per step
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;

  // RL actions are read one step ahead (a dependent HBM load at the top of every step would be fully exposed:
  // a replica-wave has at most a few companions on its SIMD): a_own = this lane's command, a_red = the action
  // of column `ii` (what the WaveAttenuation reward averages)
  const bool use_act = !FAST && actions != nullptr;
  const bool rl_lane = sl.ctrl == FS_CTRL_RL;
  const int own_col = sl.rl_index < 0 ? 0 : sl.rl_index;
  const bool red_lane = ii < s.num_rl && i < N;
  float a_own_next = 0.0f, a_red_next = 0.0f;
  if (use_act && num_steps > 0) {
    const float* a0 = actions + size_t(rr) * s.num_rl;
    if (rl_lane) a_own_next = a0[own_col];
    if (red_lane) a_red_next = a0[ii];
  }

  for (int step = 0; step < num_steps; ++step) {
    // ---- RL action of this lane (envs/base.py:599-615) -------------------
    const float* act = use_act ? actions + size_t(step) * act_stride + size_t(rr) * s.num_rl : nullptr;
    const float a_own = a_own_next, a_red = a_red_next;
    if (use_act && step + 1 < num_steps) {
      const float* an = actions + size_t(step + 1) * act_stride + size_t(rr) * s.num_rl;
      if (rl_lane) a_own_next = an[own_col];
      if (red_lane) a_red_next = an[ii];
    }
    bool crashed = false;
    for (int sub = 0; sub < sims_per_step; ++sub) {
      const bool live = live_replica && !crashed;
      // ---- controllers (S1: all read the snapshot) -----------------------
      T acc = T(0);
      bool commanded = true;
      if (FAST) {
        acc = ctrl_idm(v, vl, h, has, sl.p);
      } else {
        T vf = T(0), hf = T(0), mean_v = T(0);
        if (flags & FLAG_NEED_FOLLOWER) {
          vf = foll_read<SEG>(v, seg, wrap_foll, N);
          hf = foll_read<SEG>(h, seg, wrap_foll, N);
        }
        if (flags & FLAG_NEED_MEAN) mean_v = seg_sum<SEG>(valid ? v : T(0)) / T(N);
        const bool have_rl = rl_lane && (act != nullptr);
        T a_rl = have_rl ? T(a_own) : T(0);
        if (sorted && act != nullptr) {                  // accel.py:103-107: the k-th RL vehicle in sorted order
          const int col = order_rank(true);
          if (have_rl) a_rl = T(act[col]);
        }
        // "on an edge" (base_controller.py:98-99) and Flow's coordinate of x, from the segment cursor
        bool on_edge = true;
        T xa = x;
        const bool gated = s.junction_mode && sl.ctrl != FS_CTRL_RL && sl.ctrl != FS_CTRL_SIM;
        if (use_seg) {
          xa = cur.flow_x(x);
          if (gated) on_edge = !cur.internal(s);
        } else if (gated) {
          T u = x - tfloor(x / qj) * qj;
          on_edge = !(u >= quarter);
        }
        acc = control_accel_on<T, CSET>(s, sl, flags, v, vl, h, has, vf, hf, mean_v, on_edge, have_rl, a_rl,
                                        live && i < N, rr, ii, nctr, cst, commanded);
        if (sorted && live) xs = xa;                     // accel.py:150-169 additional_command: position before the move
      }
      // ---- apply_acceleration + SUMO integration (S4-S9) ------------------
      T next_vel = tmax(v + acc * dt, T(0));          // vehicle/traci.py:962
      T vc = v + (next_vel - v) * s.ramp;             // slowDown(.., 1e-3)
      T v_new = vc;
      if (!FAST && (flags & FLAG_NEED_SUMO)) {
        T v_sumo = sumo_idm_speed(v, vl, h, has, dt, sl);
        if (sl.speed_mode & 1) vc = tmin(vc, v_sumo);
        if (sl.speed_mode & 2) vc = tmin(vc, v + sl.max_accel * dt);
        if (sl.speed_mode & 4) vc = tmax(vc, v - sl.max_decel * dt);
        v_new = commanded ? vc : v_sumo;
      }
      if (!FAST && s.junction_on) {                   // S-J: right of way at the crossing
        // stream a blocks the box while one of its vehicles is inside, has not cleared it with its tail,
        // or reaches it within time_gap; stream a yields only to a vehicle of stream b inside the box
        const bool major_busy = seg_any<SEG>(valid && (x >= s.ja_in - s.j_time_gap * v) && (x < s.ja_out + sl.length), seg);
        const bool minor_in_box = seg_any<SEG>(valid && (x >= s.jb_in) && (x < s.jb_out + sl.length), seg);
        T cap = T(3.0e38);
        if ((x >= s.jb_in - s.j_lookahead) && (x < s.jb_in) && major_busy)
          cap = tmin(cap, sumo_idm_speed(v, T(0), s.jb_in - x, true, dt, sl));
        if ((x >= s.ja_in - s.j_lookahead) && (x < s.ja_in) && minor_in_box)
          cap = tmin(cap, sumo_idm_speed(v, T(0), s.ja_in - x, true, dt, sl));
        if ((sl.speed_mode & 1) || !commanded) v_new = tmin(v_new, cap);
      }
      T x_new = (!FAST && s.integrator == FS_BALLISTIC) ? x + (v + v_new) / T(2) * dt : x + v_new * dt;
      x_new = x_new >= L ? x_new - L : x_new;
      if (FAST) {
        x = x_new;
        v = v_new;
        tcount += 1;
      } else if (live) {
        prev_v = v;
        last_acc = acc;
        x = x_new;
        v = v_new;
        tcount += 1;
        nctr += 1u;
      }
      if (use_seg) cur.follow(s, segtab, x);
      // ---- vehicle update: new neighbour snapshot (vehicle/traci.py:219-250)
      xl = lead_read<SEG>(x, seg, wrap_lead);
      vl = lead_read<SEG>(v, seg, wrap_lead);
      d = xl - x;
      d = d < T(0) ? d + L : d;
      h = has ? d - len_lead : T(1000);
      // ---- check_collision (S12) ----------------------------------------
      bool c = has && seg_any<SEG>(valid && (h < s.crash_gap), seg);
      if (!FAST && s.junction_on)                     // S-J: both streams on the crossing point at once
        c = c || (seg_any<SEG>(valid && (x >= s.za_lo) && (x < s.za_hi), seg) &&
                  seg_any<SEG>(valid && (x >= s.zb_lo) && (x < s.zb_hi), seg));
      if (ma_wa || ma_acc) c = false;                  // multiagent/base.py:188-190: crash = 0
      crashed = crashed || (c && live);
    }

    // ---- get_state / compute_reward / done (envs/base.py:387-412) ---------
    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      const int oi = FAST ? ii : (sorted ? order_rank(false) : perm_i);
      const T xo = use_seg ? cur.flow_x(x) : x;
      if (ma_wa || ma_acc) {
        write_ma(orow, xo, h, d);
      } else if (env == FS_ENV_WAVE_ATTENUATION_PO) {
        // wave_attenuation.py:248-269; written by the RL vehicle's lane
        if (valid && sl.ctrl == FS_CTRL_RL && sl.rl_index == 0) {
          orow[0] = float(v / T(15));
          orow[1] = float((vl - v) / T(15));
          orow[2] = float(d / s.po_max_length);
        }
      } else if (valid) {
        orow[oi] = float(v / s.max_speed);               // accel.py:118-119
        orow[N + oi] = float(xo / L);                    // accel.py:120-121
      }
      // reward
      T reward;
      const bool bad = seg_any<SEG>(valid && (v < T(-100)), seg) || crashed;
      if (env == FS_ENV_ACCEL || ma_acc) {
        if (!FAST && s.evaluate) {
          reward = seg_sum<SEG>(valid ? v : T(0)) / T(N);                    // accel.py:111-112
        } else {                                                            // rewards.py:6-59
          T dv = valid ? v - s.target_velocity : T(0);
          T cost = tsqrt(seg_sum<SEG>(dv * dv));
          reward = tmax(s.max_cost - cost, T(0)) / (s.max_cost + T(1.1920928955078125e-07));
          reward = bad ? T(0) : reward;
        }
      } else {                                                              // wave_attenuation.py:113-139
        if (act == nullptr) {
          reward = T(0);
        } else {
          T a = T(0);
          if (red_lane) {
            a = T(a_red);
            if (s.clip_actions) a = tmin(tmax(a, s.act_lo), s.act_hi);
            a = tabs(a);
          }
          T mean_v = seg_sum<SEG>(valid ? v : T(0)) / T(N);
          T mean_a = seg_sum<SEG>(a) / T(s.num_rl);
          reward = T(4.0) * mean_v / T(20);
          if (mean_a > T(0)) reward = reward + T(4) * (T(0) - mean_a);
          reward = bad ? T(0) : reward;
        }
      }
      if (valid && ii == 0) {
        *rrow = float(reward);
        *drow = done_flag(tcount >= s.step_limit, crashed);              // envs/base.py:398-400
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
  }

  if (num_steps == 0) {   // observation of the current state only (Env.reset, envs/base.py:544-551)
    const int oi = FAST ? ii : (sorted ? order_rank(false) : perm_i);
    const T xo = use_seg ? cur.flow_x(x) : x;
    if (ma_wa || ma_acc) {
      write_ma(orow, xo, h, d);
    } else if (env == FS_ENV_WAVE_ATTENUATION_PO) {
      if (valid && sl.ctrl == FS_CTRL_RL && sl.rl_index == 0) {
        orow[0] = float(v / T(15));
        orow[1] = float((vl - v) / T(15));
        orow[2] = float(d / s.po_max_length);
      }
    } else if (valid) {
      orow[oi] = float(v / s.max_speed);
      orow[N + oi] = float(xo / L);
    }
    return;
  }

  // ---- write the state back -------------------------------------------------
  if (valid && live_replica) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    if (sorted) s.sort_key[idx] = xs;
    if (!FAST) {
      if (flags & FLAG_HAS_LAC) s.ctrl_state[idx] = cst;
      if (s.track_aux) { s.prev_vel[idx] = prev_v; s.accel[idx] = last_acc; }
    }
    if (ii == 0) {
      s.time[rr] = tcount;
      if (!FAST && (flags & FLAG_HAS_NOISE)) s.noise_ctr[rr] = nctr;
    }
  }
}

// ---------------------------------------------------------------------------
// k_rollout_idm: the rollout kernel of the headline configuration
// ---------------------------------------------------------------------------
// Same arithmetic as k_steps (same helpers, same order), restricted to what the host checks
// before choosing it (Sim::rollout_idm_ok): every slot an IDMController without noise or
// fail-safe, speed_mode "aggressive", no junction mode, Euler, sims_per_step 1, N > 1, AccelEnv
// head with the desired_velocity reward, observation written EVERY step.  Differences are
// purely structural, to keep the step loop one straight-line block the scheduler can overlap:
//   * idle lanes store to a scratch word instead of being masked off (no exec-mask regions);
//   * the per-replica reward tail (sqrt, divide) is deferred: lane j of a segment keeps the
//     reduced sum of squares of step j of a block of PERIOD = min(SEG, 32) steps: the PERIOD per-lane terms
//     are summed over the lanes TOGETHER (transposed_sum, ~3 VALU per step instead of a full butterfly each),
//     and the PERIOD rewards are finished and stored at once, one step per lane; the collision / bad-speed
//     facts of those steps travel as per-lane bit masks and are OR-reduced once per flush.
//   * BADCHK = false drops the "any speed < -100" test of rewards.py:46: with next_vel = max(.., 0) and a ramp in
//     (0, 1] a speed >= -100 can never fall below -100 again, so the test can only fire on speeds that were put
//     there from outside; the host selects BADCHK = true whenever the initial speeds or an fs_set_state upload
//     since the last full reset contained a value < -100 (Sim::neg_speed_possible).
// x / c for a divisor c that is constant over the launch.  FASTDIV (float only) replaces the
// IEEE division sequence by q0 = x*rc, r = fma(-q0, c, x), q = fma(r, rc, q0) with rc = RN(1/c).
// The host enables it per handle only after checking, for EVERY float mantissa of x, that the
// result equals x / c for each divisor used (Sim::fastdiv_ok, 2^23 cases per divisor): the check
// is scale-invariant, so it covers every x whose remainder r stays a normal number (|x| >= 2^-70).
// Where it is used, and why smaller dividends cannot change a result there:
//   x / L        the observation: x is 0 or >= ulp(L)/2, never tiny.
//   v / v0       feeds pw = (v/v0)^4 only: for v < 2^-70 both the exact and the approximate ratio
//                give pw = 0 (underflow), so 1 - pw - q*q is identical.
//   num / 2sqrt(ab)  feeds dyn = v*T + dq: |num| < 2^-70 makes |dq| < 2^-70; then either v*T >= 2^-40
//                and dq is below half an ulp of it (dyn = v*T either way), or dyn < 2^-39 and
//                s* = s0 + max(0, dyn) = s0 exactly because s0 >= 1e-3 (host-checked).
// The speed observation v / max_speed is an OUTPUT and speeds decay through the denormal range
// whenever a vehicle comes to rest (39 % of wave-steps on C2 hold one below 2^-90), so it keeps the
// IEEE sequence; guarding it per step cost more than it saved (measured 6.81 vs 7.12 G env-steps/s).
template <bool FASTDIV>
__device__ __forceinline__ float div_const(float x, float c, float rc) {
  if (FASTDIV) {
    float q0 = x * rc;
    float r = __builtin_fmaf(-q0, c, x);
    return __builtin_fmaf(r, rc, q0);
  }
  return x / c;
}
template <bool FASTDIV>
__device__ __forceinline__ double div_const(double x, double c, double) { return x / c; }

// n / d for operands that need none of the IEEE sequence's scaling or special-case fix-up
// (both finite, non-zero, |value| within [2^-60, 2^60]): the same refinement hipcc emits for `/`
// -- v_rcp_f32, one Newton step on the reciprocal, quotient, two residual corrections -- without
// v_div_scale (x2), v_div_fmas' post-scale and v_div_fixup.  With unscaled operands those four are
// the identity, so the result is bit-identical to n / d.  Used for s* / h only: s* >= s0 >= 1e-3
// (host-checked) and 1e-3 <= |h| <= loop length.
__device__ __forceinline__ float div_core(float n, float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y0, 1.0f);
  const float y = __builtin_fmaf(e, y0, y0);
  const float q0 = n * y;
  const float r0 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(r1, y, q1);
}
__device__ __forceinline__ double div_core(double n, double d) { return n / d; }
// div_core in two halves, for a divisor that stays while the dividends change (a vehicle's 2 sqrt(a b), v0, maxSpeed): the
// refined reciprocal is a function of the divisor alone -- kept, it saves v_rcp_f32 and two FMAs per quotient; the same
// operations in the same order, so div_core_by(n, d, div_core_recip(d)) == div_core(n, d) bit for bit
__device__ __forceinline__ float div_core_recip(float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float e = __builtin_fmaf(-d, y0, 1.0f);
  return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_core_by(float n, float d, float y) {
  const float q0 = n * y;
  const float r0 = __builtin_fmaf(-d, q0, n);
  const float q1 = __builtin_fmaf(r0, y, q0);
  const float r1 = __builtin_fmaf(-d, q1, n);
  return __builtin_fmaf(r1, y, q1);
}

// ---- the IDM-set controllers of the open-network kernels with their divisions as div_core (CSET = 1, float32) ----------
// k_steps_open / k_steps_wide evaluate, per vehicle and sub-step, the IDM law and the SUMO car-following speed once or
// twice: nine IEEE divisions (ten instructions each and two wait states for v_div_fmas' vcc) and three square roots of
// launch constants.  Here: div_core (eight instructions, bit-identical for operands in its range) and the roots taken
// once per launch (FdSlot).  Host-checked premises (Sim::open_div_ok): s0 and minGap in [1e-3, 1e6] -- the dividends
// s* and ss stay in range; v0, maxSpeed (the bottleneck's action-driven one is clamped to [0.01, 23]), the speed limit
// and both 2 sqrt(a b) in [2^-20, 2^20].  The remaining dividends may be tiny or zero -- v (v - v_lead) and v itself as
// a vehicle comes to rest -- where div_core's last bits may differ from the IEEE quotient's: the consumers cannot tell
// (the argument of div_const above: a quotient below 2^-58 vanishes next to v T or next to s0 / minGap >= 1e-3, and
// (v / v0)^4 underflows to zero either way).
struct FdSlot { float ts_idm, ts_sumo; };
__device__ __forceinline__ FdSlot make_fd(const Slot<float>& sl) {
  return FdSlot{2.0f * tsqrt(sl.p[2] * sl.p[3]), 2.0f * tsqrt(sl.max_accel * sl.max_decel)};
}
__device__ __forceinline__ float idm_fd(float v, float vl, float h, bool has, const float* p, float delta, float ts, bool delta4) {
  const float hh = tabs(h) < 1e-3f ? 1e-3f : h;                       // ctrl_idm, same operation order
  const float dyn = v * p[1] + div_core(v * (v - vl), ts);
  const float m = hmax(0.0f, dyn);
  const float s_star = has ? p[5] + m : 0.0f;
  const float q = div_core(s_star, hh);
  const float ratio = div_core(v, p[0]);
  float pw;
  if (delta4) { const float r2 = ratio * ratio; pw = r2 * r2; } else pw = pow_delta(ratio, delta);    // (wave-uniform)
  return p[2] * (1.0f - pw - q * q);
}
__device__ __forceinline__ float sumo_acc_fd(float v, float vl, float h, bool has, const Slot<float>& s, float max_speed,
                                             float ts) {
  const float gap = hmax(h, 1e-3f);                                   // sumo_idm_speed, same operation order
  const float m = hmax(0.0f, v * s.sumo_tau + div_core(v * (v - vl), ts));
  const float ss = s.sumo_min_gap + m;
  const float qq = div_core(ss, gap);
  const float q = has ? qq : 0.0f;
  const float r = div_core(v, max_speed);
  const float r2 = r * r;
  return s.max_accel * (1.0f - r2 * r2 - q * q);
}
__device__ __forceinline__ float sumo_speed_fd(float v, float vl, float h, bool has, float dt, const Slot<float>& s,
                                               float max_speed, float ts) {
  return hmax(0.0f, v + sumo_acc_fd(v, vl, h, has, s, max_speed, ts) * dt);
}
// control_accel_on<float, 1, true> (its branch-free form) over idm_fd.  (VIEW: DevView<float>, or DevView<double> for the
// FS_MIXED instantiation of the open-network kernel, CSET = 2: float32 models inside the float64 kernel)
template <typename VIEW>
__device__ __forceinline__ float control_accel_fd(const VIEW& sv, const Slot<float>& sl, const FdSlot& fd, int flags,
                                                  float v, float vl, float h, bool has, bool on_edge, bool have_rl, float a_rl,
                                                  bool& commanded, float noise_g) {
  struct { float dt, act_lo, act_hi; int clip_actions; } s = {float(sv.dt), float(sv.act_lo), float(sv.act_hi), sv.clip_actions};
  const bool is_rl = sl.ctrl == FS_CTRL_RL, is_sim = sl.ctrl == FS_CTRL_SIM;
  const bool delta4 = (flags & FLAG_DELTA4) != 0;
  float a = idm_fd(v, vl, h, has, sl.p, (is_rl || is_sim) ? 4.0f : sl.p[4], fd.ts_idm, delta4);
  if (flags & FLAG_HAS_NOISE) a = (sl.noise > 0.0f) ? a + sl.noise * noise_g : a;     // base_controller.py:109-110
  if (flags & FLAG_HAS_FAILSAFE) {                                                     // base_controller.py:113-116
    const float a1 = failsafe_instantaneous(a, v, h, has, s.dt), a2 = failsafe_safe_velocity(a, v, vl, h, s.dt, sl.delay);
    const float af = sl.failsafe == FS_FAILSAFE_INSTANTANEOUS ? a1 : (sl.failsafe == FS_FAILSAFE_SAFE_VELOCITY ? a2 : a);
    a = has ? af : a;
  }
  const float lo = s.clip_actions ? s.act_lo : -3.0e38f, hi = s.clip_actions ? s.act_hi : 3.0e38f;
  const float ar = hmin(hmax(a_rl, lo), hi);
  commanded = is_rl ? have_rl : (is_sim ? false : on_edge);
  return is_rl ? (have_rl ? ar : 0.0f) : (is_sim ? 0.0f : a);
}

// x / c, correctly rounded to float for every float x (any sign, zero, denormal) and every normal float c:
// q = RN64(x * RN64(1/c)) is within 2^-52 of the quotient, one fma pair makes it the correctly rounded float64
// quotient up to 1 ulp64 and EXACT whenever the quotient is representable (the only way to sit on a float32
// rounding boundary, ties of the denormal range included); otherwise the quotient of two 24-bit numbers stays
// >= 2^-47 (relative) away from every boundary, so the second rounding cannot differ from a single one.
__device__ __forceinline__ float div_via_f64(float x, double c, double rc) {
  const double xd = double(x);
  double q = xd * rc;
  const double r = __builtin_fma(-q, c, xd);
  q = __builtin_fma(r, rc, q);
  return float(q);
}
// x / c for a compile-time-known c (an OUTPUT: must be the correctly rounded quotient): float through div_via_f64
__device__ __forceinline__ float div_out(float x, double c) { return div_via_f64(x, c, 1.0 / c); }
__device__ __forceinline__ double div_out(double x, double c) { return x / c; }


template <typename T, int SEG, bool DELTA4, bool FASTDIV, bool BADCHK>
__global__ __launch_bounds__(1024) void k_rollout_idm(DevView<T> s, int num_steps, float* __restrict__ obs,
                                                      float* __restrict__ rew, uint8_t* __restrict__ done,
                                                      float* __restrict__ dump) {
  // Waves are independent (no LDS, no barrier); the block size only decides how many of them share a
  // CU.  Measured at R = 4096 (2048 waves, 1500 steps): 64 threads 0.657 ms, 256: 0.626, 512: 0.621,
  // 768: 0.737, 1024: 0.909 -- packing 4 waves per SIMD onto half the CUs is SLOWER, i.e. the loop is
  // bound by VALU issue (~4 cycles per instruction of this mix), not by exposed latency; 512 keeps every
  // CU busy with 2 waves per SIMD (Sim::rollout_block, profiles/r01_sweep_block.log).
  constexpr int RPW = 64 / SEG;
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

  T p[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) p[k] = s.p[k * N + ii];
  const T len_lead = lead_read<SEG>(s.length[ii], seg, wrap_lead);
  const T base_len = s.ring_len[rr];
  const T L = base_len + T(4) * s.jlen;
  int tcount = s.time[rr];
  T x = s.pos[idx];
  T v = s.vel[idx];
  T xl = lead_read<SEG>(x, seg, wrap_lead);
  T vl = lead_read<SEG>(v, seg, wrap_lead);
  T d = xl - x;
  d = d < T(0) ? d + L : d;
  T h = d - len_lead;

  const T dt = s.dt, ramp = s.ramp;
  const T two_sqrt_ab = T(2) * tsqrt(p[2] * p[3]);
  const T rc_L = T(1) / L, rc_v0 = T(1) / p[0], rc_ab = T(1) / two_sqrt_ab;
  // Observation stores: one wave-uniform base pointer that advances by a whole [R, 2N] block per step (scalar
  // add) plus a 32-bit per-lane byte offset: one v_lshl_add_u64 per store instead of carrying a 64-bit pointer per
  // lane.  (Forcing the SGPR-base `global_store_dword voff, vdata, saddr` form with a one-instruction asm was
  // measured slower, 11.38 vs 11.49 G env-steps/s: the volatile asm pins the scheduler.)  Idle lanes (i >= N, or a replica index past R) are exact clones of slot N-1 /
  // replica R-1 (same state, same leader, same parameters: see ii / rr / wrap_lead above), so they store the SAME
  // values to the SAME addresses as the lane they shadow instead of being masked off.
  const unsigned row = 2u * unsigned(N);
  const unsigned off_b = (unsigned(rr) * row + unsigned(ii)) * 4u;      // BYTE offsets: the host guarantees < 2^32
  const unsigned off_b2 = off_b + unsigned(N) * 4u;
  char* ob = reinterpret_cast<char*>(obs);
  const size_t ob_step = size_t(s.R) * row * sizeof(float);
  (void)dump;
  constexpr int PERIOD = SEG < 32 ? SEG : 32;     // steps whose reward tail is finished together
  unsigned crash_bits = 0u, bad_bits = 0u;

  // PERIOD steps at a time: each keeps its per-lane (v - v_target)^2 in a register; the PERIOD vectors are then
  // summed over the lanes together (transposed_sum) and lane j finishes the reward of step j
  for (int base = 0; base < num_steps; base += PERIOD) {
    T sq[PERIOD];
#pragma unroll
    for (int slot = 0; slot < PERIOD; ++slot) {
      sq[slot] = T(0);
      if (base + slot < num_steps) {                       // wave-uniform
        // IDMController.get_accel (ctrl_idm, car_following_models.py:464-482)
        T hh = tabs(h) < T(1e-3) ? T(1e-3) : h;
        T num = v * (v - vl);
        T dq = div_const<FASTDIV>(num, two_sqrt_ab, rc_ab);
        T ratio = div_const<FASTDIV>(v, p[0], rc_v0);
        T dyn = v * p[1] + dq;
        // max(0, dyn) written so that it compiles to one v_max_f32 (operand order only matters for a NaN dyn,
        // which no finite state produces; k_steps keeps np.maximum's NaN propagation)
        T s_star = p[5] + tmax(dyn, T(0));
        T q = FASTDIV ? div_core(s_star, hh) : s_star / hh;
        T pw;
        if (DELTA4) { T r2 = ratio * ratio; pw = r2 * r2; } else { pw = pow_delta(ratio, p[4]); }
        T acc = p[2] * (T(1) - pw - q * q);
        // apply_acceleration + integration (S4-S9)
        T next_vel = tmax(v + acc * dt, T(0));
        v = v + (next_vel - v) * ramp;
        T x_new = x + v * dt;
        x = x_new >= L ? x_new - L : x_new;
        tcount += 1;
        // new neighbour snapshot (S10) and collision check (S12)
        xl = lead_read<SEG>(x, seg, wrap_lead);
        vl = lead_read<SEG>(v, seg, wrap_lead);
        d = xl - x;
        d = d < T(0) ? d + L : d;
        h = d - len_lead;
        // collision (S12) and the v < -100 guard (rewards.py:46) are per-lane facts of this step: each lane
        // records them in bit `slot` of a mask and the per-replica "any" is taken for all PERIOD steps at once
        // at the flush (an OR-butterfly over the segment), instead of two ballots per step.  (Collecting the
        // ballots as scalar masks instead was measured slower: 10.24 vs 10.45 G env-steps/s.)
        // (idle lanes are clones of a valid lane: their flags repeat that lane's and need no masking)
        const unsigned bit = 1u << slot;
        crash_bits |= (h < s.crash_gap) ? bit : 0u;
        if (BADCHK) bad_bits |= (v < T(-100)) ? bit : 0u;
        // AccelEnv.get_state (accel.py:116-123)
        *reinterpret_cast<float*>(ob + off_b) = float(v / s.max_speed);   // an output: IEEE division (see div_const)
        *reinterpret_cast<float*>(ob + off_b2) = float(div_const<FASTDIV>(x, L, rc_L));
        ob += ob_step;
        // rewards.desired_velocity, first half: this lane's term of the sum of squares (rewards.py:53-54)
        T dv = valid ? v - s.target_velocity : T(0);
        sq[slot] = dv * dv;
      }
    }
    const T racc = transposed_sum<SEG, PERIOD>(sq, lane);
    const int last = (num_steps - base < PERIOD ? num_steps - base : PERIOD) - 1;   // last step of this block
    const unsigned crash_any = seg_or<SEG>(crash_bits);
    const unsigned bad_any = (BADCHK ? seg_or<SEG>(bad_bits) : 0u) | crash_any;
    crash_bits = 0u;
    bad_bits = 0u;
    const int j = i & (PERIOD - 1);                         // the step of the block this lane finishes
    if (rvalid && i < PERIOD && j <= last) {
      const bool my_bad = (bad_any >> j) & 1u;
      const bool my_crash = (crash_any >> j) & 1u;
      const int t_i = tcount - (last - j);                 // time counter after that step
      T cost = tsqrt(racc);
      T reward = tmax(s.max_cost - cost, T(0)) / (s.max_cost + T(1.1920928955078125e-07));   // rewards.py:59
      reward = my_bad ? T(0) : reward;                                                        // rewards.py:46
      const size_t o = size_t(base + j) * s.R + rr;
      rew[o] = float(reward);
      done[o] = done_flag(t_i >= s.step_limit, my_crash);                                   // envs/base.py:398-400
    }
  }
  if (valid) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    if (ii == 0) s.time[rr] = tcount;
  }
}

// ---------------------------------------------------------------------------
// k_steps_ml: multi-lane ring (RingNetwork lanes > 1; LaneChangeAccelEnv head)
// ---------------------------------------------------------------------------
// Same lane mapping as k_steps (lane = vehicle, segment = replica), but the leader of a vehicle
// is no longer a fixed neighbour: it is the nearest vehicle ahead IN THE SAME LANE and changes
// when a vehicle changes lane.  Every sub-step each lane scans the other N-1 slots of its
// segment (cross-lane reads through the LDS crossbar, ds_bpermute) and keeps the smallest arc
// distance -- oracle/refsim.py MultiLaneRingOracle.neighbours, ML2.  O(N) per vehicle instead
// of a sort: N <= 64 and the scan is 2 bpermutes + ~8 VALU per candidate.
template <typename T>
__device__ __forceinline__ T bperm(T v, int src_lane) { return __shfl(v, src_lane, 64); }

template <typename T, int SEG, bool LC /* some vehicle changes lane on its own (ML7) */,
          bool LCPO = false /* the LaneChangeAccelPOEnv head (ML8): an instantiation of its own, its per-lane neighbour search
                               would push the float64 kernels of the other heads past 256 VGPRs */>
__global__ __launch_bounds__(64) void k_steps_ml(DevView<T> s, int num_steps_arg, const uint8_t* __restrict__ mask,
                                                 const float* __restrict__ actions, size_t act_stride,
                                                 float* __restrict__ obs, float* __restrict__ rew,
                                                 uint8_t* __restrict__ done, int obs_every_step) {
  constexpr int RPW = 64 / SEG;
  const int lane_id = threadIdx.x;
  const int seg = lane_id / SEG;
  const int i = lane_id % SEG;
  const int segbase = seg * SEG;
  const int r = blockIdx.x * RPW + seg;
  const int N = s.N;
  const bool rvalid = r < s.R;
  const bool valid = rvalid && i < N;
  const int rr = rvalid ? r : s.R - 1;
  const int ii = i < N ? i : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const int flags = s.flags;
  const bool lcpo_env = LCPO;                                         // LaneChangeAccelEnv's actions and reward, own head
  const bool lc_env = (s.env == FS_ENV_LANE_CHANGE_ACCEL) || lcpo_env;
  const int act_w = s.num_rl * (lc_env ? 2 : 1);

  Slot<T> sl;
  sl.ctrl = s.ctrl[ii];
  sl.failsafe = s.failsafe[ii];
  sl.speed_mode = s.speed_mode[ii];
  sl.rl_index = s.rl_index[ii];
  sl.pis_index = s.pis_index[ii];
#pragma unroll
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) sl.p[k] = s.p[k * N + ii];
  sl.noise = s.noise[ii];
  sl.delay = s.delay[ii];
  sl.max_accel = s.max_accel[ii];
  sl.max_decel = s.max_decel[ii];
  sl.length = s.length[ii];
  sl.sumo_tau = s.sumo_tau[ii];
  sl.sumo_min_gap = s.sumo_min_gap[ii];
  sl.sumo_max_speed = s.sumo_max_speed[ii];
  FS_SEGTAB_LDS(T, segrows);
  const SegTab<T> segtab = load_segtab(s, int(threadIdx.x), segrows_st, segrows_fs, segrows_sl);

  const T base_len = s.ring_len[rr];
  const T L = base_len + T(4) * s.jlen;
  const T quarter = base_len / T(4);
  const T qj = quarter + s.jlen;
  const bool live_replica = rvalid && (mask == nullptr || mask[rr] != 0);
  const int num_steps = (mask != nullptr && __ballot(live_replica) == 0ull) ? 0 : num_steps_arg;   // as in k_steps
  int tcount = s.time[rr];
  uint32_t nctr = (flags & FLAG_HAS_NOISE) ? s.noise_ctr[rr] : 0u;

  T x = s.pos[idx];
  T v = s.vel[idx];
  int ln = s.lane[idx];
  int last_lc = s.last_lc[idx];
  T prev_v = v, last_acc = T(0);
  T cst = (flags & FLAG_HAS_LAC) ? s.ctrl_state[idx] : T(0);
  if (s.track_aux) { prev_v = s.prev_vel[idx]; last_acc = s.accel[idx]; }
  // sort_vehicles (LaneChangeAccelEnv, lane_change_accel.py:100-139 through AccelEnv.sorted_ids): as in k_steps the
  // vehicles are ordered by the position recorded at the last additional_command, ties in id order
  // get_ids() order = obs_perm (identity unless InitialConfig.shuffle handed the start places to shuffled ids)
  const bool sorted = s.sort_vehicles != 0;
  const int perm_i = s.obs_perm != nullptr ? s.obs_perm[ii] : ii;
  T xs = sorted ? s.sort_key[idx] : T(0);
  auto order_rank = [&](bool rl_only) -> int {
    int rk = 0;
    const int me_rl = (sl.ctrl == FS_CTRL_RL) ? 1 : 0;
    for (int j = 0; j < N; ++j) {
      const T xj = seg_read<SEG>(xs, j, seg);
      const int pj = seg_read_i<SEG>(perm_i, j, seg);
      const int rlj = seg_read_i<SEG>(me_rl, j, seg);
      const bool before = (xj < xs) || (xj == xs && pj < perm_i);
      if (before && (!rl_only || rlj != 0)) rk += 1;
    }
    return rk;
  };

  const T dt = s.dt;
  const int obs_dim = lcpo_env ? (4 * s.num_lanes + 1) * s.num_rl
                                : (lc_env ? 3 * N : ((s.env == FS_ENV_WAVE_ATTENUATION_PO) ? 3 : 2 * N));
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;

  // own-lane neighbour scan (ML2): leader slot / arc / speed / length, follower slot
  int lead = -1, foll = -1;
  T dlead = T(0), vl = T(0), len_lead = T(0), h = T(1000), vf = T(0);
  bool has = false;
  // ML7 (autonomous lane changing, the lane-drop network's rule M11 on a ring): vehicles whose type lets SUMO change
  // lanes (lane_change_mode with a strategic / cooperative / speed-gain / keep-right bit; not the RL vehicles, whose
  // changes are commanded) want the adjacent lane whose leader gap beats their headway by lc_min_gain, if the gaps to
  // the new leader and follower are at least the SUMO-IDM desired gaps; wishes are formed on every neighbour
  // snapshot, ONE change per replica and sub-step (largest gain, lowest slot) is executed with the move.
  constexpr bool lc_on = LC;
  const bool my_lc_auto = lc_on && s.lc_auto[ii] != 0 && sl.ctrl != FS_CTRL_RL;
  int lc_want = -1;
  T lc_gain = T(0);
  auto scan = [&]() {
    const T big = T(3.0e38);
    T best = big, bestf = big;
    lead = -1;
    foll = -1;

    // slot j is the same for every lane of the wave: x_j / lane_j come by v_readlane (no LDS round trip per
    // candidate), and the tests are bitwise so that they stay selects instead of nested exec-mask branches
    for (int j = 0; j < N; ++j) {
      const T xj = seg_read<SEG>(x, j, seg);
      const int lj = seg_read_i<SEG>(ln, j, seg);
      T dij = xj - x;                                   // arc from me to j
      const bool wrapf = (dij < T(0)) | ((dij == T(0)) & (j < ii));
      dij = wrapf ? dij + L : dij;
      T dji = x - xj;                                   // arc from j to me
      const bool wrapb = (dji < T(0)) | ((dji == T(0)) & (ii < j));
      dji = wrapb ? dji + L : dji;
      const bool same = (lj == ln) & (j != ii);
      // np.argmin takes the FIRST minimum in slot order; j ascends, so a strict < keeps the smaller slot on a tie
      const bool tl = same & (dij < best), tf = same & (dji < bestf);
      best = tl ? dij : best;
      lead = tl ? j : lead;
      bestf = tf ? dji : bestf;
      foll = tf ? j : foll;
    }
    has = lead >= 0;
    const int lsrc = segbase + (has ? lead : ii);
    dlead = best;
    vl = bperm(v, lsrc);
    len_lead = bperm(sl.length, lsrc);
    h = has ? dlead - len_lead : T(1000);
    if (lc_on) {                                        // ML7 wishes on this snapshot (operation order of M11)
      const bool ok0 = my_lc_auto && (tcount - last_lc >= s.lc_cooldown);
      const T two_sqrt = T(2) * tsqrt(sl.max_accel * sl.max_decel);
      T best_gain = -big;
      int best_lane = -1;
      // one more pass of the neighbour scan per side (not fused into the own-lane pass: the float64 instantiation
      // would need more than 256 VGPRs and re-materialise lane-held values from AGPRs, tests/test_codegen.py)
#pragma unroll 1
      for (int q = 0; q < 2; ++q) {                     // right first, so that left wins a tie
        const int tl = ln + (2 * q - 1);
        const bool valid_t = ok0 && tl >= 0 && tl < s.num_lanes;
        T abest = big, abestf = big;
        int alead = -1, afoll = -1;
        for (int j = 0; j < N; ++j) {
          const T xj = seg_read<SEG>(x, j, seg);
          const int lj = seg_read_i<SEG>(ln, j, seg);
          T dij = xj - x;
          const bool wrapf = (dij < T(0)) | ((dij == T(0)) & (j < ii));
          dij = wrapf ? dij + L : dij;
          T dji = x - xj;
          const bool wrapb = (dji < T(0)) | ((dji == T(0)) & (ii < j));
          dji = wrapb ? dji + L : dji;
          const bool adj = (lj == tl) & (j != ii);
          const bool al = adj & (dij < abest), af = adj & (dji < abestf);
          abest = al ? dij : abest;
          alead = al ? j : alead;
          abestf = af ? dji : abestf;
          afoll = af ? j : afoll;
        }
        const bool has_l = alead >= 0, has_f = afoll >= 0;
        const int ls = segbase + (has_l ? alead : ii), fs_ = segbase + (has_f ? afoll : ii);
        const T vl2 = bperm(v, ls), ll2 = bperm(sl.length, ls), vf2 = bperm(v, fs_);
        const T gap_l = has_l ? abest - ll2 : T(1000.0);
        const T gap_f = has_f ? abestf - sl.length : T(1000.0);
        const T v_l = has_l ? vl2 : T(0), v_f = has_f ? vf2 : T(0);
        const T need_l = sl.sumo_min_gap + tmax(T(0), v * sl.sumo_tau + v * (v - v_l) / two_sqrt);
        const T need_f = sl.sumo_min_gap + tmax(T(0), v_f * sl.sumo_tau + v_f * (v_f - v) / two_sqrt);
        const bool safe = (!has_l || gap_l >= need_l) && (!has_f || gap_f >= need_f);
        const T gain = gap_l - h;
        const bool take = valid_t && safe && (gain >= s.lc_min_gain) && (gain >= best_gain);
        best_gain = take ? gain : best_gain;
        best_lane = take ? tl : best_lane;
      }
      lc_want = best_lane;
      lc_gain = best_lane >= 0 ? best_gain : T(0);
    }
    if (!has) vl = T(-1001);                            // get_speed(None): the reference's error value
  };
  scan();

  // the observation of the state the lanes hold (orow: the row of this step)
  auto write_obs = [&]() {
    if constexpr (LCPO) {
      // ML8, LaneChangeAccelPOEnv (lane_change_accel.py:218-262 over vehicle/traci.py:776-867): per RL vehicle and lane q the
      // nearest leader / follower among ALL vehicles of lane q, the vehicle itself included (the reference's walk round
      // the loop ends on its own edge: alone in its lane it is its own leader and follower, one lap away).  Arc ahead =
      // x_j - x (+ L when negative or j is me: a vehicle of another lane at my position is a leader, bisect_left); arc
      // behind = x - x_j (+ L when <= 0).  Ties: leader the first, follower the last in slot order.  Every lane of
      // the wave scans (one pass per lane of the road, slot j by v_readlane as in scan()), the RL lanes store.
      const T big = T(3.0e38);
      const int lanes = s.num_lanes;
      const bool rl = valid && sl.ctrl == FS_CTRL_RL;
      const int col = sl.rl_index < 0 ? 0 : sl.rl_index;
      float* blk = orow + size_t(4) * lanes * col;
#pragma unroll 1
      for (int q = 0; q < lanes; ++q) {
        // (two passes, leaders then followers: one pass holds more values than the float64 instantiation has VGPRs for,
        // and its surplus would go to AGPRs -- tests/test_codegen.py keeps the v_readlane kernels out of them)
        T best = big, bestf = big;
        int lj = -1, fj = -1;
#pragma unroll 1
        for (int j = 0; j < N; ++j) {
          const T xj = seg_read<SEG>(x, j, seg);
          const int lane_j = seg_read_i<SEG>(ln, j, seg);
          T a = xj - x;
          a = ((a < T(0)) | (j == ii)) ? a + L : a;
          const bool tl = (lane_j == q) & (a < best);
          best = tl ? a : best;
          lj = tl ? j : lj;
        }
#pragma unroll 1
        for (int j = 0; j < N; ++j) {
          const T xj = seg_read<SEG>(x, j, seg);
          const int lane_j = seg_read_i<SEG>(ln, j, seg);
          T b = x - xj;
          b = (b <= T(0)) ? b + L : b;
          const bool tf = (lane_j == q) & (b <= bestf);
          bestf = tf ? b : bestf;
          fj = tf ? j : fj;
        }
        const bool hq = lj >= 0;
        const T v_l = bperm(v, segbase + (hq ? lj : ii)), len_l = bperm(sl.length, segbase + (hq ? lj : ii));
        const T v_f = bperm(v, segbase + (hq ? fj : ii));
        if (rl) {
          blk[q] = float(hq ? best - len_l : T(1000));
          blk[lanes + q] = float(hq ? bestf - sl.length : T(1000));
          blk[2 * lanes + q] = float(hq ? v_l / s.max_speed : T(0));
          blk[3 * lanes + q] = float(hq ? v_f / s.max_speed : T(0));
        }
      }
      if (rl) orow[size_t(4) * lanes * s.num_rl + col] = float(v);
      return;
    }
    const int oi = sorted ? order_rank(false) : perm_i;
    if (lc_env) {                                                  // lane_change_accel.py:100-117
      if (valid) {
        orow[oi] = float(v / s.max_speed);
        orow[N + oi] = float(x / L);
        orow[2 * N + oi] = float(T(ln) / T(s.num_lanes));
      }
    } else if (s.env == FS_ENV_WAVE_ATTENUATION_PO) {
      if (valid && sl.ctrl == FS_CTRL_RL && sl.rl_index == 0) {
        orow[0] = float(v / T(15));
        orow[1] = float((vl - v) / T(15));
        orow[2] = float((has ? dlead : T(0)) / s.po_max_length);
      }
    } else if (valid) {
      orow[oi] = float(v / s.max_speed);
      orow[N + oi] = float(x / L);
    }
  };

  for (int step = 0; step < num_steps; ++step) {
    const float* act = actions ? actions + size_t(step) * act_stride + size_t(rr) * act_w : nullptr;
    bool crashed = false;
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      const bool live = live_replica && !crashed;
      // ---- controllers on the snapshot (S1) -------------------------------
      T hf = T(0), mean_v = T(0);
      vf = T(0);
      if (flags & FLAG_NEED_FOLLOWER) {
        const int fsrc = segbase + (foll >= 0 ? foll : ii);
        vf = bperm(v, fsrc);
        hf = bperm(h, fsrc);
      }
      if (flags & FLAG_NEED_MEAN) mean_v = seg_sum<SEG>(valid ? v : T(0)) / T(N);
      const bool have_rl = (sl.ctrl == FS_CTRL_RL) && (act != nullptr);
      // sort_vehicles (lane_change_accel.py:137-139): the k-th RL vehicle in sorted order takes action pair k
      const int rl_place = (sorted && act != nullptr) ? order_rank(true) : (sl.rl_index < 0 ? 0 : sl.rl_index);
      const int acol = rl_place * (lc_env ? 2 : 1);
      T a_rl = have_rl ? T(act[acol]) : T(0);
      bool commanded = false;
      T xa_unused;
      T acc = control_accel(s, segtab, sl, flags, v, vl, h, has, vf, hf, mean_v, x, quarter, qj, have_rl, a_rl,
                            live && i < N, rr, ii, nctr, cst, commanded, xa_unused);
      if (sorted && live) xs = x;                    // accel.py:150-169 additional_command: the position before the move
      // ---- RL lane-change command (ML3) -----------------------------------
      int new_ln = ln;
      if (lc_env && have_rl) {
        const T dirv = T(act[acol + 1]);
        int direction = dirv > T(0.5) ? 1 : (dirv < T(-0.5) ? -1 : 0);
        const T last = s.last_lc_quirk ? h : T(last_lc);
        if (T(tcount + 1) <= s.lc_duration + last) direction = 0;      // lane_change_accel.py:143-147
        int target = ln + direction;
        target = target < 0 ? 0 : (target > s.num_lanes - 1 ? s.num_lanes - 1 : target);   // traci.py:982-984
        new_ln = target;
      }
      if (lc_env) {
        // refuse a change that would overlap a vehicle of the target lane (lane_change_mode != 0)
        bool clash = false;
        // (only a vehicle that is about to change lane can clash: skipped when no lane of the wave is)
        if (s.lane_change_mode != 0 && __ballot(new_ln != ln) != 0ull) {
          for (int j = 0; j < N; ++j) {
            const T xj = seg_read<SEG>(x, j, seg);
            const int lj = seg_read_i<SEG>(ln, j, seg);
            const T lenj = seg_read<SEG>(sl.length, j, seg);
            T dij = xj - x;
            const bool wrapf = (dij < T(0)) | ((dij == T(0)) & (j < ii));
            dij = wrapf ? dij + L : dij;
            const T dji = dij == T(0) ? T(0) : L - dij;
            clash = clash | ((j != ii) & (lj == new_ln) & ((dij < lenj) | (dji < sl.length)));
          }
        }
        if (clash || !live) new_ln = ln;
      }
      if (lc_on) {                                       // ML7: the one autonomous change of this sub-step
        const bool want = lc_want >= 0 && valid && live;
        const T gsel = want ? lc_gain : T(-3.0e38);
        const T gmax = seg_max<SEG>(gsel);
        const unsigned long long wb = seg_ballot<SEG>(want && gsel == gmax, seg);
        const int win = wb ? __ffsll((long long)wb) - 1 : -1;
        if (i == win) new_ln = lc_want;
      }
      // ---- apply_acceleration + integration (S4-S9) -----------------------
      T next_vel = tmax(v + acc * dt, T(0));
      T vc = v + (next_vel - v) * s.ramp;
      T v_new = vc;
      if (flags & FLAG_NEED_SUMO) {
        T v_sumo = sumo_idm_speed(v, vl, h, has, dt, sl);
        if (sl.speed_mode & 1) vc = tmin(vc, v_sumo);
        if (sl.speed_mode & 2) vc = tmin(vc, v + sl.max_accel * dt);
        if (sl.speed_mode & 4) vc = tmax(vc, v - sl.max_decel * dt);
        v_new = commanded ? vc : v_sumo;
      }
      T x_new = (s.integrator == FS_BALLISTIC) ? x + (v + v_new) / T(2) * dt : x + v_new * dt;
      x_new = x_new >= L ? x_new - L : x_new;
      if (live) {
        prev_v = v;
        last_acc = acc;
        x = x_new;
        v = v_new;
        tcount += 1;
        nctr += 1u;
        if (new_ln != ln) { ln = new_ln; last_lc = tcount; }        // vehicle/traci.py:205-209
      }
      // ---- new neighbour snapshot + collision check ------------------------
      scan();
      const bool c = seg_any<SEG>(valid && has && (h < s.crash_gap), seg);
      crashed = crashed || (c && live);
    }

    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      write_obs();
      T reward;
      const bool bad = seg_any<SEG>(valid && (v < T(-100)), seg) || crashed;
      if (s.env == FS_ENV_ACCEL || lc_env) {
        if (!lc_env && s.evaluate) {
          reward = seg_sum<SEG>(valid ? v : T(0)) / T(N);
        } else {
          T dv = valid ? v - s.target_velocity : T(0);
          T cost = tsqrt(seg_sum<SEG>(dv * dv));
          reward = tmax(s.max_cost - cost, T(0)) / (s.max_cost + T(1.1920928955078125e-07));
          reward = bad ? T(0) : reward;
        }
        if (lc_env) {                                                // lane_change_accel.py:92-96, in slot order
          const T last = s.last_lc_quirk ? h : T(last_lc);
          const bool hit = valid && sl.ctrl == FS_CTRL_RL && (last == T(tcount));
          unsigned long long b = __ballot(hit);
          if (SEG < 64) b = (b >> (seg * (SEG & 63))) & ((1ull << (SEG & 63)) - 1ull);
          const int cnt = __popcll(b);
          for (int k = 0; k < cnt; ++k) reward = reward - T(0.1);
        }
      } else {
        if (act == nullptr) {
          reward = T(0);
        } else {
          T a = T(0);
          if (ii < s.num_rl && i < N) {
            a = T(act[ii]);
            if (s.clip_actions) a = tmin(tmax(a, s.act_lo), s.act_hi);
            a = tabs(a);
          }
          T mean_v = seg_sum<SEG>(valid ? v : T(0)) / T(N);
          T mean_a = seg_sum<SEG>(a) / T(s.num_rl);
          reward = T(4.0) * mean_v / T(20);
          if (mean_a > T(0)) reward = reward + T(4) * (T(0) - mean_a);
          reward = bad ? T(0) : reward;
        }
      }
      if (valid && ii == 0) {
        *rrow = float(reward);
        *drow = done_flag(tcount >= s.step_limit, crashed);
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
  }

  if (num_steps == 0) {
    write_obs();
    return;
  }

  if (valid && live_replica) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    s.lane[idx] = ln;
    s.last_lc[idx] = last_lc;
    if (sorted) s.sort_key[idx] = xs;
    if (flags & FLAG_HAS_LAC) s.ctrl_state[idx] = cst;
    if (s.track_aux) { s.prev_vel[idx] = prev_v; s.accel[idx] = last_acc; }
    if (ii == 0) {
      s.time[rr] = tcount;
      if (flags & FLAG_HAS_NOISE) s.noise_ctr[rr] = nctr;
    }
  }
}

// Env.reset placement (envs/base.py:430, 494-518): selected replicas go back to
// their initial state; vehicles are inserted without moving (S13).
template <typename T>
__global__ void k_reset(DevView<T> s, const uint8_t* __restrict__ mask) {
  const size_t n = size_t(s.R) * s.N;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const int r = int(e / s.N);
    if (mask != nullptr && mask[r] == 0) continue;
    s.pos[e] = s.init_pos[e];
    s.vel[e] = s.init_vel[e];
    s.prev_vel[e] = s.init_vel[e];
    s.accel[e] = T(0);
    s.ctrl_state[e] = T(0);
    if (e % s.N == 0 && s.init_ring_len != nullptr) const_cast<T*>(s.ring_len)[r] = s.init_ring_len[r];
    if (s.sort_vehicles) {                         // accel.py:171-183: absolute_position = get_x_by_id at reset
      T xa = s.init_pos[e];
      if (s.nseg > 0) {
        bool internal;
        segment_lookup_args(s, s.init_pos[e], internal, xa);
      }
      s.sort_key[e] = xa;
    }
    // (the lane-change envs step on k_steps_ml, which reads the lane fields, on a one-lane ring too)
    if (s.num_lanes > 1 || s.env == FS_ENV_LANE_CHANGE_ACCEL || s.env == FS_ENV_LANE_CHANGE_ACCEL_PO) {
      s.lane[e] = s.init_lane[e];
      s.last_lc[e] = -(1 << 30);
    }
    if (s.n_pis > 0 && s.pis_index[e % s.N] >= 0) s.pis_n[size_t(r) * s.n_pis + s.pis_index[e % s.N]] = 0;
    if (e % s.N == 0) s.time[r] = 0;
  }
}

}  // namespace fs
