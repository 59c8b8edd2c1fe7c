// flowsim_queue.h -- gfx950 rollout kernel of the two-route merge (FS_NET_MERGE) in QUEUE order.
//
// k_steps_open (flowsim_open.h) keeps vehicle SLOT i in lane i and finds every vehicle's neighbours again whenever the
// order of the vehicles may have changed (rank by position, masks in rank order).  On a merge without lane changes a
// route is a queue -- vehicles enter upstream, leave downstream, nobody overtakes (flow/networks/merge.py: one lane per
// edge) -- so this kernel keeps the vehicles of a replica in the lanes of ONE wave in driving order:
//   lanes 0 .. nA-1        queue A, head first: every vehicle beyond the merge point (D, a prefix of nD lanes) followed
//                          by the vehicles of route 0 (highway) that are still upstream of it (U0)
//   lanes 63 .. 64-n1      queue U1, head in lane 63: the vehicles of route 1 (ramp) upstream of the merge point
// The leader (M5, flow/core/kernel/vehicle/traci.py:219-242) of a vehicle is the previous lane of its queue -- one DPP
// move -- and the leader of U1's head is the last vehicle of D; the vehicles whose leader I am (the follower candidates
// of the sticky rule O1, vehicle/traci.py:243-250) are the next lane of my queue and, for the last vehicle of D, the
// head of U1.  The slot a vehicle occupies in the state arrays (M1: lowest free slot of its type) is a LABEL it carries;
// everything the reference orders by id (reward sums, observation columns) is handed to the label's lane through LDS
// once per env step.
//
// A sub-step is straight-line code (controllers, movement, the snapshot from the neighbouring lanes) followed by ONE
// wave-uniform test: did anything happen that changes which vehicle sits in which lane, or what a lane is?
//   collision / tie    a vehicle is no longer strictly behind the previous lane of its queue: both queues are re-sorted
//                      by (x descending, equal x: lower slot first), the order oracle/opennet.py states;
//   join               the head of U1 has passed the merge point: it enters A at the place its position gives (the
//                      head of U0 passing it only moves the end of D);
//   arrival (M4)       the head of A leaves, A moves down one lane;
//   insertion (M2/M3)  an inflow is due: the new vehicle becomes the tail of its queue, checked against the old tail.
// Only then the event code runs (wave-uniform branches, one ds_bpermute per vehicle register) and the per-lane facts
// that depend on the arrangement -- which queue, leader yes / no, follower candidates, the thresholds of the event test
// itself -- are computed again; between events they are loop-carried constants.  oracle/queuenet.py restates this
// bookkeeping on the CPU and proves it equal to the all-pairs rules of oracle/opennet.py at every sub-step; the
// arithmetic is k_steps_open<float, ., 2, CSET = 1>'s, operation for operation (idm_fd, sumo_speed_fd, the same Philox
// draws), so the two kernels are bit-identical (tests/test_queue_gpu.py).  Scope (Sim::queue_ok): float32, IDM / RL /
// Sim-car-following slots without fail-safes, one vehicle length, Euler, the MultiAgentMergePOEnv head (FS_ENV_MERGE_MA:
// C5 of BASELINE.json), scheduled inflows, no replica mask, one or more steps per launch; everything else steps on
// k_steps_open.
#pragma once

namespace fs {

// LDS traffic of ONE wave: the hardware executes a wave's DS instructions in order, so all that is needed between a
// lane's write and another lane's read is that the compiler keeps the program order
__device__ __forceinline__ void q_fence() { asm volatile("" ::: "memory"); }

// A fact of a lane that only the events change -- which queue it is in, whether its vehicle has a leader, is an IDM
// vehicle, ... -- is kept as a bit of a wave-uniform 64-bit MASK (an SGPR pair) and consumed by v_cndmask directly.  As
// loop-carried `bool`s the same facts are lane masks too, but hipcc merges every one of them with the exec mask around
// the event branch (s_andn2 / s_and / s_or per mask and sub-step: 40 scalar instructions of nothing) and builds the
// selects from 0 / 1 integers.  (The masks these selects read are written by SALU instructions.)
__device__ __forceinline__ float selm(unsigned long long m, float a, float b) {          // bit `lane` of m ? a : b
  float r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}
__device__ __forceinline__ int selm(unsigned long long m, int a, int b) {
  int r;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
  return r;
}

// per-slot parameters, one row per label in LDS (read again by a lane whenever its vehicle changes)
struct alignas(16) QueueRow {
  float p0, p1, p2, p5;                 // IDM v0, T, a, s0
  float ts_idm, noise, max_accel, tau;  // 2 sqrt(a b) of the controller, sigma, SUMO accel / tau
  float min_gap, ts_sumo, adt, ddt;     // SUMO minGap, 2 sqrt(accel decel), speed-mode clamps (3e38: off)
  float sumo_max;                       // the vType maxSpeed
  int ctrl, speed_mode, rl_index;
};

template <bool NOISE, bool ACT>
__global__ __launch_bounds__(64) void k_merge_queue(DevView<float> s, OpenView<float> o, QueueConsts qc, int num_steps,
                                                    const float* __restrict__ actions, size_t act_stride,
                                                    float* __restrict__ obs, float* __restrict__ rew,
                                                    uint8_t* __restrict__ done, int obs_every_step) {
  using T = float;
  const T BIGV = 3.0e38f;
  const int lane = threadIdx.x;
  const int r = blockIdx.x;                       // one replica per wave (grid = R)
  const int N = s.N;
  const bool slot_ok = lane < N;                  // SLOT view: lane i speaks for slot i of the state arrays
  const int li = slot_ok ? lane : N - 1;
  const size_t base = size_t(r) * N;

  __shared__ OpenTabsLds<T> tabs_mem;
  __shared__ QueueRow rows[64];
  __shared__ float scr_f[64], scr_g[64], act_row[64];
  __shared__ int scr_i[64], scr_j[64];
  OpenTabs<T, true> tb;
  tb.load(o, lane, false, &tabs_mem);

  // ---- the slot tables as LDS rows --------------------------------------------------------------------------
  {
    QueueRow q;
    q.p0 = s.p[0 * N + li]; q.p1 = s.p[1 * N + li]; q.p2 = s.p[2 * N + li]; q.p5 = s.p[5 * N + li];
    q.noise = s.noise[li]; q.max_accel = s.max_accel[li];
    q.tau = s.sumo_tau[li]; q.min_gap = s.sumo_min_gap[li];
    const float p3 = s.p[3 * N + li], max_decel = s.max_decel[li];
    q.ts_idm = 2.0f * tsqrt(q.p2 * p3);
    q.ts_sumo = 2.0f * tsqrt(q.max_accel * max_decel);
    q.speed_mode = s.speed_mode[li];
    q.adt = (q.speed_mode & 2) ? q.max_accel * s.dt : 3.0e38f;
    q.ddt = (q.speed_mode & 4) ? max_decel * s.dt : 3.0e38f;
    q.sumo_max = s.sumo_max_speed[li];
    q.ctrl = s.ctrl[li]; q.rl_index = s.rl_index[li];
    rows[lane] = q;
  }
  const int slot_type = o.slot_type[li];                                   // (slot view)
  const bool slot_is_rl = slot_ok && s.ctrl[li] == FS_CTRL_RL;
  const int slot_rl_index = s.rl_index[li];
  __syncthreads();

  // ---- replica scalars (one replica per wave: all of them wave-uniform) ------------------------------------
  int tcount = s.time[r];
  uint32_t nctr = s.noise_ctr[r];
  int32_t* cnt = o.counters + size_t(r) * 8;
  int sim_steps = cnt[CNT_SIM_STEPS], seq_ctr = cnt[CNT_SEQ];
  int n_arr = cnt[CNT_ARRIVED], n_dep = cnt[CNT_DEPARTED], tot_arr = cnt[CNT_TOTAL_ARRIVED],
      tot_dep = cnt[CNT_TOTAL_DEPARTED];
  int emit_l = (lane < FS_MAX_INFLOWS) ? o.emitted[size_t(r) * FS_MAX_INFLOWS + lane] : 0;
  const bool my_flow = lane < o.n_inflows;             // lane f keeps inflow f's schedule (M2)
  const double my_per = my_flow ? o.flow_tab_d[lane] : 0.0;
  const double my_begin = my_flow ? o.flow_tab_d[64 + lane] : 0.0, my_end = my_flow ? o.flow_tab_d[128 + lane] : 0.0;
  const int my_number = my_flow ? o.flow_tab_i[128 + lane] : 0;
  // ... and its insertion constants (M3): coordinate, speed, and minGap / tau / 2 sqrt(accel decel) of its vehicle type
  const int fl = my_flow ? lane : 0;
  const float f_xdep = o.lane_tab[TAB_FL_XDEP * 64 + fl], f_vdep = o.lane_tab[TAB_FL_VDEP * 64 + fl];
  const float f_ts = o.lane_tab[TAB_FL_TWOSQRT * 64 + fl], f_gap = o.lane_tab[TAB_FL_MINGAP * 64 + fl],
              f_tau = o.lane_tab[TAB_FL_TAU * 64 + fl];
  const int f_typ = o.flow_tab_i[fl], f_route = o.flow_tab_i[64 + fl];
  // inflows that are due but found no room (M3: retried every sub-step): bit f.  The sub-step watches their gap
  // itself -- one test for all of them -- and runs the insertion code only when one of them fits
  unsigned long long pend_m = 0ull;
  int tlA = -1, tlU = -1;                              // lane of the vehicle an insertion on route 0 / 1 is checked against
  // the first sub-step index n (n = sim_steps - 1 of the sub-step's `now = n * sim_step`) at which some inflow is due:
  // the schedule is float64 (M2); the hot loop compares integers
  int due_n = 0;
  const double inv_dt_d = 1.0 / o.dt_d;
  auto due_index = [&](double t) -> int {
    if (!(t > 0.0)) return 0;
    const double q = t * inv_dt_d;                   // (a first guess: the two loops below make n exact whatever it is)
    if (!(q < 2.0e9)) return 0x7fffffff;
    int n = int(q);
    while (double(n) * o.dt_d < t) n += 1;
    while (n > 0 && double(n - 1) * o.dt_d >= t) n -= 1;
    return n;
  };

  auto my_due_of = [&](int k_me) -> int {
    const double due_t = my_begin + double(k_me) * my_per;
    const bool open_me = my_flow && (due_t <= my_end) && (my_number < 0 || k_me < my_number);
    return open_me ? due_index(due_t) : 0x7fffffff;
  };
  int my_due = my_due_of(emit_l);

  // ---- the vehicle this lane holds (slot view first: lane i = slot i) ---------------------------------------
  T x = s.pos[base + li];
  T v = s.vel[base + li];
  if (s.st16 != nullptr) state16_load(s, base + li, x, v);
  int route = slot_ok ? s.lane[base + li] : -1;
  int lab = lane;
  int seq = o.seq[base + li];
  int origin = o.origin[base + li];
  int foll = o.foll[base + li];
  T foll_h = o.foll_h[base + li];
  T prev_v = s.prev_vel[base + li], last_acc = s.accel[base + li];
  T vmax = o.vmax[base + li];
  float a_me = 0.0f;                                   // this step's action of my vehicle (ACT)
  T g0 = 0.0f, g1 = 0.0f, g2 = 0.0f, g3 = 0.0f;        // the four draws of my vehicle's current Philox block (NOISE)
  int nz_reload = 1;                                   // (wave-uniform) the block must be evaluated before the next draw
  int nA = 0, n1 = 0, nD = 0;
  unsigned long long alive_lab = __ballot(route >= 0);             // bit i: slot i holds a vehicle
  unsigned long long arr_rl = 0ull;                                 // RL slots that arrived in the last sub-step
  {
    const int a0 = o.arrived_rl[base + li];
    arr_rl = __ballot(slot_ok && a0 != 0);
  }

  // launch constants the sub-step reads: in VGPRs (uniform values the compiler would otherwise keep in SGPRs it does
  // not have: k_steps_open's loop spills ~60 of them to VGPR lanes and reads them back with v_readlane)
  const T dt = in_vgpr(float(s.dt)), ramp = in_vgpr(float(s.ramp));
  const T merge_x = in_vgpr(float(o.merge_x)), box_in = in_vgpr(float(o.box_in)), end_x = in_vgpr(float(o.end_x));
  const T tgap = in_vgpr(float(s.j_time_gap)), appr_lo = in_vgpr(float(o.box_in) - float(s.j_lookahead));
  const T LEN = in_vgpr(qc.veh_len);
  const T clip_lo = in_vgpr(s.clip_actions ? float(s.act_lo) : -3.0e38f), clip_hi = in_vgpr(s.clip_actions ? float(s.act_hi) : 3.0e38f);
  const int jm = __builtin_amdgcn_readfirstlane(s.junction_mode), jct = __builtin_amdgcn_readfirstlane(s.junction_on);
  const T c0 = in_vgpr(0.0f), c1000 = in_vgpr(1000.0f), cm1001 = in_vgpr(-1001.0f), cBIG = in_vgpr(3.0e38f);
  const int im1 = __builtin_bit_cast(int, in_vgpr(__builtin_bit_cast(float, -1)));

  // ---- my vehicle's parameters (by label) and the facts of my lane: recomputed by the events only --------------
  T p_v0 = 1.0f, p_T = 0.0f, p_a = 0.0f, p_s0 = 0.0f, p_ts = 1.0f, p_sig = 0.0f;          // IDM
  T u_acc = 0.0f, u_tau = 0.0f, u_gap = 1.0f, u_ts = 1.0f, u_adt = BIGV, u_ddt = BIGV, u_vmax = 1.0f;   // SUMO's model
  T y_pts = 1.0f, y_v0 = 1.0f, y_uts = 1.0f, y_vmax = 1.0f;     // div_core_recip of p_ts, p_v0, u_ts, u_vmax (set with them)
  T ia_lo = BIGV, ia_hi = BIGV, ib_lo = BIGV, ib_hi = BIGV;      // the internal stretches of my route
  int rl_col = 0;
  // (wave-uniform masks, bit = lane)
  unsigned long long mA = 0ull, mU = 0ull, mUh = 0ull, mA1 = 0ull, mU1 = 0ull, mHas = 0ull, mNl = 0ull, mCU0 = 0ull,
                     mE1 = 0ull, mE2 = 0ull, mKidm = 0ull, mKrl = 0ull, mSm1 = 0ull, mNoisy = 0ull, mHaveRl = 0ull;
  int s21i = 0, lab_c1 = -1, lab_c2 = -1;
  T ev_thr = BIGV, thr_mib = BIGV;
  int td = 0;                                           // lane of the last vehicle of D (0 if D is empty)

  auto load_params = [&]() {
    const QueueRow q = rows[lab & 63];
    p_v0 = q.p0; p_T = q.p1; p_a = q.p2; p_s0 = q.p5; p_ts = q.ts_idm; p_sig = q.noise;
    u_acc = q.max_accel; u_tau = q.tau; u_gap = q.min_gap; u_ts = q.ts_sumo; u_adt = q.adt; u_ddt = q.ddt;
    y_pts = div_core_recip(p_ts); y_v0 = div_core_recip(p_v0); y_uts = div_core_recip(u_ts);
    const bool k_idm = q.ctrl == FS_CTRL_IDM;
    mKidm = __ballot(k_idm);
    mKrl = __ballot(q.ctrl == FS_CTRL_RL);
    mSm1 = __ballot((q.speed_mode & 1) != 0);
    mNoisy = __ballot(k_idm && q.noise > 0.0f);
    rl_col = q.rl_index < 0 ? 0 : q.rl_index;
    const bool r1 = route == 1;
    ia_lo = r1 ? qc.in_lo[1][0] : qc.in_lo[0][0]; ia_hi = r1 ? qc.in_hi[1][0] : qc.in_hi[0][0];
    ib_lo = r1 ? qc.in_lo[1][1] : qc.in_lo[0][1]; ib_hi = r1 ? qc.in_hi[1][1] : qc.in_hi[0][1];
  };
  // every register that belongs to the vehicle, pulled from lane `src` where `take`
  auto gather_all = [&](int src, bool take) {
#define FS_Q_G(reg_) do { const auto t_ = bperm(reg_, src); reg_ = take ? t_ : reg_; } while (0)
    FS_Q_G(x); FS_Q_G(v); FS_Q_G(lab); FS_Q_G(route); FS_Q_G(seq); FS_Q_G(origin); FS_Q_G(foll); FS_Q_G(foll_h);
    FS_Q_G(prev_v); FS_Q_G(last_acc); FS_Q_G(vmax);
    if (ACT) FS_Q_G(a_me);
    if (NOISE) { FS_Q_G(g0); FS_Q_G(g1); FS_Q_G(g2); FS_Q_G(g3); }
#undef FS_Q_G
  };
  // what a lane is, after the arrangement changed (nA, n1 are current; the vehicles are in their lanes)
  auto classes = [&](bool reload) {
    const bool isA = lane < nA, isU = lane >= 64 - n1;
    const bool alive = isA | isU;
    if (!alive) { route = -1; lab = lane; x = 0.0f; v = 0.0f; vmax = 1.0f; }     // (values nobody reads, kept finite)
    if (reload) load_params();
    u_vmax = tmin(vmax, o.speed_limit);               // M10
    y_vmax = div_core_recip(u_vmax);
    nD = __popcll(__ballot(isA && x >= merge_x));
    td = nD > 0 ? nD - 1 : 0;
    const bool uh = isU && lane == 63;                 // U1's head: its leader is the last vehicle of D
    const bool has = isA ? lane > 0 : (isU && (lane < 63 || nD > 0));
    mA = __ballot(isA);
    mU = __ballot(isU);
    mUh = __ballot(uh);
    mA1 = __ballot(isA && lane > 0);
    mU1 = __ballot(isU && lane < 63);
    mHas = __ballot(has);
    mNl = __ballot(alive && !has);
    mCU0 = __ballot(isA && lane >= nD);
    // the event thresholds: arrival for A; the merge point for the heads of U0 and U1
    ev_thr = isA ? end_x : BIGV;
    ev_thr = (isA && lane == nD) ? merge_x : ev_thr;
    ev_thr = uh ? merge_x : ev_thr;
    thr_mib = uh ? box_in : BIGV;
    // O1: the vehicles whose leader I am: the next lane of my queue, and U1's head if I am the last vehicle of D
    const int lab_up = dpp_i<DPP_WAVE_SHR1>(lab), lab_dn = dpp_i<DPP_WAVE_SHL1>(lab);
    const int seq_up = dpp_i<DPP_WAVE_SHR1>(seq), seq_dn = dpp_i<DPP_WAVE_SHL1>(seq);
    const int lab_h = read_lane_i(lab, 63), seq_h = read_lane_i(seq, 63);
    const bool c1 = isA ? (lane + 1 < nA) : (isU && lane - 1 >= 64 - n1);
    const int sc1 = isA ? seq_dn : seq_up;
    lab_c1 = isA ? lab_dn : lab_up;
    lab_c2 = lab_h;
    const bool c2 = isA && (nD > 0) && (lane == nD - 1) && (n1 > 0);
    const bool e1 = c1 && (has || sc1 > seq), e2 = c2 && (has || seq_h > seq);
    mE1 = __ballot(e1);
    mE2 = __ballot(e2);
    s21i = (e1 && e2 && seq_h < sc1) ? -1 : 0;         // equal headways: the earlier id registers (vehicle/traci.py:243-250)
    // M3: the nearest vehicle ahead of an insertion is the tail of the route's queue (route 1 without a queue: of D)
    tlA = nA > 0 ? nA - 1 : -1;
    tlU = n1 > 0 ? 64 - n1 : (nD > 0 ? nD - 1 : -1);
  };

  // ---- (re)build the queues from whatever the lanes hold: A and U1 sorted by (x descending, lower slot first) -
  auto resort = [&]() {
    const bool al = route >= 0;
    const bool inA = al && (route == 0 || x >= merge_x);
    const bool inU = al && !inA;
    const unsigned long long bA = __ballot(inA), bU = __ballot(inU), bAl = bA | bU;
    int cA = 0, cU = 0;
    for (unsigned long long u = bAl; u; u &= u - 1ull) {
      const int j = __ffsll((long long)u) - 1;
      const T xj = read_lane(x, j);
      const int lj = read_lane_i(lab, j);
      const int ahead = int(xj > x) | (int(xj == x) & int(lj < lab));
      const bool jA = (bA >> j) & 1ull;               // (wave-uniform)
      cA += jA ? ahead : 0;
      cU += jA ? 0 : ahead;
    }
    nA = __popcll(bA);
    n1 = __popcll(bU);
    const int dead_rank = nA + __popcll(~bAl & ((1ull << lane) - 1ull));
    const int target = inA ? cA : (inU ? 63 - cU : dead_rank);
    // the inverse permutation: lane t learns which lane holds the vehicle that belongs to it
    const int src = __builtin_amdgcn_ds_permute(target << 2, lane);
    gather_all(src, true);
  };
  resort();
  classes(true);

  // ---- M5 / O1 from the structure: leader speed and headway, my sticky follower entry --------------------------
  T vl = -1001.0f, h = 1000.0f;
  auto snapshot = [&](T x_up, T x_dn, T& h_n, T& vl_n, int& foll_n, T& foll_h_n) {
    const T v_up = dpp<DPP_WAVE_SHR1>(v), v_dn = dpp<DPP_WAVE_SHL1>(v);
    const T x_t = read_lane(x, td), v_t = read_lane(v, td);     // the last vehicle of D: leader of U1's head
    const T x_h = read_lane(x, 63);                              // U1's head: follower candidate of that vehicle
    T x_l = selm(mA, x_up, x_dn), v_l = selm(mA, v_up, v_dn);
    x_l = selm(mUh, in_vgpr(x_t), x_l);
    v_l = selm(mUh, in_vgpr(v_t), v_l);
    h_n = selm(mHas, (x_l - x) - LEN, c1000);           // vehicle/traci.py:237
    vl_n = selm(mHas, v_l, cm1001);                     // get_speed(None): the accessor's error value
    // O1 (vehicle/traci.py:243-250): a candidate whose leader I am has the headway (x_me - x_c) - length_me
    const T xc1 = selm(mA, x_dn, x_up);
    const T f1 = selm(mE1, (x - xc1) - LEN, cBIG);
    const T f2 = selm(mE2, (x - x_h) - LEN, cBIG);
    // candidate 2 registers if its headway is smaller, or equal with the earlier id (s21i): as integer sign masks
    const T d = f2 - f1;
    const unsigned db = fbits(d), mag = db & 0x7fffffffu;
    const bool pick2 = int(db | ((mag - 1u) & unsigned(s21i))) < 0;
    const T bestf = hmin(f1, f2);
    const int bj = pick2 ? lab_c2 : lab_c1;
    const T start_h = selm(mNl, c1000, foll_h);
    const int start_f = selm(mNl, im1, foll);
    const bool better = bestf < start_h;                // (start_h <= 3e38 = "no candidate")
    foll_n = better ? bj : start_f;
    foll_h_n = better ? bestf : start_h;
  };
  {
    T fh_;
    int f_;
    snapshot(dpp<DPP_WAVE_SHR1>(x), dpp<DPP_WAVE_SHL1>(x), h, vl, f_, fh_);       // (no follower update at launch start)
  }

  // a value of the alive vehicles handed to the lane of their SLOT (others: `dflt`) -- through LDS, one wave
  auto to_slots_f = [&](T val, bool alive, T dflt) -> T {
    scr_f[lane] = dflt;
    q_fence();
    if (alive) scr_f[lab & 63] = val;
    q_fence();
    const T out = scr_f[lane];
    q_fence();
    return out;
  };
  auto to_slots_i = [&](int val, bool alive, int dflt) -> int {
    scr_i[lane] = dflt;
    q_fence();
    if (alive) scr_i[lab & 63] = val;
    q_fence();
    const int out = scr_i[lane];
    q_fence();
    return out;
  };

  // both routes' segment tables (launch constants) in VGPRs: read from the kernel-argument segment inside the head they came
  // back as vector memory loads, one L2 round trip per observation
  T sg_st[2][6], sg_fs[2][6], sg_sl[2][6];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      sg_st[rt][q] = in_vgpr(qc.seg_start[rt][q]);
      sg_fs[rt][q] = in_vgpr(qc.seg_flow[rt][q]);
      sg_sl[rt][q] = in_vgpr(qc.seg_slope[rt][q]);
    }
  }
  const int obs_dim = o.obs_dim;
  const double ms64 = double(s.max_speed), rc_ms64 = 1.0 / ms64, nl64 = double(o.net_length), rc_nl64 = 1.0 / nl64;
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(r) * obs_dim;
  float* rrow = rew + r;
  uint8_t* drow = done + r;

  // the vehicle in this lane leaves the network: its final state goes to its slot now (the slot's values of record
  // while it is free, as k_steps_open leaves them)
  auto retire = [&](bool mine) {
    if (mine) {
      const size_t e = base + size_t(lab & 63);
      if (s.st16 != nullptr) state16_store(s, e, x, v);
      else { s.pos[e] = x; s.vel[e] = v; }
      s.lane[e] = -1;
      s.prev_vel[e] = prev_v;
      s.accel[e] = last_acc;
      o.seq[e] = seq;
      o.origin[e] = origin;
      o.foll[e] = foll;
      o.foll_h[e] = foll_h;
      o.vmax[e] = vmax;
      o.lead[e] = -1;
      o.headway[e] = 1000.0f;
    }
  };

#ifdef FS_QDIAG
  unsigned long long dg_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};    // cycles: 0 events, 1 resort + join, 2 arrivals, 3 insertions, 4 classes, 5 noise, 6 head, 7 total
  int dg_n[4] = {0, 0, 0, 0};                               // calls: events, arrivals, insertions, noise blocks
#define FS_QT(var_) const unsigned long long var_ = __builtin_readcyclecounter()
#define FS_QA(slot_, t0_) dg_t[slot_] += __builtin_readcyclecounter() - (t0_)
#else
#define FS_QT(var_)
#define FS_QA(slot_, t0_)
#endif
  // ---- the events of a sub-step (cold: a wave-uniform branch of the loop below) -----------------------------------
  auto events = [&](int step, bool structural, bool try_insert) {
    FS_QT(q_ev0);
    bool moved = false, reload = false;
    bool isA = lane < nA, isU = lane >= 64 - n1;
    arr_rl = 0ull;
    unsigned long long just_arrived = 0ull;            // by SLOT: a slot freed now is free from the next sub-step on
    n_arr = 0;
    n_dep = 0;
    if (structural) {
      {
        const T x_up = dpp<DPP_WAVE_SHR1>(x), x_dn = dpp<DPP_WAVE_SHL1>(x);
        const bool bad = (isA && lane > 0 && !(x < x_up)) || (isU && lane < 63 && !(x < x_dn));
        if (__ballot(bad) != 0ull) {                   // a vehicle caught up with the previous lane: re-sort
          resort();
          isA = lane < nA; isU = lane >= 64 - n1;
          moved = reload = true;
        }
      }
      // the head of U1 has passed the merge point: it joins A at the place its position gives
      while (n1 > 0 && read_lane(x, 63) >= read_lane(merge_x, 0)) {
        const T xe = read_lane(x, 63);
        const int le = read_lane_i(lab, 63);
        const int k = __popcll(__ballot(isA && ((x > xe) || (x == xe && lab < le))));
        const bool up = (lane > k && lane <= nA) || (lane > 64 - n1 && lane <= 63);
        const int src = lane == k ? 63 : (up ? lane - 1 : lane);
        gather_all(src, true);
        nA += 1;
        n1 -= 1;
        isA = lane < nA; isU = lane >= 64 - n1;
        moved = reload = true;
      }
      FS_QA(1, q_ev0);
      FS_QT(q_ar0);
      // M4: arrivals are the head of A
      const bool arrived = isA && (x >= end_x);
      const int na = __popcll(__ballot(arrived));
      n_arr = na;
      tot_arr += na;
      if (na > 0) {
        retire(arrived);
        for (int j = 0; j < na; ++j) {
          const int lj = read_lane_i(lab, j) & 63;
          just_arrived |= 1ull << lj;
          if (rows[lj].ctrl == FS_CTRL_RL) arr_rl |= 1ull << lj;
        }
        alive_lab &= ~just_arrived;
        // queue A moves up by the vehicles that left: lane i takes lane i + 1 (DPP: no LDS round trip per register)
        for (int j = 0; j < na; ++j) {
          nA -= 1;
          const bool take = lane < nA;
#define FS_Q_S(reg_) do { const auto t_ = dpp<DPP_WAVE_SHL1>(reg_); reg_ = take ? t_ : reg_; } while (0)
#define FS_Q_SI(reg_) do { const int t_ = dpp_i<DPP_WAVE_SHL1>(reg_); reg_ = take ? t_ : reg_; } while (0)
          FS_Q_S(x); FS_Q_S(v); FS_Q_SI(lab); FS_Q_SI(route); FS_Q_SI(seq); FS_Q_SI(origin); FS_Q_SI(foll); FS_Q_S(foll_h);
          FS_Q_S(prev_v); FS_Q_S(last_acc); FS_Q_S(vmax);
          if (ACT) FS_Q_S(a_me);
          if (NOISE) { FS_Q_S(g0); FS_Q_S(g1); FS_Q_S(g2); FS_Q_S(g3); }
#undef FS_Q_SI
#undef FS_Q_S
        }
        isA = lane < nA;
        moved = reload = true;
#ifdef FS_QDIAG
        dg_n[1] += 1;
#endif
      }
      FS_QA(2, q_ar0);
    }
    // ---- M2 / M3: insertions in InFlows order ---------------------------------------------------------------
    bool inserted = false;
    FS_QT(q_in0);
    if (try_insert) {
      // (M2: the float64 schedule begin + k period <= now = n dt as the integer statement my_due <= n; my_due = the first
      // sub-step index at which my inflow's next vehicle is due, recomputed only when the inflow emits)
      const int n_now = sim_steps - 1;
      unsigned fm = unsigned(__ballot(my_due <= n_now)) & 0xffu;
      FS_QA(8, q_in0);
      FS_QT(q_i2);
      while (fm != 0u) {
        const int f = __ffs(int(fm)) - 1;
        fm &= fm - 1u;
        const int k = read_lane_i(emit_l, f);
        const int typ = read_lane_i(f_typ, f), route_f = read_lane_i(f_route, f);
        const T x_dep = read_lane(f_xdep, f), v_dep = read_lane(f_vdep, f);
        const T two_sqrt = read_lane(f_ts, f), min_gap_f = read_lane(f_gap, f), tau_f = read_lane(f_tau, f);
        // M1: the lowest free slot of the type (a slot freed in this sub-step is not free yet)
        const unsigned long long fb = __ballot(slot_ok && slot_type == typ) & ~alive_lab & ~just_arrived;
        const int slot = fb ? __ffsll((long long)fb) - 1 : 0;
        // M3: the nearest vehicle ahead on the route is the tail of its queue (no queue of its own: the tail of D)
        const int n_d = __popcll(__ballot(isA && x >= merge_x));
        int tl = -1;
        bool tail_in_a = true;
        if (route_f == 0) tl = nA > 0 ? nA - 1 : -1;
        else if (n1 > 0) { tl = 64 - n1; tail_in_a = false; }
        else tl = n_d > 0 ? n_d - 1 : -1;
        const bool has_lead = tl >= 0;
        int tj = has_lead ? tl : 0;
        {
          // vehicles AT the tail's position (a collision state): oracle/opennet.py checks against the lowest slot
          const T xt = read_lane(x, tj);
          const unsigned long long tie = __ballot((tail_in_a ? isA : isU) && x == xt);
          if (has_lead && __popcll(tie) > 1) {
            int best = 64;
            for (unsigned long long u = tie; u; u &= u - 1ull) {
              const int j = __ffsll((long long)u) - 1;
              const int lj = read_lane_i(lab, j);
              if (lj < best) { best = lj; tj = j; }
            }
          }
        }
        const T back_j = read_lane(x, tj) - read_lane(LEN, 0);
        const T v_lead = read_lane(v, tj);
        const T gap = back_j - x_dep;
        const T dq = div_core(v_dep * (v_dep - v_lead), two_sqrt);
        const T need = min_gap_f + tmax(0.0f, v_dep * tau_f + dq);
        const bool ok = (fb != 0ull) && (!has_lead || gap >= need);
        if (ok) {
          const int nl_ = route_f == 0 ? nA : 63 - n1;       // the new tail of the queue
          if (lane == nl_) {
            x = x_dep;
            v = v_dep;
            prev_v = 0.0f;                             // previous_speeds.get(veh_id, 0)
            last_acc = 0.0f;
            route = route_f;
            lab = slot;
            seq = seq_ctr;
            origin = f * (1 << 20) + k;
            foll = -1;
            foll_h = BIGV;
            vmax = rows[slot & 63].sumo_max;
          }
          if (route_f == 0) nA += 1; else n1 += 1;
          alive_lab |= 1ull << slot;
          seq_ctr += 1;
          n_dep += 1;
          tot_dep += 1;
          isA = lane < nA; isU = lane >= 64 - n1;
          inserted = true;
          if (lane == f) emit_l = k + 1;
        }
      }
      FS_QA(9, q_i2);
      FS_QT(q_i3);
      if (inserted) {
        moved = reload = true;
        nz_reload = 1;                                 // the newcomers' draws: the block is evaluated again
        my_due = my_due_of(emit_l);
      }
      // with the counters as the insertions left them: who is still due (watched by the sub-step from now on), and
      // when the next vehicle of the others is
      const bool still_due = my_due <= n_now;
      pend_m = __ballot(still_due) & 0xffull;
      // (the inflows sit in lanes 0 .. 7: the minimum of their indices by three DPP steps)
      int dn_ = still_due ? 0x7fffffff : my_due;
      { const int w_ = dpp_i<DPP_QUAD_XOR1>(dn_); dn_ = w_ < dn_ ? w_ : dn_; }
      { const int w_ = dpp_i<DPP_QUAD_XOR2>(dn_); dn_ = w_ < dn_ ? w_ : dn_; }
      { const int w_ = dpp_i<DPP_ROW_HALF_MIRROR>(dn_); dn_ = w_ < dn_ ? w_ : dn_; }
      due_n = __builtin_amdgcn_readfirstlane(dn_);
      FS_QA(10, q_i3);
    }
    FS_QA(3, q_in0);
#ifdef FS_QDIAG
    dg_n[2] += inserted ? 1 : 0;
    dg_n[0] += 1;
#endif
    FS_QT(q_cl0);
    // the end of D moves when the head of U0 passes the merge point: no vehicle changes lane, the lanes' facts do
    if (structural || inserted) classes(reload);
    FS_QA(4, q_cl0);
    if (ACT) {
      if (moved) {                                     // my vehicle may be another one now: its action column (the
        a_me = act_row[rl_col & 63];                   // step's row waits in LDS: a global load here was an L2 round trip)
        q_fence();
      }
      mHaveRl = mKrl & __ballot(!(a_me != a_me));      // NaN: no action for this vehicle this step
    }
    FS_QA(0, q_ev0);
  };

  asm volatile("" :: "v"(x), "v"(v), "v"(route), "v"(seq), "v"(origin), "v"(foll), "v"(foll_h), "v"(prev_v), "v"(last_acc), "v"(vmax));
#ifdef FS_QDIAG
  const unsigned long long dg_start = __builtin_readcyclecounter();
#endif
  // (the action row of a step is loaded one step ahead: read where it is used, it was an L2 / HBM round trip at the top of
  // every step; the load itself is unconditional -- a select on its value would wait for it at once)
  const int act_lane = lane < s.num_rl ? lane : 0;
  float a_pref = 0.0f;
  if (ACT && num_steps > 0) a_pref = actions[size_t(r) * s.num_rl + act_lane];
  for (int step = 0; step < num_steps; ++step) {
    if (ACT) {                                         // the step's action row, by RL column, in LDS
      act_row[lane] = lane < s.num_rl ? a_pref : 0.0f;
      q_fence();
      a_me = act_row[rl_col & 63];
      q_fence();
      const int nxt = step + 1 < num_steps ? step + 1 : step;
      a_pref = actions[size_t(nxt) * act_stride + size_t(r) * s.num_rl + act_lane];
    }
    if (ACT) mHaveRl = mKrl & __ballot(!(a_me != a_me));          // NaN: no action for this vehicle this step
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      // ---- this sub-step's acceleration noise (S14): one Philox block serves four sub-steps ---------------------
      T g_now = 0.0f;
      if (NOISE) {
        if ((nctr & 3u) == 0u || nz_reload != 0) {       // (wave-uniform: the counter is the replica's)
          FS_QT(q_nz0);
#ifdef FS_QDIAG
          dg_n[3] += 1;
#endif
          T g[4];
          gauss4<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(r), uint32_t(lab & 63), nctr >> 2, g, s.noise_exact != 0);
          g0 = g[0]; g1 = g[1]; g2 = g[2]; g3 = g[3];
          nz_reload = 0;
          FS_QA(5, q_nz0);
        }
        const uint32_t ph = nctr & 3u;
        const T lo_ = (ph & 1u) ? g1 : g0, hi_ = (ph & 1u) ? g3 : g2;
        g_now = (ph & 2u) ? hi_ : lo_;
      }
      // ---- facts of the snapshot that end up in scalar registers, taken first and consumed further down ----------
      // M6: who is in or about to enter the junction; base_controller.py:98-99: who is on a junction-internal edge
      unsigned long long busy_m = 0ull;
      if (jct) {
        const T thr_mb = selm(mCU0, box_in - tgap * v, cBIG);
        const unsigned long long mb_m = ballot_here(x >= thr_mb);      // a route-0 vehicle inside the junction or about to enter
        const unsigned long long mib_m = ballot_here(x >= thr_mib);    // the head of U1 inside the junction
        busy_m = (mb_m != 0ull ? mU : 0ull) | (mib_m != 0ull ? mA : 0ull);   // (a vehicle upstream of the junction is in U1 or U0)
      }
      unsigned long long cmd_m = mKidm;                  // commanded: an IDM vehicle on an edge, an RL vehicle with an action
      if (jm) cmd_m &= ballot_here(!sm_true(sm_in(x, ia_lo, ia_hi) | sm_in(x, ib_lo, ib_hi)));
      if (ACT) cmd_m |= mHaveRl;
      const unsigned long long obey_m = mSm1 | ~cmd_m;
      // ---- controllers on the snapshot (S1): idm_fd / control_accel_fd, branch-free ----------------------------
      T acc;
      {
        const float hh = tabs(h) < 1e-3f ? 1e-3f : h;
        const float dyn = v * p_T + div_core_by(v * (v - vl), p_ts, y_pts);
        const float m = hmax(0.0f, dyn);
        const float s_star = selm(mHas, p_s0 + m, c0);
        const float q = div_core(s_star, hh);
        const float ratio = div_core_by(v, p_v0, y_v0);
        const float r2 = ratio * ratio;
        float a = p_a * (1.0f - r2 * r2 - q * q);
        if (NOISE) a = selm(mNoisy, a + p_sig * g_now, a);           // base_controller.py:109-110
        float ar = c0;
        if (ACT) ar = selm(mHaveRl, hmin(hmax(a_me, clip_lo), clip_hi), c0);
        acc = selm(mKidm, a, ar);
      }
      // ---- M7: apply_acceleration + SUMO integration (sumo_speed_fd, k_steps_open's FD form) ---------------------
      // (common to SUMO's speed behind the leader and towards the stop line)
      const float u_rr = div_core_by(v, u_vmax, y_vmax);
      const float u_r2 = u_rr * u_rr;
      const float u_free = 1.0f - u_r2 * u_r2;
      const float v_tau = v * u_tau;
      auto sumo_speed = [&](T dvv, T h_, bool has_, unsigned long long has_m) -> T {
        const float gap = hmax(h_, 1e-3f);
        const float m = hmax(0.0f, v_tau + div_core_by(dvv, u_ts, y_uts));
        const float ss = u_gap + m;
        const float qq = div_core(ss, gap);
        const float q = has_ ? qq : selm(has_m, qq, c0);
        const float a_s = u_acc * (u_free - q * q);
        return hmax(0.0f, v + a_s * dt);
      };
      const T v_sumo = sumo_speed(v * (v - vl), h, false, mHas);
      const float next_vel = hmax(v + acc * dt, 0.0f);
      float vc = v + (next_vel - v) * ramp;
      vc = hmin(vc, selm(mSm1, v_sumo, cBIG));
      vc = hmin(vc, v + u_adt);
      vc = hmax(vc, v - u_ddt);
      T v_new = selm(cmd_m, vc, v_sumo);
      if (jct) {                                         // M6: right of way at the merge
        const T stop = sumo_speed(v * (v - 0.0f), box_in - x, true, 0ull);
        const T cap_b = selm(busy_m & obey_m, stop, cBIG);
        const T cap = sm_true(sm_in(x, appr_lo, box_in)) ? cap_b : cBIG;
        v_new = hmin(v_new, cap);
      }
      // (every lane moves: a lane without a vehicle holds values nobody reads)
      prev_v = v;
      last_acc = acc;
      x = x + v_new * dt;
      v = v_new;
      tcount += 1;
      nctr += 1u;
      sim_steps += 1;
      // ---- did anything happen?  One test; the snapshot is taken as if not, and taken again if so -----------------
      const T x_up = dpp<DPP_WAVE_SHR1>(x), x_dn = dpp<DPP_WAVE_SHL1>(x);           // lane - 1 / lane + 1
      const T xa = selm(mA1, x_up, selm(mU1, x_dn, cBIG));
      const unsigned quiet_b = fbits(x - xa) & fbits(x - ev_thr);                  // sign set: behind the previous lane, before my threshold
      const unsigned long long ev_m = ballot_here(int(quiet_b) >= 0);
      // M3: does a waiting inflow fit now?  Lane f tests inflow f against the tail of its route's queue
      unsigned long long fit_m = 0ull;
      if (pend_m != 0ull) {
        const int ja = tlA < 0 ? 0 : tlA, ju = tlU < 0 ? 0 : tlU;
        const T xt = f_route == 0 ? read_lane(x, ja) : read_lane(x, ju);
        const T vt = f_route == 0 ? read_lane(v, ja) : read_lane(v, ju);
        const bool none = f_route == 0 ? tlA < 0 : tlU < 0;
        const T gap = (xt - LEN) - f_xdep;
        const T need = f_gap + hmax(0.0f, f_vdep * f_tau + div_core(f_vdep * (f_vdep - vt), f_ts));
        fit_m = ballot_here(none || gap >= need) & pend_m;
      }
      T h_n, vl_n, foll_h_n;
      int foll_n;
      snapshot(x_up, x_dn, h_n, vl_n, foll_n, foll_h_n);
      const bool due_now = sim_steps - 1 >= due_n;
      if (ev_m != 0ull || due_now || fit_m != 0ull) {
        // (a join or an arrival can make room for a waiting inflow in this very sub-step: the tails the test read are old)
        events(step, ev_m != 0ull, due_now || fit_m != 0ull || (ev_m != 0ull && pend_m != 0ull));
        snapshot(dpp<DPP_WAVE_SHR1>(x), dpp<DPP_WAVE_SHL1>(x), h_n, vl_n, foll_n, foll_h_n);
      } else {
        n_arr = 0;
        n_dep = 0;
        arr_rl = 0ull;
      }
      h = h_n;
      vl = vl_n;
      foll = foll_n;
      foll_h = foll_h_n;
    }

    // ---- get_state / compute_reward / done ------------------------------------------------------------------
    const bool emit = obs_every_step || (step == num_steps - 1);
    FS_QT(q_hd0);
    if (emit) {
      const bool isA = lane < nA, isU = lane >= 64 - n1, alive = isA | isU, uh = isU && lane == 63;
      const bool has = ((mHas >> lane) & 1ull) != 0ull, k_rl = ((mKrl >> lane) & 1ull) != 0ull;
      // the five features of my vehicle (flow/envs/multiagent/merge.py:108-140)
      // Flow's coordinate of x (O5, route_lookup's arithmetic): both routes' tables are launch constants in registers
      // -- five compares and selects each, no table walk through LDS (that walk was half of the head's time)
      T fx;
      {
        T fxr[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          T st = sg_st[rt][0], fs0 = sg_fs[rt][0], sl = sg_sl[rt][0];
#pragma unroll
          for (int q = 1; q < 6; ++q) {
            const bool hit = x >= sg_st[rt][q];
            st = hit ? sg_st[rt][q] : st;
            fs0 = hit ? sg_fs[rt][q] : fs0;
            sl = hit ? sg_sl[rt][q] : sl;
          }
          fxr[rt] = fs0 + sl * (x - st);
        }
        fx = route == 1 ? fxr[1] : fxr[0];
      }
      const T fx_up = dpp<DPP_WAVE_SHR1>(fx), fx_dn = dpp<DPP_WAVE_SHL1>(fx);
      const T fx_t = read_lane(fx, td);
      T fx_l = isA ? fx_up : fx_dn;
      fx_l = uh ? fx_t : fx_l;
      // ONE exchange hands everything the head needs by SLOT to the slots' lanes (the follower is a slot: which lane
      // holds it now; the reward's sums run in slot order): four scatters, one wait, the reads, one wait -- a free slot's
      // entry is stale and masked by `alive_lab` (the first version cleared and waited per value: fourteen LDS round trips)
      const T dvt = v - s.target_velocity;
      const bool use = alive && k_rl && has && (v > 0.0f);
      const T t_headway = tmax(h / (use ? v : 1.0f), 0.0f);
      const T term = tmin((t_headway - 1.0f) / 1.0f, 0.0f);
      if (alive) {
        scr_i[lab & 63] = lane;
        scr_f[lab & 63] = s.evaluate ? v : dvt * dvt;
        scr_g[lab & 63] = term;
        scr_j[lab & 63] = use ? 1 : 0;
      }
      q_fence();
      const bool slot_alive = ((alive_lab >> lane) & 1ull) != 0ull;
      const int fo = alive ? foll : -1;
      const int fl_ = scr_i[fo & 63];
      const T sum_s_ = scr_f[lane], term_s_ = scr_g[lane];
      const int use_s_ = scr_j[lane];
      q_fence();
      FS_QA(11, q_hd0);
      FS_QT(q_h2);
      const int fl = (fo >= 0 && ((alive_lab >> (fo & 63)) & 1ull) != 0ull) ? fl_ : -1;
      const T sum_s = slot_alive ? sum_s_ : 0.0f, term_s = slot_alive ? term_s_ : 0.0f;
      const int use_s = slot_alive ? use_s_ : 0;
      T v_f = bperm(v, fl >= 0 ? fl : lane), h_f = bperm(h, fl >= 0 ? fl : lane);
      if (fo >= 0 && fl < 0) {                          // a recorded follower that has left: its slot's values of record
        v_f = s.vel[base + size_t(fo)];
        h_f = 1000.0f;
      }
      const bool hl = alive && has;
      const T this_speed = alive ? v : -1001.0f;
      const T lead_speed = hl ? vl : s.max_speed;
      const T lead_head = hl ? fx_l - fx - LEN : o.net_length;
      const T follow_speed = fo >= 0 ? v_f : 0.0f;
      const T follow_head = fo >= 0 ? h_f : o.net_length;
      T f5[5];                                                        // (launch-constant divisors: the exact float64 route)
      f5[0] = div_via_f64(this_speed, ms64, rc_ms64);
      f5[1] = div_via_f64(lead_speed - this_speed, ms64, rc_ms64);
      f5[2] = div_via_f64(lead_head, nl64, rc_nl64);
      f5[3] = div_via_f64(this_speed - follow_speed, ms64, rc_ms64);
      f5[4] = div_via_f64(follow_head, nl64, rc_nl64);
      if (alive && k_rl) {
        const int col = rows[lab & 63].rl_index;
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * col + q] = f5[q];
      }
      if (slot_is_rl && !((alive_lab >> lane) & 1ull)) {              // (slot view) an RL slot without a vehicle
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * slot_rl_index + q] = 0.0f;
      }
      FS_QA(12, q_h2);
      FS_QT(q_h3);
      // reward (flow/envs/multiagent/merge.py:142-171 over rewards.desired_velocity), sums in SLOT order
      const int n_alive = nA + n1;
      T reward;
      if (s.evaluate) {
        const T sum_v = seg_sum<64>(sum_s);
        reward = n_alive > 0 ? sum_v / T(n_alive) : 0.0f;
      } else {
        const T mc_lane = tb.template t_gather<TAB_MAX_COST>(n_alive & 63);
        const T max_cost = n_alive < 64 ? mc_lane : o.max_cost_full;
        const T cost = tsqrt(seg_sum<64>(sum_s));
        T cost1 = tmax(max_cost - cost, 0.0f) / (max_cost + 1.1920928955078125e-07f);
        const bool bad = (__ballot(alive && (v < -100.0f)) != 0ull) || n_alive == 0;
        cost1 = bad ? 0.0f : cost1;
        T cost2 = 0.0f;
        for (unsigned long long u = __ballot(use_s != 0); u; u &= u - 1ull) cost2 = cost2 + read_lane(term_s, __ffsll((long long)u) - 1);
        reward = tmax(cost1 + 0.1f * cost2, 0.0f);
      }
      FS_QA(13, q_h3);
      if (lane == 0) {
        *rrow = reward;
        *drow = done_flag(tcount >= s.step_limit, false);          // multiagent/base.py:188-190: crash = 0
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
    FS_QA(6, q_hd0);
  }
#ifdef FS_QDIAG
  dg_t[7] = __builtin_readcyclecounter() - dg_start;
  if (blockIdx.x == 7 && lane == 0)
    printf("QDIAG total %llu events %llu (n %d) resort+join %llu arrivals %llu (n %d) insert %llu (n %d) classes %llu noise %llu (n %d) head %llu\n",
           dg_t[7], dg_t[0], dg_n[0], dg_t[1], dg_t[2], dg_n[1], dg_t[3], dg_n[2], dg_t[4], dg_t[5], dg_n[3], dg_t[6]);
  if (blockIdx.x == 7 && lane == 0)
    printf("QDIAG2 insert: schedule %llu loop %llu after %llu | head: exchange %llu features %llu reward %llu\n",
           dg_t[8], dg_t[9], dg_t[10], dg_t[11], dg_t[12], dg_t[13]);
#endif

  // ---- the state back to its slots -----------------------------------------------------------------------------
  __threadfence();
  {
    const bool isA = lane < nA, isU = lane >= 64 - n1, alive = isA | isU, uh = isU && lane == 63;
    const bool has = ((mHas >> lane) & 1ull) != 0ull;
    const int lab_up = dpp_i<DPP_WAVE_SHR1>(lab), lab_dn = dpp_i<DPP_WAVE_SHL1>(lab);
    const int lab_t = read_lane_i(lab, td);
    int lab_l = isA ? lab_up : lab_dn;
    lab_l = uh ? lab_t : lab_l;
    if (alive) {
      const size_t e = base + size_t(lab & 63);
      if (s.st16 != nullptr) state16_store(s, e, x, v);
      else { s.pos[e] = x; s.vel[e] = v; }
      s.lane[e] = route;
      s.prev_vel[e] = prev_v;
      s.accel[e] = last_acc;
      o.seq[e] = seq;
      o.origin[e] = origin;
      o.foll[e] = foll;
      o.foll_h[e] = foll_h;
      o.vmax[e] = vmax;
      o.lead[e] = has ? lab_l : -1;
      o.headway[e] = h;
    }
    if (slot_ok) o.arrived_rl[base + lane] = int((arr_rl >> lane) & 1ull);
    if (lane == 0) {
      s.time[r] = tcount;
      s.noise_ctr[r] = nctr;
      cnt[CNT_SIM_STEPS] = sim_steps;
      cnt[CNT_SEQ] = seq_ctr;
      cnt[CNT_ARRIVED] = n_arr;
      cnt[CNT_DEPARTED] = n_dep;
      cnt[CNT_TOTAL_ARRIVED] = tot_arr;
      cnt[CNT_TOTAL_DEPARTED] = tot_dep;
    }
    if (lane < FS_MAX_INFLOWS) o.emitted[size_t(r) * FS_MAX_INFLOWS + lane] = emit_l;
  }
}

}  // namespace fs
