// flowsim_queue.h -- gfx950 rollout kernel of the two-route merge (FS_NET_MERGE) in QUEUE order.
//
// k_steps_open (flowsim_open.h) keeps vehicle SLOT i in lane i and finds every vehicle's neighbours again whenever the
// order of the vehicles may have changed (rank by position, masks in rank order).  On a merge without lane changes a
// route is a queue -- vehicles enter upstream, leave downstream, nobody overtakes (flow/networks/merge.py: one lane per
// edge) -- so this kernel keeps the vehicles of a replica in the lanes of ONE wave in driving order:
//   lanes 0 .. nA-1        queue A, head first: every vehicle beyond the merge point (D, a prefix) followed by the
//                          vehicles of route 0 (highway) that are still upstream of it (U0)
//   lanes 63 .. 64-n1      queue U1, head in lane 63: the vehicles of route 1 (ramp) upstream of the merge point
// The leader (M5, flow/core/kernel/vehicle/traci.py:219-242) of a vehicle is the previous lane of its queue -- one DPP
// move -- and the leader of U1's head is the last vehicle of D; the vehicles whose leader I am (the follower candidates
// of the sticky rule O1, vehicle/traci.py:243-250) are the next lane of my queue and, for the last vehicle of D, the
// head of U1.  The slot a vehicle occupies in the state arrays (M1: lowest free slot of its type) is a LABEL it carries;
// everything the reference orders by id (reward sums, observation columns) is handed to the label's lane through LDS
// once per env step.  Events move vehicles between lanes (wave-uniform branches, one ds_bpermute per register):
//   arrival (M4)       the head of A leaves, A moves down one lane;
//   join               the head of U1 has passed the merge point: it enters A at the place its position gives;
//   insertion (M2/M3)  the new vehicle becomes the tail of its queue (checked against the old tail, M3);
//   collision / tie    a vehicle is no longer strictly behind the previous lane: both queues are re-sorted by
//                      (x descending, equal x: lower slot first), the order oracle/opennet.py states.
// oracle/queuenet.py restates this bookkeeping on the CPU and proves it equal to the all-pairs rules of
// oracle/opennet.py at every sub-step; the arithmetic is k_steps_open<float, ., 2, CSET = 1>'s, operation for operation
// (control_accel_fd, sumo_speed_fd, the same Philox draws), so the two kernels are bit-identical
// (tests/test_queue_gpu.py).  Scope (Sim::queue_ok): float32, IDM / RL / Sim-car-following slots, the
// MultiAgentMergePOEnv head (FS_ENV_MERGE_MA: C5 of BASELINE.json), scheduled inflows, no replica mask, one or more
// steps per launch; everything else steps on k_steps_open.
#pragma once

namespace fs {

// LDS traffic of ONE wave: the hardware executes a wave's DS instructions in order, so all that is needed between a
// lane's write and another lane's read is that the compiler keeps the program order
__device__ __forceinline__ void q_fence() { asm volatile("" ::: "memory"); }

// per-slot parameters, one row per label in LDS (read again by a lane whenever its vehicle changes)
struct alignas(16) QueueRow {
  float p0, p1, p2, p4;            // IDM v0, T, a, delta
  float p5, noise, max_accel, length;   // IDM s0, sigma, SUMO accel, vehicle length
  float tau, min_gap, ts_idm, ts_sumo;  // SUMO tau / minGap, 2 sqrt(a b) of the controller and of SUMO's model
  float adt, ddt, sumo_max, delay;      // speed-mode clamps (3e38: off), the vType maxSpeed, fail-safe delay
  int ctrl, failsafe, speed_mode, rl_index;
};

template <bool MA>
__global__ __launch_bounds__(64) void k_merge_queue(DevView<float> s, OpenView<float> o, int num_steps,
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
  const int flags = s.flags & ~(FLAG_NEED_FOLLOWER | FLAG_NEED_MEAN | FLAG_HAS_LAC);

  __shared__ OpenTabsLds<T> tabs_mem;
  __shared__ QueueRow rows[64];
  __shared__ float scr_f[64];
  __shared__ int scr_i[64];
  OpenTabs<T, true> tb;
  tb.load(o, lane, false, &tabs_mem);
  RouteCursor<T, OpenTabs<T, true>> cur;

  // ---- the slot tables as LDS rows --------------------------------------------------------------------------
  {
    QueueRow q;
    q.p0 = s.p[0 * N + li]; q.p1 = s.p[1 * N + li]; q.p2 = s.p[2 * N + li]; q.p4 = s.p[4 * N + li];
    q.p5 = s.p[5 * N + li]; q.noise = s.noise[li]; q.max_accel = s.max_accel[li]; q.length = s.length[li];
    q.tau = s.sumo_tau[li]; q.min_gap = s.sumo_min_gap[li];
    const float p3 = s.p[3 * N + li], max_decel = s.max_decel[li];
    q.ts_idm = 2.0f * tsqrt(q.p2 * p3);
    q.ts_sumo = 2.0f * tsqrt(q.max_accel * max_decel);
    q.speed_mode = s.speed_mode[li];
    q.adt = (q.speed_mode & 2) ? q.max_accel * s.dt : 3.0e38f;
    q.ddt = (q.speed_mode & 4) ? max_decel * s.dt : 3.0e38f;
    q.sumo_max = s.sumo_max_speed[li];
    q.delay = s.delay[li];
    q.ctrl = s.ctrl[li]; q.failsafe = s.failsafe[li]; q.rl_index = s.rl_index[li];
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
      tot_dep = cnt[CNT_TOTAL_DEPARTED], tot_drop = cnt[CNT_TOTAL_DROPPED];
  int emit_l = (lane < FS_MAX_INFLOWS) ? o.emitted[size_t(r) * FS_MAX_INFLOWS + lane] : 0;
  const bool my_flow = lane < o.n_inflows;             // lane f keeps inflow f's schedule (M2)
  const double my_per = my_flow ? o.flow_tab_d[lane] : 0.0;
  const double my_begin = my_flow ? o.flow_tab_d[64 + lane] : 0.0, my_end = my_flow ? o.flow_tab_d[128 + lane] : 0.0;
  const int my_number = my_flow ? o.flow_tab_i[128 + lane] : 0;
  double next_due = -1.0e300;

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
  float a_me = 0.0f;                                   // this step's action of my vehicle (ma_apply_actions)
  NoiseBlock<T> nzb;
  nzb.init();
  int nA = 0, n1 = 0, nD = 0;
  unsigned long long alive_lab = __ballot(route >= 0);             // bit i: slot i holds a vehicle
  unsigned long long arr_rl = 0ull;                                 // RL slots that arrived in the last sub-step
  {
    const int a0 = o.arrived_rl[base + li];
    arr_rl = __ballot(slot_ok && a0 != 0);
  }

  // my vehicle's parameters (by label) -- reloaded after every event that moves vehicles between lanes
  Slot<float> sl;
  FdSlot fd;
  float fd_adt, fd_ddt;
  sl.pis_index = -1;
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) sl.p[k] = 0.0f;
  auto load_params = [&]() {
    const QueueRow q = rows[lab & 63];
    sl.p[0] = q.p0; sl.p[1] = q.p1; sl.p[2] = q.p2; sl.p[4] = q.p4; sl.p[5] = q.p5;
    sl.noise = q.noise; sl.max_accel = q.max_accel; sl.length = q.length; sl.sumo_tau = q.tau; sl.sumo_min_gap = q.min_gap;
    sl.sumo_max_speed = q.sumo_max; sl.delay = q.delay; sl.max_decel = 0.0f;
    sl.ctrl = q.ctrl; sl.failsafe = q.failsafe; sl.speed_mode = q.speed_mode; sl.rl_index = q.rl_index;
    fd.ts_idm = q.ts_idm; fd.ts_sumo = q.ts_sumo;
    fd_adt = q.adt; fd_ddt = q.ddt;
  };
  auto seg_route = [&]() -> int { return route < 0 ? 0 : route; };

  // every register that belongs to the vehicle, pulled from lane `src` where `take`
  auto gather_all = [&](int src, bool take) {
#define FS_Q_G(reg_) do { const auto t_ = bperm(reg_, src); reg_ = take ? t_ : reg_; } while (0)
    FS_Q_G(x); FS_Q_G(v); FS_Q_G(lab); FS_Q_G(route); FS_Q_G(seq); FS_Q_G(origin); FS_Q_G(foll); FS_Q_G(foll_h);
    FS_Q_G(prev_v); FS_Q_G(last_acc); FS_Q_G(vmax); FS_Q_G(cur.k); FS_Q_G(a_me);
    FS_Q_G(nzb.g[0]); FS_Q_G(nzb.g[1]); FS_Q_G(nzb.g[2]); FS_Q_G(nzb.g[3]);
#undef FS_Q_G
  };
  auto after_move = [&]() {             // the lanes hold other vehicles now: their parameters and segment rows
    const bool al = (lane < nA) || (lane >= 64 - n1);
    if (!al) { route = -1; lab = lane; cur.k = 0; }
    load_params();
    cur.k &= 15;
    cur.refresh(o, tb, seg_route());
  };

  // ---- (re)build the queues from whatever the lanes hold: A and U1 sorted by (x descending, lower slot first) -
  auto resort = [&]() {
    const bool al = route >= 0;
    const bool inA = al && (route == 0 || x >= o.merge_x);
    const bool inU = al && !inA;
    const unsigned long long mA = __ballot(inA), mU = __ballot(inU), mAl = mA | mU;
    int cA = 0, cU = 0;
    for (unsigned long long u = mAl; u; u &= u - 1ull) {
      const int j = __ffsll((long long)u) - 1;
      const T xj = read_lane(x, j);
      const int lj = read_lane_i(lab, j);
      const int ahead = int(xj > x) | (int(xj == x) & int(lj < lab));
      const bool jA = (mA >> j) & 1ull;               // (wave-uniform)
      cA += jA ? ahead : 0;
      cU += jA ? 0 : ahead;
    }
    nA = __popcll(mA);
    n1 = __popcll(mU);
    const int dead_rank = nA + __popcll(~mAl & ((1ull << lane) - 1ull));
    const int target = inA ? cA : (inU ? 63 - cU : dead_rank);
    // the inverse permutation: lane t learns which lane holds the vehicle that belongs to it
    const int src = __builtin_amdgcn_ds_permute(target << 2, lane);
    gather_all(src, true);
    after_move();
  };

  cur.k = 0;
  resort();
  cur.restart(o, tb, seg_route(), x);

  // ---- M5 / O1 from the structure ---------------------------------------------------------------------------
  int lead = -1;
  T vl = -1001.0f, h = 1000.0f;
  bool has = false;
  auto neighbours = [&](bool live, bool follow) {
    const bool isA = lane < nA, isU = lane >= 64 - n1, alive = isA | isU;
    nD = __popcll(__ballot(isA && x >= o.merge_x));
    const T len = sl.length;
    const T x_up = dpp<DPP_WAVE_SHR1>(x), x_dn = dpp<DPP_WAVE_SHL1>(x);           // lane - 1 / lane + 1
    const T v_up = dpp<DPP_WAVE_SHR1>(v), v_dn = dpp<DPP_WAVE_SHL1>(v);
    const T len_up = dpp<DPP_WAVE_SHR1>(len), len_dn = dpp<DPP_WAVE_SHL1>(len);
    const int lab_up = dpp_i<DPP_WAVE_SHR1>(lab), lab_dn = dpp_i<DPP_WAVE_SHL1>(lab);
    const int seq_up = dpp_i<DPP_WAVE_SHR1>(seq), seq_dn = dpp_i<DPP_WAVE_SHL1>(seq);
    // the last vehicle of D (leader of U1's head) and U1's head (follower candidate of that vehicle)
    const int td = nD > 0 ? nD - 1 : 0;
    const T x_t = read_lane(x, td), v_t = read_lane(v, td), len_t = read_lane(len, td);
    const int lab_t = read_lane_i(lab, td);
    const T x_h = read_lane(x, 63);
    const int lab_h = read_lane_i(lab, 63), seq_h = read_lane_i(seq, 63);
    const bool u_head = lane == 63;
    T x_l = isA ? x_up : x_dn, v_l = isA ? v_up : v_dn, len_l = isA ? len_up : len_dn;
    int lab_l = isA ? lab_up : lab_dn;
    x_l = u_head ? x_t : x_l; v_l = u_head ? v_t : v_l; len_l = u_head ? len_t : len_l; lab_l = u_head ? lab_t : lab_l;
    has = isA ? lane > 0 : (isU && (lane < 63 || nD > 0));
    vl = has ? v_l : -1001.0f;                          // get_speed(None): the accessor's error value
    h = has ? (x_l - x) - len_l : 1000.0f;              // vehicle/traci.py:237
    lead = has ? lab_l : -1;
    if (!follow) return;
    // O1: the vehicles whose leader I am: the next lane of my queue, and U1's head if I am the last vehicle of D
    const bool c1 = isA ? (lane + 1 < nA) : (isU && lane - 1 >= 64 - n1);
    const T xc1 = isA ? x_dn : x_up;
    const int sc1 = isA ? seq_dn : seq_up, lc1 = isA ? lab_dn : lab_up;
    const bool c2 = isA && (nD > 0) && (lane == nD - 1) && (n1 > 0);
    const T ch1 = (x - xc1) - len, ch2 = (x - x_h) - len;
    const bool e1 = c1 && (has || sc1 > seq), e2 = c2 && (has || seq_h > seq);
    T bestf = BIGV;
    int bseq = 0x7fffffff, bj = -1;
    if (e1) { bestf = ch1; bseq = sc1; bj = lc1; }
    if (e2 && (ch2 < bestf || (ch2 == bestf && seq_h < bseq))) { bestf = ch2; bseq = seq_h; bj = lab_h; }
    const bool no_lead = alive && !has;
    const T start_h = no_lead ? 1000.0f : foll_h;
    const int start_f = no_lead ? -1 : foll;
    const bool better = (bestf < start_h) && (bestf < BIGV);
    if (alive && live) {
      foll = better ? bj : start_f;
      foll_h = better ? bestf : start_h;
    }
  };
  neighbours(false, false);

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

  const T dt = s.dt;
  const int obs_dim = o.obs_dim;
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(r) * obs_dim;
  float* rrow = rew + r;
  uint8_t* drow = done + r;
  const bool use_act = actions != nullptr && o.ma_apply_actions != 0;

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

  asm volatile("" :: "v"(x), "v"(v), "v"(route), "v"(seq), "v"(origin), "v"(foll), "v"(foll_h), "v"(prev_v), "v"(last_acc), "v"(vmax));
  for (int step = 0; step < num_steps; ++step) {
    if (use_act) {                                     // the step's action row, by RL column, in LDS
      const float* act = actions + size_t(step) * act_stride + size_t(r) * s.num_rl;
      scr_f[lane] = lane < s.num_rl ? act[lane] : 0.0f;
      q_fence();
      a_me = scr_f[(sl.rl_index < 0 ? 0 : sl.rl_index) & 63];
      q_fence();
    }
    bool crashed = false;
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      const bool live = !crashed;
      bool isA = lane < nA, isU = lane >= 64 - n1, alive = isA | isU;
      // ---- controllers on the snapshot (S1) ----------------------------------------------------------------
      const bool internal = cur.internal(o, seg_route());
      const bool on_edge = s.junction_mode ? !internal : true;
      const bool is_rl = sl.ctrl == FS_CTRL_RL;
      bool have_rl = false;
      T a_rl = 0.0f;
      if (use_act) {
        const bool cand = is_rl && alive;
        have_rl = cand && !(a_me != a_me);             // NaN: no action for this vehicle this step
        a_rl = have_rl ? a_me : 0.0f;
      }
      bool commanded = false;
      T g_now = 0.0f;
      if (flags & FLAG_HAS_NOISE) g_now = nzb.draw(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(r), uint32_t(lab & 63), nctr);
      const T acc = control_accel_fd(s, sl, fd, flags, v, vl, h, has, on_edge, have_rl, a_rl, commanded, g_now);
      // ---- M7: apply_acceleration + SUMO integration (k_steps_open's FD form) --------------------------------
      const T vmax_eff = tmin(vmax, o.speed_limit);   // M10
      const float next_vel = hmax(v + acc * dt, 0.0f);
      float vc = v + (next_vel - v) * s.ramp;
      const T v_sumo = sumo_speed_fd(v, vl, h, has, dt, sl, vmax_eff, fd.ts_sumo);
      vc = hmin(vc, (sl.speed_mode & 1) ? v_sumo : 3.0e38f);
      vc = hmin(vc, v + fd_adt);
      vc = hmax(vc, v - fd_ddt);
      T v_new = commanded ? vc : v_sumo;
      if (s.junction_on) {                             // M6: right of way at the merge
        const bool in_reach = alive && (x < o.merge_x);
        const bool major_busy = __ballot(in_reach && route == 0 && (x >= o.box_in - s.j_time_gap * v)) != 0ull;
        const bool minor_in_box = __ballot(in_reach && route == 1 && (x >= o.box_in)) != 0ull;
        const bool approaching = alive && (x >= o.box_in - s.j_lookahead) && (x < o.box_in);
        const bool yields = approaching && ((route == 1 && major_busy) || (route == 0 && minor_in_box));
        const T stop = sumo_speed_fd(v, 0.0f, o.box_in - x, true, dt, sl, vmax_eff, fd.ts_sumo);
        const T cap = yields ? stop : BIGV;
        v_new = hmin(v_new, ((sl.speed_mode & 1) || !commanded) ? cap : BIGV);
      }
      const T x_new = (s.integrator == FS_BALLISTIC) ? x + (v + v_new) / 2.0f * dt : x + v_new * dt;
      const bool mv = live && alive;
      prev_v = mv ? v : prev_v;
      last_acc = mv ? acc : last_acc;
      x = mv ? x_new : x;
      v = mv ? v_new : v;
      if (mv) cur.follow(o, tb, seg_route(), x);
      if (live) {
        tcount += 1;
        nctr += 1u;
        sim_steps += 1;
      }
      // ---- events: the order of the lanes ---------------------------------------------------------------------
      bool moved = false;
      if (live) {
        const T x_up = dpp<DPP_WAVE_SHR1>(x), x_dn = dpp<DPP_WAVE_SHL1>(x);
        const bool bad = (isA && lane > 0 && !(x < x_up)) || (isU && lane < 63 && !(x < x_dn));
        if (__ballot(bad) != 0ull) {                   // a vehicle caught up with the previous lane: re-sort
          resort();
          moved = true;
          isA = lane < nA; isU = lane >= 64 - n1; alive = isA | isU;
        }
        // the head of U1 has passed the merge point: it joins A at the place its position gives
        while (n1 > 0 && read_lane(x, 63) >= o.merge_x) {
          const T xe = read_lane(x, 63);
          const int le = read_lane_i(lab, 63);
          const int k = __popcll(__ballot(isA && ((x > xe) || (x == xe && lab < le))));
          const bool up = (lane > k && lane <= nA) || (lane > 64 - n1 && lane <= 63);
          const int src = lane == k ? 63 : (up ? lane - 1 : lane);
          gather_all(src, true);
          nA += 1;
          n1 -= 1;
          moved = true;
          isA = lane < nA; isU = lane >= 64 - n1; alive = isA | isU;
        }
        // M4: arrivals are the head of A
        const bool arrived = isA && (x >= o.end_x);
        const int na = __popcll(__ballot(arrived));
        arr_rl = 0ull;
        unsigned long long just_arrived = 0ull;      // by SLOT: a slot freed now is free from the next sub-step on
        n_arr = na;
        n_dep = 0;
        tot_arr += na;
        if (na > 0) {
          retire(arrived);
          for (int j = 0; j < na; ++j) {
            const int lj = read_lane_i(lab, j) & 63;
            just_arrived |= 1ull << lj;
            if (read_lane_i(route, j) >= 0 && rows[lj].ctrl == FS_CTRL_RL) arr_rl |= 1ull << lj;
          }
          alive_lab &= ~just_arrived;
          nA -= na;
          gather_all(lane + na, lane < nA);
          moved = true;
          isA = lane < nA; alive = isA | isU;
        }
        if (moved) after_move();
        // ---- M2 / M3: insertions in InFlows order ---------------------------------------------------------------
        const double now = double(sim_steps - 1) * o.dt_d;
        if (next_due <= now) {
          auto schedule = [&](double& t_mine) -> bool {
            const int k_me = emit_l;
            const double due_t = my_begin + double(k_me) * my_per;
            const bool open_me = my_flow && (due_t <= my_end) && (my_number < 0 || k_me < my_number);
            t_mine = open_me ? due_t : 1.0e300;
            return open_me && (due_t <= now);
          };
          double t_mine;
          const bool due_me = schedule(t_mine);
          unsigned fm = unsigned(__ballot(due_me)) & 0xffu;
          bool fresh = false;                          // this lane received a vehicle in this sub-step
          bool inserted = false;
          while (fm != 0u) {
            const int f = __ffs(int(fm)) - 1;
            fm &= fm - 1u;
            const int k = read_lane_i(emit_l, f);
            const int typ = tb.template fi<0>(f);
            const int route_f = tb.template fi<1>(f);
            const T x_dep = tb.template t<TAB_FL_XDEP>(f), v_dep = tb.template t<TAB_FL_VDEP>(f);
            const T two_sqrt = tb.template t<TAB_FL_TWOSQRT>(f), min_gap_f = tb.template t<TAB_FL_MINGAP>(f),
                    tau_f = tb.template t<TAB_FL_TAU>(f);
            // M1: the lowest free slot of the type (a slot freed in this sub-step is not free yet)
            const unsigned long long fb = __ballot(slot_ok && slot_type == typ) & ~alive_lab & ~just_arrived;
            const int slot = fb ? __ffsll((long long)fb) - 1 : 0;
            // M3: the nearest vehicle ahead on the route is the tail of its queue (no queue of its own: the tail of D)
            const int n_d = __popcll(__ballot(isA && x >= o.merge_x));
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
            const T back_j = read_lane(x, tj) - read_lane(sl.length, tj);
            const T v_lead = read_lane(v, tj);
            const T gap = back_j - x_dep;
            const T dq = div_core(v_dep * (v_dep - v_lead), two_sqrt);
            const T need = min_gap_f + tmax(0.0f, v_dep * tau_f + dq);
            const bool ok = (fb != 0ull) && (!has_lead || gap >= need);
            if (ok) {
              const int nl = route_f == 0 ? nA : 63 - n1;       // the new tail of the queue
              if (lane == nl) {
                x = x_dep;
                v = v_dep;
                prev_v = 0.0f;                         // previous_speeds.get(veh_id, 0)
                last_acc = 0.0f;
                route = route_f;
                lab = slot;
                seq = seq_ctr;
                origin = f * (1 << 20) + k;
                foll = -1;
                foll_h = BIGV;
                fresh = true;
              }
              if (route_f == 0) nA += 1; else n1 += 1;
              alive_lab |= 1ull << slot;
              seq_ctr += 1;
              n_dep += 1;
              tot_dep += 1;
              isA = lane < nA; isU = lane >= 64 - n1; alive = isA | isU;
              inserted = true;
              if (lane == f) emit_l = k + 1;
            }
          }
          if (inserted) {
            load_params();
            if (fresh) {
              vmax = sl.sumo_max_speed;
              cur.restart(o, tb, seg_route(), x);
            }
            nzb.loaded = false;                        // the newcomers' draws: the block is evaluated again
            moved = true;
          }
          double t_after;
          schedule(t_after);                           // with the counters as the insertions left them
          next_due = seg_min<64>(t_after);
        }
        if (use_act && moved) {                        // my vehicle may be another one now: its action column
          const float* act = actions + size_t(step) * act_stride + size_t(r) * s.num_rl;
          const int col = sl.rl_index < 0 ? 0 : sl.rl_index;
          a_me = act[col < s.num_rl ? col : 0];
        }
      }
      // ---- O1: new neighbour snapshot, sticky followers, collision check --------------------------------------
      neighbours(live, true);
      if (!MA) {
        bool c = __ballot((lane < nA || lane >= 64 - n1) && has && (h < s.crash_gap)) != 0ull;
        if (s.junction_on) {
          const bool inside = (lane < nA || lane >= 64 - n1) && (x >= o.box_in) && (x < o.merge_x);
          c = c || ((__ballot(inside && route == 0) != 0ull) && (__ballot(inside && route == 1) != 0ull));
        }
        crashed = crashed || (c && live);
      }
    }

    // ---- get_state / compute_reward / done ------------------------------------------------------------------
    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      const bool isA = lane < nA, isU = lane >= 64 - n1, alive = isA | isU;
      const bool is_rl = sl.ctrl == FS_CTRL_RL;
      // the five features of my vehicle (flow/envs/multiagent/merge.py:108-140)
      const T fx = cur.flow_x(x);
      const T fx_up = dpp<DPP_WAVE_SHR1>(fx), fx_dn = dpp<DPP_WAVE_SHL1>(fx);
      const T fx_t = read_lane(fx, nD > 0 ? nD - 1 : 0);
      T fx_l = isA ? fx_up : fx_dn;
      fx_l = lane == 63 ? fx_t : fx_l;
      // the follower is a SLOT: which lane holds it now
      const int lane_of = to_slots_i(lane, alive, -1);               // (slot view: lane of slot i, -1 if free)
      scr_i[lane] = lane_of;
      q_fence();
      const int fo = alive ? foll : -1;
      const int fl = fo >= 0 ? scr_i[fo & 63] : -1;
      q_fence();
      T v_f = bperm(v, fl >= 0 ? fl : lane), h_f = bperm(h, fl >= 0 ? fl : lane);
      if (fo >= 0 && fl < 0) {                          // a recorded follower that has left: its slot's values of record
        v_f = s.vel[base + size_t(fo)];
        h_f = 1000.0f;
      }
      const T this_speed = alive ? v : -1001.0f;
      const T lead_speed = (alive && has) ? vl : s.max_speed;
      const T lead_head = (alive && has) ? fx_l - fx - sl.length : o.net_length;
      const T follow_speed = fo >= 0 ? v_f : 0.0f;
      const T follow_head = fo >= 0 ? h_f : o.net_length;
      T f5[5];
      f5[0] = this_speed / s.max_speed;
      f5[1] = (lead_speed - this_speed) / s.max_speed;
      f5[2] = lead_head / o.net_length;
      f5[3] = (this_speed - follow_speed) / s.max_speed;
      f5[4] = follow_head / o.net_length;
      if (alive && is_rl) {
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * sl.rl_index + q] = f5[q];
      }
      if (slot_is_rl && !((alive_lab >> lane) & 1ull)) {              // (slot view) an RL slot without a vehicle
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * slot_rl_index + q] = 0.0f;
      }
      // reward (flow/envs/multiagent/merge.py:142-171 over rewards.desired_velocity), sums in SLOT order
      const int n_alive = nA + n1;
      T reward;
      if (s.evaluate) {
        const T sum_v = seg_sum<64>(to_slots_f(v, alive, 0.0f));
        reward = n_alive > 0 ? sum_v / T(n_alive) : 0.0f;
      } else {
        const T mc_lane = tb.template t_gather<TAB_MAX_COST>(n_alive & 63);
        const T max_cost = n_alive < 64 ? mc_lane : o.max_cost_full;
        const T dv = v - s.target_velocity;
        const T cost = tsqrt(seg_sum<64>(to_slots_f(dv * dv, alive, 0.0f)));
        T cost1 = tmax(max_cost - cost, 0.0f) / (max_cost + 1.1920928955078125e-07f);
        const bool bad = (__ballot(alive && (v < -100.0f)) != 0ull) || n_alive == 0;
        cost1 = bad ? 0.0f : cost1;
        const bool use = alive && is_rl && has && (v > 0.0f);
        const T t_headway = tmax(h / (use ? v : 1.0f), 0.0f);
        const T term = tmin((t_headway - 1.0f) / 1.0f, 0.0f);
        const T term_s = to_slots_f(term, use, 0.0f);
        const int use_s = to_slots_i(1, use, 0);
        T cost2 = 0.0f;
        for (unsigned long long u = __ballot(use_s != 0); u; u &= u - 1ull) cost2 = cost2 + read_lane(term_s, __ffsll((long long)u) - 1);
        reward = tmax(cost1 + 0.1f * cost2, 0.0f);
        reward = crashed ? 0.0f : reward;
      }
      if (lane == 0) {
        *rrow = reward;
        *drow = done_flag(tcount >= s.step_limit, crashed);
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
  }

  // ---- the state back to its slots -----------------------------------------------------------------------------
  __threadfence();
  {
    const bool alive = (lane < nA) || (lane >= 64 - n1);
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
      o.lead[e] = lead;
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
      cnt[CNT_TOTAL_DROPPED] = tot_drop;
    }
    if (lane < FS_MAX_INFLOWS) o.emitted[size_t(r) * FS_MAX_INFLOWS + lane] = emit_l;
  }
}

}  // namespace fs
