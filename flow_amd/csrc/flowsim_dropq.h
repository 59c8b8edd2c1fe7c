// flowsim_dropq.h -- gfx950 rollout kernel of the lane-drop network (FS_NET_BOTTLENECK, 4 -> 2 -> 1 lanes) in QUEUE order.
//
// k_steps_wide (flowsim_wide.h) keeps vehicle SLOT t in thread t of a workgroup and, every sub-step, ranks all slots by
// position, scatters them into rank order and searches masks over ranks for leaders (~9 barriers, ~1400 instructions per
// wave and sub-step).  With lane changing off (lane_change_mode = 0: every shipped bottleneck RL experiment,
// examples/exp_configs/rl/singleagent/singleagent_bottleneck.py:33-53) a vehicle keeps the entry lane ("path") it was
// inserted on, and a path is a queue: vehicles enter upstream, leave downstream, nobody overtakes.  So here
//   * one replica = one workgroup of 4 waves, WAVE p HOLDS PATH p in driving order, head in lane 0: the nearest vehicle
//     ahead on the own path (M5, flow/core/kernel/vehicle/traci.py:219-242) is the previous lane -- one DPP move;
//   * the other candidates of the zipper rule M8 (a vehicle within zipper_distance of a join, or past it, follows the
//     nearest vehicle ahead on every lane that joins its own there) come out of the other waves' queues, mirrored in LDS
//     after the move: a vehicle upstream of the zone needs only the REARMOST vehicle of the partner path beyond the
//     first join and of the other two paths beyond the second (two prefix lengths per path, one ballot each); a vehicle
//     inside a zone or beyond a join does a 6-step binary search for its place in the sorted partner queue(s);
//   * the slot a vehicle occupies in the state arrays (M1: lowest free slot of its type) is a LABEL it carries;
//   * the observation of BottleneckDesiredVelocityEnv (flow/envs/bottleneck.py:868-924: vehicles and mean speeds per
//     lane-segment) is one LDS integer atomic per vehicle and quantity -- speeds in units of 2^-16 m/s, so the sum is
//     exact and its order does not matter (oracle/opennet.py, spec['cell_sum'] = 'fixed', states the same sum);
//   * a sub-step has TWO barriers: after the move (mirrors, arrivals) and before the observation is read back.
// Events inside a wave -- an arrival moves the queue down one lane, a vehicle that is no longer strictly behind the
// previous lane (a collision) re-sorts the path by (x descending, lower slot first) -- are wave-uniform branches; the
// insertion bookkeeping (schedule, random entry lanes M9, slot pools M1, id counters) is computed redundantly by all
// four waves from the same published values, so it needs no barrier of its own.
//
// Scope (Sim::dropq_ok): float32, every vehicle driven by SUMO's car-following model (SimCarFollowingController or an
// RLController whose actions are maxSpeed shifts: FLAG_NO_FLOW_CTRL), one vehicle length, Euler, four entry lanes,
// lane changing and follower tracking off, scheduled inflows, no replica mask, >= 1 step per launch.  A path holds at
// most 64 vehicles: an insertion into a full path is refused and the handle is flagged (fs_* calls then fail) -- with
// the bench's 256 slots a path holds ~40.  Everything else steps on k_steps_wide / k_steps_open<., 64, 4>.
#pragma once

namespace fs {

struct DropRow { float tau, min_gap, max_accel, ts_sumo, sumo_max; int is_rl, type, pad; };   // per label (slot)
// a queue entry as the other waves see it.  (key, x) read as ONE 64-bit unsigned integer orders the candidates of the
// leader choice -- the nearest first, and of two at the same position the higher slot (positions are >= 0: the bits of
// a float order like the float) -- with a single v_cmp_lt_u64 where (x <, x ==, slot >) took three compares and two
// scalar operations on their masks
struct alignas(16) DropXL { unsigned key; float x; float v; int pad; };            // key = 0xffff - slot

struct DropQLds {
  OpenTabsLds<float> tabs;
  DropRow row[256];
  DropXL xl[4][64];            // path p's queue, head first: (position, label, speed) after the move
  int n[4], ng1[4], ng2[4];    // vehicles on path p; of them beyond the first / the second join (prefixes of the queue)
  int arr_n[4], arr_lab[4][8]; // arrivals of this sub-step: count and labels per path
  int crash[2][4];
  int acc[2][4][64];           // observation cells [gym step & 1]: human count, RL count, human speed sum, RL speed sum (2^-16 m/s)
  float act[4][64];            // the step's action row, a copy per wave (written and read by that wave only: no barrier)
  unsigned long long alive_w[4], tmask[8][4];   // slots in use at launch start; slots of vehicle type t
  // staging for the (re)build of the queues from the slot arrays
  float st_x[256], st_v[256], st_vmax[256], st_prev[256];
  int st_path[256], st_seq[256], st_origin[256], st_lab[256];
};

template <bool DV>
__global__ __launch_bounds__(256) void k_drop_queue(DevView<float> s, OpenView<float> o, QueueConsts qc, int* __restrict__ qflag,
                                                    int num_steps, const float* __restrict__ actions, size_t act_stride,
                                                    float* __restrict__ obs, float* __restrict__ rew,
                                                    uint8_t* __restrict__ done, int obs_every_step) {
  using T = float;
  using ull = unsigned long long;
  constexpr int P = 4;
  const T BIGV = 3.0e38f;
  __shared__ DropQLds L;
  const int tid = threadIdx.x;
  const int w = tid >> 6;                         // wave = path (entry lane)
  const int l = tid & 63;
  const int rr = blockIdx.x;                      // one replica per workgroup
  const int N = s.N;
  const bool slot_ok = tid < N;                   // SLOT view: thread t speaks for slot t of the state arrays
  const int ti = slot_ok ? tid : N - 1;
  const size_t base = size_t(rr) * N;

  OpenTabs<T, true> tb;
  tb.load(o, l, DV, &L.tabs);
  {
    DropRow q;
    q.tau = s.sumo_tau[ti]; q.min_gap = s.sumo_min_gap[ti]; q.max_accel = s.max_accel[ti];
    q.ts_sumo = 2.0f * tsqrt(s.max_accel[ti] * s.max_decel[ti]);
    q.sumo_max = s.sumo_max_speed[ti];
    q.is_rl = s.ctrl[ti] == FS_CTRL_RL ? 1 : 0;
    q.type = o.slot_type[ti];
    q.pad = 0;
    L.row[tid] = q;
  }
  const int slot_type = o.slot_type[ti];
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const ull m = __ballot(slot_ok && slot_type == t);
    if (l == 0) L.tmask[t][w] = m;
  }
  if (tid < 4) { L.n[tid] = 0; L.arr_n[tid] = 0; L.crash[0][tid] = 0; L.crash[1][tid] = 0; }
  L.acc[0][w][l] = 0;
  L.acc[1][w][l] = 0;

  // ---- replica scalars: every wave keeps its own copy and updates it the same way ----------------------------
  int tcount = s.time[rr];
  int32_t* cnt = o.counters + size_t(rr) * 8;
  int sim_steps = cnt[CNT_SIM_STEPS], seq_ctr = cnt[CNT_SEQ];
  int n_arr = cnt[CNT_ARRIVED], n_dep = cnt[CNT_DEPARTED], tot_arr = cnt[CNT_TOTAL_ARRIVED],
      tot_dep = cnt[CNT_TOTAL_DEPARTED], tot_drop = cnt[CNT_TOTAL_DROPPED];
  const uint32_t episode = uint32_t(o.episode[rr]);
  int emit_l = (l < FS_MAX_INFLOWS) ? o.emitted[size_t(rr) * FS_MAX_INFLOWS + l] : 0;
  int hist_l = (l < 20) ? o.arr_hist[size_t(rr) * 20 + l] : 0;           // arrivals of sub-step % 20 == lane
  const bool my_flow = l < o.n_inflows;                                    // lane f of every wave keeps inflow f (M2)
  const int fl = my_flow ? l : 0;
  const double my_per = o.flow_tab_d[fl], my_begin = o.flow_tab_d[64 + fl], my_end = o.flow_tab_d[128 + fl];
  const int my_number = o.flow_tab_i[128 + fl];
  const float f_xdep = o.lane_tab[TAB_FL_XDEP * 64 + fl], f_vdep = o.lane_tab[TAB_FL_VDEP * 64 + fl];
  const float f_ts = o.lane_tab[TAB_FL_TWOSQRT * 64 + fl], f_gap = o.lane_tab[TAB_FL_MINGAP * 64 + fl],
              f_tau = o.lane_tab[TAB_FL_TAU * 64 + fl];
  const int f_typ = o.flow_tab_i[fl], f_route = o.flow_tab_i[64 + fl];
  // the first sub-step index n (now = n * sim_step, n = sim_steps - 1) at which my inflow's next vehicle is due: the
  // schedule is float64 (M2), the sub-steps compare integers
  const double inv_dt_d = 1.0 / o.dt_d;
  auto due_index = [&](int k) -> int {
    const double t = my_begin + double(k) * my_per;
    if (!my_flow || !(t <= my_end) || !(my_number < 0 || k < my_number)) return 0x7fffffff;
    if (!(t > 0.0)) return 0;
    const double q = t * inv_dt_d;                 // (a first guess: the two loops below make n exact whatever it is)
    if (!(q < 2.0e9)) return 0x7fffffff;
    int n = int(q);
    while (double(n) * o.dt_d < t) n += 1;
    while (n > 0 && double(n - 1) * o.dt_d >= t) n -= 1;
    return n;
  };
  int my_due_n = due_index(emit_l);

  // (launch constants of the hot path in VECTOR registers: the loop's scalar state does not fit the SGPR file, and what
  // the compiler spills it reloads with a v_readlane each time)
  const T dt = in_vgpr(s.dt);
  const T m1 = in_vgpr(o.m1), m2 = in_vgpr(o.m2), zip = in_vgpr(o.zip_d), end_x = in_vgpr(o.end_x), LEN = in_vgpr(qc.veh_len),
          vlim = in_vgpr(o.speed_limit);
  const T crash_gap = in_vgpr(s.crash_gap);
  const T act_lo = in_vgpr(s.clip_actions ? s.act_lo : -3.0e38f), act_hi = in_vgpr(s.clip_actions ? s.act_hi : 3.0e38f);
  auto shift_of = [&](T xx) -> int { return (xx >= m1 ? 1 : 0) + (xx >= m2 ? 1 : 0); };
  // the route segments (one table: edges and junction-internal stretches alternate): segment of a coordinate
  const int nseg = o.nseg[0];
  T sst[15];                                     // the starts of segments 1 .. 15 (3e38 beyond the table): registers
#pragma unroll
  for (int q = 1; q < 16; ++q) sst[q - 1] = in_vgpr(q < nseg ? o.lane_tab[TAB_SEG_START * 64 + q] : 3.0e38f);
  auto seg_of = [&](T xx) -> int {
    int k = 0;
#pragma unroll
    for (int q = 0; q < 15; ++q) k += (xx >= sst[q]) ? 1 : 0;
    return k;
  };
  const unsigned seg_internal = o.seg_internal[0];

  // ---- build the queues from the slot arrays: rank of every vehicle within its path ---------------------------
  T x = 0.0f, v = 0.0f, vmax = 1.0f, prev_v = 0.0f;
  int lab = 0, seq = 0, origin = -1;
  int n = 0;                                      // vehicles in MY wave's queue (wave-uniform)
  {
    T sx = s.pos[base + ti], sv = s.vel[base + ti];
    if (s.st16 != nullptr) state16_load(s, base + ti, sx, sv);
    const int sp = slot_ok ? s.lane[base + ti] : -1;
    L.st_x[tid] = sx;
    L.st_path[tid] = sp;
    const ull am = __ballot(sp >= 0);
    if (l == 0) L.alive_w[w] = am;
    __syncthreads();
    int rank = 0;
    for (int j = 0; j < N; ++j) {
      const T xj = L.st_x[j];
      const int pj = L.st_path[j];
      rank += (pj == sp) & ((xj > sx) | ((xj == sx) & (j < tid))) ? 1 : 0;
    }
    __syncthreads();
    if (sp >= 0 && sp < P && rank < 64) {
      const int d = sp * 64 + rank;
      L.st_x[d] = sx; L.st_v[d] = sv; L.st_vmax[d] = o.vmax[base + ti]; L.st_prev[d] = s.prev_vel[base + ti];
      L.st_lab[d] = tid; L.st_seq[d] = o.seq[base + ti]; L.st_origin[d] = o.origin[base + ti];
      atomicAdd(&L.n[sp], 1);
    } else if (sp >= 0) {
      atomicOr(qflag, 1);                         // more than 64 vehicles on one path (or a path this kernel does not know)
    }
    __syncthreads();
    n = L.n[w];
    n = __builtin_amdgcn_readfirstlane(n);
    const int d = w * 64 + l;
    if (l < n) {
      x = L.st_x[d]; v = L.st_v[d]; vmax = L.st_vmax[d]; prev_v = L.st_prev[d];
      lab = L.st_lab[d]; seq = L.st_seq[d]; origin = L.st_origin[d];
    }
  }
  ull aw0 = L.alive_w[0], aw1 = L.alive_w[1], aw2 = L.alive_w[2], aw3 = L.alive_w[3];      // slots in use (all waves: same)
  ull ar0 = 0ull, ar1 = 0ull, ar2 = 0ull, ar3 = 0ull;                                       // RL slots that arrived in the last sub-step

  // my vehicle's parameters (by label)
  T u_tau = 1.0f, u_gap = 1.0f, u_acc = 1.0f, u_ts = 1.0f, y_uts = 1.0f;      // (y_uts = div_core_recip(u_ts))
  bool is_rl = false;
#ifdef FS_QDIAG
  int dq_n[2] = {0, 0};            // sub-steps with a due inflow; full searches
#endif
  int c1 = 0, c2 = 0, c3 = 0;      // vehicles of paths w ^ 1, w ^ 2, w ^ 3 ahead of mine at the last snapshot (count_near3)
  T row_smax = 1.0f;                 // the vType maxSpeed of my vehicle's slot (what a new vehicle starts with)
  auto load_params = [&]() {
    const DropRow q = L.row[lab & 255];
    u_tau = q.tau; u_gap = q.min_gap; u_acc = q.max_accel; u_ts = q.ts_sumo;
    y_uts = div_core_recip(u_ts);
    is_rl = q.is_rl != 0;
    row_smax = q.sumo_max;
  };
  load_params();
  auto gather_all = [&](int src, bool take) {
#define FS_D_G(reg_) do { const auto t_ = bperm(reg_, src); reg_ = take ? t_ : reg_; } while (0)
    FS_D_G(x); FS_D_G(v); FS_D_G(lab); FS_D_G(seq); FS_D_G(origin); FS_D_G(vmax); FS_D_G(prev_v);
    FS_D_G(c1); FS_D_G(c2); FS_D_G(c3);
#undef FS_D_G
  };
  // my path re-sorted by (x descending, lower slot first) -- after a collision
  auto resort = [&]() {
    const bool al = l < n;
    int c = 0;
    for (int j = 0; j < n; ++j) {
      const T xj = read_lane(x, j);
      const int lj = read_lane_i(lab, j);
      c += (int(xj > x) | (int(xj == x) & int(lj < lab)));
    }
    const int target = al ? c : l;
    const int src = __builtin_amdgcn_ds_permute(target << 2, l);
    gather_all(src, true);
    load_params();
  };

  // ---- M5 / M8: leader, headway, collision --------------------------------------------------------------------
  T h = 1000.0f, vl = -1001.0f;
  bool has = false;
  int lead_lab = -1;
  // the number of vehicles of the three other paths AHEAD of mine: searches in their sorted mirrors, the three side by
  // side.  Three pivots per round (steps 16, 4, 1: three dependent LDS round trips for 64 entries, where a binary search
  // takes six); a mirror is sorted, so the pivots that are ahead form a prefix and their number is the advance.  The
  // searches compare positions only; vehicles AT my position (rare: ahead iff their slot is lower) are counted afterwards,
  // by the whole wave, when some lane met one
  // the queue lengths and join prefixes of the three OTHER paths (t = 1: my partner at the first join, t = 2, 3: the other
  // pair), as their owners published them
  int g1T1 = 0, g2T2 = 0, g2T3 = 0, aT1 = 0, aT2 = 0, aT3 = 0;      // (wave-uniform values in VECTOR registers)
  int pn_l = 0, pg1_l = 0, pg2_l = 0, parr_l = 0;     // lane q (mod 4): n / ng1 / ng2 / arrivals of path q
  const int wu = __builtin_amdgcn_readfirstlane(w);
  auto ties = [&](int q, int nq, int& lo, bool tie) {
    int k = lo;
#pragma unroll 1
    for (int it = 0; it < 64 && __ballot(tie) != 0ull; ++it) {
      const DropXL ek = L.xl[q][k & 63];
      tie = tie && k < nq && ek.x == x;
      if (tie && int(0xffffu - ek.key) < lab) lo = k + 1;              // (equal positions are sorted by slot)
      k += 1;
    }
  };
  auto count_ahead3 = [&](int na_, int nb_, int nc_, int& ca, int& cb_, int& cc) {
    const int qa = w ^ 1, qb = w ^ 2, qc_ = w ^ 3;
    int la_ = 0, lb_ = 0, lc_ = 0;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 2) {
      int ta = 0, tb2 = 0, tc2 = 0;
#pragma unroll
      for (int m = 1; m <= 3; ++m) {
        const int ia = la_ + m * step - 1, ib = lb_ + m * step - 1, ic = lc_ + m * step - 1;
        const T xa = L.xl[qa][ia & 63].x, xb = L.xl[qb][ib & 63].x, xc = L.xl[qc_][ic & 63].x;
        ta += int(ia < na_) & int(xa > x);                             // (bitwise: no branches)
        tb2 += int(ib < nb_) & int(xb > x);
        tc2 += int(ic < nc_) & int(xc > x);
      }
      la_ += ta * step;
      lb_ += tb2 * step;
      lc_ += tc2 * step;
    }
    const DropXL ea = L.xl[qa][la_ & 63], eb = L.xl[qb][lb_ & 63], ec = L.xl[qc_][lc_ & 63];   // the first NOT strictly ahead
    // (the searches end at 63 at most: a full path whose 64 vehicles are all ahead is one more)
    la_ += int(la_ < na_) & int(ea.x > x);
    lb_ += int(lb_ < nb_) & int(eb.x > x);
    lc_ += int(lc_ < nc_) & int(ec.x > x);
    const bool ta = la_ < na_ && ea.x == x, tb_ = lb_ < nb_ && eb.x == x, tc = lc_ < nc_ && ec.x == x;
    if (__ballot(ta || tb_ || tc) != 0ull) {
      ties(qa, na_, la_, ta);
      ties(qb, nb_, lb_, tb_);
      ties(qc_, nc_, lc_, tc);
    }
    ca = la_; cb_ = lb_; cc = lc_;
  };
  // The same three counts from the counts of the last snapshot: between two sub-steps a count moves by the arrivals of
  // that path and by the few vehicles that passed or were passed on another lane, so the FOUR mirror entries around the
  // old count decide it in one LDS round trip -- the entries strictly ahead form a prefix of a sorted mirror: the window
  // holds the boundary if it shows an entry that is ahead (or starts at the head) and one that is not (or reaches the
  // tail).  Any lane whose window does not hold it sends the wave through the full search (the count is a property of the
  // mirrors, not of how it is found: both ways give the same number).
  auto count_near3 = [&](bool alive, int& ca, int& cb_, int& cc) {
    const int qa = w ^ 1, qb = w ^ 2, qc_ = w ^ 3;
    // (entries beyond a queue hold -3e38: never ahead, never equal -- no index tests; the window stays inside the array.
    // Everything below is a compare into VCC consumed by the next instruction: a predicate combined on the scalar unit
    // costs this wave, alone on its SIMD, a round trip through the SGPR file each time)
    // (a count never exceeds its queue: it was one a sub-step ago, and what left since is taken off)
    const int sa = min(max(ca - 2, 0), 60), sb = min(max(cb_ - 2, 0), 60), sc = min(max(cc - 2, 0), 60);
    const DropXL* pa = &L.xl[qa][sa];
    const DropXL* pb = &L.xl[qb][sb];
    const DropXL* pc = &L.xl[qc_][sc];
    int ka = 0, kb = 0, kc = 0, ga = 0, gb = 0, gc = 0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const T xa = pa[m].x, xb = pb[m].x, xc = pc[m].x;
      ka += xa > x ? 1 : 0; ga += xa >= x ? 1 : 0;
      kb += xb > x ? 1 : 0; gb += xb >= x ? 1 : 0;
      kc += xc > x ? 1 : 0; gc += xc >= x ? 1 : 0;
    }
    // the window does not hold the boundary: nothing ahead in it and it does not start at the head, or all four ahead and
    // it does not end at the array's end
    int bad = (ka == 0 ? sa : 0) | (ka == 4 ? sa - 60 : 0) | (kb == 0 ? sb : 0) | (kb == 4 ? sb - 60 : 0) |
              (kc == 0 ? sc : 0) | (kc == 4 ? sc - 60 : 0);
    bad = alive ? bad : 0;
    if (__ballot(bad != 0) != 0ull) {
#ifdef FS_QDIAG
      dq_n[1] += 1;
#endif
      count_ahead3(read_lane_i(pn_l, wu ^ 1), read_lane_i(pn_l, wu ^ 2), read_lane_i(pn_l, wu ^ 3), ca, cb_, cc);
      return;
    }
    int la_ = sa + ka, lb_ = sb + kb, lc_ = sc + kc;
    int eq = (ga - ka) | (gb - kb) | (gc - kc);            // vehicles AT my position in a window
    eq = alive ? eq : 0;
    if (__ballot(eq != 0) != 0ull) {
      ties(qa, read_lane_i(pn_l, wu ^ 1), la_, ga != ka);
      ties(qb, read_lane_i(pn_l, wu ^ 2), lb_, gb != kb);
      ties(qc_, read_lane_i(pn_l, wu ^ 3), lc_, gc != kc);
    }
    ca = la_; cb_ = lb_; cc = lc_;
  };
  auto read_counts = [&]() {
    // (ONE round trip: every lane reads the four words of its path l & 3; the scalars are lane reads -- six uniform LDS
    // reads, each waited for, were a tenth of the sub-step)
    pn_l = L.n[l & 3]; pg1_l = L.ng1[l & 3]; pg2_l = L.ng2[l & 3]; parr_l = L.arr_n[l & 3];
    // what every sub-step's neighbour update uses of them, read to vector registers directly (uniform addresses: the same
    // round trip) -- a lane read would take each through the scalar file and back
    g1T1 = L.ng1[wu ^ 1]; g2T2 = L.ng2[wu ^ 2]; g2T3 = L.ng2[wu ^ 3];
    aT1 = L.arr_n[wu ^ 1]; aT2 = L.arr_n[wu ^ 2]; aT3 = L.arr_n[wu ^ 3];
  };
  auto neighbours = [&](bool& crash) {
    const bool alive = l < n;
    const int la = (x + zip >= m1 ? 1 : 0) + (x + zip >= m2 ? 1 : 0);
    const T x_up = dpp<DPP_WAVE_SHR1>(x), v_up = dpp<DPP_WAVE_SHR1>(v);
    const int lab_up = dpp_i<DPP_WAVE_SHR1>(lab);
    // the best candidate so far as (key, x): the vehicle ahead on my own path, if any (lanes beyond the queue compute
    // along; what they find is never used)
    const bool up = l > 0;
    unsigned bk = up ? 0xffffu - unsigned(lab_up) : 0u;
    T bx = up ? x_up : BIGV, bv = up ? v_up : 0.0f;
    int bp = w;
    // (the counts of the last snapshot, less the vehicles that left those paths since)
    int cnt1 = c1 - aT1, cnt2 = c2 - aT2, cnt3 = c3 - aT3;
    count_near3(alive, cnt1, cnt2, cnt3);
    c1 = cnt1; c2 = cnt2; c3 = cnt3;
    // the partner path (t = 1): every vehicle ahead once I look across the first join (la >= 1), else its rearmost
    // vehicle beyond that join; the other pair: every vehicle ahead once I look across the second join, else its
    // rearmost vehicle beyond it
    const int ci1 = (la >= 1 ? cnt1 : g1T1) - 1, ci2 = (la == 2 ? cnt2 : g2T2) - 1, ci3 = (la == 2 ? cnt3 : g2T3) - 1;
    const DropXL e1 = L.xl[w ^ 1][ci1 < 0 ? 0 : ci1], e2 = L.xl[w ^ 2][ci2 < 0 ? 0 : ci2], e3 = L.xl[w ^ 3][ci3 < 0 ? 0 : ci3];
#define FS_D_CAND(e_, ci_, q_) do {                                                                                   \
      const T ex_ = ci_ >= 0 ? e_.x : BIGV;                                                                           \
      const ull ke_ = (ull(__float_as_uint(ex_)) << 32) | ull(e_.key), kb_ = (ull(__float_as_uint(bx)) << 32) | ull(bk); \
      const bool take_ = ke_ < kb_;                       /* the nearest; equal x: the higher slot */                  \
      bx = take_ ? ex_ : bx; bk = take_ ? e_.key : bk; bv = take_ ? e_.v : bv; bp = take_ ? (q_) : bp;                 \
    } while (0)
    FS_D_CAND(e1, ci1, w ^ 1);
    FS_D_CAND(e2, ci2, w ^ 2);
    FS_D_CAND(e3, ci3, w ^ 3);
#undef FS_D_CAND
    const bool any = bx < BIGV;
    has = any;
    h = any ? (bx - x) - LEN : 1000.0f;                 // vehicle/traci.py:237
    vl = any ? bv : -1001.0f;
    lead_lab = any ? int(0xffffu - bk) : -1;
    const int sh_l = (bx >= m1 ? 1 : 0) + (bx >= m2 ? 1 : 0);
    const int other_lane = (w ^ bp) >> sh_l;                                // M8: a collision needs one physical lane
    const T h_same = other_lane == 0 ? h : 1000.0f;
    crash = __ballot(alive && (h_same < crash_gap)) != 0ull;
  };
  auto publish = [&]() {
    const bool alive = l < n;
    DropXL e;
    e.x = alive ? x : -BIGV;                            // (beyond the queue: behind everybody, whoever compares)
    e.key = 0xffffu - unsigned(lab); e.v = v; e.pad = 0;
    L.xl[w][l] = e;
    const int c1 = __popcll(__ballot(alive && x >= m1)), c2 = __popcll(__ballot(alive && x >= m2));
    if (l == 0) { L.n[w] = n; L.ng1[w] = c1; L.ng2[w] = c2; }
  };
  // O6: the lane-segments of my vehicle at its position: the one it is observed in (bottleneck.py:868-924) and the one
  // whose action shifts its maxSpeed in the next sub-step (:926-969) -- one lookup serves both
  int acell = -1;
  auto cells = [&](bool want_obs, int& ocell) {
    const bool alive = l < n;
    const int seg_k = seg_of(x);
    const bool eligible = alive && !((seg_internal >> seg_k) & 1u);
    const int my_lane = w >> shift_of(x);
    // cell_of<0> / cell_of<1> (flowsim_open.h) with their table rows read together: the groups of my edge (at most three
    // observation groups, two action groups: Sim::dropq_ok), each (start, lo, hi, first cell | lanes | first lane | last)
    const int range = L.tabs.ctab_i[2][seg_k & 63];
    const int og0 = range & 0xff, ocnt = (range >> 8) & 0xff, ag0 = (range >> 16) & 0xff, acnt = (range >> 24) & 0xff;
    const CellRow<T> o0 = L.tabs.cpack[0][(og0) & 63], o1 = L.tabs.cpack[0][(og0 + (1 < ocnt ? 1 : 0)) & 63],
                     o2 = L.tabs.cpack[0][(og0 + (2 < ocnt ? 2 : 0)) & 63];
    const CellRow<T> a0 = L.tabs.cpack[1][(ag0) & 63], a1 = L.tabs.cpack[1][(ag0 + (1 < acnt ? 1 : 0)) & 63];
    // (a group's tests each turn the candidate into -1 through VCC; the groups of an edge are its segments -- disjoint
    // stretches (lo, hi], and the position-0 rule picks the last one only when no stretch holds the position -- so the
    // first hit of the table walk is the only hit: a maximum.  A row index beyond the edge's groups repeats an earlier row)
    auto hit = [&](const CellRow<T>& row, bool last_rule) -> int {
      const T pos = x - row.start;
      const unsigned meta = unsigned(row.meta);
      const int rel = my_lane - int((meta >> 16) & 0xffu);
      int c = int(meta & 0xffu) + rel;
      c = unsigned(rel) < ((meta >> 8) & 0xffu) ? c : -1;
      int c_in = pos > row.lo ? c : -1;
      c_in = pos <= row.hi ? c_in : -1;
      if (last_rule) {                                   // searchsorted(..) - 1 == -1: the last segment
        int c_last = (meta >> 24) != 0u ? c : -1;
        c_last = pos == 0.0f ? c_last : -1;
        c_in = max(c_in, c_last);
      }
      return c_in;
    };
    int ac = max(hit(a0, false), hit(a1, false));
    ac = acnt > 0 ? ac : -1;
    acell = eligible ? ac : -1;
    int oc = -1;
    if (want_obs) {
      oc = max(max(hit(o0, true), hit(o1, true)), hit(o2, true));
      oc = ocnt > 0 ? oc : -1;
      oc = eligible ? oc : -1;
    }
    ocell = oc;
  };

  // get_outflow_rate over the last `window` sub-steps (vehicle/traci.py:500-505): the arrivals inside the observation's /
  // the reward's window are kept as running sums (this sub-step's arrivals enter, the sub-step that leaves the window goes)
  auto window_sum = [&](int window) -> int {
    const int nn = tcount < window ? tcount : window;
    const int ago = (((tcount - 1 - l) % 20) + 20) % 20;
    const T mine = (l < 20 && ago < nn) ? T(hist_l) : 0.0f;
    return __builtin_amdgcn_readfirstlane(int(seg_sum<64>(mine)));   // small integers: exact in any order
  };
  int out_obs = window_sum(o.obs_window), out_rew = window_sum(o.rew_window);
  auto outflow = [&](int window, int total_i) -> T {
    const int nn = tcount < window ? tcount : window;
    const T rate = (3600.0f * T(total_i)) / (T(nn > 0 ? nn : 1) * dt);
    return nn > 0 ? rate : 0.0f;
  };

  const int obs_dim = o.obs_dim;
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;

  // the vehicle in this lane leaves the network: its final state goes to its slot now
  auto retire = [&](bool mine) {
    if (mine) {
      const size_t e = base + size_t(lab & 255);
      if (s.st16 != nullptr) state16_store(s, e, x, v);
      else { s.pos[e] = x; s.vel[e] = v; }
      s.lane[e] = -1;
      s.prev_vel[e] = prev_v;
      s.accel[e] = 0.0f;
      o.seq[e] = seq;
      o.origin[e] = origin;
      o.vmax[e] = vmax;
      o.lead[e] = -1;
      o.headway[e] = 1000.0f;
    }
  };

  // ---- the snapshot of the launch's first sub-step ------------------------------------------------------------------
  if (actions != nullptr && l < s.num_rl) L.act[w][l] = actions[size_t(rr) * s.num_rl + l];
  publish();
  __syncthreads();
  read_counts();
  {
    bool c_;
    neighbours(c_);
    int oc_;
    if (DV) cells(false, oc_);
  }
  __syncthreads();

#ifdef FS_QDIAG
  unsigned long long dq_t[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 0 move, 1 order/arrivals, 2 publish+barrier, 3 counts+bookkeeping, 4 insertions, 5 neighbours, 6 cells, 7 barrier 2, 8 head
  const unsigned long long dq_start = __builtin_readcyclecounter();
#define FS_DT(var_) const unsigned long long var_ = __builtin_readcyclecounter()
#define FS_DA(slot_, t0_) dq_t[slot_] += __builtin_readcyclecounter() - (t0_)
#else
#define FS_DT(var_)
#define FS_DA(slot_, t0_)
#endif
  for (int step = 0; step < num_steps; ++step) {
    const int ab = step & 1;
    // the next step's action row: the load is issued here and its value goes to the wave's LDS copy after the sub-steps
    // (stored at once, it was an L2 / HBM round trip at the top of every step; used inside the sub-step loop, the compiler
    // waits for it in the loop's preheader)
    const bool act_next = actions != nullptr && step + 1 < num_steps && l < s.num_rl;
    float a_pref = 0.0f;
    if (act_next) a_pref = actions[size_t(step + 1) * act_stride + size_t(rr) * s.num_rl + l];
    bool crashed = false;
    const bool emit = obs_every_step || (step == num_steps - 1);
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      const bool live = !crashed;
      bool alive = l < n;
      FS_DT(d0);
      // ---- O6: BottleneckDesiredVelocityEnv._apply_rl_actions (bottleneck.py:926-969) -----------------------------
      if (DV && actions != nullptr) {
        const float a_cell = L.act[w][acell >= 0 ? acell : 0];
        T a = acell >= 0 ? a_cell : 0.0f;
        a = tmin(tmax(a, act_lo), act_hi);
        T nxt = tmin(tmax(vmax + a, 0.01f), 23.0f);
        nxt = acell >= 0 ? nxt : 23.0f;
        if (live && alive && is_rl) vmax = nxt;
      }
      // ---- M7: every vehicle follows SUMO's model (sumo_speed_fd) ---------------------------------------------------
      {
        const T u_vmax = tmin(vmax, vlim);                 // M10
        const float gap = hmax(h, 1e-3f);
        const float m = hmax(0.0f, v * u_tau + div_core_by(v * (v - vl), u_ts, y_uts));
        const float ss = u_gap + m;
        const float qq = div_core(ss, gap);
        const float q = has ? qq : 0.0f;
        const float r_ = div_core(v, u_vmax);
        const float r2 = r_ * r_;
        const float a_s = u_acc * (1.0f - r2 * r2 - q * q);
        const T v_new = hmax(0.0f, v + a_s * dt);
        const bool mv = live && alive;
        prev_v = mv ? v : prev_v;
        x = mv ? x + v_new * dt : x;
        v = mv ? v_new : v;
      }
      if (live) { tcount += 1; sim_steps += 1; }
      FS_DA(0, d0);
      FS_DT(d1);
      // ---- the order of my path: a collision re-sorts it; arrivals (M4) leave from the head -----------------------
      int na = 0;
      if (live) {
        const T x_up = dpp<DPP_WAVE_SHR1>(x);
        if (__ballot(alive && l > 0 && !(x < x_up)) != 0ull) resort();
        const bool arrived = alive && (x >= end_x);
        na = __popcll(__ballot(arrived));
        if (na > 0) {
          retire(arrived);
          if (arrived && l < 8) L.arr_lab[w][l] = lab;
          if (na > 8) atomicOr(qflag, 2);
          n -= na;
          gather_all(l + na, l < n);
          load_params();
          alive = l < n;
        }
      }
      if (l == 0) L.arr_n[w] = na;
      FS_DA(1, d1);
      FS_DT(d2);
      publish();
      lds_barrier();
      FS_DA(2, d2);
      FS_DT(d3);
      // ---- every wave: the arrivals of all paths, then the insertions (M2 / M3 / M9 / M1) in InFlows order ---------
      read_counts();
      ull ja0 = 0ull, ja1 = 0ull, ja2 = 0ull, ja3 = 0ull;      // slots freed in this sub-step: free from the next one on
      if (live) {
        int na_all = 0;
        ar0 = ar1 = ar2 = ar3 = 0ull;
#pragma unroll
        for (int q = 0; q < P; ++q) {
          const int nq = read_lane_i(parr_l, q);
          na_all += nq;
          for (int j = 0; j < nq && j < 8; ++j) {
            const int lj = __builtin_amdgcn_readfirstlane(L.arr_lab[q][j]) & 255;
            const ull bit = 1ull << (lj & 63);
            const bool rl_j = L.row[lj].is_rl != 0;
            if ((lj >> 6) == 0) { ja0 |= bit; if (rl_j) ar0 |= bit; }
            else if ((lj >> 6) == 1) { ja1 |= bit; if (rl_j) ar1 |= bit; }
            else if ((lj >> 6) == 2) { ja2 |= bit; if (rl_j) ar2 |= bit; }
            else { ja3 |= bit; if (rl_j) ar3 |= bit; }
          }
        }
        n_arr = na_all;
        n_dep = 0;
        tot_arr += na_all;
        {
          const int eo = tcount - 1 - o.obs_window, er = tcount - 1 - o.rew_window;      // the sub-steps that leave the windows
          out_obs += na_all - (eo >= 0 ? read_lane_i(hist_l, eo % 20) : 0);
          out_rew += na_all - (er >= 0 ? read_lane_i(hist_l, er % 20) : 0);
        }
        if (l == (tcount - 1) % 20) hist_l = na_all;
        FS_DA(3, d3);
        FS_DT(d4);
        // inflows whose next vehicle is due (lane f of every wave evaluates inflow f)
        unsigned fm = unsigned(__ballot(sim_steps - 1 >= my_due_n)) & 0xffu;
#ifdef FS_QDIAG
        dq_n[0] += fm != 0u ? 1 : 0;
#endif
        // (x, v, label) of the vehicles inserted so far in this sub-step, per path: the tail an insertion is checked against
        T ix0 = 0, ix1 = 0, ix2 = 0, ix3 = 0, iv0 = 0, iv1 = 0, iv2 = 0, iv3 = 0;
        int ic0 = 0, ic1 = 0, ic2 = 0, ic3 = 0;
        while (fm != 0u) {
          const int f = __ffs(int(fm)) - 1;
          fm &= fm - 1u;
          const int k = read_lane_i(emit_l, f);
          const int typ = read_lane_i(f_typ, f);
          int route_f = read_lane_i(f_route, f);
          const T x_dep = read_lane(f_xdep, f), v_dep = read_lane(f_vdep, f);
          const T two_sqrt = read_lane(f_ts, f), min_gap_f = read_lane(f_gap, f), tau_f = read_lane(f_tau, f);
          const bool random_lane = route_f < 0;
          if (random_lane) {                                 // M9: departLane = "random"
            uint32_t c0 = uint32_t(k), c1 = uint32_t(1000 + f), c2 = s.rep0 + uint32_t(rr), c3 = 1u + 2u * episode;
            philox4x32_10(c0, c1, c2, c3, s.seed_lo, s.seed_hi);
            route_f = __builtin_amdgcn_readfirstlane(int((uint64_t(c0 >> 8) * uint64_t(P)) >> 24));
          }
          // M1: the lowest free slot of the type (a slot freed in this sub-step is not free yet)
          const ull f0 = L.tmask[typ & 7][0] & ~aw0 & ~ja0, f1 = L.tmask[typ & 7][1] & ~aw1 & ~ja1,
                    f2 = L.tmask[typ & 7][2] & ~aw2 & ~ja2, f3 = L.tmask[typ & 7][3] & ~aw3 & ~ja3;
          int slot = -1;
          if (f0) slot = first_bit(f0);
          else if (f1) slot = 64 + first_bit(f1);
          else if (f2) slot = 128 + first_bit(f2);
          else if (f3) slot = 192 + first_bit(f3);
          slot = __builtin_amdgcn_readfirstlane(slot);
          // M3: the nearest vehicle ahead on the route: the tail of the own path, the rearmost vehicle of the partner
          // path beyond the first join, of the other two beyond the second; equal positions: the lowest slot
          // (lane q of every quad looks at path q -- one LDS round trip for the four candidates -- and the quad reduces to
          // the smallest (position, slot); the first version walked the paths one after the other: eight round trips)
          T xm, vm_;
          bool has_lead;
          {
            const int q = l & 3;
            const int t = q ^ route_f;
            const int icq = q == 0 ? ic0 : (q == 1 ? ic1 : (q == 2 ? ic2 : ic3));
            const int ci = t == 0 ? pn_l + icq - 1 : (t == 1 ? pg1_l - 1 : pg2_l - 1);
            const DropXL e = L.xl[q][ci < 0 ? 0 : (ci & 63)];
            const bool fresh = t == 0 && icq > 0;          // the vehicle inserted a moment ago on this path
            T cx = fresh ? (q == 0 ? ix0 : (q == 1 ? ix1 : (q == 2 ? ix2 : ix3))) : e.x;
            T cv = fresh ? (q == 0 ? iv0 : (q == 1 ? iv1 : (q == 2 ? iv2 : iv3))) : e.v;
            int cl = fresh ? 255 : int(0xffffu - e.key);   // (its position is the insertion point: no tie with a vehicle ahead)
            cx = ci >= 0 ? cx : BIGV;
            cl = ci >= 0 ? cl : 256;
            {
              const T ox = dpp<DPP_QUAD_XOR1>(cx), ov = dpp<DPP_QUAD_XOR1>(cv);
              const int ol = dpp_i<DPP_QUAD_XOR1>(cl);
              const bool b = (ox < cx) || (ox == cx && ol < cl);
              cx = b ? ox : cx; cv = b ? ov : cv; cl = b ? ol : cl;
            }
            {
              const T ox = dpp<DPP_QUAD_XOR2>(cx), ov = dpp<DPP_QUAD_XOR2>(cv);
              const int ol = dpp_i<DPP_QUAD_XOR2>(cl);
              const bool b = (ox < cx) || (ox == cx && ol < cl);
              cx = b ? ox : cx; cv = b ? ov : cv; cl = b ? ol : cl;
            }
            xm = read_lane(cx, 0);
            vm_ = read_lane(cv, 0);
            has_lead = xm < BIGV;
          }
          const T gap = (xm - LEN) - x_dep;
          const T dq = div_core(v_dep * (v_dep - vm_), two_sqrt);
          const T need = min_gap_f + tmax(0.0f, v_dep * tau_f + dq);
          const int n_own = read_lane_i(pn_l, route_f & 3) +
                            (route_f == 0 ? ic0 : (route_f == 1 ? ic1 : (route_f == 2 ? ic2 : ic3)));
          if (slot >= 0 && n_own >= 64 && (!has_lead || gap >= need)) atomicOr(qflag, 1);   // the path is full: refused, flagged
          const bool ok = __builtin_amdgcn_readfirstlane(int((slot >= 0) && n_own < 64 && (!has_lead || gap >= need))) != 0;
          if (ok) {
            if (w == route_f) {
              const bool fresh_lane = l == n;
              if (fresh_lane) {
                x = x_dep;
                v = v_dep;
                prev_v = 0.0f;                             // previous_speeds.get(veh_id, 0)
                lab = slot;
                seq = seq_ctr;
                origin = f * (1 << 20) + k;
                c1 = read_lane_i(pn_l, wu ^ 1); c2 = read_lane_i(pn_l, wu ^ 2); c3 = read_lane_i(pn_l, wu ^ 3);   // (all of them ahead of the entry point, as a rule)
              }
              n += 1;
              load_params();                               // (one LDS round trip for the parameters and the newcomer's maxSpeed)
              vmax = fresh_lane ? row_smax : vmax;
            }
            if (route_f == 0) { ix0 = x_dep; iv0 = v_dep; ic0 += 1; }
            else if (route_f == 1) { ix1 = x_dep; iv1 = v_dep; ic1 += 1; }
            else if (route_f == 2) { ix2 = x_dep; iv2 = v_dep; ic2 += 1; }
            else { ix3 = x_dep; iv3 = v_dep; ic3 += 1; }
            const ull bit = 1ull << (slot & 63);
            if ((slot >> 6) == 0) aw0 |= bit; else if ((slot >> 6) == 1) aw1 |= bit; else if ((slot >> 6) == 2) aw2 |= bit; else aw3 |= bit;
            seq_ctr += 1;
            n_dep += 1;
            tot_dep += 1;
          }
          // M9: a random-lane vehicle that does not fit when it is due is dropped, not retried
          const bool consumed = ok || random_lane;
          if (consumed && l == f) { emit_l = k + 1; my_due_n = due_index(k + 1); }
          if (consumed && !ok) tot_drop += 1;
        }
        aw0 &= ~ja0; aw1 &= ~ja1; aw2 &= ~ja2; aw3 &= ~ja3;
        FS_DA(4, d4);
      }
      FS_DT(d5);
      // ---- O1: the new snapshot, the collision check ---------------------------------------------------------------
      bool c = false;
      neighbours(c);
      FS_DA(5, d5);
      FS_DT(d6);
      const int cb = (step * s.sims_per_step + sub) & 1;
      if (c && l == 0) L.crash[cb][w] = 1;
      // ---- O6 get_state (bottleneck.py:868-924): every vehicle enters itself into its cell ---------------------------
      const bool last_sub = sub == s.sims_per_step - 1;
      if (DV && (actions != nullptr || (emit && last_sub))) {
        int ocell;
        cells(emit && last_sub, ocell);
        if (ocell >= 0) {
          const int vi = int(rintf(v * 65536.0f));
          atomicAdd(&L.acc[ab][is_rl ? 1 : 0][ocell], 1);
          atomicAdd(&L.acc[ab][is_rl ? 3 : 2][ocell], vi);
        }
      }
      FS_DA(6, d6);
      FS_DT(d7);
      lds_barrier();
      FS_DA(7, d7);
      {
        const bool cc = (L.crash[cb][0] | L.crash[cb][1] | L.crash[cb][2] | L.crash[cb][3]) != 0;
        crashed = crashed || (cc && live);
        if (tid < 4) L.crash[cb ^ 1][tid] = 0;             // (the other buffer is written in the next sub-step, two barriers away)
      }
    }

    // ---- get_state / compute_reward / done ------------------------------------------------------------------------
    FS_DT(d8);
    if (act_next) L.act[w][l] = a_pref;
    if (emit) {
      if (DV) {
        // (wave q writes block q of the row -- counts of humans, of RL vehicles, mean speeds of humans, of RL vehicles: the
        // four waves share the quotients; the sums of the NEXT gym step go to the other buffer, cleared here: its last
        // readers were the previous step's head, two barriers ago)
        const int C = o.n_obs_cells;
        const int kind = w & 1;                                // 0: humans, 1: RL vehicles
        const int cnt_k = L.acc[ab][kind][l];
        const T sp_k = T(L.acc[ab][2 + kind][l]) * (1.0f / 65536.0f);
        L.acc[ab ^ 1][w][l] = 0;
        const T nk = div_out(T(cnt_k), 20.0);                                           // NUM_VEHICLE_NORM
        T out_k = nk;
        if (w >= 2) out_k = div_out(cnt_k > 0 ? sp_k / (nk * 20.0f) : 0.0f, 50.0);     // (wave-uniform)
        if (l < C) orow[w * C + l] = out_k;
        if (w == 1) {
          const T of = div_out(outflow(o.obs_window, out_obs), 2000.0);
          if (l == 0) orow[4 * C] = of;
        }
      } else if (tid == 0) {
        orow[0] = 1.0f;                                    // bottleneck.py:481-483
      }
      if (w == 0) {                                        // (the wave with the shortest block of the row above)
        const T reward = outflow(o.rew_window, out_rew) / o.out_norm;     // bottleneck.py:474-478, 971-981
        if (l == 0) {
          *rrow = reward;
          *drow = done_flag(tcount >= s.step_limit, crashed);
        }
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
    FS_DA(8, d8);
  }
#ifdef FS_QDIAG
  if (blockIdx.x == 5 && l == 0)
    printf("DQDIAG wave %d total %llu move %llu order %llu publish+barrier %llu counts %llu insert %llu (n %d) full searches %d neighbours %llu cells %llu barrier2 %llu head %llu\n",
           w, __builtin_readcyclecounter() - dq_start, dq_t[0], dq_t[1], dq_t[2], dq_t[3], dq_t[4], dq_n[0], dq_n[1], dq_t[5], dq_t[6], dq_t[7], dq_t[8]);
#endif

  // ---- the state back to its slots ---------------------------------------------------------------------------------
  __threadfence();
  if (slot_ok) o.arrived_rl[base + tid] = 0;
  __syncthreads();
  {
    const bool alive = l < n;
    if (alive) {
      const size_t e = base + size_t(lab & 255);
      if (s.st16 != nullptr) state16_store(s, e, x, v);
      else { s.pos[e] = x; s.vel[e] = v; }
      s.lane[e] = w;
      s.prev_vel[e] = prev_v;
      s.accel[e] = 0.0f;
      o.seq[e] = seq;
      o.origin[e] = origin;
      o.vmax[e] = vmax;
      o.lead[e] = lead_lab;
      o.headway[e] = h;
    }
    if (slot_ok) {
      const ull arw = w == 0 ? ar0 : (w == 1 ? ar1 : (w == 2 ? ar2 : ar3));
      if ((arw >> l) & 1ull) o.arrived_rl[base + tid] = 1;
    }
    if (tid == 0) {
      s.time[rr] = tcount;
      cnt[CNT_SIM_STEPS] = sim_steps;
      cnt[CNT_SEQ] = seq_ctr;
      cnt[CNT_ARRIVED] = n_arr;
      cnt[CNT_DEPARTED] = n_dep;
      cnt[CNT_TOTAL_ARRIVED] = tot_arr;
      cnt[CNT_TOTAL_DEPARTED] = tot_dep;
      cnt[CNT_TOTAL_DROPPED] = tot_drop;
    }
    if (w == 0 && l < FS_MAX_INFLOWS) o.emitted[size_t(rr) * FS_MAX_INFLOWS + l] = emit_l;
    if (w == 0 && l < 20) o.arr_hist[size_t(rr) * 20 + l] = hist_l;
  }
}

}  // namespace fs
