// flowsim_open.h -- gfx950 step kernel for OPEN networks (FS_NET_MERGE): vehicles enter through inflows,
// leave at the end of their route, two routes converge at a priority junction.
//
// Reference behaviour restated (rule numbers = oracle/opennet.py, which is the bit-twin of this file):
//   O1  TraCIVehicle.update bookkeeping + sticky follower      flow/core/kernel/vehicle/traci.py:119-259
//   O2  MergePOEnv rl_veh / rl_queue, get_state, reward        flow/envs/merge.py:109-221
//   O3  MultiAgentMergePOEnv                                   flow/envs/multiagent/merge.py:86-171
//   O4  rewards.desired_velocity over the vehicles present     flow/core/rewards.py:6-59
//   O5  get_x_by_id through the edge-start table               flow/core/kernel/network/traci.py:273-289
//   M1-M7 the SUMO side (slots, inflow schedule, insertion, arrival, leader, right of way, movement)
//
// Lane mapping as in the closed-loop kernels: one 64-lane wave carries 64/SEG replicas, lane seg*SEG + i holds
// SLOT i of its replica in registers (a slot is free when its route is -1).  Neighbours change with every
// insertion / arrival / merge, so each sub-step every lane scans the other slots of its segment through the
// LDS crossbar (ds_bpermute), as k_steps_ml does; per-replica bookkeeping (inflow clocks, id counters, the
// rl_veh list) is computed redundantly by every lane of the segment from ballots.  All cross-lane reads are
// executed by the whole wave (loop bounds are launch constants), never under segment-divergent control flow.
#pragma once

namespace fs {

template <typename T>
struct OpenView {
  int32_t* seq;          // [R,N] place in the id list
  int32_t* origin;       // [R,N]
  int32_t* foll;         // [R,N] sticky follower slot
  int32_t* ctl_seq;      // [R,N] >= 0: in rl_veh
  int32_t* lead;         // [R,N] leader slot of the last update (get_leader)
  int32_t* arrived_rl;   // [R,N]
  T* foll_h;             // [R,N] "follower_headway"
  T* headway;            // [R,N] get_headway
  T* vmax;               // [R,N] per-vehicle maxSpeed of the SUMO car-following model (M10, setMaxSpeed)
  int32_t* arr_hist;     // [R,20] arrivals of the last 20 sub-steps (ring buffer indexed by sub-step % 20)
  int32_t* counters;     // [R,8]
  int32_t* emitted;      // [R,FS_MAX_INFLOWS]
  int32_t* generated;    // [R,FS_MAX_INFLOWS] vehicles a probabilistic inflow has generated so far (M2b)
  int32_t* episode;      // [R] resets of the replica since fs_create (-1 before the first): keys the entry-lane draws (M9)
  const uint8_t* init_alive;   // [R,N]
  const int32_t* slot_type;    // [N]
  // Launch constants the step loop reads are kept one-per-lane in a few VGPRs and fetched with v_readlane, so the
  // loop body makes no memory access for them: lane_tab[row][lane], rows = enum below (host: Sim::init_open)
  const T* lane_tab;
  const double* flow_tab_d;    // [3][64] lane f: period, begin, end of inflow f (the schedule is evaluated in double)
  const int32_t* flow_tab_i;   // [3][64] lane f: vehicle type, route, number (-1 = unlimited)
  int n_inflows, ma_apply_actions, n_rl_slots;
  int n_prob;                  // inflows with a probability instead of a period (flow_tab_d row 0 holds -(threshold + 1))
  double dt_d;
  int nseg[2];
  unsigned seg_internal[2];
  T max_cost_full;             // norm([target] * 64): the one entry that does not fit the 64-lane row
  T merge_x, box_in, end_x, net_length;
  // lane drops (M8-M10): lanes 2q, 2q+1 join at m1, the resulting lanes at m2 (two paths: m1 == m2 == merge_x)
  T m1, m2, zip_d, speed_limit;
  // bottleneck heads (O6 / O7)
  // lane-segments are stored as GROUPS of cells that differ only in the lane: lane g of the tables holds group g
  const T* cell_tab;           // [6][64] edge start / lo / hi of observation group g, of action group g
  const int32_t* cell_tab_i;   // [3][64] rows 0 / 1: first cell | lanes << 8 | first lane << 16 | is_last_segment << 24 of
                               // observation / action group g; row 2, lane k: the groups on route segment k,
                               // obs first | obs count << 8 | action first << 16 | action count << 24
  int n_obs_cells, n_act_cells, n_obs_groups, n_act_groups, obs_window, rew_window, obs_dim, track_followers;
  int obs_span, act_span;      // most groups any one segment (edge) holds
  // M11 simplified lane changing
  const int32_t* lc_auto;      // [N] 1: the slot's vehicle type changes lane on its own
  int lc_enabled, lc_cooldown;
  T lc_min_gain;
  T out_norm;                  // 2000 * scaling
};

enum {
  TAB_SEG_START = 0,    // lane r*16 + q: start of segment q of route r
  TAB_SEG_FLOW = 1,     // ... its Flow table coordinate
  TAB_SEG_SLOPE = 2,    // ... 1 or 0
  TAB_FL_XDEP = 3,      // lane f: insertion coordinate of inflow f (route start + departPos)
  TAB_FL_VDEP = 4,      // ... departSpeed
  TAB_FL_MINGAP = 5,    // ... minGap / tau / 2*sqrt(accel*decel) of its vehicle type (M3)
  TAB_FL_TAU = 6,
  TAB_FL_TWOSQRT = 7,
  TAB_MAX_COST = 8,     // lane n: norm([target] * n), n < 64
  TAB_ROWS = 9
};
enum { CELL_OBS_START = 0, CELL_OBS_LO = 1, CELL_OBS_HI = 2, CELL_ACT_START = 3, CELL_ACT_LO = 4, CELL_ACT_HI = 5 };

enum { CNT_SIM_STEPS = 0, CNT_SEQ = 1, CNT_CTL = 2, CNT_ARRIVED = 3, CNT_DEPARTED = 4, CNT_TOTAL_ARRIVED = 5,
       CNT_TOTAL_DEPARTED = 6, CNT_TOTAL_DROPPED = 7 };

// ---- launch constants of the step loop ---------------------------------------------------------------------
// Two homes for the tables, chosen by OpenTabs<T, IN_LDS>:
//   IN_LDS = false  one entry per lane in VGPRs, entry j fetched with v_readlane (no memory access in the loop);
//   IN_LDS = true   an LDS copy read with a uniform (or, for the gathers, lane-varying) address.
// The kernels use the LDS home.  The VGPR home is only sound while the table registers are never parked in AGPRs:
// hipcc brings an AGPR-held row back with v_accvgpr_read under the CURRENT exec mask right before the v_readlane,
// which then returns stale data for rows held by lanes that are inactive at that point (found by the f64 parity
// test of k_steps_wide).  The float64 kernels always spill; the float32 ones sit at 190-256 VGPRs, one edit away
// from it, and measured no faster with the VGPR home (C5 237 M, C4 15.5 M either way) -- so LDS for all of them.
template <typename T>
struct alignas(16) CellRow { T start, lo, hi; int meta; };      // one lane-segment group: a single LDS read in cell_of

template <typename T>
struct OpenTabsLds {
  CellRow<T> cpack[2][64];      // [observation | action][group]
  T tab[TAB_ROWS][64];
  T ctab[6][64];
  double ftd[3][64];
  int fti[3][64];
  int ctab_i[3][64];
};

template <typename T, bool IN_LDS>
struct OpenTabs;

template <typename T>
struct OpenTabs<T, false> {
  T tab[TAB_ROWS], ctab[6];
  double ftd[3];
  int fti[3], ctab_i[3];
  __device__ __forceinline__ void load(const OpenView<T>& o, int lane, bool cells, OpenTabsLds<T>*) {
#pragma unroll
    for (int r = 0; r < TAB_ROWS; ++r) tab[r] = o.lane_tab[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      ftd[r] = o.flow_tab_d[r * 64 + lane];
      fti[r] = o.flow_tab_i[r * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) ctab[r] = cells ? o.cell_tab[r * 64 + lane] : T(0);
#pragma unroll
    for (int r = 0; r < 3; ++r) ctab_i[r] = cells ? o.cell_tab_i[r * 64 + lane] : 0;
  }
  template <int ROW> __device__ __forceinline__ T t(int j) const { return read_lane(tab[ROW], j); }
  template <int ROW> __device__ __forceinline__ T c(int j) const { return read_lane(ctab[ROW], j); }
  template <int ROW> __device__ __forceinline__ int ci(int j) const { return read_lane_i(ctab_i[ROW], j); }
  template <int ROW> __device__ __forceinline__ double fd(int j) const { return read_lane(ftd[ROW], j); }
  template <int ROW> __device__ __forceinline__ int fi(int j) const { return read_lane_i(fti[ROW], j); }
  // entry j with a lane-varying j (whole wave active)
  template <int ROW> __device__ __forceinline__ T t_gather(int j) const { return __shfl(tab[ROW], j, 64); }
  template <int ROW> __device__ __forceinline__ T c_gather(int j) const { return __shfl(ctab[ROW], j, 64); }
  template <int ROW> __device__ __forceinline__ int ci_gather(int j) const { return __shfl(ctab_i[ROW], j, 64); }
  template <int KIND> __device__ __forceinline__ CellRow<T> cell_row(int g) const {
    CellRow<T> row;
    row.start = c_gather<KIND * 3 + 0>(g);
    row.lo = c_gather<KIND * 3 + 1>(g);
    row.hi = c_gather<KIND * 3 + 2>(g);
    row.meta = ci_gather<KIND>(g);
    return row;
  }
};

template <typename T>
struct OpenTabs<T, true> {
  const OpenTabsLds<T>* L;
  __device__ __forceinline__ void load(const OpenView<T>& o, int lane, bool cells, OpenTabsLds<T>* lds) {
#pragma unroll
    for (int r = 0; r < TAB_ROWS; ++r) lds->tab[r][lane] = o.lane_tab[r * 64 + lane];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      lds->ftd[r][lane] = o.flow_tab_d[r * 64 + lane];
      lds->fti[r][lane] = o.flow_tab_i[r * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) lds->ctab[r][lane] = cells ? o.cell_tab[r * 64 + lane] : T(0);
#pragma unroll
    for (int r = 0; r < 3; ++r) lds->ctab_i[r][lane] = cells ? o.cell_tab_i[r * 64 + lane] : 0;
#pragma unroll
    for (int kind = 0; kind < 2; ++kind) {
      CellRow<T> row;
      row.start = cells ? o.cell_tab[(kind * 3 + 0) * 64 + lane] : T(0);
      row.lo = cells ? o.cell_tab[(kind * 3 + 1) * 64 + lane] : T(0);
      row.hi = cells ? o.cell_tab[(kind * 3 + 2) * 64 + lane] : T(0);
      row.meta = cells ? o.cell_tab_i[kind * 64 + lane] : 0;
      lds->cpack[kind][lane] = row;
    }
    L = lds;
    __syncthreads();
  }
  template <int KIND> __device__ __forceinline__ CellRow<T> cell_row(int g) const { return L->cpack[KIND][g]; }
  template <int ROW> __device__ __forceinline__ T t(int j) const { return L->tab[ROW][j]; }
  template <int ROW> __device__ __forceinline__ T c(int j) const { return L->ctab[ROW][j]; }
  template <int ROW> __device__ __forceinline__ int ci(int j) const { return L->ctab_i[ROW][j]; }
  template <int ROW> __device__ __forceinline__ double fd(int j) const { return L->ftd[ROW][j]; }
  template <int ROW> __device__ __forceinline__ int fi(int j) const { return L->fti[ROW][j]; }
  template <int ROW> __device__ __forceinline__ T t_gather(int j) const { return L->tab[ROW][j]; }
  template <int ROW> __device__ __forceinline__ T c_gather(int j) const { return L->ctab[ROW][j]; }
  template <int ROW> __device__ __forceinline__ int ci_gather(int j) const { return L->ctab_i[ROW][j]; }
};

// (internal?, Flow table coordinate) of coordinate x on route r (O5): both routes are walked with wave-uniform
// loops and the lane keeps the result of its own route.
template <int NR, typename T, typename TABS>
__device__ __forceinline__ void route_lookup(const OpenView<T>& o, const TABS& tb, T x, int route, bool& internal,
                                             T& flow_x, int& seg_k) {
  internal = false;
  flow_x = T(0);
  seg_k = 0;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    int k = 0;
    T st = tb.template t<TAB_SEG_START>(r * 16), fs0 = tb.template t<TAB_SEG_FLOW>(r * 16),
      sl = tb.template t<TAB_SEG_SLOPE>(r * 16);
    for (int q = 1; q < o.nseg[r]; ++q) {
      const T sq = tb.template t<TAB_SEG_START>(r * 16 + q);
      const bool hit = x >= sq;
      k = hit ? q : k;
      st = hit ? sq : st;
      fs0 = hit ? tb.template t<TAB_SEG_FLOW>(r * 16 + q) : fs0;
      sl = hit ? tb.template t<TAB_SEG_SLOPE>(r * 16 + q) : sl;
    }
    if (NR == 1 || route == r) {
      internal = (o.seg_internal[r] >> k) & 1u;
      flow_x = fs0 + sl * (x - st);
      seg_k = k;
    }
  }
}
template <int NR, typename T, typename TABS>
__device__ __forceinline__ void route_lookup(const OpenView<T>& o, const TABS& tb, T x, int route, bool& internal,
                                             T& flow_x) {
  int k_unused;
  route_lookup<NR>(o, tb, x, route, internal, flow_x, k_unused);
}

// The route segment of a vehicle, FOLLOWED from sub-step to sub-step (vehicles only move forward) instead of searched
// by every reader: after a move the cached segment holds unless x has passed the next start; a vehicle inserted into
// the slot starts again from its route's first segment.  The table row of the current segment is cached in registers
// and re-read from the LDS tables (plain loads with a per-lane index: fine under any exec mask) only on a change.
// (o.nseg / o.seg_internal live in the kernel-argument segment: indexed with a lane's route they are a MEMORY load per
// use -- one per sub-step for `internal` -- so the two entries are read as scalars and selected)
template <typename T>
__device__ __forceinline__ int nseg_of(const OpenView<T>& o, int r) { return r == 0 ? o.nseg[0] : o.nseg[1]; }
template <typename T>
__device__ __forceinline__ unsigned seg_internal_of(const OpenView<T>& o, int r) { return r == 0 ? o.seg_internal[0] : o.seg_internal[1]; }

template <typename T, typename TABS>
struct RouteCursor {
  int k;
  T st, fs0, sl, next;
  __device__ __forceinline__ void refresh(const OpenView<T>& o, const TABS& tb, int r) {
    st = tb.template t_gather<TAB_SEG_START>(r * 16 + k);
    fs0 = tb.template t_gather<TAB_SEG_FLOW>(r * 16 + k);
    sl = tb.template t_gather<TAB_SEG_SLOPE>(r * 16 + k);
    const bool more = k + 1 < nseg_of(o, r);
    const T nx = tb.template t_gather<TAB_SEG_START>(r * 16 + (more ? k + 1 : k));
    next = more ? nx : T(3.0e38);
  }
  __device__ __forceinline__ void restart(const OpenView<T>& o, const TABS& tb, int r, T x) {
    k = 0;
    refresh(o, tb, r);
    while (x >= next) {
      k += 1;
      refresh(o, tb, r);
    }
  }
  __device__ __forceinline__ void follow(const OpenView<T>& o, const TABS& tb, int r, T x) {
    while (x >= next) {
      k += 1;
      refresh(o, tb, r);
    }
  }
  __device__ __forceinline__ bool internal(const OpenView<T>& o, int r) const { return (seg_internal_of(o, r) >> k) & 1u; }
  __device__ __forceinline__ T flow_x(T x) const { return fs0 + sl * (x - st); }
};

// O6: the lane-segment ("cell") of a vehicle at coordinate x on route segment seg_k, lane my_lane; -1 = none.  KIND 0:
// observation cells, 1: action cells.  Only the groups of the vehicle's own edge are tried (row 2 of the integer
// table says which), gathered with a lane-varying index: call with the whole wave active.
template <int KIND, typename T, typename TABS>
__device__ __forceinline__ int cell_of(const TABS& tb, int span, T x, int seg_k, int my_lane, bool eligible) {
  const int range = tb.template ci_gather<2>(seg_k);
  const int g0 = (range >> (KIND * 16)) & 0xff, cnt = (range >> (KIND * 16 + 8)) & 0xff;
  // (every test turns the candidate into -1 through VCC, and the groups of an edge are its segments -- disjoint stretches
  // (lo, hi], the position-0 rule picking the last one only where no stretch holds the position -- so the first hit of
  // the walk is the only hit: a maximum.  Written as `eligible & inside & ...` the eight tests were seven scalar
  // combinations of fresh masks per group, ~14 cycles each for a wave alone on its SIMD.  A row index beyond the edge's
  // groups repeats its first row)
  int cell = -1;
  for (int r = 0; r < span; ++r) {
    const int g = g0 + (r < cnt ? r : 0);
    const CellRow<T> row = tb.template cell_row<KIND>(g);
    const T pos = x - row.start;
    const unsigned meta = unsigned(row.meta);
    const int rel = my_lane - int((meta >> 16) & 0xffu);
    int c = int(meta & 0xffu) + rel;
    c = unsigned(rel) < ((meta >> 8) & 0xffu) ? c : -1;
    int c_in = pos > row.lo ? c : -1;
    c_in = pos <= row.hi ? c_in : -1;
    if (KIND == 0) {                                       // searchsorted(..) - 1 == -1: the last segment
      int c_last = (meta >> 24) != 0u ? c : -1;
      c_last = pos == T(0) ? c_last : -1;
      c_in = max(c_in, c_last);
    }
    cell = max(cell, c_in);
  }
  cell = cnt > 0 ? cell : -1;
  cell = eligible ? cell : -1;
  return cell;
}

// P = number of paths (entry lanes): 2 = MergeNetwork (each path has its own segment table), 4 = BottleneckNetwork
// (one table; lanes 2q / 2q+1 join at m1, the two resulting lanes at m2)
// CSET = 1: every slot is an IDM / RL / Sim-car-following controller (FLAG_IDM_SET), see control_accel_on
// PROB: some inflow is probabilistic (M2b) -- its own instantiation: the per-sub-step trial and the schedule's second
// form cost the deterministic configurations 7 % (C5) when they were run-time branches of one kernel
// PO: the MergePOEnv head (rl_veh / rl_queue bookkeeping, places -> action columns); the other merge head
// (MultiAgentMergePOEnv, C5) is its own instantiation without any of it
template <typename T, int SEG, int P, int CSET = 0, bool PROB = false, bool PO = false>
__global__ __launch_bounds__(64) void k_steps_open(DevView<T> s, OpenView<T> o, int num_steps,
                                                   const uint8_t* __restrict__ mask,
                                                   const float* __restrict__ actions, size_t act_stride,
                                                   float* __restrict__ obs, float* __restrict__ rew,
                                                   uint8_t* __restrict__ done, int obs_every_step,
                                                   int after_reset) {
  constexpr int RPW = 64 / SEG;
  constexpr int NR = (P == 2) ? 2 : 1;            // segment tables
  const T BIGV = T(3.0e38);
  const int lane_id = threadIdx.x;
  const int seg = lane_id / SEG;
  const int i = lane_id % SEG;
  const int segbase = seg * SEG;
  const int r = blockIdx.x * RPW + seg;
  const int N = s.N;
  const bool rvalid = r < s.R;
  const bool slot_ok = i < N;
  const bool valid = rvalid && slot_ok;
  const int rr = rvalid ? r : s.R - 1;
  const int ii = slot_ok ? i : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const int flags = CSET == 1 ? (s.flags & ~(FLAG_NEED_FOLLOWER | FLAG_NEED_MEAN | FLAG_HAS_LAC)) : s.flags;
  const int env = s.env;
  // the env families are tied to the network (validated by fs_create): P == 2 merge heads, P == 4 bottleneck heads;
  // making that a compile-time fact keeps each instantiation free of the other family's code and registers
  constexpr bool po_env = (P == 2) && PO;                 // (the host launches PO = (env == FS_ENV_MERGE_PO))
  const bool bn_env = (P == 4) && (SEG == 64) && (env == FS_ENV_BOTTLENECK_DV || env == FS_ENV_BOTTLENECK);
  const bool dv_env = (P == 4) && (SEG == 64) && (env == FS_ENV_BOTTLENECK_DV);
  const bool track_foll = (P == 2) ? true : (o.track_followers != 0);

  Slot<T> sl;
  sl.ctrl = s.ctrl[ii];
  sl.failsafe = s.failsafe[ii];
  sl.speed_mode = s.speed_mode[ii];
  sl.rl_index = s.rl_index[ii];
  sl.pis_index = -1;
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
  // CSET = 1 in float32: the controllers' divisions as div_core, their square roots taken here (flowsim_kernels.h idm_fd)
  constexpr bool FD = CSET == 1 && std::is_same<T, float>::value;
  // CSET = 2 (FS_MIXED, T = double): the same float32 car-following models, evaluated on the rounded speeds and gaps; their
  // accelerations enter the float64 integration.  Everything else -- positions, geometry, every decision -- is this
  // kernel's float64 arithmetic.
  constexpr bool MXC = CSET == 2;
  static_assert(!MXC || std::is_same<T, double>::value, "CSET = 2 is the float64 kernel's FS_MIXED form");
  FdSlot fd = FdSlot{0.0f, 0.0f};
  float fd_adt = 3.0e38f, fd_ddt = 3.0e38f;
  Slot<float> slf;
  slf.ctrl = sl.ctrl; slf.failsafe = sl.failsafe; slf.speed_mode = sl.speed_mode; slf.rl_index = sl.rl_index; slf.pis_index = sl.pis_index;
#pragma unroll
  for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) slf.p[k] = float(sl.p[k]);
  slf.noise = float(sl.noise); slf.delay = float(sl.delay); slf.max_accel = float(sl.max_accel); slf.max_decel = float(sl.max_decel);
  slf.length = float(sl.length); slf.sumo_tau = float(sl.sumo_tau); slf.sumo_min_gap = float(sl.sumo_min_gap);
  slf.sumo_max_speed = float(sl.sumo_max_speed);
  if constexpr (FD || MXC) fd = make_fd(slf);
  if constexpr (FD) {
    fd_adt = (sl.speed_mode & 2) ? sl.max_accel * s.dt : 3.0e38f;
    fd_ddt = (sl.speed_mode & 4) ? sl.max_decel * s.dt : 3.0e38f;
  }
  const int my_type = o.slot_type[ii];
  const bool is_rl = sl.ctrl == FS_CTRL_RL;
  constexpr bool TABS_IN_LDS = true;               // see OpenTabs above
  __shared__ typename std::conditional<TABS_IN_LDS, OpenTabsLds<T>, int>::type tabs_mem;
  OpenTabs<T, TABS_IN_LDS> tb;
  tb.load(o, lane_id, bn_env, reinterpret_cast<OpenTabsLds<T>*>(&tabs_mem));
  static_assert(TABS_IN_LDS, "RouteCursor reads the tables with per-lane indices under divergent control flow");
  RouteCursor<T, OpenTabs<T, TABS_IN_LDS>> cur;

  const bool live_replica = rvalid && (mask == nullptr || mask[rr] != 0);
  int tcount = s.time[rr];
  uint32_t nctr = s.noise_ctr[rr];
  int32_t* cnt = o.counters + size_t(rr) * 8;
  int sim_steps = cnt[CNT_SIM_STEPS], seq_ctr = cnt[CNT_SEQ], ctl_ctr = cnt[CNT_CTL];
  int n_arr = cnt[CNT_ARRIVED], n_dep = cnt[CNT_DEPARTED], tot_arr = cnt[CNT_TOTAL_ARRIVED],
      tot_dep = cnt[CNT_TOTAL_DEPARTED], tot_drop = cnt[CNT_TOTAL_DROPPED];
  // vehicles emitted so far by inflow f of this replica: held by lane f of the replica's segment (SEG >= 8)
  int emit_l = (i < FS_MAX_INFLOWS) ? o.emitted[size_t(rr) * FS_MAX_INFLOWS + i] : 0;
  // M2b probabilistic inflows (InFlows.add(probability=p), params.py:1103-1105; SUMO: one Bernoulli(p * step length)
  // trial per flow and step between begin and end): lane f of the segment makes inflow f's trial of the sub-step -- a
  // Philox word keyed by (sub-step, 2000 + f, global replica, episode) against a 32-bit threshold -- and counts the
  // vehicles generated; vehicle k of the flow is due once k < generated
  constexpr bool prob_any = PROB;
  const bool my_flow = i < o.n_inflows;            // lane f of a segment keeps inflow f's schedule (and evaluates it, M2)
  const double my_per = my_flow ? o.flow_tab_d[i] : 0.0;
  const bool my_prob = prob_any && my_per < 0.0;
  const uint32_t my_thr = my_prob ? uint32_t(-my_per - 1.0) : 0u;
  const double my_begin = my_flow ? o.flow_tab_d[64 + i] : 0.0, my_end = my_flow ? o.flow_tab_d[128 + i] : 0.0;
  const int my_number = my_flow ? o.flow_tab_i[128 + i] : 0;
  int gen_l = (prob_any && i < FS_MAX_INFLOWS) ? o.generated[size_t(rr) * FS_MAX_INFLOWS + i] : 0;
  const uint32_t episode = uint32_t(o.episode[rr]);
  double next_due = -1.0e300;                                   // unknown yet: the first sub-step evaluates the schedule

  T x = s.pos[idx];
  T v = s.vel[idx];
  if (s.st16 != nullptr) state16_load(s, idx, x, v);       // FS_F16S: the state of record is the half arrays
  int route = slot_ok ? s.lane[idx] : -1;
  int seq = o.seq[idx];
  int origin = o.origin[idx];
  int foll = o.foll[idx];
  T foll_h = o.foll_h[idx];
  int ctl_seq = slot_ok ? o.ctl_seq[idx] : -1;
  int arrived_rl = o.arrived_rl[idx];
  T prev_v = s.prev_vel[idx], last_acc = s.accel[idx];
  T cst = s.ctrl_state[idx];
  T vmax = o.vmax[idx];
  const auto seg_route = [&]() -> int { return (NR == 1 || route < 0) ? 0 : route; };   // segment table of my route
  cur.restart(o, tb, seg_route(), x);
  const bool lc_on = (P > 2) && (o.lc_enabled != 0);
  const bool my_lc_auto = lc_on && (o.lc_auto[ii] != 0);
  int last_lc = lc_on ? s.last_lc[idx] : 0;
  int lc_want = -1;                 // path wanted after the next move (M11); recomputed with every neighbour update
  T lc_gain = T(0);
  int hist_l = (bn_env && i < 20) ? o.arr_hist[size_t(rr) * 20 + i] : 0;     // arrivals of sub-step % 20 == lane
  bool just_arrived = false;
  auto shift_of = [&](T xx) -> int { return (xx >= o.m1 ? 1 : 0) + (xx >= o.m2 ? 1 : 0); };

  const T dt = s.dt;
  const int num_rl = s.num_rl;
  const int obs_dim = o.obs_dim;
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;

  // slots j that hold a vehicle in at least one replica of this wave, as a wave-uniform SEG-bit mask
  auto occupied_slots = [&]() -> unsigned long long {
    unsigned long long u = __ballot(route >= 0);
    if (SEG < 64) u |= u >> 32;
    if (SEG < 32) u |= u >> 16;
    if (SEG < 16) u |= u >> 8;
    return SEG == 64 ? u : (u & ((1ull << (SEG & 63)) - 1ull));
  };
  // ---- M5 + O1: neighbours through the ORDER of the vehicles ------------------------------------------------
  // Every sub-step each lane needs its leader (nearest vehicle ahead on its route) and, for the sticky follower
  // rule, the vehicles whose leader it is.  Instead of comparing all pairs, the lanes are ranked once by position
  // (one v_readlane + compare + add-with-carry per occupied slot); everything else is O(1) per lane on 64-bit
  // masks laid out in that order:
  //   rank       place of the vehicle in the order "x ascending, equal x: higher slot first"; free slots rank last
  //   sorted_slot  lane segbase+p holds the slot whose rank is p (one ds_permute push; the ranks of a segment are a
  //              permutation of 0..SEG-1)
  //   B[q], R1, R2  bit p: the vehicle of rank p is on path q / has passed the first / the second join (ballots over
  //              the (path, joins) key pushed to the rank's lane the same way)
  // leader(i)   = lowest set bit above rank_i of (B_route(i) | SH)
  // followers(X) = for each route, the highest set bit of B_r below rank_X, if that vehicle's leader is X
  int lead = -1;
  T vl = T(-1001), h = T(1000);
  bool has = false, lead_same_lane = false;
  // ---- the neighbour STRUCTURE is kept from sub-step to sub-step ---------------------------------------------------
  // Who my leader is, who could register as my follower and who stands next to me on the other lanes are functions of
  // (a) the ORDER of the vehicles by position, (b) each vehicle's key = (path, joins upstream of it) and (c) my
  // look-ahead region `la`.  Insertions, arrivals and lane changes change (b) (a free slot's key is 0xffff), crossing a
  // join or coming within zipper_distance of one changes (b) / (c), and the order changes only when some vehicle catches
  // up with its leader (see `neighbours` below for why nothing else has to be watched).  These are cheap to test after a
  // move (two integer compares and the sign of the leader gap that is needed anyway); while no lane of the wave reports a
  // change -- most sub-steps -- the masks and the slot numbers derived from them are those of the last evaluation, and a
  // step only re-reads the positions and speeds of the neighbours it already knows.  nb_structure() is the full
  // evaluation and the only writer of the kept values, so the two paths cannot drift apart.
#ifdef FS_PHASE_TIMERS
  long long nb_struct_cycles = 0;
  int nb_struct_calls = 0;
#endif
  int nb_key = -1, nb_la = -1;      // my key and look-ahead region at the last nb_structure (-1: none yet)
  int nb_succ = -1;                 // slot of the vehicle ranked directly above me (-1: I am the front vehicle / free)
  int nb_pred = -1;                 // ... directly below me (-1: I am the last vehicle / free)
  int nb_rank = 0;                  // my rank at the last nb_structure
  T nb_len_lead = T(0);             // length of my leader
  int nb_cand[P], nb_cseq[P];       // O1: per path, the nearest vehicle behind me (slot, id-list place) ...
  bool nb_celig[P];                 // ... and whether its leader is me
#pragma unroll
  for (int q = 0; q < P; ++q) { nb_cand[q] = -1; nb_cseq[q] = 0; nb_celig[q] = false; }
  auto nb_structure = [&]() {
    const bool alive = route >= 0;
    const T xr = alive ? x : BIGV;
    const unsigned long long segmask = SEG == 64 ? ~0ull : ((1ull << (SEG & 63)) - 1ull);
    const unsigned long long am = seg_ballot<SEG>(alive, seg);
    const int n_alive = __popcll(am);
    const int dead_rank = n_alive + __popcll(~am & segmask & ((1ull << i) - 1ull));
    const int my_key = alive ? (route | (shift_of(x) << 8)) : 0xffff;
    int rank = 0, sorted_slot, skey;
    {
      // float64 (round 3): the same machinery on the FLOAT32 image of the position -- rounding is monotone, so images that
      // ascend strictly order the float64 positions too; equal images (equal positions, or positions less than a float32 ulp
      // apart) are the "tie" below and are then counted exactly on the float64 positions.
      // float32: the order "x ascending, equal x: higher slot first" is that of the unsigned 64-bit key
      // (order-preserving image of x) : (SEG-1-slot).  Two vehicles with the SAME x are rare (two inflows releasing
      // at one coordinate in one sub-step), so the count runs on the 32-bit images alone -- v_readlane, v_cmp_lt_u32,
      // add-with-carry per slot, four slots per iteration; a free slot's image is the largest and never counts --
      // and is checked: tied vehicles get the same count, their pushes collide, and a rank lane stays unmarked.
      // Only then the exact 64-bit count runs.  (Written with || the two-compare form had compiled to two nested
      // exec-mask branches per slot, 125 cycles each: 35 % of C5.  x + 0 turns a -0.0 into +0.0, whose integer
      // images would otherwise differ.)
      const uint32_t xb = __float_as_uint(float(xr) + 0.0f);
      const uint32_t ord = (xb & 0x80000000u) ? ~xb : (xb | 0x80000000u);
      // The order rarely changes by more than vehicles of different paths passing each other (a queue on the minor
      // route is passed by every vehicle of the major one): unless a vehicle has just been inserted, the ranks of the
      // last evaluation are tried with every inverted ADJACENT pair exchanged, and the try is PROVEN before it is used
      // -- the keys pushed to their rank lanes must fill lanes 0 .. n_alive-1 and ascend strictly there (a strictly
      // ascending arrangement of the keys is unique, so it is the ranking the count would give).  Anything else --
      // newcomers, ties, a vehicle that passed two others -- fails the proof and is counted in full.
      bool counted = true;
      if (__ballot(alive && (nb_key < 0 || nb_key == 0xffff)) == 0ull) {
        const T x_succ = bperm(x, segbase + (nb_succ >= 0 ? nb_succ : ii));
        const T x_pred = bperm(x, segbase + (nb_pred >= 0 ? nb_pred : ii));
        const int up = (alive && nb_succ >= 0 && x_succ < x) ? 1 : 0;
        const int down = (alive && nb_pred >= 0 && x < x_pred) ? 1 : 0;
        const int rank_try = alive ? nb_rank + up - down : dead_rank;
        const uint32_t xs = uint32_t(__builtin_amdgcn_ds_permute((segbase + rank_try) << 2, int(alive ? ord : 0xffffffffu)));
        const uint32_t nx = uint32_t(dpp_i<DPP_WAVE_SHL1>(int(xs)));
        const bool bad = (i < n_alive) && ((xs == 0u) || ((i + 1 < n_alive) && !(xs < nx)));
        counted = __ballot(bad) != 0ull;
        rank = rank_try;
      }
      if (counted) {
        const unsigned long long occ = occupied_slots();
        int r0 = 0, r1 = 0, r2 = 0, r3 = 0;
        for (int q = 0; q < SEG; q += 4) {
          if (((occ >> q) & 0xFull) == 0ull) continue;
          r0 += (uint32_t(seg_read_i<SEG>(int(ord), q, seg)) < ord) ? 1 : 0;
          r1 += (uint32_t(seg_read_i<SEG>(int(ord), q + 1, seg)) < ord) ? 1 : 0;
          r2 += (uint32_t(seg_read_i<SEG>(int(ord), q + 2, seg)) < ord) ? 1 : 0;
          r3 += (uint32_t(seg_read_i<SEG>(int(ord), q + 3, seg)) < ord) ? 1 : 0;
        }
        rank = alive ? (r0 + r1) + (r2 + r3) : dead_rank;
        const int marked = __builtin_amdgcn_ds_permute((segbase + rank) << 2, 1);
        if (__ballot(marked == 0) != 0ull) {                      // some rank lane got no vehicle: a tie somewhere
          int rk = 0;
          if constexpr (sizeof(T) == 4) {
            const unsigned long long key = ((unsigned long long)ord << 32) | (unsigned long long)uint32_t(SEG - 1 - ii);
            for (unsigned long long u = occ; u; u &= u - 1ull) {
              const int j = __ffsll((long long)u) - 1;
              const unsigned long long kj = ((unsigned long long)uint32_t(seg_read_i<SEG>(int(ord), j, seg)) << 32) |
                                            (unsigned long long)uint32_t(SEG - 1 - j);
              rk += (kj < key) ? 1 : 0;
            }
          } else {                                                // float64: the exact count on the positions themselves
            for (unsigned long long u = occ; u; u &= u - 1ull) {
              const int j = __ffsll((long long)u) - 1;
              const T xj = seg_read<SEG>(xr, j, seg);
              rk += (int(xj < xr) | (int(xj == xr) & int(j > ii)));     // bitwise: no short-circuit branches
            }
          }
          rank = alive ? rk : dead_rank;
        }
      }
    }
    sorted_slot = __builtin_amdgcn_ds_permute((segbase + rank) << 2, i);
    // the same push for (path, joins upstream of the vehicle): once the lanes hold these in rank order the
    // per-path / per-region masks are plain ballots
    skey = __builtin_amdgcn_ds_permute((segbase + rank) << 2, my_key);
    const bool s_alive = skey != 0xffff;
    const unsigned long long bit = 1ull << rank;
    unsigned long long B[P];
#pragma unroll
    for (int q = 0; q < P; ++q) B[q] = seg_ballot<SEG>(s_alive && (skey & 0xff) == q, seg);
    const unsigned long long R1 = seg_ballot<SEG>(s_alive && (skey >> 8) >= 1, seg);
    const unsigned long long R2 = (P == 2) ? R1 : seg_ballot<SEG>(s_alive && (skey >> 8) >= 2, seg);
    unsigned long long ALL = 0ull, own = 0ull;
#pragma unroll
    for (int q = 0; q < P; ++q) {
      ALL |= B[q];
      own = (route == q) ? B[q] : own;
    }
    const unsigned long long pair = (P == 2) ? ALL : (route < 2 ? (B[0] | B[1 % P]) : (B[2 % P] | B[3 % P]));
    // M5 / M8: a vehicle of region g (number of joins upstream of IT) is on my lane if our paths agree after
    // max(g, la) joins, la = joins upstream of the point zipper_distance ahead of me
    const int la = shift_of(x + o.zip_d);
    const unsigned long long cl0 = la == 0 ? own : (la == 1 ? pair : ALL);
    const unsigned long long cl1 = la <= 1 ? pair : ALL;
    const unsigned long long above = rank >= 63 ? 0ull : (~0ull << (rank + 1));
    const unsigned long long m = alive ? (((~R1 & cl0) | (R1 & ~R2 & cl1) | (R2 & ALL)) & above) : 0ull;
    has = m != 0ull;
    const int lead_pos = has ? __ffsll((long long)m) - 1 : 0;
    const int lslot = __shfl(sorted_slot, segbase + lead_pos, 64);
    lead = has ? lslot : -1;
    const int lsrc = segbase + (has ? lslot : ii);
    nb_len_lead = bperm(sl.length, lsrc);
    T x_l = T(0);
    if (P > 2) {                                        // M8: does the leader share my PHYSICAL lane (collision check)
      const int p_l = __shfl(route, lsrc, 64);
      x_l = bperm(x, lsrc);
      const int sh_l = shift_of(x_l);
      lead_same_lane = has && ((route >> sh_l) == (p_l >> sh_l));
    } else {
      lead_same_lane = has;
    }
    // the vehicle ranked directly above me: while I stay behind it (and nothing else changes) the order stands
    {
      const int sidx = rank + 1 < SEG ? rank + 1 : SEG - 1;
      const int ss = __shfl(sorted_slot, segbase + sidx, 64);
      nb_succ = (alive && rank + 1 < n_alive) ? ss : -1;
      const int pp = __shfl(sorted_slot, segbase + (rank > 0 ? rank - 1 : 0), 64);
      nb_pred = (alive && rank > 0) ? pp : -1;
      nb_rank = rank;
    }
    nb_key = my_key;
    nb_la = la;
    if (lc_on) {
      // ---- M11: which adjacent lane (if any) this vehicle would like to continue on after the next move ----------
      const T h_now = has ? (x_l - x) - nb_len_lead : T(1000);          // (the headway nb_refresh is about to set)
      const bool internal = cur.internal(o, seg_route());
      const int g = shift_of(x);
      const int my_path = route < 0 ? 0 : route;
      const int lane = my_path >> g, n_lanes = P >> g;
      const bool ok0 = alive && my_lc_auto && !internal && g < 2 && la == g && n_lanes > 1 &&
                       (tcount - last_lc >= o.lc_cooldown);
      const unsigned long long below = bit - 1ull;
      const T two_sqrt = T(2) * tsqrt(sl.max_accel * sl.max_decel);
      T best_gain = -BIGV;
      int best_path = -1;
#pragma unroll
      for (int dl = -1; dl <= 1; dl += 2) {               // right first, so that left wins a tie
        const int tl = lane + dl;
        const bool valid_t = ok0 && tl >= 0 && tl < n_lanes;
        const int p2 = valid_t ? (tl << g) : 0;
        unsigned long long own2 = 0ull;
#pragma unroll
        for (int q = 0; q < P; ++q) own2 = (p2 == q) ? B[q] : own2;
        const unsigned long long pair2 = p2 < 2 ? (B[0] | B[1 % P]) : (B[2 % P] | B[3 % P]);
        const unsigned long long c0 = la == 0 ? own2 : (la == 1 ? pair2 : ALL);
        const unsigned long long c1 = la <= 1 ? pair2 : ALL;
        const unsigned long long ml = valid_t ? (((~R1 & c0) | (R1 & ~R2 & c1) | (R2 & ALL)) & above) : 0ull;
        // the lanes that feed the target lane at my position: path q with q >> g == tl
        const unsigned long long feed = g == 0 ? own2 : pair2;
        const unsigned long long mf = valid_t ? (feed & below) : 0ull;
        const bool has_l = ml != 0ull, has_f = mf != 0ull;
        const int lpos = has_l ? __ffsll((long long)ml) - 1 : 0;
        const int fpos = has_f ? 63 - __clzll((long long)mf) : 0;
        const int ls = __shfl(sorted_slot, segbase + lpos, 64), fs_ = __shfl(sorted_slot, segbase + fpos, 64);
        const T xl2 = bperm(x, segbase + ls), vl2 = bperm(v, segbase + ls), ll2 = bperm(sl.length, segbase + ls);
        const T xf2 = bperm(x, segbase + fs_), vf2 = bperm(v, segbase + fs_);
        const T gap_l = has_l ? (xl2 - x) - ll2 : T(1000.0);
        const T gap_f = has_f ? (x - xf2) - sl.length : T(1000.0);
        const T v_l = has_l ? vl2 : T(0), v_f = has_f ? vf2 : T(0);
        const T need_l = sl.sumo_min_gap + tmax(T(0), v * sl.sumo_tau + v * (v - v_l) / two_sqrt);
        const T need_f = sl.sumo_min_gap + tmax(T(0), v_f * sl.sumo_tau + v_f * (v_f - v) / two_sqrt);
        const bool safe = (!has_l || gap_l >= need_l) && (!has_f || gap_f >= need_f);
        const T gain = gap_l - h_now;
        const bool take = valid_t && safe && (gain >= o.lc_min_gain) && (gain >= best_gain);
        best_gain = take ? gain : best_gain;
        best_path = take ? p2 : best_path;
      }
      lc_want = best_path;
      lc_gain = best_path >= 0 ? best_gain : T(0);
    }
    // ---- O1: who could register as my follower (vehicle/traci.py:243-250): per path the nearest vehicle behind me
    if (track_foll) {
      const unsigned long long below = bit - 1ull;
#pragma unroll
      for (int r = 0; r < P; ++r) {
        const unsigned long long mb = B[r] & below;
        const bool has_c = alive && mb != 0ull;
        const int q = has_c ? 63 - __clzll((long long)mb) : 0;
        const int cslot = __shfl(sorted_slot, segbase + q, 64);
        const int c_lead = __shfl(lead, segbase + cslot, 64);
        nb_cseq[r] = __shfl(seq, segbase + cslot, 64);
        nb_cand[r] = has_c ? cslot : -1;
        nb_celig[r] = has_c && (c_lead == ii);
      }
    }
  };
  // The snapshot proper: leader speed and headway from the kept leader slot; the sticky follower entry of THIS vehicle
  // (vehicle/traci.py:232-250) from the kept candidates -- a candidate whose leader I am has the headway
  // (x_me - x_c) - length_me, the very expression its own lane evaluates, so its x is all that has to be fetched.
  //
  // WHEN the kept structure stands.  Which vehicle can be whose leader depends on the classes (path, joins upstream,
  // look-ahead region) of the two and on which of them is ahead.  Between two vehicles that matter to each other the one
  // behind has the other among its leader candidates, i.e. AT or BEYOND its leader: their order cannot change before a
  // vehicle has caught up with its own leader (vehicles only move forward; classes that differ in region or look-ahead
  // are ordered by position, so they cannot swap without a class change first).  Vehicles that do not matter to each
  // other -- different paths, both upstream of the zipper zone -- may pass each other freely: a queue on the minor route
  // is passed by every vehicle of the major one, which changes ranks and mask bits but no leader, no follower and no
  // slot number.  So the structure is re-evaluated when a class changed somewhere in the wave (insertion, arrival, lane
  // change, a join or zipper zone reached) or a vehicle is no longer strictly behind its leader; the positions just
  // fetched for the headway decide that, and the follower state is only touched once it is settled.
  // (lane-drop network, P > 2: the follower candidates are not kept from sub-step to sub-step -- with followers tracked
  // there every sub-step evaluates in full -- which keeps those instantiations inside the 256 VGPRs the code-generation
  // guard holds the float32 kernels to, tests/test_codegen.py)
  constexpr bool KEEP_FOLL = P == 2;
  auto neighbours = [&](bool live, bool follow) {
    const bool alive = route >= 0;
    if (!KEEP_FOLL) {
#pragma unroll
      for (int q = 0; q < P; ++q) { nb_cand[q] = -1; nb_cseq[q] = 0; nb_celig[q] = false; }
    }
    T x_l, v_l, x_c[P];
    auto fetch = [&]() {
      const int lsrc = segbase + (has ? lead : ii);
      x_l = bperm(x, lsrc);
      v_l = bperm(v, lsrc);
#pragma unroll
      for (int r = 0; r < P; ++r) x_c[r] = follow ? bperm(x, segbase + (nb_cand[r] >= 0 ? nb_cand[r] : ii)) : T(0);
    };
    fetch();
    const int key_now = alive ? (route | (shift_of(x) << 8)) : 0xffff;
    const bool changed = (key_now != nb_key) | (shift_of(x + o.zip_d) != nb_la) | (has & !(x < x_l));
    // M11 reads gaps on the adjacent lanes every sub-step: with lane changing on, the full evaluation always runs
    if (lc_on || (!KEEP_FOLL && follow) || __ballot(changed) != 0ull) {
#ifdef FS_PHASE_TIMERS
      const long long t0_ = clock64();
      nb_structure();
      fetch();
#if FS_PHASE_TIMERS != 3
      nb_struct_cycles += clock64() - t0_;
      nb_struct_calls += 1;
#endif
#else
      nb_structure();
      fetch();
#endif
    }
    vl = has ? v_l : T(-1001);                          // get_speed(None): the accessor's error value
    h = has ? (x_l - x) - nb_len_lead : T(1000);        // vehicle/traci.py:237
    if (!follow) return;
    const bool no_lead = alive && !has;
    const T start_h = no_lead ? T(1000) : foll_h;
    const int start_f = no_lead ? -1 : foll;
    T bestf = BIGV;
    int bseq = 0x7fffffff, bj = -1;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      const T c_h = (x - x_c[r]) - sl.length;
      const bool elig = nb_celig[r] && (has || nb_cseq[r] > seq);
      if (elig && (c_h < bestf || (c_h == bestf && nb_cseq[r] < bseq))) { bestf = c_h; bseq = nb_cseq[r]; bj = nb_cand[r]; }
    }
    const bool better = (bestf < start_h) && (bestf < BIGV);
    if (alive && live) {
      foll = better ? bj : start_f;
      foll_h = better ? bestf : start_h;
    }
  };
  // place of this slot in rl_veh (-1 if not in it): rank by order of joining, ghosts included (O2)
  auto ctl_rank = [&]() -> int {
    unsigned long long b = seg_ballot<SEG>(ctl_seq >= 0, seg);
    int rank = 0;
    for (int t = 0; t < num_rl; ++t) {
      const bool bit = b != 0ull;
      const int j = bit ? __ffsll((long long)b) - 1 : 0;
      b &= b - 1ull;
      const int cj = __shfl(ctl_seq, segbase + j, 64);
      if (bit && cj < ctl_seq) rank += 1;
    }
    return ctl_seq >= 0 ? rank : -1;
  };
  // the five features of the vehicle in this slot (merge.py:128-156)
  auto five = [&](T* f5) {
    const bool alive = route >= 0;
    const T fx = cur.flow_x(x);
    const int ld = alive ? lead : -1;
    const int fo = alive ? foll : -1;
    const int lsrc = segbase + (ld >= 0 ? ld : ii), fsrc = segbase + (fo >= 0 ? fo : ii);
    const T v_l = bperm(v, lsrc), fx_l = bperm(fx, lsrc), v_f = bperm(v, fsrc), h_f = bperm(h, fsrc);
    const T this_speed = alive ? v : T(-1001);
    const T lead_speed = ld >= 0 ? v_l : s.max_speed;
    const T lead_head = ld >= 0 ? fx_l - fx - sl.length : o.net_length;
    const T follow_speed = fo >= 0 ? v_f : T(0);
    const T follow_head = fo >= 0 ? h_f : o.net_length;
    f5[0] = this_speed / s.max_speed;
    f5[1] = (lead_speed - this_speed) / s.max_speed;
    f5[2] = lead_head / o.net_length;
    f5[3] = (this_speed - follow_speed) / s.max_speed;
    f5[4] = follow_head / o.net_length;
  };
  // get_outflow_rate over the last `window` sub-steps (vehicle/traci.py:500-505); the history is one count per lane
  auto outflow = [&](int window) -> T {
    const int n = tcount < window ? tcount : window;
    const int ago = (((tcount - 1 - i) % 20) + 20) % 20;
    const T mine = (i < 20 && ago < n) ? T(hist_l) : T(0);
    const T total = seg_sum<SEG>(mine);                   // small integers: exact in any order
    const T rate = (T(3600) * total) / (T(n > 0 ? n : 1) * dt);
    return n > 0 ? rate : T(0);
  };
  auto write_obs = [&](int rank) {
    if (bn_env) {
      if (env == FS_ENV_BOTTLENECK) {                    // bottleneck.py:481-483
        if (valid && ii == 0) orow[0] = 1.0f;
        return;
      }
      // ---- O6 get_state (bottleneck.py:868-924): which observation cell am I in ...
      const bool alive = route >= 0;
      const bool internal = cur.internal(o, seg_route());
      const int seg_k = cur.k;
      const int my_lane = (route < 0 ? 0 : route) >> shift_of(x);
      const int ocell = cell_of<0>(tb, o.obs_span, x, seg_k, my_lane, alive && !internal);
      // ... then lane c collects cell c: who is in it (one ballot per cell and class), then their speeds in slot order
      unsigned long long mh = 0ull, mr = 0ull;
      for (int c = 0; c < o.n_obs_cells; ++c) {
        const unsigned long long bh = __ballot(ocell == c && !is_rl);
        const unsigned long long br = __ballot(ocell == c && is_rl);
        if (lane_id == c) { mh = bh; mr = br; }
      }
      const int cnt_h = __popcll(mh), cnt_r = __popcll(mr);
      T sp_h = T(0), sp_r = T(0);
      while (__ballot(mh != 0ull) != 0ull) {
        const int j = mh ? __ffsll((long long)mh) - 1 : 0;
        const T vj = bperm(v, j);
        if (mh) sp_h = sp_h + vj;
        mh &= mh - 1ull;
      }
      while (__ballot(mr != 0ull) != 0ull) {
        const int j = mr ? __ffsll((long long)mr) - 1 : 0;
        const T vj = bperm(v, j);
        if (mr) sp_r = sp_r + vj;
        mr &= mr - 1ull;
      }
      const int C = o.n_obs_cells;
      const T nh = T(cnt_h) / T(20), nr = T(cnt_r) / T(20);          // NUM_VEHICLE_NORM
      const T mean_h = (cnt_h > 0 ? sp_h / (nh * T(20)) : T(0)) / T(50);
      const T mean_r = (cnt_r > 0 ? sp_r / (nr * T(20)) : T(0)) / T(50);
      const T of = outflow(o.obs_window) / T(2000.0);
      if (rvalid && lane_id < C) {
        orow[lane_id] = float(nh);
        orow[C + lane_id] = float(nr);
        orow[2 * C + lane_id] = float(mean_h);
        orow[3 * C + lane_id] = float(mean_r);
      }
      if (rvalid && lane_id == 0) orow[4 * C] = float(of);
      return;
    }
    T f5[5];
    five(f5);
    if (po_env) {
      const int n_ctl = __popcll(seg_ballot<SEG>(ctl_seq >= 0, seg));
      if (valid && rank >= 0 && rank < num_rl) {
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * rank + q] = float(f5[q]);
      }
      if (rvalid && i < num_rl && i >= n_ctl) {          // unfilled entries stay 0 (merge.py:126)
#pragma unroll
        for (int q = 0; q < 5; ++q) orow[5 * i + q] = 0.0f;
      }
    } else if (valid && is_rl) {
      const bool alive = route >= 0;
#pragma unroll
      for (int q = 0; q < 5; ++q) orow[5 * sl.rl_index + q] = alive ? float(f5[q]) : 0.0f;
    }
  };
  auto store_snapshot = [&]() {
    if (valid && live_replica) {
      o.lead[idx] = lead;
      o.headway[idx] = h;
    }
  };

  neighbours(false, false);
  NoiseBlock<T> nzb;
  nzb.init();

  // -DFS_PHASE_TIMERS (scripts/phase_open.py): cycles per section of the sub-step, summed per wave and left in the
  // replica's counters (which makes the handle useless for anything else -- a measurement build)
#ifdef FS_PHASE_TIMERS
  long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long ph_last = clock64();
#define FS_TICK(p_) do { const long long n_ = clock64(); ph_t[p_] += n_ - ph_last; ph_last = n_; } while (0)
#else
#define FS_TICK(p_) do {} while (0)
#endif

  if (num_steps == 0) {
    // after_reset: update(reset=True) registers the followers of the initial placement (vehicle/traci.py:219-250)
    if (after_reset) {
      neighbours(live_replica, track_foll);
      if (valid && live_replica) {
        o.foll[idx] = foll;
        o.foll_h[idx] = foll_h;
      }
    }
    write_obs(po_env ? ctl_rank() : -1);
    store_snapshot();
    return;
  }

  // Everything loaded above is waited for HERE.  Vector loads and stores share one counter (vmcnt) and complete in order:
  // left to hipcc, the wait for a value whose first use is inside the step loop (vmax, ctrl_state, ...) sits at that use --
  // s_waitcnt vmcnt(0) -- where from the second step on it waits for the previous step's observation stores to land.
  asm volatile("" :: "v"(x), "v"(v), "v"(route), "v"(seq), "v"(origin), "v"(foll), "v"(foll_h), "v"(ctl_seq), "v"(arrived_rl),
               "v"(prev_v), "v"(last_acc), "v"(cst), "v"(vmax), "v"(last_lc));
  for (int step = 0; step < num_steps; ++step) {
    const float* act = actions ? actions + size_t(step) * act_stride + size_t(rr) * num_rl : nullptr;
    // the step's actions, read ONCE (they are the same for its sims_per_step sub-steps; a load inside the sub-step loop
    // is a memory round trip on the critical path of every sub-step): the multi-agent head's own column per RL slot,
    // MergePOEnv's row one column per lane (its places -> columns mapping changes with the sub-steps)
    float act_own = 0.0f, act_row = 0.0f;
    if (act != nullptr) {
      if (po_env || dv_env) { if (i < num_rl) act_row = act[i]; }           // (num_rl <= SEG, fs_create)
      else if (o.ma_apply_actions && sl.ctrl == FS_CTRL_RL) act_own = act[sl.rl_index < 0 ? 0 : sl.rl_index];
    }
    bool crashed = false;
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      const bool live = live_replica && !crashed;
      const bool alive = route >= 0;
      // ---- controllers on the snapshot (S1) ------------------------------------------------------------
      T vf = T(0), hf = T(0), mean_v = T(0);
      if (flags & FLAG_NEED_FOLLOWER) {
        const int fsrc = segbase + (foll >= 0 ? foll : 0);
        vf = bperm(v, fsrc);
        hf = bperm(h, fsrc);
      }
      if (flags & FLAG_NEED_MEAN) {
        const int n_alive = __popcll(seg_ballot<SEG>(alive, seg));
        mean_v = seg_sum<SEG>(alive ? v : T(0)) / T(n_alive > 0 ? n_alive : 1);
      }
      const bool internal = cur.internal(o, seg_route());
      const int seg_k = cur.k;
      const bool on_edge = s.junction_mode ? !internal : true;
      // RL command (envs/base.py:355 runs before additional_command: the rl_veh list of the last sub-step)
      bool have_rl = false;
      T a_rl = T(0);
      int po_place = -1;                                    // my place in rl_veh as the last additional_command left it
      if (po_env) {
        po_place = ctl_rank();
        const int rank = po_place;
        have_rl = (act != nullptr) && is_rl && alive && rank >= 0 && rank < num_rl;
        const float a = bperm(act_row, segbase + (have_rl ? rank : 0));
        if (have_rl) a_rl = T(a);
      } else if (o.ma_apply_actions && act != nullptr && is_rl && alive) {
        const float a = act_own;
        have_rl = !(a != a);                             // NaN: no action for this vehicle this step
        a_rl = have_rl ? T(a) : T(0);
      }
      bool commanded = false;
      T g_now = T(0);
      if (flags & FLAG_HAS_NOISE) g_now = nzb.draw(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr, s.noise_exact != 0);
      T acc;
      if constexpr (FD) acc = control_accel_fd(s, sl, fd, flags, v, vl, h, has, on_edge, have_rl, a_rl, commanded, g_now);
      else if constexpr (MXC) acc = T(control_accel_fd(s, slf, fd, flags, float(v), float(vl), float(h), has, on_edge, have_rl,
                                                       float(a_rl), commanded, float(g_now)));
      else acc = control_accel_on<T, CSET, true>(s, sl, flags, v, vl, h, has, vf, hf, mean_v, on_edge, have_rl, a_rl,
                                                 live && slot_ok, rr, ii, nctr, cst, commanded, g_now);
      FS_TICK(0);
      // ---- O6: BottleneckDesiredVelocityEnv._apply_rl_actions (bottleneck.py:926-969) -------------------
      if (dv_env && act != nullptr) {
        const int my_lane = (route < 0 ? 0 : route) >> shift_of(x);
        const int acell = cell_of<1>(tb, o.act_span, x, seg_k, my_lane, alive && !internal);
        const float a_cell = bperm(act_row, segbase + (acell >= 0 ? acell : 0));
        T a = acell >= 0 ? T(a_cell) : T(0);
        if (s.clip_actions) a = tmin(tmax(a, s.act_lo), s.act_hi);
        T nxt = tmin(tmax(vmax + a, T(0.01)), T(23.0));
        nxt = acell >= 0 ? nxt : T(23.0);
        if (live && alive && is_rl) vmax = nxt;
      }
      // ---- MergePOEnv.additional_command (merge.py:189-221) --------------------------------------------
      if (po_env) {
        const bool alive_rl = alive && is_rl;
        // `for veh_id in self.rl_veh: if veh_id not in rl_ids: self.rl_veh.remove(veh_id)` (merge.py:206-208) removes
        // from the list it iterates: the entry behind a removed one is skipped, so of a RUN of consecutive entries
        // that have left only the 1st, 3rd, ... go in this pass; the others stay listed (ghost rows of error values
        // in get_state, merge.py:128-156) for another sub-step.  D = the departed entries as a mask over list places
        // (a seg_or of 1 << place), run = the departed entries directly in front of mine.
        {
          const int place = po_place;
          const bool gone = place >= 0 && !alive_rl;
          const unsigned D = seg_or<SEG>(gone ? (1u << (place & 31)) : 0u);
          const unsigned below = place > 0 ? ((1u << (place & 31)) - 1u) : 0u;
          const unsigned holes = ~D & below;                       // places in front of mine that have NOT left
          const int run = holes != 0u ? (place - 1) - (31 - __clz(int(holes))) : place;
          if (live && gone && (run & 1) == 0) ctl_seq = -1;
          if (live && place < 0) ctl_seq = -1;                     // (not listed)
        }
        const int n_ctl = __popcll(seg_ballot<SEG>(ctl_seq >= 0, seg));
        const int free_places = num_rl - n_ctl > 0 ? num_rl - n_ctl : 0;
        const bool queued = alive_rl && ctl_seq < 0;
        unsigned long long qb = seg_ballot<SEG>(queued, seg);
        int qrank = 0;
        for (int t = 0; t < o.n_rl_slots; ++t) {
          const bool bit = qb != 0ull;
          const int j = bit ? __ffsll((long long)qb) - 1 : 0;
          qb &= qb - 1ull;
          const int sj = __shfl(seq, segbase + j, 64);
          if (bit && sj < seq) qrank += 1;
        }
        const bool take = queued && (qrank < free_places) && live;
        const int n_take = __popcll(seg_ballot<SEG>(take, seg));
        if (take) ctl_seq = ctl_ctr + qrank;
        ctl_ctr += n_take;
      }
      // ---- M7: apply_acceleration + SUMO integration ---------------------------------------------------
      Slot<T> sm = sl;
      sm.sumo_max_speed = tmin(vmax, o.speed_limit);     // M10
      T v_sumo, v_new;
      if constexpr (FD) {
        // (one-instruction clamps, k_rollout_loop's form: a slot whose speed-mode bit is clear holds 3e38 in the clamp's place)
        const float next_vel = hmax(v + acc * dt, 0.0f);
        float vc = v + (next_vel - v) * s.ramp;
        v_sumo = sumo_speed_fd(v, vl, h, has, dt, sl, sm.sumo_max_speed, fd.ts_sumo);
        vc = hmin(vc, (sl.speed_mode & 1) ? v_sumo : 3.0e38f);
        vc = hmin(vc, v + fd_adt);
        vc = hmax(vc, v - fd_ddt);
        v_new = commanded ? vc : v_sumo;
      } else {
        T next_vel = tmax(v + acc * dt, T(0));
        T vc = v + (next_vel - v) * s.ramp;
        if constexpr (MXC)
          v_sumo = tmax(T(0), v + T(sumo_acc_fd(float(v), float(vl), float(h), has, slf, float(sm.sumo_max_speed), fd.ts_sumo)) * dt);
        else v_sumo = sumo_idm_speed(v, vl, h, has, dt, sm);
        if (sl.speed_mode & 1) vc = tmin(vc, v_sumo);
        if (sl.speed_mode & 2) vc = tmin(vc, v + sl.max_accel * dt);
        if (sl.speed_mode & 4) vc = tmax(vc, v - sl.max_decel * dt);
        v_new = commanded ? vc : v_sumo;
      }
      if (s.junction_on) {                               // M6: right of way at the merge
        const bool in_reach = alive && (x < o.merge_x);
        const bool major_busy = seg_any<SEG>(in_reach && route == 0 && (x >= o.box_in - s.j_time_gap * v), seg);
        const bool minor_in_box = seg_any<SEG>(in_reach && route == 1 && (x >= o.box_in), seg);
        const bool approaching = alive && (x >= o.box_in - s.j_lookahead) && (x < o.box_in);
        const bool yields = approaching && ((route == 1 && major_busy) || (route == 0 && minor_in_box));
        T stop;
        if constexpr (FD) stop = sumo_speed_fd(v, 0.0f, o.box_in - x, true, dt, sl, sm.sumo_max_speed, fd.ts_sumo);
        else if constexpr (MXC)
          stop = tmax(T(0), v + T(sumo_acc_fd(float(v), 0.0f, float(o.box_in - x), true, slf, float(sm.sumo_max_speed), fd.ts_sumo)) * dt);
        else stop = sumo_idm_speed(v, T(0), o.box_in - x, true, dt, sm);
        const T cap = yields ? stop : BIGV;
        if constexpr (FD) {
          v_new = hmin(v_new, ((sl.speed_mode & 1) || !commanded) ? cap : BIGV);
        } else {
          if ((sl.speed_mode & 1) || !commanded) v_new = tmin(v_new, cap);
        }
      }
      if (lc_on) {                                       // M11: the one lane change of this step, with the move
        const bool want = lc_want >= 0 && alive && live;
        const T gsel = want ? lc_gain : -BIGV;
        const T gmax = seg_max<SEG>(gsel);
        const unsigned long long wb = seg_ballot<SEG>(want && gsel == gmax, seg);
        const int win = wb ? __ffsll((long long)wb) - 1 : -1;
        if (slot_ok && ii == win) {
          route = lc_want;
          last_lc = tcount + 1;
        }
      }
      FS_TICK(1);
      T x_new = (s.integrator == FS_BALLISTIC) ? x + (v + v_new) / T(2) * dt : x + v_new * dt;
      const bool mv = live && alive;
      if (mv) {
        prev_v = v;
        last_acc = acc;
        x = x_new;
        v = v_new;
      }
      if (mv) cur.follow(o, tb, seg_route(), x);
      if (live) {
        tcount += 1;
        nctr += 1u;
        sim_steps += 1;
      }
      // ---- M4: arrivals -------------------------------------------------------------------------------
      const bool arrived = mv && (x >= o.end_x);
      if (live) arrived_rl = (arrived && is_rl) ? 1 : 0;
      if (arrived) route = -1;
      just_arrived = arrived;
      {
        const int na = __popcll(seg_ballot<SEG>(arrived, seg));
        if (live) { n_arr = na; n_dep = 0; }
        tot_arr += na;
        if (bn_env && live && i == (tcount - 1) % 20) hist_l = na;
      }
      FS_TICK(2);
      // ---- M2 / M3: insertions in InFlows order -------------------------------------------------------
      // (a rolled loop over the inflows; every per-flow constant comes out of a lane table, so the loop keeps no
      // scalar registers alive across the step loop)
      const double now = double(sim_steps - 1) * o.dt_d;
      // `next_due` = earliest scheduled time of the replica's next vehicles (their count / end limits aside): while
      // it lies ahead in every replica of the wave the whole loop is skipped -- most sub-steps
      if (prob_any) {                                    // M2b: this sub-step's trial of every probabilistic inflow
        uint32_t c0 = uint32_t(sim_steps - 1), c1 = uint32_t(2000 + i), c2 = s.rep0 + uint32_t(rr), c3 = 1u + 2u * episode;
        philox4x32_10(c0, c1, c2, c3, s.seed_lo, s.seed_hi);
        const bool gen = my_prob && live && (now >= my_begin) && (now <= my_end) &&
                         (my_number < 0 || gen_l < my_number) && (c0 < my_thr);
        gen_l += gen ? 1 : 0;
        if (seg_any<SEG>(gen, seg)) next_due = -1.0e300;  // a vehicle became due: the schedule is looked at
      }
      const bool any_due = __ballot(live && (next_due <= now)) != 0ull;
#if defined(FS_PHASE_TIMERS) && FS_PHASE_TIMERS == 3     // (diagnostic: how often and how long the schedule is looked at)
      const long long due_t0_ = clock64();
#endif
      if (any_due) {
        // Lane f of a segment evaluates inflow f of its replica -- all inflows at once, from the schedule constants the lane
        // holds -- and only the inflows that are due in SOME replica of the wave are walked (in InFlows order).  (The first
        // version walked every inflow and read its schedule from the tables: C5's highway inflow runs at the capacity of
        // its insertion rule, some vehicle is due in 43 % of the sub-steps, and the walk was 27 % of the sub-step.)
        // `t_mine`: when the inflow is next worth looking at, given its counters.
        auto schedule = [&](double& t_mine) -> bool {
          const int k_me = emit_l;
          const double due_t = my_prob ? (k_me < gen_l ? -1.0e300 : 1.0e300) : my_begin + double(k_me) * my_per;
          const bool open_me = my_flow && (my_prob || ((due_t <= my_end) && (my_number < 0 || k_me < my_number)));
          t_mine = open_me ? due_t : 1.0e300;
          return open_me && (my_prob ? (k_me < gen_l) : (due_t <= now));
        };
        double t_mine;
        const bool due_me = schedule(t_mine) && live;
        const unsigned long long due_w = __ballot(due_me);
        const unsigned long long due_seg = seg_ballot<SEG>(due_me, seg);       // bit f: inflow f is due in MY replica
        unsigned fm = 0u;
#pragma unroll
        for (int sg = 0; sg < RPW; ++sg) fm |= unsigned(due_w >> (sg * (SEG & 63))) & 0xffu;
        while (fm != 0u) {
          const int f = __ffs(int(fm)) - 1;
          fm &= fm - 1u;
          const int k = seg_read_i<SEG>(emit_l, f, seg);
          const bool due = ((due_seg >> f) & 1ull) != 0ull;
          // (all of the inflow's table entries first: seven LDS reads in flight together, one round trip -- read where they
          // are used, each waits out its own behind the ballots and reductions in between)
          const int typ = tb.template fi<0>(f);
          int route_f = tb.template fi<1>(f);
          const T x_dep = tb.template t<TAB_FL_XDEP>(f), v_dep = tb.template t<TAB_FL_VDEP>(f);
          const T two_sqrt = tb.template t<TAB_FL_TWOSQRT>(f), min_gap_f = tb.template t<TAB_FL_MINGAP>(f),
                  tau_f = tb.template t<TAB_FL_TAU>(f);
          const bool random_lane = route_f < 0;
          if (random_lane) {                               // M9: departLane = "random"
            uint32_t c0 = uint32_t(k), c1 = uint32_t(1000 + f), c2 = s.rep0 + uint32_t(rr), c3 = 1u + 2u * episode;
            philox4x32_10(c0, c1, c2, c3, s.seed_lo, s.seed_hi);
            route_f = int((uint64_t(c0 >> 8) * uint64_t(P)) >> 24);
          }
          const bool alive_now = route >= 0;
          const bool free_slot = !alive_now && slot_ok && (my_type == typ) && !just_arrived && ctl_seq < 0;   // (a listed ghost keeps its slot)
          const unsigned long long fb = seg_ballot<SEG>(free_slot, seg);
          const int slot = fb ? __ffsll((long long)fb) - 1 : 0;
          const int sj = tmax(shift_of(x), shift_of(x_dep + o.zip_d));
          const bool cand = alive_now && ((route >> sj) == (route_f >> sj));
          const T xm = seg_min<SEG>(cand ? x : BIGV);
          const unsigned long long cb = seg_ballot<SEG>(cand && x == xm, seg);
          const bool has_lead = cb != 0ull;
          const int j = has_lead ? __ffsll((long long)cb) - 1 : 0;
          const T back_j = bperm(x - sl.length, segbase + j);
          const T v_lead = bperm(v, segbase + j);
          const T gap = back_j - x_dep;
          T dq;                                            // (FD: the divisor is a slot type's 2 sqrt(a b), Sim::open_div_ok)
          if constexpr (FD) dq = div_core(v_dep * (v_dep - v_lead), two_sqrt);
          else dq = v_dep * (v_dep - v_lead) / two_sqrt;
          const T need = min_gap_f + tmax(T(0), v_dep * tau_f + dq);
          const bool ok = live && due && (fb != 0ull) && (!has_lead || gap >= need);
          if (ok && slot_ok && ii == slot) {
            x = x_dep;
            v = v_dep;
            prev_v = T(0);                                 // previous_speeds.get(veh_id, 0)
            cst = T(0);
            last_acc = T(0);
            route = route_f;
            last_lc = -(1 << 30);
            vmax = sl.sumo_max_speed;
            seq = seq_ctr;
            origin = f * (1 << 20) + k;
            foll = -1;
            foll_h = BIGV;
            ctl_seq = -1;
            cur.restart(o, tb, seg_route(), x);
          }
          if (ok) {
            seq_ctr += 1;
            n_dep += 1;
            tot_dep += 1;
          }
          // M9: a random-lane vehicle that does not fit when it is due is dropped, not retried
          const bool consumed = ok || (random_lane && live && due);
          if (consumed && i == f) emit_l = k + 1;
          if (consumed && !ok) tot_drop += 1;
        }
        double t_after;
        schedule(t_after);                               // with the counters as the insertions left them
        next_due = seg_min<SEG>(t_after);
      }
#if defined(FS_PHASE_TIMERS) && FS_PHASE_TIMERS == 3
      if (any_due) { nb_struct_cycles += clock64() - due_t0_; nb_struct_calls += 1; }
#endif
      FS_TICK(3);
      // ---- O1: new neighbour snapshot, sticky followers, collision check --------------------------------
      neighbours(live, track_foll);
      bool c = seg_any<SEG>((route >= 0) && has && lead_same_lane && (h < s.crash_gap), seg);
      if (s.junction_on) {
        const bool inside = (route >= 0) && (x >= o.box_in) && (x < o.merge_x);
        c = c || (seg_any<SEG>(inside && route == 0, seg) && seg_any<SEG>(inside && route == 1, seg));
      }
      if (env == FS_ENV_MERGE_MA) c = false;             // multiagent/base.py:188-190: crash = 0
      crashed = crashed || (c && live);
      FS_TICK(4);
    }

    // ---- get_state / compute_reward / done ---------------------------------------------------------------
    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      const int rank = po_env ? ctl_rank() : -1;
      write_obs(rank);
      const bool alive = route >= 0;
      const int n_alive = __popcll(seg_ballot<SEG>(alive, seg));
      T reward;
      if (bn_env) {                                      // bottleneck.py:474-478, 971-981
        reward = outflow(o.rew_window) / o.out_norm;
      } else if (s.evaluate) {                           // merge.py:161-162
        const T sum_v = seg_sum<SEG>(alive ? v : T(0));
        reward = n_alive > 0 ? sum_v / T(n_alive) : T(0);
      } else {
        // O4; n_alive differs between the replicas of a wave: a gather (ds_bpermute), not a v_readlane
        const T mc_lane = tb.template t_gather<TAB_MAX_COST>(n_alive & 63);
        const T max_cost = n_alive < 64 ? mc_lane : o.max_cost_full;
        const T dv = alive ? v - s.target_velocity : T(0);
        const T cost = tsqrt(seg_sum<SEG>(dv * dv));
        T cost1 = tmax(max_cost - cost, T(0)) / (max_cost + T(1.1920928955078125e-07));
        const bool bad = seg_any<SEG>(alive && (v < T(-100)), seg) || n_alive == 0;
        cost1 = bad ? T(0) : cost1;
        // small time headways (merge.py:172-180), summed in rl_veh order (MergePO) / slot order (multi-agent)
        const bool use = alive && is_rl && has && (v > T(0)) && (po_env ? rank >= 0 : true);
        const T t_headway = tmax(h / (use ? v : T(1)), T(0));
        const T term = tmin((t_headway - T(1)) / T(1), T(0));
        T cost2 = T(0);
        unsigned long long ub = seg_ballot<SEG>(use, seg);
        if (SEG == 64 && !po_env) {
          // one replica per wave: the mask is wave-uniform, so the terms are walked in slot order with v_readlane
          // (no LDS round trip per term) and only as many times as there are terms
          for (unsigned long long u = ub; u; u &= u - 1ull) cost2 = cost2 + read_lane(term, __ffsll((long long)u) - 1);
        } else {
          const int lim = po_env ? num_rl : o.n_rl_slots;
          for (int t = 0; t < lim; ++t) {
            const unsigned long long b = po_env ? seg_ballot<SEG>(use && rank == t, seg) : ub;
            const bool bit = b != 0ull;
            const int j = bit ? __ffsll((long long)b) - 1 : 0;
            ub &= ub - 1ull;
            const T tj = bperm(term, segbase + j);
            if (bit) cost2 = cost2 + tj;
          }
        }
        reward = tmax(cost1 + T(0.1) * cost2, T(0));
        reward = crashed ? T(0) : reward;
      }
      if (valid && ii == 0) {
        *rrow = float(reward);
        *drow = done_flag(tcount >= s.step_limit, crashed);
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
    FS_TICK(7);
  }

  if (valid && live_replica) {
    if (s.st16 != nullptr) state16_store(s, idx, x, v);
    else {
      s.pos[idx] = x;
      s.vel[idx] = v;
    }
    s.lane[idx] = route;
    s.prev_vel[idx] = prev_v;
    s.accel[idx] = last_acc;
    s.ctrl_state[idx] = cst;
    o.seq[idx] = seq;
    o.origin[idx] = origin;
    o.foll[idx] = foll;
    o.foll_h[idx] = foll_h;
    o.ctl_seq[idx] = ctl_seq;
    o.arrived_rl[idx] = arrived_rl;
    o.vmax[idx] = vmax;
    if (lc_on) s.last_lc[idx] = last_lc;
    o.lead[idx] = lead;
    o.headway[idx] = h;
    if (ii == 0) {
      s.time[rr] = tcount;
      s.noise_ctr[rr] = nctr;
      cnt[CNT_SIM_STEPS] = sim_steps;
      cnt[CNT_SEQ] = seq_ctr;
      cnt[CNT_CTL] = ctl_ctr;
      cnt[CNT_ARRIVED] = n_arr;
      cnt[CNT_DEPARTED] = n_dep;
      cnt[CNT_TOTAL_ARRIVED] = tot_arr;
      cnt[CNT_TOTAL_DEPARTED] = tot_dep;
      cnt[CNT_TOTAL_DROPPED] = tot_drop;
#ifdef FS_PHASE_TIMERS
      for (int q = 0; q < 8; ++q) cnt[q] = int(ph_t[q] >> 6);
      cnt[5] = int(nb_struct_cycles >> 6);          // (part of section 4)
      cnt[6] = nb_struct_calls;                     // full evaluations of the neighbour structure
#endif
    }
  }
#undef FS_TICK
  if (rvalid && live_replica && i < FS_MAX_INFLOWS) o.emitted[size_t(rr) * FS_MAX_INFLOWS + i] = emit_l;
  if (prob_any && rvalid && live_replica && i < FS_MAX_INFLOWS) o.generated[size_t(rr) * FS_MAX_INFLOWS + i] = gen_l;
  if (bn_env && rvalid && live_replica && i < 20) o.arr_hist[size_t(rr) * 20 + i] = hist_l;
}

// Env.reset of an open network: the initial vehicles back in their slots, every other slot free, clocks and
// id counters restarted (restart_instance: SUMO starts again at time 0, envs/base.py:430-470); S13: one step
// has run when reset returns.
template <typename T>
__global__ void k_reset_open(DevView<T> s, OpenView<T> o, const uint8_t* __restrict__ mask) {
  const int N = s.N;
  const size_t n = size_t(s.R) * N;
  for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < n; e += size_t(gridDim.x) * blockDim.x) {
    const int r = int(e / N), i = int(e % N);
    if (mask != nullptr && mask[r] == 0) continue;
    const uint8_t* al = o.init_alive + size_t(r) * N;
    const bool a = al[i] != 0;
    int ids = 0;                                   // id-list place = number of initial vehicles in lower slots
    for (int j = 0; j < i; ++j) ids += al[j] ? 1 : 0;
    s.pos[e] = s.init_pos[e];
    s.vel[e] = s.init_vel[e];
    s.prev_vel[e] = s.init_vel[e];
    s.accel[e] = T(0);
    s.ctrl_state[e] = T(0);
    s.lane[e] = a ? s.init_lane[e] : -1;
    o.seq[e] = a ? ids : 0;
    const int old_origin = o.origin[e];
    const bool po_keep = s.env == FS_ENV_MERGE_PO;
    o.origin[e] = a ? -1 - i : -1;
    o.foll[e] = -1;
    o.foll_h[e] = T(3.0e38);
    // MergePOEnv never clears rl_veh (merge.py:223-231 resets only leader / follower): the vehicles it lists at the end
    // of an episode stay listed into the next one -- an initial vehicle that is placed again keeps its place, every
    // other entry is a vehicle that no longer exists (a ghost row until additional_command has removed it, subject to
    // the skipping above).  A slot that an initial vehicle needs cannot also hold such a ghost: that entry is dropped.
    {
      const int old_ctl = o.ctl_seq[e];
      const bool same_vehicle = a && old_origin == -1 - i;
      o.ctl_seq[e] = (po_keep && old_ctl >= 0 && (same_vehicle || !a)) ? old_ctl : -1;
    }
    o.arrived_rl[e] = 0;
    o.vmax[e] = s.sumo_max_speed[i];
    s.last_lc[e] = -(1 << 30);
    o.lead[e] = -1;
    o.headway[e] = T(1000);
    if (i < FS_MAX_INFLOWS) {
      o.emitted[size_t(r) * FS_MAX_INFLOWS + i] = 0;
      o.generated[size_t(r) * FS_MAX_INFLOWS + i] = 0;
    }
    if (i == 0) {
      int total = 0;
      for (int j = 0; j < N; ++j) total += al[j] ? 1 : 0;
      int32_t* cnt = o.counters + size_t(r) * 8;
      cnt[CNT_SIM_STEPS] = 1;
      cnt[CNT_SEQ] = total;
      for (int q = 2; q < 8; ++q)
        if (!(q == CNT_CTL && s.env == FS_ENV_MERGE_PO)) cnt[q] = 0;      // (the join counter orders rl_veh: it goes on)
      for (int q = 0; q < 20; ++q) o.arr_hist[size_t(r) * 20 + q] = 0;
      s.time[r] = 0;
      o.episode[r] += 1;                           // a new episode draws new entry lanes (the reference re-seeds SUMO)
    }
  }
  // replicas with fewer than FS_MAX_INFLOWS slots: the remaining inflow counters
  if (N < FS_MAX_INFLOWS)
    for (size_t e = size_t(blockIdx.x) * blockDim.x + threadIdx.x; e < size_t(s.R) * FS_MAX_INFLOWS;
         e += size_t(gridDim.x) * blockDim.x) {
      const int r = int(e / FS_MAX_INFLOWS);
      if (mask == nullptr || mask[r] != 0) {
        o.emitted[e] = 0;
        o.generated[e] = 0;
      }
    }
}

}  // namespace fs
