// flowsim_wide.h -- the open-network step kernel for replicas with MORE THAN 64 vehicle slots (lane-drop
// BottleneckNetwork: at C4's demand the queue upstream of the drops holds ~200 vehicles).
//
// Same rules, in the same order of floating-point operations, as k_steps_open<T, 64, 4> (flowsim_open.h; rule
// numbers = oracle/opennet.py).  What changes is the mapping: one replica = one WORKGROUP of W waves, thread t
// holds slot t in registers, and everything the 64-slot kernel did through the wave (ballots, ds_bpermute,
// DPP reductions) goes through LDS instead:
//   * per-slot values other threads gather (x, v, length, route, leader, headway, seq) are mirrored in LDS arrays;
//   * a mask over slots / ranks is W 64-bit words, one ballot per wave, published to LDS and read back by everyone;
//   * a reduction is one per-wave partial (DPP / ballot inside the wave) + a combine over the W partials, in the
//     order of the xor-butterfly's next levels ((w0 + w1) + (w2 + w3)) so that oracle/rewards.py tree_sum holds.
// A value is published before a lds_barrier() and consumed after it; scratch words alternate between two
// buffers so that a fast wave's next publication never lands on words a slow wave is still reading.  Control flow
// around every barrier is block-uniform: a block holds ONE replica, so `live`, `due`, `crashed` are uniform.
#pragma once

namespace fs {

template <typename T>
struct alignas(16) CellQuad { T a, b, c, d; };

template <typename T, int W>
struct WideLds {
  static constexpr int NS = 64 * W;
  T x[NS];                      // position of the vehicle in slot j; BIGV when the slot is free
  T v[NS];
  T h[NS];                      // headway of the last neighbour update
  T len[NS];                    // launch constant
  int route[NS];
  int lead[NS];
  int seq[NS];
  int sorted_slot[NS];          // [rank] -> slot
  int skey[NS];                 // [rank] -> path | joins << 8, 0xffff for a free slot
  unsigned long long okey[NS];                        // the 64-bit ordering keys of the vehicles, compacted
  unsigned long long okey2[NS];                       // ... and in the order of the updated ranking (its proof)
  T cx[sizeof(T) == 4 ? 1 : NS];                      // float64, when the proof fails: their positions ...
  int cslot[sizeof(T) == 4 ? 1 : NS];                 // ... and slots, compacted (the exact count)
  unsigned long long rmask[10][W];       // masks over RANKS: path 0..P-1 (P <= 8), then passed the first / the second join
  unsigned long long comb[24][W];        // [path * 3 + look-ahead region]: the ranks that hold a leader candidate of that class
  unsigned long long words[2][2][W];
  T red_t[2][4][W];
  int red_i[2][4][W];
  unsigned long long cell[2][128][W];    // observation cells: [human | rl][cell][wave] = members in that wave
  static constexpr int CELL_CAP = 32;    // speeds a cell's member list holds (a lane-segment of ~100 m: <= ~20 vehicles)
  alignas(16) T cellv[2][128][sizeof(T) == 4 ? 1 : CELL_CAP]; // float64: [human | rl][cell][k]: speed of the k-th member in slot order
  int acc[4][128];                       // float32: per cell, vehicles (human, RL) and speed sums in 2^-16 m/s (human, RL)
  // launch constants, read with uniform / gathered addresses -- OpenTabs<T, IN_LDS> of flowsim_open.h, and why
  OpenTabsLds<T> tabs;
  int emitted[FS_MAX_INFLOWS];           // vehicles emitted so far by inflow f
  int generated[FS_MAX_INFLOWS];         // vehicles generated so far by a probabilistic inflow f (M2b)
  int hist[20];                          // arrivals of sub-step % 20
  float act[4][64];                      // the RL actions of this step, a copy per wave (written and read by that wave only)
};

// Workgroup barrier for data exchanged through LDS only.  __syncthreads() is a full fence: it also waits for the
// wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)) -- the observation rows written a few hundred instructions
// earlier, an HBM round trip on the critical path of every sub-step.  Nothing in this kernel reads back what it stored
// to global memory, so the barriers only order the LDS traffic.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int first_bit(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int last_bit(unsigned long long m) { return 63 - __clzll((long long)m); }

// P = entry lanes of the lane-drop network: 4 (4 -> 2 -> 1 lanes) or 8 (scaling 2: 8 -> 4 -> 2)
template <typename T, int W, int CSET = 0, int P = 4>
__global__ __launch_bounds__(64 * W) void k_steps_wide(DevView<T> s, OpenView<T> o, int num_steps,
                                                       const uint8_t* __restrict__ mask,
                                                       const float* __restrict__ actions, size_t act_stride,
                                                       float* __restrict__ obs, float* __restrict__ rew,
                                                       uint8_t* __restrict__ done, int obs_every_step,
                                                       int after_reset) {
  constexpr int NS = 64 * W;
  using ull = unsigned long long;
  __shared__ WideLds<T, W> L;
  const T BIGV = T(3.0e38);
  const int tid = threadIdx.x;
  const int w = tid >> 6;                 // wave of the block
  const int l = tid & 63;                 // lane of the wave
  const int rr = blockIdx.x;              // one replica per block
  const int N = s.N;
  const bool slot_ok = tid < N;
  const int ii = slot_ok ? tid : N - 1;
  const size_t idx = size_t(rr) * N + ii;
  const int flags = CSET == 1 ? (s.flags & ~(FLAG_NEED_FOLLOWER | FLAG_NEED_MEAN | FLAG_HAS_LAC)) : s.flags;
  const int env = s.env;
  const bool dv_env = env == FS_ENV_BOTTLENECK_DV;
  const bool track_foll = o.track_followers != 0;

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
  FdSlot fd = FdSlot{0.0f, 0.0f};
  if constexpr (FD) fd = make_fd(sl);
  float y_ts_sumo = 1.0f;                 // div_core_recip of the slot's 2 sqrt(accel decel): M11's four quotients
  if constexpr (FD) y_ts_sumo = div_core_recip(fd.ts_sumo);
  const int my_type = o.slot_type[ii];
  const bool is_rl = sl.ctrl == FS_CTRL_RL;
  constexpr bool TABS_IN_LDS = true;
  OpenTabs<T, TABS_IN_LDS> tb;
  tb.load(o, l, dv_env, &L.tabs);
  static_assert(TABS_IN_LDS, "RouteCursor reads the tables with per-lane indices under divergent control flow");
  RouteCursor<T, OpenTabs<T, TABS_IN_LDS>> cur;
  if (tid < FS_MAX_INFLOWS) {
    L.emitted[tid] = o.emitted[size_t(rr) * FS_MAX_INFLOWS + tid];
    L.generated[tid] = o.generated[size_t(rr) * FS_MAX_INFLOWS + tid];
  }
  // M2b (flowsim_open.h): thread f makes the per-sub-step trial of probabilistic inflow f
  const bool prob_any = o.n_prob > 0;
  const bool my_flow = prob_any && tid < o.n_inflows;
  const double my_per = my_flow ? o.flow_tab_d[tid] : 0.0;
  const bool my_prob = my_per < 0.0;
  const uint32_t my_thr = my_prob ? uint32_t(-my_per - 1.0) : 0u;
  const double my_begin = my_flow ? o.flow_tab_d[64 + tid] : 0.0, my_end = my_flow ? o.flow_tab_d[128 + tid] : 0.0;
  const int my_number = my_flow ? o.flow_tab_i[128 + tid] : 0;
  if (tid < 20) L.hist[tid] = o.arr_hist[size_t(rr) * 20 + tid];

  const bool live_replica = mask == nullptr || mask[rr] != 0;
  const uint32_t episode = uint32_t(o.episode[rr]);
  int tcount = s.time[rr];
  uint32_t nctr = s.noise_ctr[rr];
  int32_t* cnt = o.counters + size_t(rr) * 8;
  int sim_steps = cnt[CNT_SIM_STEPS], seq_ctr = cnt[CNT_SEQ];
  const int ctl_ctr = cnt[CNT_CTL];
  int n_arr = cnt[CNT_ARRIVED], n_dep = cnt[CNT_DEPARTED], tot_arr = cnt[CNT_TOTAL_ARRIVED],
      tot_dep = cnt[CNT_TOTAL_DEPARTED], tot_drop = cnt[CNT_TOTAL_DROPPED];

  T x = s.pos[idx];
  T v = s.vel[idx];
  int route = slot_ok ? s.lane[idx] : -1;
  int seq = o.seq[idx];
  int origin = o.origin[idx];
  int foll = o.foll[idx];
  T foll_h = o.foll_h[idx];
  int arrived_rl = o.arrived_rl[idx];
  T prev_v = s.prev_vel[idx], last_acc = s.accel[idx];
  T cst = s.ctrl_state[idx];
  T vmax = o.vmax[idx];
  cur.restart(o, tb, 0, x);                     // one segment table: every lane drives the same edges
  const bool lc_on = o.lc_enabled != 0;
  const bool my_lc_auto = lc_on && (o.lc_auto[ii] != 0);
  int last_lc = lc_on ? s.last_lc[idx] : 0;
  int lc_want = -1;
  T lc_gain = T(0);
  bool just_arrived = false;
  auto shift_of = [&](T xx) -> int { return (xx >= o.m1 ? 1 : 0) + (xx >= o.m2 ? 1 : 0); };

  const T dt = s.dt;
  const int obs_dim = o.obs_dim;
  const size_t step_rows = obs_every_step ? size_t(s.R) : 0;
  float* orow = obs + size_t(rr) * obs_dim;
  float* rrow = rew + rr;
  uint8_t* drow = done + rr;

  L.len[tid] = sl.length;
  if (sizeof(T) == 4) { L.okey[tid] = 0ull; L.okey2[tid] = 0ull; }      // (stamp 0: never a ranking update's)
  for (int k = tid; k < 2 * 128 * W; k += NS) (&L.cell[0][0][0])[k] = 0ull;
  for (int k = tid; k < 4 * 128; k += NS) (&L.acc[0][0])[k] = 0;
  int phase = 0;                           // scratch buffer of the next publication (block-uniform)
  double next_due = -1.0e300;              // see the inflow loop (unknown yet: the first sub-step looks)

  // -DFS_PHASE_TIMERS (scripts/phase_open.py c4): cycles per section of the sub-step, left in the replica's counters
#ifdef FS_PHASE_TIMERS
  long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long ph_last = clock64();
#if FS_PHASE_TIMERS == 2      // the fine sections of the neighbour update and the observation (phase_open.py c4fine)
#define FS_TICKF(p_) do { const long long n_ = clock64(); ph_t[p_] += n_ - ph_last; ph_last = n_; } while (0)
#define FS_TICKW(p_) do {} while (0)
#else
#define FS_TICKW(p_) do { const long long n_ = clock64(); ph_t[p_] += n_ - ph_last; ph_last = n_; } while (0)
#define FS_TICKF(p_) do {} while (0)
#endif
#else
#define FS_TICKW(p_) do {} while (0)
#define FS_TICKF(p_) do {} while (0)
#endif
  // ---- M5 + O1: neighbours through the ORDER of the vehicles (see k_steps_open) ----------------------------
  int lead = -1;
  T vl = T(-1001), h = T(1000);
  bool has = false, lead_same_lane = false;
  // `crash_out`: does any vehicle of the replica sit closer than crash_gap behind its leader on its own lane;
  // `arrived_now` / `na_out`: the arrivals of this sub-step are counted through the first barrier of the update
  int lc_winner = -1;                      // the slot that changes lane with the next move (block-uniform; M11's arbitration)
  int nb_rank = -1;                        // my rank at the last update (-1: none yet / the slot was free)
  unsigned gen = 0u;                       // stamp of the ranking try (block-uniform)
  int n_new = 0;                           // vehicles inserted in this sub-step (block-uniform) ...
  int new_slot[FS_MAX_INFLOWS];            // ... and their slots
#pragma unroll
  for (int q = 0; q < FS_MAX_INFLOWS; ++q) new_slot[q] = -1;
  int out_obs = 0, out_rew = 0;            // arrivals inside the observation's / the reward's outflow window (block-uniform)
  auto neighbours = [&](bool live, bool follow, bool& crash_out, bool arrived_now, bool count_arrivals, int& na_out) {
    const bool alive = route >= 0;
    const T xr = alive ? x : BIGV;
    L.x[tid] = xr;
    L.v[tid] = v;
    L.route[tid] = route;
    // rank: x ascending, equal x: higher slot first; free slots after the vehicles (any fixed order).  Only the
    // vehicles are compared: their keys are first compacted (ballot prefix inside the wave + the wave totals), so the
    // count runs over n_alive entries instead of all NS slots; a free slot ranks n_alive + (free slots below it).
    FS_TICKF(0);       // everything outside the update and the observation
    const ull am = __ballot(alive);
    const int bA = phase & 1;
    phase += 1;
    {
      const int na_w = __popcll(__ballot(arrived_now));
      if (l == 0) {
        L.red_i[bA][0][w] = __popcll(am);
        L.red_i[bA][1][w] = na_w;
      }
    }
    lds_barrier();
    int base = 0, n_alive = 0;
    na_out = 0;
#pragma unroll
    for (int ww = 0; ww < W; ++ww) {
      const int cw = L.red_i[bA][0][ww];
      base += ww < w ? cw : 0;
      n_alive += cw;
      na_out += L.red_i[bA][1][ww];
    }
    if (count_arrivals) {
      // get_outflow_rate's windows (vehicle/traci.py:500-505) as running sums: this sub-step's arrivals enter, the
      // entry that has left the window goes (read here, between two barriers; the ring itself is written after the next)
      const int eo = tcount - 1 - o.obs_window, er = tcount - 1 - o.rew_window;
      out_obs += na_out - (eo >= 0 ? L.hist[eo % 20] : 0);
      out_rew += na_out - (er >= 0 ? L.hist[er % 20] : 0);
    }
    FS_TICKF(1);       // first barrier (publish + wait)
    const int place = base + __popcll(am & ((1ull << l) - 1ull));    // vehicles in lower slots
    int rank = 0;
    bool rank_bad = false;                 // float32: this thread saw the updated ranking fail its proof
    ull rank_key = 0ull;
    {
      // The order is that of ONE unsigned 64-bit key -- float64 (round 3): the key holds the FLOAT32 image of the position,
      // which orders two vehicles correctly whenever their images differ (rounding is monotone); two neighbours of the
      // updated ranking with EQUAL images fail the proof below and the block counts exactly, on the float64 positions --
      // float32: the order is that of ONE unsigned 64-bit key, (order-preserving image of x) : (NS-1-slot), so a
      // pair costs a v_cmp_lt_u64 and an add-with-carry instead of two float compares and three mask operations
      // (x + 0 turns a -0.0 into +0.0, whose integer images would otherwise differ)
      const uint32_t xb = __float_as_uint(float(xr) + 0.0f);
      const uint32_t ord = (xb & 0x80000000u) ? ~xb : (xb | 0x80000000u);
      // Between two updates the order changes little (vehicles of neighbouring lanes pass each other, a few places at
      // most), so the ranking is UPDATED instead of counted: every vehicle writes its new key at the place its OLD rank
      // gives (the vehicles inserted in this sub-step in front -- they stand at the network's entry --, everyone else
      // moved up by their number; arrivals were the front vehicles), reads the keys RW places either side and moves by
      // the number of them that are now on the other side of it.  The result is PROVEN before it is used: the keys
      // written to their new places must fill places 0 .. n_alive-1 with this update's stamp and ascend strictly there
      // (a strictly ascending arrangement of the full keys is unique: it is the ranking the count gives).  Anything
      // else -- a vehicle that moved further than RW places, an insertion behind a standing vehicle -- fails the proof
      // and the block counts as before (the loop over the n_alive compacted keys was 19 % of C4's step).
      // The stamp sits between the position image and the slot, so it never decides a comparison of two fresh keys.
      constexpr int RW = 4;
      gen += 1u;
      const unsigned stamp = gen & 0xffffffu;
      const ull key = (ull(ord) << 32) | (ull(stamp) << 8) | ull(uint32_t(NS - 1 - tid));
      {
        int newer = 0;                                    // inserted vehicles that rank below me
        bool i_am_new = false;
#pragma unroll
        for (int q = 0; q < FS_MAX_INFLOWS; ++q) {
          i_am_new = i_am_new || (q < n_new && new_slot[q] == tid);
          newer += (q < n_new && new_slot[q] > tid) ? 1 : 0;
        }
        const int pos = i_am_new ? newer : nb_rank + n_new;
        bool lost = alive && (pos < 0 || pos >= n_alive || (!i_am_new && nb_rank < 0));
        if (alive && !lost) L.okey[pos] = key;
        lds_barrier();
        if (count_arrivals && tid == 0) L.hist[(tcount - 1) % 20] = na_out;
        int delta = 0;
#pragma unroll
        for (int d = 1; d <= RW; ++d) {
          const int ja = pos + d, jb = pos - d;
          const bool ina = alive && !lost && ja < n_alive, inb = alive && !lost && jb >= 0;
          const ull ka = L.okey[ina ? ja : 0], kb = L.okey[inb ? jb : 0];
          lost = lost || (ina && (unsigned(ka >> 8) & 0xffffffu) != stamp) || (inb && (unsigned(kb >> 8) & 0xffffffu) != stamp);
          delta += (ina && ka < key) ? 1 : 0;
          delta -= (inb && kb > key) ? 1 : 0;
        }
        FS_TICKF(2);   // keys at the old places, barrier, window
        int rank_try = pos + delta;
        lost = lost || (alive && (rank_try < 0 || rank_try >= n_alive));
        if (alive && !lost) L.okey2[rank_try] = key;
        lds_barrier();
        const ull k0 = L.okey2[tid], k1 = L.okey2[tid + 1 < NS ? tid + 1 : tid];
        rank_bad = lost || ((tid < n_alive) && ((unsigned(k0 >> 8) & 0xffffffu) != stamp || ((tid + 1 < n_alive) && !(k0 < k1))));
        if (sizeof(T) == 8) rank_bad = rank_bad || ((tid + 1 < n_alive) && unsigned(k0 >> 32) == unsigned(k1 >> 32));
        rank = lost ? 0 : rank_try;
        rank_key = key;
      }
    }
    const int my_key = alive ? (route | (shift_of(x) << 8)) : 0xffff;
    if (!alive) rank = n_alive + (tid - place);
    L.sorted_slot[rank] = tid;             // (a ranking that fails its proof leaves rubbish here: written again below)
    L.skey[rank] = my_key;
    {
      const int bP = phase & 1;
      phase += 1;
      const ull bw_ = __ballot(rank_bad);
      if (l == 0) L.words[bP][1][w] = bw_;
      lds_barrier();
      ull ball = 0ull;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) ball |= L.words[bP][1][ww];
      if (ball != 0ull) {                                 // block-uniform: count in full
        int rk = 0;
        if constexpr (sizeof(T) == 4) {
          if (alive) L.okey[place] = rank_key;
          lds_barrier();
#pragma unroll 8
          for (int j = 0; j < n_alive; ++j) rk += (L.okey[j] < rank_key) ? 1 : 0;
        } else {                                          // float64: x ascending, equal x: higher slot first
          if (alive) {
            L.cx[place] = xr;
            L.cslot[place] = tid;
          }
          lds_barrier();
#pragma unroll 8
          for (int j = 0; j < n_alive; ++j) {
            const T xj = L.cx[j];
            rk += ((xj < xr) || (xj == xr && L.cslot[j] > tid)) ? 1 : 0;
          }
        }
        rank = alive ? rk : rank;
        L.sorted_slot[rank] = tid;
        L.skey[rank] = my_key;
        lds_barrier();
      }
    }
    nb_rank = alive ? rank : -1;
    n_new = 0;
    FS_TICKF(3);       // proof, scatter, flag barrier (+ the count when it failed)
    FS_TICKW(3);
    const int skey = L.skey[tid];          // the key of the vehicle whose rank is my thread index
    const bool s_alive = skey != 0xffff;
    {
      ull bw[P + 2];
#pragma unroll
      for (int q = 0; q < P; ++q) bw[q] = __ballot(s_alive && (skey & 0xff) == q);
      bw[P] = __ballot(s_alive && (skey >> 8) >= 1);
      bw[P + 1] = __ballot(s_alive && (skey >> 8) >= 2);
      if (l == 0) {
#pragma unroll
        for (int q = 0; q < P + 2; ++q) L.rmask[q][w] = bw[q];
      }
      // the ranks of this wave's word that hold a leader candidate of class (path p, look-ahead region la): see cand_w
      if (l < 3 * P) {
        const int p = l / 3, la_ = l % 3;
        ull path_w = 0ull, pair_w = 0ull, quad_w = 0ull;
#pragma unroll
        for (int q = 0; q < P; ++q) {
          path_w |= (q == p) ? bw[q] : 0ull;
          pair_w |= ((q & ~1) == (p & ~1)) ? bw[q] : 0ull;
          quad_w |= ((q & ~3) == (p & ~3)) ? bw[q] : 0ull;
        }
        const ull R1 = bw[P], R2 = bw[P + 1];
        const ull c0 = la_ == 0 ? path_w : (la_ == 1 ? pair_w : quad_w);
        const ull c1 = la_ <= 1 ? pair_w : quad_w;
        L.comb[l][w] = (~R1 & c0) | (R1 & ~R2 & c1) | (R2 & quad_w);
      }
    }
    lds_barrier();
    FS_TICKW(4);
    FS_TICKF(4);       // masks
    // the masks stay in LDS (written again two barriers into the next call) and are fetched word by word
    auto Bw = [&](int q, int ww) -> ull { return L.rmask[q][ww]; };
    // lanes of path p / of the pair p joins first / of the four paths that share p's lane after both joins
    auto pathw = [&](int p, int ww) -> ull { return L.rmask[p & (P - 1)][ww]; };
    auto pairw = [&](int p, int ww) -> ull { return L.rmask[p & (P - 2)][ww] | L.rmask[(p & (P - 2)) | 1][ww]; };
    auto quadw = [&](int p, int ww) -> ull {
      const int q0 = p & (P - 4);
      return L.rmask[q0][ww] | L.rmask[q0 + 1][ww] | L.rmask[q0 + 2][ww] | L.rmask[q0 + 3][ww];
    };
    // vehicles on "my lane" of those ahead, word ww: see k_steps_open (M5 / M8); a vehicle that has passed j joins
    // shares my lane if our paths agree after max(j, la) joins
    // (M5 / M8, see k_steps_open: (~R1 & c0) | (R1 & ~R2 & c1) | (R2 & quad) with c0 / c1 by look-ahead region -- combined
    // once per wave word when the masks were published, so a search reads one word per wave instead of ten)
    auto cand_w = [&](int p, int la_, int ww) -> ull { return L.comb[(p & (P - 1)) * 3 + la_][ww]; };
    const int rw = rank >> 6, rb = rank & 63;
    auto above_w = [&](int ww) -> ull { return ww < rw ? 0ull : (ww > rw ? ~0ull : (rb == 63 ? 0ull : (~0ull << (rb + 1)))); };
    auto below_w = [&](int ww) -> ull { return ww < rw ? ~0ull : (ww > rw ? 0ull : ((1ull << rb) - 1ull)); };
    const int la = shift_of(x + o.zip_d);
    const int my_path = route < 0 ? 0 : route;
    // The nearest candidate ahead (lowest set bit above my rank) / the nearest feeder behind (highest set bit below it) of
    // a mask over ranks: looked for in the word of my rank and the next one (the previous one) first -- two LDS reads
    // instead of W, half the 64-bit logic -- and over all W words only when some lane of the wave found nothing there
    // although words beyond hold vehicles (a vehicle whose lane class has nobody within 64 .. 127 ranks: rare).  The
    // position is a property of the masks: both walks give the same one.
    const int last_w = n_alive > 0 ? (n_alive - 1) >> 6 : 0;              // the last word that holds a vehicle
    const int w_up = rw + 1 < W ? rw + 1 : rw, w_dn = rw > 0 ? rw - 1 : 0;
    bool walk_all = false;
    auto near_above = [&](int p, int la_, bool on) -> int {
      const int row = (p & (P - 1)) * 3 + la_;
      const ull a0 = rb == 63 ? 0ull : (L.comb[row][rw] & (~0ull << (rb + 1)));
      const ull a1 = rw + 1 < W ? L.comb[row][w_up] : 0ull;
      const int pos = a0 != 0ull ? rw * 64 + first_bit(a0) : (a1 != 0ull ? w_up * 64 + first_bit(a1) : -1);
      walk_all = walk_all || (on && pos < 0 && rw + 2 <= last_w);
      return on ? pos : -1;
    };
    auto full_above = [&](int p, int la_, bool on) -> int {
      int pos = -1;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const ull m = on ? (cand_w(p, la_, ww) & above_w(ww)) : 0ull;
        if (pos < 0 && m != 0ull) pos = ww * 64 + first_bit(m);
      }
      return pos;
    };
    int lead_pos = near_above(my_path, la, alive);
    // ---- M11's places (lane changing on): the nearest vehicle ahead and the nearest feeder behind on either adjacent lane
    const bool internal_lc = lc_on ? cur.internal(o, 0) : false;
    const int g = shift_of(x);
    const int lane = my_path >> g, n_lanes = P >> g;
    const bool ok0 = lc_on && alive && my_lc_auto && !internal_lc && g < 2 && la == g && n_lanes > 1 &&
                     (tcount - last_lc >= o.lc_cooldown);
    auto feed_w = [&](int p2, int ww) -> ull { return g == 0 ? pathw(p2, ww) : pairw(p2, ww); };   // the lanes that feed the target lane at my position
    auto near_below = [&](int p2, bool on) -> int {
      const ull b0 = feed_w(p2, rw) & ((1ull << rb) - 1ull);
      const ull b1 = rw > 0 ? feed_w(p2, w_dn) : 0ull;
      const int pos = b0 != 0ull ? rw * 64 + last_bit(b0) : (b1 != 0ull ? w_dn * 64 + last_bit(b1) : -1);
      walk_all = walk_all || (on && pos < 0 && rw >= 2);
      return on ? pos : -1;
    };
    auto full_below = [&](int p2, bool on) -> int {
      int pos = -1;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const ull mf = on ? (feed_w(p2, ww) & below_w(ww)) : 0ull;
        if (mf != 0ull) pos = ww * 64 + last_bit(mf);      // words ascend: the last hit is the highest rank
      }
      return pos;
    };
    int lpos_d[2] = {-1, -1}, fpos_d[2] = {-1, -1}, p2_d[2] = {0, 0};
    bool valid_d[2] = {false, false};
    if (lc_on) {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const int tl = lane + (d == 0 ? -1 : 1);
        valid_d[d] = ok0 && tl >= 0 && tl < n_lanes;
        p2_d[d] = valid_d[d] ? (tl << g) : 0;
        lpos_d[d] = near_above(p2_d[d], la, valid_d[d]);
        fpos_d[d] = near_below(p2_d[d], valid_d[d]);
      }
    }
    if (__ballot(walk_all) != 0ull) {                      // (wave-uniform, rare) the walks over all W words
      lead_pos = full_above(my_path, la, alive);
      if (lc_on) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          lpos_d[d] = full_above(p2_d[d], la, valid_d[d]);
          fpos_d[d] = full_below(p2_d[d], valid_d[d]);
        }
      }
    }
    has = lead_pos >= 0;
    const int lslot = L.sorted_slot[has ? lead_pos : 0];
    lead = has ? lslot : -1;
    const int lsrc = has ? lslot : ii;
    const T x_l = L.x[lsrc];
    const T v_l = L.v[lsrc];
    const T len_lead = L.len[lsrc];
    vl = has ? v_l : T(-1001);
    h = has ? (x_l - x) - len_lead : T(1000);
    {
      const int p_l = L.route[lsrc];
      const int sh_l = shift_of(x_l);
      lead_same_lane = has && ((route >> sh_l) == (p_l >> sh_l));
    }
    if (lc_on) {
      // ---- M11: which adjacent lane (if any) this vehicle would like to continue on after the next move ----------
      T two_sqrt;                                          // (FD: the slot's constant, its quotients as div_core -- the
      if constexpr (FD) two_sqrt = fd.ts_sumo;             // insertion test's form of the same expression, Sim::open_div_ok)
      else two_sqrt = T(2) * tsqrt(sl.max_accel * sl.max_decel);
      auto by_ts = [&](T n) -> T {
        if constexpr (FD) return div_core_by(n, two_sqrt, y_ts_sumo);
        else return n / two_sqrt;
      };
      T best_gain = -BIGV;
      int best_path = -1;
#pragma unroll
      for (int d = 0; d < 2; ++d) {                        // right first, so that left wins a tie
        const bool valid_t = valid_d[d];
        const int p2 = p2_d[d];
        const int lpos = lpos_d[d], fpos = fpos_d[d];
        const bool has_l = lpos >= 0, has_f = fpos >= 0;
        const int ls = L.sorted_slot[has_l ? lpos : 0], fs_ = L.sorted_slot[has_f ? fpos : 0];
        const T xl2 = L.x[ls], vl2 = L.v[ls], ll2 = L.len[ls];
        const T xf2 = L.x[fs_], vf2 = L.v[fs_];
        const T gap_l = has_l ? (xl2 - x) - ll2 : T(1000.0);
        const T gap_f = has_f ? (x - xf2) - sl.length : T(1000.0);
        const T v_l2 = has_l ? vl2 : T(0), v_f = has_f ? vf2 : T(0);
        const T need_l = sl.sumo_min_gap + tmax(T(0), v * sl.sumo_tau + by_ts(v * (v - v_l2)));
        const T need_f = sl.sumo_min_gap + tmax(T(0), v_f * sl.sumo_tau + by_ts(v_f * (v_f - v)));
        const bool safe = (!has_l || gap_l >= need_l) && (!has_f || gap_f >= need_f);
        const T gain = gap_l - h;
        const bool take = valid_t && safe && (gain >= o.lc_min_gain) && (gain >= best_gain);
        best_gain = take ? gain : best_gain;
        best_path = take ? p2 : best_path;
      }
      lc_want = best_path;
      lc_gain = best_path >= 0 ? best_gain : T(0);
    }
    FS_TICKW(5);
    FS_TICKF(5);       // leader search
    // publish what the follower rule and the crash check read of OTHER slots
    L.lead[tid] = lead;
    L.h[tid] = h;
    L.seq[tid] = seq;
    const int b1 = phase & 1;
    phase += 1;
    {
      const ull cw = __ballot(alive && has && lead_same_lane && (h < s.crash_gap));
      if (l == 0) L.words[b1][0][w] = cw;
    }
    // M11's arbitration (largest gain, lowest slot on a tie) rides on this barrier: the wishes were made just above; the
    // winner changes lane with the NEXT move if that sub-step is live (it had a publication and a barrier of its own there)
    if (lc_on) {
      const bool want = lc_want >= 0 && alive;
      const T gsel = want ? lc_gain : -BIGV;
      const T gmax_w = seg_max<64>(gsel);
      const ull wb = __ballot(want && gsel == gmax_w);
      if (l == 0) {
        L.red_t[b1][3][w] = gmax_w;
        L.red_i[b1][3][w] = wb ? w * 64 + first_bit(wb) : -1;
      }
    }
    lds_barrier();
    ull call = 0ull;
#pragma unroll
    for (int ww = 0; ww < W; ++ww) call |= L.words[b1][0][ww];
    crash_out = call != 0ull;
    if (lc_on) {
      int win = -1;
      T gbest = -BIGV;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const int cw = L.red_i[b1][3][ww];
        const T gw = L.red_t[b1][3][ww];
        if (cw >= 0 && (win < 0 || gw > gbest)) {        // waves ascend: the first one wins a tie
          win = cw;
          gbest = gw;
        }
      }
      lc_winner = win;
    }
    if (!follow) return;
    // ---- O1: the sticky follower entry of THIS vehicle (vehicle/traci.py:232-250) ----------------------
    const bool no_lead = alive && !has;
    const T start_h = no_lead ? T(1000) : foll_h;
    const int start_f = no_lead ? -1 : foll;
    T bestf = BIGV;
    int bseq = 0x7fffffff, bj = -1;
#pragma unroll
    for (int r = 0; r < P; ++r) {
      int q = -1;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const ull mb = Bw(r, ww) & below_w(ww);
        if (mb != 0ull) q = ww * 64 + last_bit(mb);
      }
      const bool has_c = alive && q >= 0;
      const int cslot = L.sorted_slot[has_c ? q : 0];
      const int c_lead = L.lead[cslot];
      const T c_h = L.h[cslot];
      const int c_seq = L.seq[cslot];
      const bool elig = has_c && (c_lead == ii) && (has || c_seq > seq);
      if (elig && (c_h < bestf || (c_h == bestf && c_seq < bseq))) { bestf = c_h; bseq = c_seq; bj = cslot; }
    }
    const bool better = (bestf < start_h) && (bestf < BIGV);
    if (alive && live) {
      foll = better ? bj : start_f;
      foll_h = better ? bestf : start_h;
    }
  };
  // get_outflow_rate over the last `window` sub-steps (vehicle/traci.py:500-505): `total_i` arrivals in the window
  auto window_sum = [&](int window) -> int {             // the ring walked once (launch start; the sub-steps keep sums)
    const int n = tcount < window ? tcount : window;
    const int newest = (((tcount - 1) % 20) + 20) % 20;  // entry of the last sub-step; entry q is (newest - q) mod 20 old
    int total_i = 0;
    for (int q = 0; q < 20; ++q) {
      int ago = newest - q;
      ago += ago < 0 ? 20 : 0;
      total_i += (ago < n) ? L.hist[q] : 0;
    }
    return total_i;
  };
  auto outflow = [&](int window, int total_i) -> T {
    const int n = tcount < window ? tcount : window;
    const T total = T(total_i);                           // small integers: the float sum of k_steps_open is exact
    const T rate = (T(3600) * total) / (T(n > 0 ? n : 1) * dt);
    return n > 0 ? rate : T(0);
  };
  auto write_obs = [&]() {
    if (env == FS_ENV_BOTTLENECK) {                      // bottleneck.py:481-483
      if (tid == 0) orow[0] = 1.0f;
      return;
    }
    // ---- O6 get_state (bottleneck.py:868-924): which observation cell am I in ...
    const bool alive = route >= 0;
    const bool internal = cur.internal(o, 0);
    const int seg_k = cur.k;
    const int my_lane = (route < 0 ? 0 : route) >> shift_of(x);
    const int ocell = cell_of<0>(tb, o.obs_span, x, seg_k, my_lane, alive && !internal);
    // ... then every vehicle enters itself in its cell's membership words (one LDS atomic per vehicle instead of two
    // ballots per cell; the words were cleared by their owner after the previous observation)
    const int C = o.n_obs_cells;
    FS_TICKF(6);       // tail of the update (crash word, followers) + the cell lookup
    const int cls = is_rl ? 1 : 0;
    if constexpr (sizeof(T) == 4) {
      // float32: the vehicles of a cell are counted and their speeds added by LDS integer atomics, the speeds in units of
      // 2^-16 m/s -- exact, so in any order (k_drop_queue's sum; oracle/opennet.py cell_sum = 'fixed', its float32 default
      // beyond 64 slots).  One barrier; the member lists below (two barriers, a list walk per cell) keep the reference's
      // order of additions for float64
      if (ocell >= 0) {
        atomicAdd(&L.acc[cls][ocell], 1);
        atomicAdd(&L.acc[2 + cls][ocell], int(rintf(float(v) * 65536.0f)));
      }
      lds_barrier();
      const int c = l * W + w;                           // (the W waves' cells side by side)
      if (c < C) {
        const int cnt_h = L.acc[0][c], cnt_r = L.acc[1][c];
        const T sp_h = T(L.acc[2][c]) * T(1.0f / 65536.0f), sp_r = T(L.acc[3][c]) * T(1.0f / 65536.0f);
        L.acc[0][c] = 0; L.acc[1][c] = 0; L.acc[2][c] = 0; L.acc[3][c] = 0;   // (the next entries come several barriers later)
        const T nh = div_out(T(cnt_h), 20.0), nr = div_out(T(cnt_r), 20.0);          // NUM_VEHICLE_NORM
        const T mean_h = div_out(cnt_h > 0 ? sp_h / (nh * T(20)) : T(0), 50.0);
        const T mean_r = div_out(cnt_r > 0 ? sp_r / (nr * T(20)) : T(0), 50.0);
        orow[c] = float(nh);
        orow[C + c] = float(nr);
        orow[2 * C + c] = float(mean_h);
        orow[3 * C + c] = float(mean_r);
      }
      const T of = div_out(outflow(o.obs_window, out_obs), 2000.0);
      if (tid == 64) orow[4 * C] = float(of);
      return;
    }
    if (ocell >= 0) atomicOr(&L.cell[cls][ocell][w], 1ull << l);
    lds_barrier();
    // ... and writes its speed at its place in the cell's member list: the place is its ordinal among the members of its
    // class in slot order (members in lower waves' words + lower lanes of its own word), so the list holds the speeds in
    // the order the reference adds them (its loop over the ids) and the cell's owner reads them back four per LDS read
    // instead of walking the masks with one dependent LDS read per member (28 % of C4's step)
    if (ocell >= 0) {
      int k = 0;
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        const ull word = L.cell[cls][ocell][ww];
        k += ww < w ? __popcll(word) : (ww == w ? __popcll(word & ((1ull << l) - 1ull)) : 0);
      }
      if (k < WideLds<T, W>::CELL_CAP) L.cellv[cls][ocell][k] = v;
    }
    lds_barrier();
    // cell c is collected by lane c / W of wave c % W: the W waves' collections run side by side
    const int my_cell = l * W + w;
    if (my_cell < C) {
      int cnt_h = 0, cnt_r = 0;
      T sp_h = T(0), sp_r = T(0);
      ull mh[W], mr[W];
#pragma unroll
      for (int ww = 0; ww < W; ++ww) {
        mh[ww] = L.cell[0][my_cell][ww];
        mr[ww] = L.cell[1][my_cell][ww];
        L.cell[0][my_cell][ww] = 0ull;                  // (the next entries come several barriers later)
        L.cell[1][my_cell][ww] = 0ull;
        cnt_h += __popcll(mh[ww]);
        cnt_r += __popcll(mr[ww]);
      }
      constexpr int CAP = WideLds<T, W>::CELL_CAP;
      auto list_sum = [&](int which, int cnt) -> T {
        T sp = T(0);
        for (int k0 = 0; k0 < cnt; k0 += 4) {
          const CellQuad<T> q = *reinterpret_cast<const CellQuad<T>*>(&L.cellv[which][my_cell][k0]);
          sp = sp + q.a;
          if (k0 + 1 < cnt) sp = sp + q.b;
          if (k0 + 2 < cnt) sp = sp + q.c;
          if (k0 + 3 < cnt) sp = sp + q.d;
        }
        return sp;
      };
      if (cnt_h <= CAP && cnt_r <= CAP) {
        sp_h = list_sum(0, cnt_h);
        sp_r = list_sum(1, cnt_r);
      } else {
        // a cell with more members than a list holds: the walk over the membership masks, in slot order
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          for (ull m = mh[ww]; m != 0ull; m &= m - 1ull) sp_h = sp_h + L.v[ww * 64 + first_bit(m)];
          for (ull m = mr[ww]; m != 0ull; m &= m - 1ull) sp_r = sp_r + L.v[ww * 64 + first_bit(m)];
        }
      }
      const T nh = div_out(T(cnt_h), 20.0), nr = div_out(T(cnt_r), 20.0);          // NUM_VEHICLE_NORM
      const T mean_h = div_out(cnt_h > 0 ? sp_h / (nh * T(20)) : T(0), 50.0);
      const T mean_r = div_out(cnt_r > 0 ? sp_r / (nr * T(20)) : T(0), 50.0);
      orow[my_cell] = float(nh);
      orow[C + my_cell] = float(nr);
      orow[2 * C + my_cell] = float(mean_h);
      orow[3 * C + my_cell] = float(mean_r);
    }
    const T of = div_out(outflow(o.obs_window, out_obs), 2000.0);
    if (tid == 64) orow[4 * C] = float(of);            // (each single-lane store costs its wave ~0.5 us: one per wave)
    // (the next write of the cell words is a whole sub-step, i.e. several barriers, away)
  };
  auto store_snapshot = [&]() {
    if (slot_ok && live_replica) {
      o.lead[idx] = lead;
      o.headway[idx] = h;
    }
  };

  // the actions of a step are read from global memory one step early (by the threads that own a column) and handed
  // over through LDS: the barriers of the neighbour update in between make them visible, no thread waits for HBM
  if (actions != nullptr && num_steps > 0 && l < s.num_rl) L.act[w][l] = actions[size_t(rr) * s.num_rl + l];
  bool crash_now = false;
  int na_unused = 0;
  neighbours(false, false, crash_now, false, false, na_unused);
  out_obs = window_sum(o.obs_window);       // (L.hist was loaded before the barriers of the update above)
  out_rew = window_sum(o.rew_window);
  FS_TICKW(7);

  if (num_steps == 0) {
    if (after_reset) {
      neighbours(live_replica, track_foll, crash_now, false, false, na_unused);
      if (slot_ok && live_replica) {
        o.foll[idx] = foll;
        o.foll_h[idx] = foll_h;
      }
    }
    write_obs();
    store_snapshot();
    return;
  }

  // Everything loaded above is waited for HERE.  Vector loads and stores share one counter (vmcnt) and complete in order:
  // left to hipcc, the wait for a value whose first use is inside the step loop (vmax, ctrl_state, ...) sits at that use --
  // s_waitcnt vmcnt(0) -- where from the second step on it waits for the previous step's observation stores to land.
  asm volatile("" :: "v"(x), "v"(v), "v"(route), "v"(seq), "v"(origin), "v"(foll), "v"(foll_h), "v"(arrived_rl), "v"(prev_v),
               "v"(last_acc), "v"(cst), "v"(vmax));
  for (int step = 0; step < num_steps; ++step) {
    const bool have_act = actions != nullptr;
    // the next step's action row: loaded here, stored to the wave's LDS copy after the sub-steps (stored at once, the
    // row's wave sat out an L2 / HBM round trip -- and the previous step's observation stores, same counter -- at the top
    // of every step while the others waited at the first barrier)
    const bool act_next = actions != nullptr && step + 1 < num_steps && l < s.num_rl;
    float a_pref = 0.0f;
    if (act_next) a_pref = actions[size_t(step + 1) * act_stride + size_t(rr) * s.num_rl + l];
    bool crashed = false;
    for (int sub = 0; sub < s.sims_per_step; ++sub) {
      const bool live = live_replica && !crashed;
      const bool alive = route >= 0;
      // ---- controllers on the snapshot (S1) ------------------------------------------------------------
      T vf = T(0), hf = T(0), mean_v = T(0);
      if (flags & FLAG_NEED_FOLLOWER) {
        const int fsrc = foll >= 0 ? foll : 0;
        vf = L.v[fsrc];
        hf = L.h[fsrc];
      }
      if (flags & FLAG_NEED_MEAN) {
        const int bm = phase & 1;
        phase += 1;
        const int na_w = __popcll(__ballot(alive));
        const T sv_w = seg_sum<64>(alive ? v : T(0));
        if (l == 0) {
          L.red_i[bm][0][w] = na_w;
          L.red_t[bm][0][w] = sv_w;
        }
        lds_barrier();
        int n_alive = 0;
        T part[W];
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          n_alive += L.red_i[bm][0][ww];
          part[ww] = L.red_t[bm][0][ww];
        }
        T tot = part[0];
        if (W == 2) tot = part[0] + part[1];
        if (W == 4) tot = (part[0] + part[1]) + (part[2 % W] + part[3 % W]);
        mean_v = tot / T(n_alive > 0 ? n_alive : 1);
      }
      const bool internal = cur.internal(o, 0);
      const int seg_k = cur.k;
      const bool on_edge = s.junction_mode ? !internal : true;
      // (the lane-drop heads hand no acceleration to an RL vehicle: with SUMO-driven humans -- every shipped bottleneck
      // experiment -- no vehicle is commanded and S4-S8's command path is dead: block-uniform skip)
      // (BottleneckAccelEnv: one acceleration column per RL slot, NaN = no command for that vehicle this step)
      const bool ma_cmd = o.ma_apply_actions != 0 && !dv_env && have_act;       // block-uniform
      const bool any_cmd = !(flags & FLAG_NO_FLOW_CTRL) || ma_cmd;
      bool have_rl = false;
      T a_rl = T(0);
      if (ma_cmd && is_rl && alive) {
        const float a = L.act[w][(sl.rl_index < 0 ? 0 : sl.rl_index) & 63];
        have_rl = !(a != a);
        a_rl = have_rl ? T(a) : T(0);
      }
      bool commanded = false;
      T acc = T(0);
      if (any_cmd) {
        if constexpr (FD) {
          const T g_now = (flags & FLAG_HAS_NOISE) ? gauss<T>(s.seed_lo, s.seed_hi, s.rep0 + uint32_t(rr), uint32_t(ii), nctr, s.noise_exact != 0) : T(0);
          acc = control_accel_fd(s, sl, fd, flags, v, vl, h, has, on_edge, have_rl, float(a_rl), commanded, g_now);
        } else {
          acc = control_accel_on<T, CSET>(s, sl, flags, v, vl, h, has, vf, hf, mean_v, on_edge, have_rl, a_rl, live && slot_ok,
                                          rr, ii, nctr, cst, commanded);
        }
      }
      FS_TICKW(6);      // (section 6 = tail of neighbours + controllers)
      // ---- O6: BottleneckDesiredVelocityEnv._apply_rl_actions (bottleneck.py:926-969) -------------------
      if (dv_env && have_act) {
        const int my_lane = (route < 0 ? 0 : route) >> shift_of(x);
        const int acell = cell_of<1>(tb, o.act_span, x, seg_k, my_lane, alive && !internal);
        T a = acell >= 0 ? T(L.act[w][acell & 63]) : T(0);
        if (s.clip_actions) a = tmin(tmax(a, s.act_lo), s.act_hi);
        T nxt = tmin(tmax(vmax + a, T(0.01)), T(23.0));
        nxt = acell >= 0 ? nxt : T(23.0);
        if (live && alive && is_rl) vmax = nxt;
      }
      // ---- M7: apply_acceleration + SUMO integration ---------------------------------------------------
      Slot<T> sm = sl;
      sm.sumo_max_speed = tmin(vmax, o.speed_limit);     // M10
      T v_sumo;
      if constexpr (FD) v_sumo = sumo_speed_fd(v, vl, h, has, dt, sl, sm.sumo_max_speed, fd.ts_sumo);
      else v_sumo = sumo_idm_speed(v, vl, h, has, dt, sm);
      T v_new = v_sumo;
      if (any_cmd) {
        T next_vel = tmax(v + acc * dt, T(0));
        T vc = v + (next_vel - v) * s.ramp;
        if (sl.speed_mode & 1) vc = tmin(vc, v_sumo);
        if (sl.speed_mode & 2) vc = tmin(vc, v + sl.max_accel * dt);
        if (sl.speed_mode & 4) vc = tmax(vc, v - sl.max_decel * dt);
        v_new = commanded ? vc : v_sumo;
      }
      T x_new = (s.integrator == FS_BALLISTIC) ? x + (v + v_new) / T(2) * dt : x + v_new * dt;
      const bool mv = live && alive;
      const bool arrived = mv && (x_new >= o.end_x);
      // ---- M11: the winner of the last update's arbitration changes lane with this move ---------------------------
      if (lc_on && live && slot_ok && ii == lc_winner) {
        route = lc_want;
        last_lc = tcount + 1;
      }
      if (mv) {
        prev_v = v;
        last_acc = acc;
        x = x_new;
        v = v_new;
      }
      if (mv) cur.follow(o, tb, 0, x);
      if (live) {
        tcount += 1;
        nctr += 1u;
        sim_steps += 1;
      }
      // ---- M4: arrivals (counted through the first barrier of the neighbour update below) ----------------
      if (live) arrived_rl = (arrived && is_rl) ? 1 : 0;
      if (arrived) route = -1;
      just_arrived = arrived;
      if (live) n_dep = 0;
      FS_TICKW(1);      // (section 1 = actions + integration + arbitration + move)
      // ---- M2 / M3: insertions in InFlows order -------------------------------------------------------
      const double now = double(sim_steps - 1) * o.dt_d;
      if (prob_any) {                                    // block-uniform
        if (my_prob && live) {
          uint32_t c0 = uint32_t(sim_steps - 1), c1 = uint32_t(2000 + tid), c2 = s.rep0 + uint32_t(rr), c3 = 1u + 2u * episode;
          philox4x32_10(c0, c1, c2, c3, s.seed_lo, s.seed_hi);
          const int gl = L.generated[tid];
          if ((now >= my_begin) && (now <= my_end) && (my_number < 0 || gl < my_number) && (c0 < my_thr))
            L.generated[tid] = gl + 1;
        }
        lds_barrier();
      }
      // `next_due` (block-uniform): the earliest time an inflow has to be looked at again; a schedule that lies ahead
      // skips the loop and its table reads -- most sub-steps (probabilistic inflows: always looked at)
      const bool look = live && (prob_any || next_due <= now);
      double nd = 1.0e300;
      for (int f = 0; look && f < o.n_inflows; ++f) {
        // (the schedule entries first, read together: one LDS round trip)
        const int k = L.emitted[f];
        const double per_f = tb.template fd<0>(f);
        const double begin_f = tb.template fd<1>(f), end_f = tb.template fd<2>(f);
        const int number = tb.template fi<2>(f);
        const bool prob_f = per_f < 0.0;
        const double due_t = begin_f + double(k) * per_f;
        const bool open_f = prob_f || ((due_t <= end_f) && (number < 0 || k < number));
        const bool due = prob_f ? (k < L.generated[f]) : (due_t <= now) && open_f;
        if (!(due && live)) {                            // block-uniform
          const double mine = prob_f ? -1.0e300 : (open_f ? due_t : 1.0e300);
          nd = mine < nd ? mine : nd;
          continue;
        }
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
        const bool free_slot = !alive_now && slot_ok && (my_type == typ) && !just_arrived;
        const int sj = tmax(shift_of(x), shift_of(x_dep + o.zip_d));
        const bool cand = alive_now && ((route >> sj) == (route_f >> sj));
        // per wave: its free slots, its rearmost candidate leader (lowest slot on equal x) and that one's back / speed
        const int bi = phase & 1;
        phase += 1;
        {
          const ull fbw = __ballot(free_slot);
          const T xm_w = seg_min<64>(cand ? x : BIGV);
          const ull cbw = __ballot(cand && x == xm_w);
          const int jl = cbw ? first_bit(cbw) : 0;
          const T back_w = bperm(x - sl.length, jl);
          const T vlead_w = bperm(v, jl);
          if (l == 0) {
            L.words[bi][0][w] = fbw;
            L.red_i[bi][0][w] = cbw ? 1 : 0;
            L.red_t[bi][0][w] = xm_w;
            L.red_t[bi][1][w] = back_w;
            L.red_t[bi][2][w] = vlead_w;
          }
        }
        lds_barrier();
        int slot = -1;
        bool has_lead = false;
        T xm = BIGV, back_j = T(0), v_lead = T(0);
#pragma unroll
        for (int ww = 0; ww < W; ++ww) {
          const ull fbw = L.words[bi][0][ww];
          if (slot < 0 && fbw != 0ull) slot = ww * 64 + first_bit(fbw);
          const bool hw = L.red_i[bi][0][ww] != 0;
          const T xw = L.red_t[bi][0][ww];
          if (hw && (!has_lead || xw < xm)) {              // strictly smaller: the lower wave keeps a tie
            has_lead = true;
            xm = xw;
            back_j = L.red_t[bi][1][ww];
            v_lead = L.red_t[bi][2][ww];
          }
        }
        const T gap = back_j - x_dep;
        T dq;                                            // (FD: the divisor is a slot type's 2 sqrt(a b), Sim::open_div_ok)
        if constexpr (FD) dq = div_core(v_dep * (v_dep - v_lead), two_sqrt);
        else dq = v_dep * (v_dep - v_lead) / two_sqrt;
        const T need = min_gap_f + tmax(T(0), v_dep * tau_f + dq);
        const bool ok = (slot >= 0) && (!has_lead || gap >= need);
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
          cur.restart(o, tb, 0, x);
        }
        if (ok) {
          seq_ctr += 1;
          n_dep += 1;
          tot_dep += 1;
#pragma unroll
          for (int q = 0; q < FS_MAX_INFLOWS; ++q) new_slot[q] = (q == n_new) ? slot : new_slot[q];
          n_new += 1;
        }
        // M9: a random-lane vehicle that does not fit when it is due is dropped, not retried
        const bool consumed = ok || random_lane;
        if (consumed && tid == 0) L.emitted[f] = k + 1;    // every thread read k before this iteration's barrier
        if (consumed && !ok) tot_drop += 1;
        {                                                // when this inflow is next worth looking at (k_steps_open's rule)
          const double t_next = begin_f + double(k + 1) * per_f;
          const bool more = (t_next <= end_f) && (number < 0 || k + 1 < number);
          const double mine = prob_f ? -1.0e300 : (consumed ? (more ? t_next : 1.0e300) : due_t);
          nd = mine < nd ? mine : nd;
        }
      }
      if (look) next_due = nd;
      FS_TICKW(2);
      // ---- O1: new neighbour snapshot, sticky followers, collision check --------------------------------
      bool c = false;
      int na = 0;
      neighbours(live, track_foll, c, arrived, live, na);
      if (live) n_arr = na;
      tot_arr += na;
      crashed = crashed || (c && live);
    }

    if (act_next) L.act[w][l] = a_pref;
    // ---- get_state / compute_reward / done ---------------------------------------------------------------
    const bool emit = obs_every_step || (step == num_steps - 1);
    if (emit) {
      write_obs();
      const T reward = outflow(o.rew_window, out_rew) / o.out_norm;       // bottleneck.py:474-478, 971-981
      if (tid == 64 * (W - 1)) {
        *rrow = float(reward);
        *drow = done_flag(tcount >= s.step_limit, crashed);
      }
      orow += step_rows * obs_dim;
      rrow += step_rows;
      drow += step_rows;
    }
    FS_TICKW(7);
    FS_TICKF(7);       // cell collection, divisions, stores, reward
  }

  if (slot_ok && live_replica) {
    s.pos[idx] = x;
    s.vel[idx] = v;
    s.lane[idx] = route;
    s.prev_vel[idx] = prev_v;
    s.accel[idx] = last_acc;
    s.ctrl_state[idx] = cst;
    o.seq[idx] = seq;
    o.origin[idx] = origin;
    o.foll[idx] = foll;
    o.foll_h[idx] = foll_h;
    o.ctl_seq[idx] = -1;
    o.arrived_rl[idx] = arrived_rl;
    o.vmax[idx] = vmax;
    if (lc_on) s.last_lc[idx] = last_lc;
    o.lead[idx] = lead;
    o.headway[idx] = h;
    if (tid == 0) {
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
#endif
    }
  }
#undef FS_TICKW
#undef FS_TICKF
  lds_barrier();
  if (live_replica && tid < FS_MAX_INFLOWS) {
    o.emitted[size_t(rr) * FS_MAX_INFLOWS + tid] = L.emitted[tid];
    o.generated[size_t(rr) * FS_MAX_INFLOWS + tid] = L.generated[tid];
  }
  if (live_replica && tid < 20) o.arr_hist[size_t(rr) * 20 + tid] = L.hist[tid];
}

}  // namespace fs
