// flowsim_sim.h -- host side of libflowsim.so shared by its translation units: the handle (SimBase / Sim<T>),
// allocation, table upload, kernel choice.  The library is compiled as several objects (flow_amd/build.py): flowsim.hip
// holds the C ABI, validation and everything but the step launches; flowsim_part.hip is compiled once per
// (precision, lanes-per-replica) pair and holds Sim<T>::launch_seg<SEG> / launch_wide<W> with the step kernels they
// instantiate (defined in flowsim_launch.h, which only the parts include).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "flowsim.h"
#include "flowsim_kernels.h"
#include "flowsim_open.h"
#include "flowsim_pair.h"
#include "flowsim_fig8.h"
#include "flowsim_ringrl.h"
#include "flowsim_policy.h"
#include "flowsim_wide.h"
#include "flowsim_queue_consts.h"


namespace fsim {

inline thread_local std::string g_err;

inline int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(FS_ERR_HIP, std::string(#expr) + " failed: " + hipGetErrorString(e_));   \
  } while (0)

struct SimBase {
  virtual ~SimBase() {}
  fs_config cfg{};
  std::vector<fs_vehicle_spec> veh;
  std::vector<fs_segment> segs;
  std::vector<fs_inflow> inflows;
  std::vector<fs_cell> obs_cells, act_cells;
  std::vector<int32_t> obs_perm;
  std::vector<uint8_t> init_alive;
  int obs_dim = 0;
  int act_dim = 0;
  int seg = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // host-API staging buffers (device)
  float* d_actions = nullptr;
  float* d_obs = nullptr;
  float* d_rew = nullptr;
  uint8_t* d_done = nullptr;
  uint8_t* d_mask = nullptr;
  float* d_dump = nullptr;      // scratch words the idle lanes of k_rollout_idm store to
  std::vector<void*> allocs;
  int after_reset = 0;          // open networks: the next zero-step launch follows a reset (update(reset=True))
  bool force_generic = false;   // FLOWSIM_FORCE_GENERIC=1: never take the specialised kernels (tests)
  bool no_fastdiv = false;      // FLOWSIM_NO_FASTDIV=1: keep the IEEE division sequence in k_rollout_idm
  int rollout_block = 512;      // threads per block of k_rollout_idm (FLOWSIM_ROLLOUT_BLOCK overrides; sweep: docs/HISTORY.md)
  bool f16s = false;            // FS_F16S: the state between launches is kept as halves (DevView::st16)
  bool mixed = false;           // FS_MIXED: float64 state, float32 controller arithmetic (k_rollout_pair<double>)
  bool no_pair = false;         // FLOWSIM_NO_PAIR=1: keep k_rollout_idm (one vehicle per lane) for the float rollout
  bool no_loop_kernel = false;  // FLOWSIM_NO_LOOP_KERNEL=1: keep the generic k_steps for segment-table loops (tests)
  bool no_loop_full = false;    // FLOWSIM_NO_LOOP_FULL=1: keep the run-time-flag instantiation of k_rollout_loop (tests)
  bool no_ring_rl = false;      // FLOWSIM_NO_RING_RL=1: keep the generic k_steps for IDM + RL rings (tests)
  bool no_queue = false;        // FLOWSIM_NO_QUEUE=1: keep k_steps_open / k_steps_wide for the open networks (tests)
  int* d_qflag = nullptr;       // k_drop_queue: bit 0 = a path held more than its 64 lanes (the launch's results are invalid)
  bool qflag_armed = false;     // a queue-order launch ran since the flag was last read
  // after a stream synchronisation: did a queue-order launch overflow a path?  (sticky: the handle is unusable then)
  int check_qflag() {
    if (!qflag_armed || !d_qflag) return FS_OK;
    int f = 0;
    HIP_TRY(hipMemcpy(&f, d_qflag, sizeof(int), hipMemcpyDeviceToHost));
    if (f != 0)
      return fail(FS_ERR_UNSUPPORTED, "k_drop_queue: a path of the lane-drop network held more than 64 vehicles (or more than 8 "
                                      "arrived in one sub-step): the results of this handle are invalid -- create it "
                                      "with FLOWSIM_NO_QUEUE=1 (slot-order kernels)");
    qflag_armed = false;
    return FS_OK;
  }
  bool has_user_ctrl = false;   // a FS_CTRL_USER slot: only a library built with the user's controller may launch it
  int pair_block = 256;         // threads per block of k_rollout_pair (FLOWSIM_PAIR_BLOCK overrides)
  const char* last_kernel = "";  // family of the step kernel the last launch_steps call chose (fs_last_kernel)

  virtual int launch_steps(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride,
                           float* obs, float* rew, uint8_t* done, int obs_every_step) = 0;
  virtual int launch_reset(const uint8_t* mask) = 0;
  // policy in the loop (flowsim_policy.h): obs == nullptr selects the eager form (act only)
  virtual int launch_policy(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs,
                            float* act, float* logp, float* rew, uint8_t* done) = 0;
  uint32_t* d_pol_ctr = nullptr;   // [R] actions sampled so far per replica (the policy's draw counters)
  virtual int get_state(int field, void* dst, size_t bytes) = 0;
  virtual int set_state(int field, const void* src, size_t bytes) = 0;
  virtual int add_vehicle(int replica, int slot, int route, double x, double speed) = 0;
};

template <typename T>
struct Sim : SimBase {
  fs::DevView<T> dv{};
  fs::OpenView<T> ov{};        // open networks (FS_NET_MERGE) only
  fs::QueueConsts qc{};        // the queue-order kernels' view of the network (flowsim_queue.h)
  bool open_net = false;
  std::vector<T> h_len;        // vehicle lengths (host copy, for FS_FIELD_HEADWAY)
  std::vector<T> h_ring_len;   // ring lengths (host copy, for the divisor verification)

  template <typename U>
  int dev_alloc(U** out, size_t count) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, (count ? count : 1) * sizeof(U)));
    allocs.push_back(p);
    *out = static_cast<U*>(p);
    return FS_OK;
  }

  template <typename U>
  int upload(const U** out, const std::vector<U>& host) {
    U* p = nullptr;
    int rc = dev_alloc(&p, host.size());
    if (rc) return rc;
    HIP_TRY(hipMemcpy(p, host.data(), host.size() * sizeof(U), hipMemcpyHostToDevice));
    *out = p;
    return FS_OK;
  }

  int init() {
    const int R = cfg.num_replicas, N = cfg.num_vehicles;
    const size_t RN = size_t(R) * N;
    int rc;
    if ((rc = dev_alloc(&dv.pos, RN))) return rc;
    if ((rc = dev_alloc(&dv.vel, RN))) return rc;
    if ((rc = dev_alloc(&dv.prev_vel, RN))) return rc;
    if ((rc = dev_alloc(&dv.accel, RN))) return rc;
    if ((rc = dev_alloc(&dv.ctrl_state, RN))) return rc;
    if ((rc = dev_alloc(&dv.lane, RN))) return rc;
    if ((rc = dev_alloc(&dv.last_lc, RN))) return rc;
    {
      std::vector<int32_t> il(RN, 0);
      if (cfg.init_lane)
        for (size_t e = 0; e < RN; ++e) il[e] = cfg.init_lane[e];
      if ((rc = upload(&dv.init_lane, il))) return rc;
    }
    if ((rc = dev_alloc(&dv.time, size_t(R)))) return rc;
    if ((rc = dev_alloc(&dv.sort_key, RN))) return rc;
    dv.sort_vehicles = cfg.sort_vehicles;
    dv.obs_perm = nullptr;
    if (!obs_perm.empty()) {
      if ((rc = upload(&dv.obs_perm, obs_perm))) return rc;
    }
    if ((rc = dev_alloc(&dv.noise_ctr, size_t(R)))) return rc;
    HIP_TRY(hipMemset(dv.noise_ctr, 0, size_t(R) * sizeof(uint32_t)));
    HIP_TRY(hipMemset(dv.time, 0, size_t(R) * sizeof(int32_t)));

    std::vector<T> ipos(RN), ivel(RN), rlen(R);
    for (size_t e = 0; e < RN; ++e) {
      ipos[e] = T(cfg.init_pos[e]);
      ivel[e] = cfg.init_vel ? T(cfg.init_vel[e]) : T(veh[e % N].initial_speed);
    }
    for (int r = 0; r < R; ++r) rlen[r] = cfg.ring_length ? T(cfg.ring_length[r]) : T(0);
    for (T vv : ivel) init_vel_negative = init_vel_negative || !(vv >= T(-100));
    neg_speed_possible = init_vel_negative;
    if ((rc = upload(&dv.init_pos, ipos))) return rc;
    if ((rc = upload(&dv.init_vel, ivel))) return rc;
    if ((rc = upload(&dv.ring_len, rlen))) return rc;
    if ((rc = upload(&dv.init_ring_len, rlen))) return rc;
    dv.st16 = nullptr;
    if (f16s && (rc = dev_alloc(&dv.st16, 3 * RN))) return rc;
    h_ring_len = rlen;

    std::vector<int32_t> ctrl(N), fsafe(N), smode(N), rli(N), pisi(N, -1);
    int n_pis = 0;
    std::vector<T> p(size_t(FS_MAX_CTRL_PARAMS) * N), noise(N), delay(N), maxa(N), maxd(N), len(N), stau(N),
        sgap(N), smax(N);
    int flags = 0;
    bool all_idm = true, idm_set = true;
    for (int i = 0; i < N; ++i) {
      const fs_vehicle_spec& v = veh[i];
      ctrl[i] = v.controller;
      fsafe[i] = v.fail_safe;
      smode[i] = v.speed_mode;
      rli[i] = v.rl_index;
      for (int k = 0; k < FS_MAX_CTRL_PARAMS; ++k) p[size_t(k) * N + i] = T(v.p[k]);
      noise[i] = T(v.noise);
      delay[i] = T(v.delay);
      maxa[i] = T(v.max_accel);
      maxd[i] = T(v.max_decel);
      len[i] = T(v.length);
      stau[i] = T(v.sumo_tau);
      sgap[i] = T(v.sumo_min_gap);
      smax[i] = T(v.sumo_max_speed);
      if (v.controller == FS_CTRL_BCM || v.controller == FS_CTRL_USER) flags |= fs::FLAG_NEED_FOLLOWER;
      if (v.controller == FS_CTRL_USER) has_user_ctrl = true;
      if (v.controller == FS_CTRL_NONLOCAL_FOLLOWER_STOPPER) flags |= fs::FLAG_NEED_MEAN;
      if (v.controller == FS_CTRL_LAC) flags |= fs::FLAG_HAS_LAC;
      if (v.controller == FS_CTRL_PISATURATION) {
        flags |= fs::FLAG_HAS_LAC;
        pisi[i] = n_pis++;
      }
      if (v.noise > 0 && v.controller != FS_CTRL_SIM && v.controller != FS_CTRL_RL) flags |= fs::FLAG_HAS_NOISE;
      if (v.fail_safe != FS_FAILSAFE_NONE) flags |= fs::FLAG_HAS_FAILSAFE;
      if (v.controller == FS_CTRL_SIM || v.controller == FS_CTRL_RL || (v.speed_mode & 1) || cfg.junction_mode)
        flags |= fs::FLAG_NEED_SUMO;
      if (v.speed_mode & 6) flags |= fs::FLAG_NEED_SUMO;
      if (v.controller == FS_CTRL_SIM || v.controller == FS_CTRL_RL || cfg.junction_mode) sumo_beyond_speed_mode = true;
      if (v.speed_mode & 7) speed_mode_any = true;
      if (v.controller == FS_CTRL_SIM) any_sim = true;
      if (v.controller != FS_CTRL_IDM) all_idm = false;
      if (v.controller != FS_CTRL_IDM && v.controller != FS_CTRL_RL && v.controller != FS_CTRL_SIM) idm_set = false;
    }
    loop_div_ok = true;      // premises of div_core in k_rollout_loop: s0 and minGap keep the dividends out of the tiny range
    for (int i = 0; i < N; ++i)
      loop_div_ok = loop_div_ok && float(veh[i].sumo_min_gap) >= 1e-3f && float(veh[i].sumo_min_gap) <= 1e6f &&
                    (veh[i].controller != FS_CTRL_IDM || (float(veh[i].p[5]) >= 1e-3f && float(veh[i].p[5]) <= 1e6f));
    loop_delta4 = true;
    for (int i = 0; i < N; ++i) loop_delta4 = loop_delta4 && (veh[i].controller != FS_CTRL_IDM || veh[i].p[4] == 4.0);
    if (loop_delta4) flags |= fs::FLAG_DELTA4;
    // premises of the open-network kernels' div_core forms (flowsim_kernels.h idm_fd / sumo_speed_fd)
    open_div_ok = loop_div_ok && (cfg.speed_limit <= 0 ||        // (0: no limit)
                                  (float(cfg.speed_limit) >= 9.5367431640625e-07f && float(cfg.speed_limit) <= 1048576.0f));
    for (int i = 0; i < N; ++i) {
      auto in_range = [](float c) { return c >= 9.5367431640625e-07f && c <= 1048576.0f; };      // [2^-20, 2^20]
      open_div_ok = open_div_ok && in_range(float(veh[i].sumo_max_speed)) &&
                    in_range(2.0f * std::sqrt(float(veh[i].max_accel) * float(veh[i].max_decel))) &&
                    (veh[i].controller != FS_CTRL_IDM ||
                     (in_range(float(veh[i].p[0])) && in_range(2.0f * std::sqrt(float(veh[i].p[2]) * float(veh[i].p[3])))));
    }
    {
      bool no_ctrl = true;
      for (int i = 0; i < N; ++i) no_ctrl = no_ctrl && (veh[i].controller == FS_CTRL_SIM || veh[i].controller == FS_CTRL_RL);
      if (no_ctrl) flags |= fs::FLAG_NO_FLOW_CTRL;
    }
    if (all_idm) flags |= fs::FLAG_ALL_IDM;
    if (idm_set) flags |= fs::FLAG_IDM_SET;
    delta4 = all_idm;
    for (int i = 0; i < N; ++i) delta4 = delta4 && (veh[i].p[4] == 4.0);
    h_len = len;
    if ((rc = upload(&dv.ctrl, ctrl))) return rc;
    if ((rc = upload(&dv.failsafe, fsafe))) return rc;
    if ((rc = upload(&dv.speed_mode, smode))) return rc;
    if ((rc = upload(&dv.rl_index, rli))) return rc;
    if ((rc = upload(&dv.pis_index, pisi))) return rc;
    dv.n_pis = n_pis;
    dv.pis_H = int(38.0 / cfg.sim_step) - 1;          // velocity_controllers.py:221
    if (dv.pis_H < 1) dv.pis_H = 1;
    if ((rc = dev_alloc(&dv.pis_hist, size_t(R) * (n_pis ? n_pis : 1) * (n_pis ? dv.pis_H : 1)))) return rc;
    if ((rc = dev_alloc(&dv.pis_n, size_t(R) * (n_pis ? n_pis : 1)))) return rc;
    HIP_TRY(hipMemset(dv.pis_n, 0, size_t(R) * (n_pis ? n_pis : 1) * sizeof(int32_t)));
    if ((rc = upload(&dv.p, p))) return rc;
    if ((rc = upload(&dv.noise, noise))) return rc;
    if ((rc = upload(&dv.delay, delay))) return rc;
    if ((rc = upload(&dv.max_accel, maxa))) return rc;
    if ((rc = upload(&dv.max_decel, maxd))) return rc;
    if ((rc = upload(&dv.length, len))) return rc;
    if ((rc = upload(&dv.sumo_tau, stau))) return rc;
    if ((rc = upload(&dv.sumo_min_gap, sgap))) return rc;
    if ((rc = upload(&dv.sumo_max_speed, smax))) return rc;

    dv.R = R;
    dv.N = N;
    dv.num_rl = cfg.num_rl;
    dv.env = cfg.env;
    dv.integrator = cfg.integrator;
    dv.sims_per_step = cfg.sims_per_step;
    dv.junction_mode = cfg.junction_mode;
    dv.clip_actions = cfg.clip_actions;
    dv.evaluate = cfg.evaluate;
    dv.track_aux = cfg.track_aux;
    dv.num_lanes = cfg.num_lanes < 1 ? 1 : cfg.num_lanes;
    dv.lane_change_mode = cfg.lane_change_mode;
    dv.last_lc_quirk = cfg.last_lc_quirk;
    dv.noise_exact = cfg.noise_exact != 0 ? 1 : 0;
    dv.lc_duration = T(cfg.lane_change_duration);
    {  // ML7: autonomous lane changing of the non-RL vehicles on a multi-lane ring
      std::vector<int32_t> lca(N, 0);
      int any_lc = 0;
      for (int i = 0; i < N; ++i) {
        lca[i] = (veh[i].lane_change_mode & 0x55) != 0 && veh[i].controller != FS_CTRL_RL;
        any_lc |= lca[i];
      }
      if ((rc = upload(&dv.lc_auto, lca))) return rc;
      dv.lc_enabled = (cfg.network == FS_NET_RING && cfg.num_lanes > 1 && any_lc) ? 1 : 0;
      dv.lc_cooldown = cfg.lane_change_cooldown_steps > 0 ? cfg.lane_change_cooldown_steps : 1;
      dv.lc_min_gain = T(cfg.lane_change_min_gain);
    }
    dv.nseg = cfg.num_segments;
    dv.seg_internal = 0u;
    for (int k = 0; k < FS_MAX_SEGMENTS; ++k) {
      const bool in = k < cfg.num_segments;
      dv.seg_start[k] = in ? T(segs[k].start) : T(0);
      dv.seg_flow_start[k] = in ? T(segs[k].flow_start) : T(0);
      dv.seg_flow_slope[k] = in ? T(segs[k].flow_slope) : T(0);
      if (in && segs[k].internal) dv.seg_internal |= (1u << k);
    }
    dv.junction_on = cfg.junction.enabled;
    dv.ja_in = T(cfg.junction.a_in);
    dv.ja_out = T(cfg.junction.a_out);
    dv.jb_in = T(cfg.junction.b_in);
    dv.jb_out = T(cfg.junction.b_out);
    dv.j_lookahead = T(cfg.junction.lookahead);
    dv.j_time_gap = T(cfg.junction.time_gap);
    dv.za_lo = T(cfg.junction.za_lo);
    dv.za_hi = T(cfg.junction.za_hi);
    dv.zb_lo = T(cfg.junction.zb_lo);
    dv.zb_hi = T(cfg.junction.zb_hi);
    if (cfg.horizon < 0) {
      dv.step_limit = INT_MAX;
    } else {
      long long lim = (long long)cfg.sims_per_step * ((long long)cfg.warmup_steps + cfg.horizon);
      dv.step_limit = lim > INT_MAX ? INT_MAX : int(lim);
    }
    dv.flags = flags;
    dv.seed_lo = uint32_t(cfg.seed & 0xFFFFFFFFull);
    dv.seed_hi = uint32_t(cfg.seed >> 32);
    dv.rep0 = uint32_t(cfg.replica_offset);
    dv.dt = T(cfg.sim_step);
    dv.ramp = T(cfg.slowdown_ramp);
    dv.jlen = T(cfg.junction_length);
    dv.crash_gap = T(cfg.crash_gap);
    dv.max_speed = T(cfg.max_speed);
    dv.target_velocity = T(cfg.target_velocity);
    {  // np.linalg.norm([target_velocity] * N) evaluated in double (rewards.py:50-51)
      double ss = 0.0;
      for (int i = 0; i < N; ++i) ss += cfg.target_velocity * cfg.target_velocity;
      dv.max_cost = T(std::sqrt(ss));
    }
    dv.act_lo = T(cfg.action_low);
    dv.act_hi = T(cfg.action_high);
    dv.po_max_length = T(cfg.po_max_length);

    open_net = (cfg.network == FS_NET_MERGE || cfg.network == FS_NET_BOTTLENECK);
    if (open_net && (rc = init_open())) return rc;

    // host-API staging
    if ((rc = dev_alloc(&d_actions, size_t(R) * (act_dim > 0 ? act_dim : 1)))) return rc;
    if ((rc = dev_alloc(&d_obs, size_t(R) * obs_dim))) return rc;
    if ((rc = dev_alloc(&d_rew, size_t(R)))) return rc;
    if ((rc = dev_alloc(&d_done, size_t(R)))) return rc;
    if ((rc = dev_alloc(&d_mask, size_t(R)))) return rc;
    if ((rc = dev_alloc(&d_dump, size_t(256)))) return rc;
    if ((rc = launch_reset(nullptr))) return rc;
    if (open_net) HIP_TRY(hipMemsetAsync(ov.episode, 0xFF, size_t(R) * sizeof(int32_t), stream));
    return FS_OK;
  }

  // ---- open networks: per-slot bookkeeping arrays, route tables, inflow table ------------------
  int init_open() {
    const int R = cfg.num_replicas, N = cfg.num_vehicles;
    const size_t RN = size_t(R) * N;
    int rc;
    if ((rc = dev_alloc(&ov.seq, RN))) return rc;
    if ((rc = dev_alloc(&ov.origin, RN))) return rc;
    if ((rc = dev_alloc(&ov.foll, RN))) return rc;
    if ((rc = dev_alloc(&ov.ctl_seq, RN))) return rc;
    if ((rc = dev_alloc(&ov.lead, RN))) return rc;
    if ((rc = dev_alloc(&ov.arrived_rl, RN))) return rc;
    if ((rc = dev_alloc(&ov.foll_h, RN))) return rc;
    if ((rc = dev_alloc(&ov.headway, RN))) return rc;
    if ((rc = dev_alloc(&ov.vmax, RN))) return rc;
    if ((rc = dev_alloc(&ov.arr_hist, size_t(R) * 20))) return rc;
    if ((rc = dev_alloc(&ov.counters, size_t(R) * 8))) return rc;
    if ((rc = dev_alloc(&ov.emitted, size_t(R) * FS_MAX_INFLOWS))) return rc;
    if ((rc = dev_alloc(&ov.generated, size_t(R) * FS_MAX_INFLOWS))) return rc;
    if ((rc = dev_alloc(&ov.episode, size_t(R)))) return rc;
    if ((rc = dev_alloc(&d_qflag, size_t(1)))) return rc;
    HIP_TRY(hipMemset(d_qflag, 0, sizeof(int)));
    HIP_TRY(hipMemset(ov.episode, 0xFF, size_t(R) * sizeof(int32_t)));     // -1: fs_create's own reset below is not an episode
    HIP_TRY(hipMemset(ov.ctl_seq, 0xFF, RN * sizeof(int32_t)));            // rl_veh starts empty (a reset keeps it, O2)
    HIP_TRY(hipMemset(ov.origin, 0xFF, RN * sizeof(int32_t)));
    HIP_TRY(hipMemset(ov.counters, 0, size_t(R) * 8 * sizeof(int32_t)));
    if ((rc = upload(&ov.init_alive, init_alive))) return rc;
    std::vector<int32_t> st(N);
    int n_rl_slots = 0;
    for (int i = 0; i < N; ++i) {
      st[i] = veh[i].type;
      if (veh[i].controller == FS_CTRL_RL) ++n_rl_slots;
    }
    if ((rc = upload(&ov.slot_type, st))) return rc;
    std::vector<T> tab(size_t(fs::TAB_ROWS) * 64, T(0));
    for (int n = 0; n <= 64; ++n) {               // np.linalg.norm([target] * n) in double (rewards.py:50-51)
      double ss = 0.0;
      for (int i = 0; i < n; ++i) ss += cfg.target_velocity * cfg.target_velocity;
      if (n < 64) tab[size_t(fs::TAB_MAX_COST) * 64 + n] = T(std::sqrt(ss));
      else ov.max_cost_full = T(std::sqrt(ss));
    }
    ov.n_inflows = cfg.num_inflows;
    ov.ma_apply_actions = cfg.ma_apply_actions;
    ov.n_rl_slots = n_rl_slots;
    std::vector<double> ftd(3 * 64, 0.0);
    std::vector<int32_t> fti(3 * 64, 0);
    ov.n_prob = 0;
    for (int f = 0; f < cfg.num_inflows; ++f) {
      ftd[f] = inflows[f].period;
      if (inflows[f].probability >= 0.0) {
        // a probabilistic inflow keeps, in the place of its period, -(threshold + 1): a vehicle is generated in a
        // sub-step when the sub-step's 32-bit Philox draw is below threshold = floor(p * sim_step * 2^32)
        double thr = std::floor(inflows[f].probability * cfg.sim_step * 4294967296.0);
        if (thr > 4294967295.0) thr = 4294967295.0;
        ftd[f] = -(thr + 1.0);
        ov.n_prob += 1;
      }
      ftd[64 + f] = inflows[f].begin;
      ftd[128 + f] = inflows[f].end;
      fti[f] = inflows[f].type;
      fti[64 + f] = inflows[f].route;
      fti[128 + f] = inflows[f].number;
      int first = 0;                               // the type's parameters: those of its first slot
      for (int i = N - 1; i >= 0; --i)
        if (veh[i].type == inflows[f].type) first = i;
      // same operations, in T, as oracle/opennet.py _insert
      // (the lane-drop network has ONE route table: its "routes" are entry lanes, -1 = a random one)
      const int rs = (cfg.network == FS_NET_MERGE && inflows[f].route == 1) ? 1 : 0;
      tab[size_t(fs::TAB_FL_XDEP) * 64 + f] = T(cfg.route_start[rs]) + T(inflows[f].depart_pos);
      tab[size_t(fs::TAB_FL_VDEP) * 64 + f] = T(inflows[f].depart_speed);
      tab[size_t(fs::TAB_FL_MINGAP) * 64 + f] = T(veh[first].sumo_min_gap);
      tab[size_t(fs::TAB_FL_TAU) * 64 + f] = T(veh[first].sumo_tau);
      tab[size_t(fs::TAB_FL_TWOSQRT) * 64 + f] = T(2) * std::sqrt(T(veh[first].max_accel) * T(veh[first].max_decel));
    }
    ov.dt_d = cfg.sim_step;
    for (int r = 0; r < 2; ++r) {
      ov.nseg[r] = 0;
      ov.seg_internal[r] = 0u;
    }
    for (const fs_segment& sg : segs) {
      const int r = sg.route, k = ov.nseg[r]++;
      tab[size_t(fs::TAB_SEG_START) * 64 + r * 16 + k] = T(sg.start);
      tab[size_t(fs::TAB_SEG_FLOW) * 64 + r * 16 + k] = T(sg.flow_start);
      tab[size_t(fs::TAB_SEG_SLOPE) * 64 + r * 16 + k] = T(sg.flow_slope);
      if (sg.internal) ov.seg_internal[r] |= (1u << k);
    }
    if ((rc = upload(&ov.lane_tab, tab))) return rc;
    if ((rc = upload(&ov.flow_tab_d, ftd))) return rc;
    if ((rc = upload(&ov.flow_tab_i, fti))) return rc;
    {   // lane drops + bottleneck heads
      const bool bn = cfg.network == FS_NET_BOTTLENECK;
      ov.m1 = T(bn ? cfg.merge1_x : cfg.merge_x);
      ov.m2 = T(bn ? cfg.merge2_x : cfg.merge_x);
      ov.zip_d = T(bn ? cfg.zipper_distance : 0.0);
      ov.speed_limit = cfg.speed_limit > 0 ? T(cfg.speed_limit) : T(3.0e38);
      ov.n_obs_cells = int(obs_cells.size());
      ov.n_act_cells = int(act_cells.size());
      ov.obs_window = cfg.obs_outflow_window;
      ov.rew_window = cfg.reward_outflow_window;
      ov.out_norm = T(cfg.outflow_norm > 0 ? cfg.outflow_norm : 2000.0);
      ov.obs_dim = obs_dim;
      std::vector<T> ct(6 * 64, T(0));
      std::vector<int32_t> cti(3 * 64, 0);
      // consecutive cells that differ only in the lane (lane l, l+1, ...) form one group
      auto group = [&](const std::vector<fs_cell>& cells, int row_start, int row_lo, int row_hi, int row_i) {
        int g = 0;
        for (size_t c = 0; c < cells.size();) {
          size_t e = c + 1;
          while (e < cells.size() && cells[e].edge_start == cells[c].edge_start && cells[e].lo == cells[c].lo &&
                 cells[e].hi == cells[c].hi && cells[e].last_segment == cells[c].last_segment &&
                 cells[e].lane == cells[c].lane + int(e - c))
            ++e;
          ct[size_t(row_start) * 64 + g] = T(cells[c].edge_start);
          ct[size_t(row_lo) * 64 + g] = T(cells[c].lo);
          ct[size_t(row_hi) * 64 + g] = T(cells[c].hi);
          cti[size_t(row_i) * 64 + g] = int(c) | (int(e - c) << 8) | (cells[c].lane << 16) |
                                         ((cells[c].last_segment ? 1 : 0) << 24);
          ++g;
          c = e;
        }
        return g;
      };
      ov.n_obs_groups = group(obs_cells, fs::CELL_OBS_START, fs::CELL_OBS_LO, fs::CELL_OBS_HI, 0);
      ov.n_act_groups = group(act_cells, fs::CELL_ACT_START, fs::CELL_ACT_LO, fs::CELL_ACT_HI, 1);
      // row 2: the groups that lie on route segment k (an edge and its groups share the start coordinate), so that a
      // vehicle only tries the lane-segments of its own edge
      ov.obs_span = ov.act_span = 0;
      auto ranges = [&](int n_groups, int row_start, int shift) {
        int span = 0;
        for (size_t k = 0; k < segs.size() && k < 64; ++k) {
          int first = -1, cnt = 0;
          for (int g = 0; g < n_groups; ++g)
            if (double(ct[size_t(row_start) * 64 + g]) == double(T(segs[k].start)) && !segs[k].internal) {
              if (first < 0) first = g;
              ++cnt;
            }
          if (first >= 0) cti[size_t(2) * 64 + k] |= (first | (cnt << 8)) << shift;
          span = cnt > span ? cnt : span;
        }
        return span;
      };
      ov.obs_span = ranges(ov.n_obs_groups, fs::CELL_OBS_START, 0);
      ov.act_span = ranges(ov.n_act_groups, fs::CELL_ACT_START, 16);
      ov.track_followers = cfg.track_followers;
      std::vector<int32_t> lca(N, 0);
      int any_lc = 0;
      for (int i = 0; i < N; ++i) {
        lca[i] = (veh[i].lane_change_mode & 0x55) != 0;
        any_lc |= lca[i];
      }
      if ((rc = upload(&ov.lc_auto, lca))) return rc;
      ov.lc_enabled = bn && any_lc;
      ov.lc_cooldown = cfg.lane_change_cooldown_steps;
      ov.lc_min_gain = T(cfg.lane_change_min_gain);
      if ((rc = upload(&ov.cell_tab, ct))) return rc;
      if ((rc = upload(&ov.cell_tab_i, cti))) return rc;
    }
    {   // flowsim_queue.h: the junction-internal stretches of each route as intervals, the one vehicle length
      qc.ok = (cfg.network == FS_NET_MERGE) ? 1 : 0;
      for (int r = 0; r < 2; ++r)
        for (int j = 0; j < 2; ++j) qc.in_lo[r][j] = qc.in_hi[r][j] = 3.0e38f;
      int n_int[2] = {0, 0}, k_of[2] = {0, 0};
      for (int r = 0; r < 2; ++r)
        for (int q = 0; q < 6; ++q) { qc.seg_start[r][q] = 3.0e38f; qc.seg_flow[r][q] = 0.0f; qc.seg_slope[r][q] = 0.0f; }
      for (size_t i = 0; i < segs.size() && qc.ok; ++i) {
        const int r = segs[i].route;
        if (r < 0 || r > 1) { qc.ok = 0; break; }
        const int k = k_of[r]++;
        if (k >= 6) { qc.ok = 0; break; }
        qc.seg_start[r][k] = float(segs[i].start);
        qc.seg_flow[r][k] = float(segs[i].flow_start);
        qc.seg_slope[r][k] = float(segs[i].flow_slope);
        if (!segs[i].internal) continue;
        if (n_int[r] >= 2) { qc.ok = 0; break; }
        float hi = 3.0e38f;                              // the next segment of the same route starts where this one ends
        for (size_t j = i + 1; j < segs.size(); ++j)
          if (segs[j].route == r) { hi = float(segs[j].start); break; }
        qc.in_lo[r][n_int[r]] = k == 0 ? -3.0e38f : float(segs[i].start);
        qc.in_hi[r][n_int[r]] = hi;
        n_int[r] += 1;
      }
      qc.veh_len = N > 0 ? float(veh[0].length) : 5.0f;          // (k_drop_queue checks the lengths itself: dropq_ok)
      for (int i = 0; i < N; ++i)
        if (float(veh[i].length) != qc.veh_len) qc.ok = 0;
    }
    ov.merge_x = T(cfg.merge_x);
    ov.box_in = T(cfg.box_in);
    ov.end_x = T(cfg.end_x);
    ov.net_length = T(cfg.net_length);
    dv.nseg = 0;                                   // the closed-loop segment table is not used
    return FS_OK;
  }

  // ---- exact division by launch constants (flowsim_kernels.h div_const) --------------------
  // The 3-operation reciprocal sequence is enabled only if, for every divisor the rollout kernel
  // will use (v0 and 2*sqrt(a*b) of every slot, the loop length of every replica), it
  // reproduces x / c for ALL 2^23 float mantissas of x.  Checked lazily, once per handle state;
  // set_state of the ring lengths invalidates it.  Only the float kernels use it.
  // rewards.py:46 ("any speed < -100 -> reward 0") can only fire on speeds put there from outside (see
  // k_rollout_idm): true while the initial speeds or an upload since the last full reset held such a value
  bool neg_speed_possible = false;
  bool init_vel_negative = false;
  int fastdiv_state = -1;       // -1 unknown, 0 no, 1 yes
  static bool fastdiv_exact_for(float c) {
    if (!(c > 0.0f) || !std::isfinite(c)) return false;
    // process-wide memo: the answer depends on the divisor only
    static std::mutex mu;
    static std::map<uint32_t, bool> memo;
    uint32_t key;
    std::memcpy(&key, &c, 4);
    {
      std::lock_guard<std::mutex> lock(mu);
      auto it = memo.find(key);
      if (it != memo.end()) return it->second;
    }
    const bool ok = fastdiv_exact_uncached(c);
    std::lock_guard<std::mutex> lock(mu);
    memo[key] = ok;
    return ok;
  }
  static bool fastdiv_exact_uncached(float c) {
    const float rc = 1.0f / c;
    for (uint32_t m = 0; m < (1u << 23); ++m) {
      const uint32_t bits = 0x3F800000u | m;
      float x;
      std::memcpy(&x, &bits, 4);
      const float q0 = x * rc;
      const float r = std::fmaf(-q0, c, x);
      if (std::fmaf(r, rc, q0) != x / c) return false;
    }
    return true;
  }
  bool fastdiv_ok() {
    if ((!std::is_same<T, float>::value && !mixed) || force_generic || no_fastdiv) return false;
    if (fastdiv_state >= 0) return fastdiv_state == 1;
    fastdiv_state = 0;
    std::vector<float> cs;
    auto add = [&cs](float c) {
      for (float e : cs)
        if (e == c) return;
      cs.push_back(c);
    };
    for (int i = 0; i < dv.N; ++i) {            // per-slot IDM divisors and the s0 >= 1e-3 premise
      add(float(veh[i].p[0]));
      add(2.0f * std::sqrt(float(veh[i].p[2]) * float(veh[i].p[3])));
      if (!(float(veh[i].p[5]) >= 1e-3f) || !(float(veh[i].p[5]) <= 1e6f)) return false;
      if (speed_mode_any) {                     // sumo_acc_pair's divisors and its minGap >= 1e-3 premise
        add(float(veh[i].sumo_max_speed));
        add(2.0f * std::sqrt(float(veh[i].max_accel) * float(veh[i].max_decel)));
        if (!(float(veh[i].sumo_min_gap) >= 1e-3f) || !(float(veh[i].sumo_min_gap) <= 1e6f)) return false;
      }
    }
    for (T b : h_ring_len) {                    // host copy: no HIP call on the launch path
      const float L = float(b) + 4.0f * float(dv.jlen);
      if (!(L >= 1.0f)) return false;        // keeps x = 0 or x >= ulp(L)/2 out of the tiny range
      add(L);
      if (cs.size() > 80) return false;      // too many distinct loop lengths to verify cheaply
    }
    for (float c : cs)
      if (!fastdiv_exact_for(c)) return false;
    fastdiv_state = 1;
    return true;
  }

  // k_rollout_loop<..., FULL>: its controller divisions by launch constants are div_const -- every divisor proven
  int loop_fastc_state = -1;
  bool loop_fastc_ok() {
    if (!(std::is_same<T, float>::value || mixed) || no_fastdiv) return false;
    if (loop_fastc_state >= 0) return loop_fastc_state == 1;
    loop_fastc_state = 0;
    if (!loop_div_ok) return false;               // s0 / minGap in [1e-3, 1e6]: tiny dividends cannot matter
    for (int i = 0; i < dv.N; ++i) {
      if (veh[i].controller == FS_CTRL_IDM) {
        if (!fastdiv_exact_for(float(veh[i].p[0]))) return false;
        if (!fastdiv_exact_for(2.0f * std::sqrt(float(veh[i].p[2]) * float(veh[i].p[3])))) return false;
      }
      if (!fastdiv_exact_for(float(veh[i].sumo_max_speed))) return false;
      if (!fastdiv_exact_for(2.0f * std::sqrt(float(veh[i].max_accel) * float(veh[i].max_decel)))) return false;
    }
    loop_fastc_state = 1;
    return true;
  }

  // k_ring_pair<..., FAST> (flowsim_ringrl.h): exponent 4 and the exact reciprocal divisions -- every divisor proven.
  // The premises on s0 / minGap are weaker than fastdiv_ok()'s (>= 0 instead of >= 1e-3): s* = s0 + max(0, dyn) below
  // 2^-60 makes q = s* / h < 2^-50, whose square vanishes against every non-zero 1 - (v/v0)^4 (>= 2^-24) and, where that
  // term is exactly zero, gives an acceleration too small to move a speed near v0 -- so an inexact quotient there
  // cannot change a result; from 2^-60 up div_core's operands are inside its proven range (|h| in [1e-3, L]).
  int ringrl_fast_state = -1;
  bool any_sim = false;                  // some slot is a SimCarFollowingController
  bool ringrl_fast_ok() {
    if (no_fastdiv || force_generic) return false;
    if (ringrl_fast_state >= 0) return ringrl_fast_state == 1;
    ringrl_fast_state = 0;
    std::vector<float> cs;
    auto add = [&cs](float c) {
      for (float e : cs)
        if (e == c) return;
      cs.push_back(c);
    };
    for (int i = 0; i < dv.N; ++i) {
      if (veh[i].controller == FS_CTRL_IDM) {
        if (veh[i].p[4] != 4.0) return false;
        add(float(veh[i].p[0]));
        add(2.0f * std::sqrt(float(veh[i].p[2]) * float(veh[i].p[3])));
        if (!(float(veh[i].p[5]) >= 0.0f) || !(float(veh[i].p[5]) <= 1e6f)) return false;
      }
      add(float(veh[i].sumo_max_speed));
      add(2.0f * std::sqrt(float(veh[i].max_accel) * float(veh[i].max_decel)));
      if (!(float(veh[i].sumo_min_gap) >= 0.0f) || !(float(veh[i].sumo_min_gap) <= 1e6f)) return false;
    }
    for (T b : h_ring_len) {
      const float L = float(b) + 4.0f * float(dv.jlen);
      if (!(L >= 1.0f)) return false;
      add(L);
      if (cs.size() > 96) return false;
    }
    for (float c : cs)
      if (!fastdiv_exact_for(c)) return false;
    ringrl_fast_state = 1;
    return true;
  }

  // the specialisations for the headline configuration (see flowsim_kernels.h)
  bool delta4 = false;
  bool loop_div_ok = false, loop_delta4 = false, open_div_ok = false;
  bool sumo_beyond_speed_mode = false;   // FLAG_NEED_SUMO for more than speed-mode bits (Sim / RL slots, junction mode)
  bool speed_mode_any = false;           // some slot carries a speed-mode clamp (bits 0-2)
  // allow_speed_mode: the caller's kernel evaluates the speed-mode clamps itself (k_rollout_pair<..., SM = true>)
  bool fast_ok(const uint8_t* mask, int num_steps, bool allow_speed_mode = false, bool allow_noise = false) const {
    const int f = dv.flags;
    const bool sumo_free = !(f & fs::FLAG_NEED_SUMO) || (allow_speed_mode && !sumo_beyond_speed_mode);
    const bool noise_free = !(f & fs::FLAG_HAS_NOISE) || allow_noise;
    return (f & fs::FLAG_ALL_IDM) && !(f & fs::FLAG_HAS_FAILSAFE) && noise_free && sumo_free &&
           dv.env == FS_ENV_ACCEL && !dv.evaluate && dv.sims_per_step == 1 && dv.integrator == FS_EULER &&
           !dv.junction_mode && !dv.track_aux && mask == nullptr && num_steps > 0 && !force_generic &&
           dv.nseg == 0 && !dv.junction_on && !dv.sort_vehicles && dv.obs_perm == nullptr;
  }

  // the merge network in queue order (flowsim_queue.h): float32, IDM / RL / Sim slots, the multi-agent head, scheduled
  // inflows, every replica stepping.  Defined in flowsim_launch.h for the one part that holds the kernel (FS_PART_QUEUE).
  bool queue_ok(const uint8_t* mask, int num_steps) const {
    if (!std::is_same<T, float>::value || !open_net || cfg.network != FS_NET_MERGE || no_queue || force_generic) return false;
    if (dv.env != FS_ENV_MERGE_MA || !(dv.flags & fs::FLAG_IDM_SET) || !open_div_ok || ov.n_prob > 0) return false;
    if (!(dv.flags & fs::FLAG_DELTA4) || (dv.flags & fs::FLAG_HAS_FAILSAFE) || dv.integrator != FS_EULER || !qc.ok) return false;
    if (mask != nullptr || num_steps < 1 || dv.N > 64) return false;
    for (const fs_inflow& f : inflows)
      if (f.route < 0 || f.route > 1) return false;
    return true;
  }
  int launch_queue(int num_steps, const float* actions, size_t act_stride, float* obs, float* rew, uint8_t* done,
                   int obs_every_step);
  // the lane-drop network in queue order (flowsim_dropq.h): one wave per entry lane, every vehicle on SUMO's model
  bool dropq_ok(const uint8_t* mask, int num_steps) const {
    if (!std::is_same<T, float>::value || !open_net || cfg.network != FS_NET_BOTTLENECK || no_queue || force_generic) return false;
    if (cfg.num_paths != 4 || ov.lc_enabled || ov.track_followers || ov.n_prob > 0 || !open_div_ok) return false;
    if (!(dv.flags & fs::FLAG_NO_FLOW_CTRL) || dv.integrator != FS_EULER) return false;
    if (dv.env != FS_ENV_BOTTLENECK_DV && dv.env != FS_ENV_BOTTLENECK) return false;
    if (dv.env == FS_ENV_BOTTLENECK && dv.num_rl > 0 && ov.ma_apply_actions) return false;   // per-vehicle RL accelerations (BottleneckAccelEnv)
    if (mask != nullptr || num_steps < 1 || dv.N > 256 || ov.nseg[0] > 16 || ov.obs_span > 3 || ov.act_span > 2) return false;
    for (int i = 0; i < dv.N; ++i)
      if (float(veh[i].length) != float(veh[0].length) || veh[i].type < 0 || veh[i].type > 7) return false;
    for (const fs_inflow& f : inflows)
      if (f.route > 3) return false;
    return true;
  }
  int launch_dropq(int num_steps, const float* actions, size_t act_stride, float* obs, float* rew, uint8_t* done,
                   int obs_every_step);

  // more than 64 slots per replica (lane-drop network): one workgroup of W waves per replica (flowsim_launch.h)
  template <int W>
  int launch_wide(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride, float* obs,
                  float* rew, uint8_t* done, int obs_every_step);

  // the policy kernels exist for rows of 16 lanes (SEG = 32: 18..32 vehicles); defined in flowsim_launch.h, instantiated
  // by the SEG = 32 objects
  int launch_policy_row16(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs, float* act,
                          float* logp, float* rew, uint8_t* done);
  // the fused policy + step kernel of segment-table loops (the figure eight, flowsim_policy.h k_loop_policy): rows of 16
  // lanes (up to 16 vehicles); defined in flowsim_launch.h, instantiated by the SEG = 16 objects
  int launch_policy_loop16(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs, float* act,
                           float* logp, float* rew, uint8_t* done);
  int launch_policy(const fs_policy* pol, int num_steps, int reset_done, const float* obs_in, float* obs, float* act,
                    float* logp, float* rew, uint8_t* done) override {
    if (!pol || pol->struct_size != sizeof(fs_policy)) return fail(FS_ERR_INVALID, "fs_policy: struct_size mismatch");
    const bool f32_or_mixed = mixed || std::is_same<T, float>::value;
    const bool loop = dv.nseg > 0;                       // a segment-table loop (figure eight) or a ring
    const char* why = nullptr;
    if (pol->num_hidden < 1 || pol->num_hidden > 3 || pol->hidden_width != 32 || pol->activation != 0)
      why = "fs_policy model (1..3 hidden layers of 32 tanh units)";
    else if (!pol->weights_dev) why = "fs_policy.weights_dev (NULL)";
    else if (loop) {
      const bool po = dv.env == FS_ENV_WAVE_ATTENUATION_PO, accel = dv.env == FS_ENV_ACCEL && !dv.evaluate;
      if (!std::is_same<T, float>::value || mixed) why = "precision (f32 on segment-table loops)";
      else if (seg != 16 || dv.N < 2) why = "num_vehicles (2..16: a row of 16 lanes per replica)";
      else if (!(po || accel) || dv.num_rl != 1) why = "env (WaveAttenuationPOEnv or AccelEnv with one RL vehicle)";
      else if (pol->obs_dim != (po ? 3 : 2 * dv.N)) why = "fs_policy.obs_dim (the environment's observation)";
      else if (!(dv.flags & fs::FLAG_IDM_SET) || (dv.flags & fs::FLAG_HAS_FAILSAFE) || dv.sims_per_step != 1 ||
               dv.integrator != FS_EULER || dv.track_aux || dv.sort_vehicles || dv.obs_perm != nullptr || !loop_div_ok ||
               (obs != nullptr && reset_done && cfg.warmup_steps != 0))
        why = "configuration (what k_rollout_loop steps: IDM / RL / Sim vehicles, Euler, track_aux = 0; resets inside a "
              "fragment: warmup_steps = 0)";
      if (why) return fail(FS_ERR_UNSUPPORTED, std::string("fs_policy: not built for this handle: ") + why);
      if (!d_pol_ctr) {
        int rc = dev_alloc(&d_pol_ctr, size_t(dv.R));
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(d_pol_ctr, 0, size_t(dv.R) * sizeof(uint32_t), stream));
      }
      return launch_policy_loop16(pol, num_steps, reset_done, obs_in, obs, act, logp, rew, done);
    }
    else if (!f32_or_mixed) why = "precision (f32 or mixed)";
    else if (seg != 32) why = "num_vehicles (18..32: a row of 16 lanes per replica)";
    else if (dv.env != FS_ENV_WAVE_ATTENUATION_PO || dv.num_rl != 1) why = "env (WaveAttenuationPOEnv with one RL vehicle)";
    else if (pol->obs_dim != 3) why = "fs_policy.obs_dim (3)";
    else if (dv.nseg != 0 || dv.junction_on || dv.num_lanes > 1 || !(dv.flags & fs::FLAG_IDM_SET) || any_sim ||
             (dv.flags & fs::FLAG_HAS_FAILSAFE) || dv.sims_per_step != 1 || dv.integrator != FS_EULER || dv.junction_mode ||
             dv.track_aux || dv.sort_vehicles || dv.obs_perm != nullptr || dv.evaluate || (dv.N % 2) != 0)
      why = "configuration (what k_ring_pair steps: single-lane ring of IDM / RL vehicles, Euler, track_aux = 0)";
    if (why) return fail(FS_ERR_UNSUPPORTED, std::string("fs_policy: not built for this handle: ") + why);
    if (!d_pol_ctr) {
      int rc = dev_alloc(&d_pol_ctr, size_t(dv.R));
      if (rc) return rc;
      HIP_TRY(hipMemsetAsync(d_pol_ctr, 0, size_t(dv.R) * sizeof(uint32_t), stream));
    }
    return launch_policy_row16(pol, num_steps, reset_done, obs_in, obs, act, logp, rew, done);
  }

  // one wave carries 64 / SEG replicas (flowsim_launch.h)
  template <int SEG>
  int launch_seg(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride, float* obs,
                 float* rew, uint8_t* done, int obs_every_step);

  int launch_steps(int num_steps, const uint8_t* mask, const float* actions, size_t act_stride, float* obs,
                   float* rew, uint8_t* done, int obs_every_step) override {
    switch (seg) {
      case 8: return launch_seg<8>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
      case 16: return launch_seg<16>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
      case 32: return launch_seg<32>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
      case 128: return launch_wide<2>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
      case 256: return launch_wide<4>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
      default: return launch_seg<64>(num_steps, mask, actions, act_stride, obs, rew, done, obs_every_step);
    }
  }

  int launch_reset(const uint8_t* mask) override {
    if (mask == nullptr) neg_speed_possible = init_vel_negative;     // every replica back at its initial speeds
    if (open_net) {
      const size_t n_open = size_t(dv.R) * dv.N;
      int blocks_open = int((n_open + 255) / 256);
      if (blocks_open > 2048) blocks_open = 2048;
      hipLaunchKernelGGL((fs::k_reset_open<T>), dim3(blocks_open), dim3(256), 0, stream, dv, ov, mask);
      if (dv.st16) hipLaunchKernelGGL((fs::k_state16_pack<T>), dim3(blocks_open), dim3(256), 0, stream, dv, mask);
      HIP_TRY(hipGetLastError());
      return FS_OK;
    }
    const size_t n = size_t(dv.R) * dv.N;
    int blocks = int((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL((fs::k_reset<T>), dim3(blocks), dim3(256), 0, stream, dv, mask);
    HIP_TRY(hipGetLastError());
    return FS_OK;
  }

  T* field_ptr(int field, size_t* count, bool* writable) {
    const size_t RN = size_t(dv.R) * dv.N;
    *writable = true;
    switch (field) {
      case FS_FIELD_POS: *count = RN; return dv.pos;
      case FS_FIELD_VEL: *count = RN; return dv.vel;
      case FS_FIELD_PREV_VEL: *count = RN; return dv.prev_vel;
      case FS_FIELD_ACCEL: *count = RN; return dv.accel;
      case FS_FIELD_CTRL_STATE: *count = RN; return dv.ctrl_state;
      case FS_FIELD_MAX_SPEED: *count = open_net ? RN : 0; return open_net ? ov.vmax : nullptr;
      case FS_FIELD_RING_LENGTH: *count = size_t(dv.R); return const_cast<T*>(dv.ring_len);
      case FS_FIELD_INIT_RING_LENGTH: *count = size_t(dv.R); return const_cast<T*>(dv.init_ring_len);
      case FS_FIELD_SORT_KEY: *count = RN; return dv.sort_key;
      case FS_FIELD_INIT_POS: *count = RN; return const_cast<T*>(dv.init_pos);
      case FS_FIELD_INIT_VEL: *count = RN; return const_cast<T*>(dv.init_vel);
      default: *count = 0; return nullptr;
    }
  }

  int get_state(int field, void* dst, size_t bytes) override {
    if (dv.st16 && (field == FS_FIELD_POS || field == FS_FIELD_VEL)) {     // the halves -> the float32 staging arrays
      hipLaunchKernelGGL((fs::k_state16_unpack<T>), dim3(256), dim3(256), 0, stream, dv);
      HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipStreamSynchronize(stream));
    if (int qrc = check_qflag()) return qrc;
    if (field == FS_FIELD_TIME) {
      if (bytes != size_t(dv.R) * sizeof(int32_t)) return fail(FS_ERR_INVALID, "FS_FIELD_TIME: wrong byte count");
      HIP_TRY(hipMemcpy(dst, dv.time, bytes, hipMemcpyDeviceToHost));
      return FS_OK;
    }
    if (field == FS_FIELD_LANE || field == FS_FIELD_LAST_LC || field == FS_FIELD_INIT_LANE) {
      const size_t RN = size_t(dv.R) * dv.N;
      if (bytes != RN * sizeof(int32_t)) return fail(FS_ERR_INVALID, "fs_get_state: wrong byte count");
      const int32_t* p = field == FS_FIELD_LANE ? dv.lane : (field == FS_FIELD_LAST_LC ? dv.last_lc : dv.init_lane);
      HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
      return FS_OK;
    }
    if (field >= FS_FIELD_ROUTE && field <= FS_FIELD_ARRIVED_RL) {
      if (!open_net) return fail(FS_ERR_INVALID, "fs_get_state: field exists for open networks only");
      const size_t RN = size_t(dv.R) * dv.N;
      const int32_t* p = nullptr;
      size_t count = RN;
      switch (field) {
        case FS_FIELD_ROUTE: p = dv.lane; break;
        case FS_FIELD_SEQ: p = ov.seq; break;
        case FS_FIELD_ORIGIN: p = ov.origin; break;
        case FS_FIELD_FOLLOWER: p = ov.foll; break;
        case FS_FIELD_CTL_SEQ: p = ov.ctl_seq; break;
        case FS_FIELD_COUNTERS: p = ov.counters; count = size_t(dv.R) * 8; break;
        default: p = ov.arrived_rl; break;
      }
      if (bytes != count * sizeof(int32_t)) return fail(FS_ERR_INVALID, "fs_get_state: wrong byte count");
      HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
      return FS_OK;
    }
    if (open_net && (field == FS_FIELD_HEADWAY || field == FS_FIELD_LEADER)) {
      // open networks keep the snapshot of the last update on the device (vehicle/traci.py:219-250)
      const size_t RN = size_t(dv.R) * dv.N;
      const size_t want = RN * (field == FS_FIELD_HEADWAY ? sizeof(T) : sizeof(int32_t));
      if (bytes != want) return fail(FS_ERR_INVALID, "fs_get_state: wrong byte count");
      HIP_TRY(hipMemcpy(dst, field == FS_FIELD_HEADWAY ? static_cast<const void*>(ov.headway)
                                                        : static_cast<const void*>(ov.lead),
                        bytes, hipMemcpyDeviceToHost));
      return FS_OK;
    }
    if (field == FS_FIELD_HEADWAY || field == FS_FIELD_LEADER) {
      // the kernels' neighbour rule, evaluated on the host from the positions (and lanes):
      // headway = (x_lead - x) mod L - len_lead (vehicle/traci.py:219-250); leader = own-lane nearest ahead
      const int R = dv.R, N = dv.N;
      const size_t RN = size_t(R) * N;
      const size_t want = RN * (field == FS_FIELD_HEADWAY ? sizeof(T) : sizeof(int32_t));
      if (bytes != want) return fail(FS_ERR_INVALID, "fs_get_state: wrong byte count");
      std::vector<T> x(RN), rl(R);
      std::vector<int32_t> ln(RN, 0);
      HIP_TRY(hipMemcpy(x.data(), dv.pos, RN * sizeof(T), hipMemcpyDeviceToHost));
      HIP_TRY(hipMemcpy(rl.data(), dv.ring_len, size_t(R) * sizeof(T), hipMemcpyDeviceToHost));
      if (dv.num_lanes > 1) HIP_TRY(hipMemcpy(ln.data(), dv.lane, RN * sizeof(int32_t), hipMemcpyDeviceToHost));
      for (int r = 0; r < R; ++r) {
        const T L = rl[r] + T(4) * dv.jlen;
        for (int i = 0; i < N; ++i) {
          int lead = -1;
          T best = T(0);
          if (dv.num_lanes > 1) {
            for (int j = 0; j < N; ++j) {
              if (j == i || ln[size_t(r) * N + j] != ln[size_t(r) * N + i]) continue;
              T d = x[size_t(r) * N + j] - x[size_t(r) * N + i];
              if (d < T(0) || (d == T(0) && j < i)) d = d + L;
              if (lead < 0 || d < best) { best = d; lead = j; }
            }
          } else if (N > 1) {
            lead = (i + 1 >= N) ? 0 : i + 1;
            best = x[size_t(r) * N + lead] - x[size_t(r) * N + i];
            if (best < T(0)) best = best + L;
          }
          if (field == FS_FIELD_LEADER) static_cast<int32_t*>(dst)[size_t(r) * N + i] = lead;
          else static_cast<T*>(dst)[size_t(r) * N + i] = lead < 0 ? T(1000) : best - h_len[lead];
        }
      }
      return FS_OK;
    }
    size_t count;
    bool writable;
    T* p = field_ptr(field, &count, &writable);
    if (!p) return fail(FS_ERR_INVALID, "fs_get_state: unknown field");
    if (bytes != count * sizeof(T)) return fail(FS_ERR_INVALID, "fs_get_state: wrong byte count");
    HIP_TRY(hipMemcpy(dst, p, bytes, hipMemcpyDeviceToHost));
    return FS_OK;
  }

  // k.vehicle.add (vehicle/traci.py:1089-1122) for a vehicle that has a slot of its own and is not in the network (an
  // initial vehicle that arrived: BottleneckAccelEnv.additional_command re-inserts its RL vehicles, bottleneck.py:733-757):
  // the slot's vehicle is back at coordinate x of entry lane / route `route` with the given speed, last in the id list
  int add_vehicle(int replica, int slot, int route, double x, double speed) override {
    HIP_TRY(hipStreamSynchronize(stream));
    if (!open_net) return fail(FS_ERR_UNSUPPORTED, "fs_add_vehicle: open networks only (a closed loop keeps its vehicles)");
    const int paths = cfg.network == FS_NET_MERGE ? 2 : (cfg.num_paths ? cfg.num_paths : 4);
    if (replica < 0 || replica >= dv.R || slot < 0 || slot >= dv.N || route < 0 || route >= paths)
      return fail(FS_ERR_INVALID, "fs_add_vehicle: replica / slot / route out of range");
    if (!(speed >= 0.0) || !(x >= 0.0) || !(x < double(ov.end_x)))
      return fail(FS_ERR_INVALID, "fs_add_vehicle: need speed >= 0 and 0 <= x < the end of the route");
    const size_t e = size_t(replica) * dv.N + slot;
    int32_t cur = 0, cnt[8];
    HIP_TRY(hipMemcpy(&cur, dv.lane + e, sizeof(int32_t), hipMemcpyDeviceToHost));
    if (cur >= 0) return fail(FS_ERR_INVALID, "fs_add_vehicle: the slot's vehicle is in the network");
    HIP_TRY(hipMemcpy(cnt, ov.counters + size_t(replica) * 8, sizeof(cnt), hipMemcpyDeviceToHost));
    const T xv = T(x), vv = T(speed), zero = T(0), big = T(3.0e38), vm = T(veh[slot].sumo_max_speed);
    const int32_t rt = route, sq = cnt[fs::CNT_SEQ], minus1 = -1, org = -1 - slot, never = -(1 << 30);
    cnt[fs::CNT_SEQ] += 1;
    HIP_TRY(hipMemcpy(dv.pos + e, &xv, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.vel + e, &vv, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.prev_vel + e, &zero, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dv.accel + e, &zero, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<int32_t*>(dv.lane) + e, &rt, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(const_cast<int32_t*>(dv.last_lc) + e, &never, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.seq + e, &sq, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.origin + e, &org, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.foll + e, &minus1, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.ctl_seq + e, &minus1, sizeof(int32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.foll_h + e, &big, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.vmax + e, &vm, sizeof(T), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(ov.counters + size_t(replica) * 8, cnt, sizeof(cnt), hipMemcpyHostToDevice));
    return launch_steps(0, nullptr, nullptr, 0, d_obs, d_rew, d_done, 0);      // the neighbour fields of the new arrangement
  }

  int set_state(int field, const void* src, size_t bytes) override {
    HIP_TRY(hipStreamSynchronize(stream));
    if (field == FS_FIELD_TIME) {
      if (bytes != size_t(dv.R) * sizeof(int32_t)) return fail(FS_ERR_INVALID, "FS_FIELD_TIME: wrong byte count");
      HIP_TRY(hipMemcpy(dv.time, src, bytes, hipMemcpyHostToDevice));
      return FS_OK;
    }
    if (field == FS_FIELD_HEADWAY || field == FS_FIELD_LEADER)
      return fail(FS_ERR_INVALID, "headway / leader are derived from positions and lanes");
    if (field >= FS_FIELD_SEQ && field <= FS_FIELD_ARRIVED_RL)
      return fail(FS_ERR_INVALID, "fs_set_state: read-only field");
    if (field == FS_FIELD_MAX_SPEED && !open_net)
      return fail(FS_ERR_INVALID, "fs_set_state: FS_FIELD_MAX_SPEED exists for open networks only");
    if (field == FS_FIELD_ROUTE) field = FS_FIELD_LANE;
    if (field == FS_FIELD_LANE || field == FS_FIELD_LAST_LC || field == FS_FIELD_INIT_LANE) {
      const size_t RN = size_t(dv.R) * dv.N;
      if (bytes != RN * sizeof(int32_t)) return fail(FS_ERR_INVALID, "fs_set_state: wrong byte count");
      const int32_t* p = field == FS_FIELD_LANE ? dv.lane : (field == FS_FIELD_LAST_LC ? dv.last_lc : dv.init_lane);
      HIP_TRY(hipMemcpy(const_cast<int32_t*>(p), src, bytes, hipMemcpyHostToDevice));
      if (open_net && field == FS_FIELD_LANE) return launch_steps(0, nullptr, nullptr, 0, d_obs, d_rew, d_done, 0);
      return FS_OK;
    }
    size_t count;
    bool writable;
    T* p = field_ptr(field, &count, &writable);
    if (!p) return fail(FS_ERR_INVALID, "fs_set_state: unknown field");
    if (bytes != count * sizeof(T)) return fail(FS_ERR_INVALID, "fs_set_state: wrong byte count");
    HIP_TRY(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    if (field == FS_FIELD_VEL || field == FS_FIELD_INIT_VEL) {
      bool neg = false;
      const T* vals = static_cast<const T*>(src);
      for (size_t e = 0; e < count; ++e) neg = neg || !(vals[e] >= T(-100));
      if (field == FS_FIELD_INIT_VEL) init_vel_negative = neg;
      neg_speed_possible = neg_speed_possible || neg;
    }
    if (dv.st16 && (field == FS_FIELD_POS || field == FS_FIELD_VEL)) {
      // one of the two staging arrays was just overwritten: bring the OTHER one up to date from the halves first, then
      // pack both (a float32 value that is no half is rounded here, as every launch boundary does)
      std::vector<T> keep(count);
      HIP_TRY(hipMemcpy(keep.data(), p, bytes, hipMemcpyDeviceToHost));
      hipLaunchKernelGGL((fs::k_state16_unpack<T>), dim3(256), dim3(256), 0, stream, dv);
      HIP_TRY(hipStreamSynchronize(stream));
      HIP_TRY(hipMemcpy(p, keep.data(), bytes, hipMemcpyHostToDevice));
      hipLaunchKernelGGL((fs::k_state16_pack<T>), dim3(256), dim3(256), 0, stream, dv, static_cast<const uint8_t*>(nullptr));
      HIP_TRY(hipGetLastError());
    }
    if (open_net && (field == FS_FIELD_POS || field == FS_FIELD_VEL))   // refresh the leader / headway snapshot
      return launch_steps(0, nullptr, nullptr, 0, d_obs, d_rew, d_done, 0);
    if (field == FS_FIELD_RING_LENGTH) {            // sets the pending lengths too
      HIP_TRY(hipMemcpy(const_cast<T*>(dv.init_ring_len), src, bytes, hipMemcpyHostToDevice));
      h_ring_len.assign(static_cast<const T*>(src), static_cast<const T*>(src) + count);
      fastdiv_state = -1;
      ringrl_fast_state = -1;
    }
    if (field == FS_FIELD_INIT_RING_LENGTH) {
      // the host copy the divisor proofs walk holds every length a replica may be running on: the current ones and,
      // appended, the pending ones (a masked reset on the device swaps them in without the host knowing which)
      const T* vals = static_cast<const T*>(src);
      h_ring_len.insert(h_ring_len.end(), vals, vals + count);
      std::sort(h_ring_len.begin(), h_ring_len.end());             // the distinct lengths only: the proofs walk this list
      h_ring_len.erase(std::unique(h_ring_len.begin(), h_ring_len.end()), h_ring_len.end());
      fastdiv_state = -1;
      ringrl_fast_state = -1;
    }
    return FS_OK;
  }
};

}  // namespace fsim
