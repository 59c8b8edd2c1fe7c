// flowsim.hip -- C ABI (include/flowsim.h) over the gfx950 kernels in
// flowsim_kernels.h.  Host code only allocates, uploads tables, picks the
// kernel instantiation and enqueues launches; every number the environment
// returns is computed on the GPU.  There is no CPU fallback: without a HIP
// device fs_create fails with FS_ERR_HIP.
#include "flowsim_sim.h"

namespace fsim {

int validate(const fs_config* c) {
  if (!c) return fail(FS_ERR_INVALID, "fs_create: cfg is NULL");
  if (c->struct_size != sizeof(fs_config))
    return fail(FS_ERR_INVALID, "fs_create: struct_size mismatch (header/library out of sync)");
  if (c->abi_version != FS_ABI_VERSION) return fail(FS_ERR_INVALID, "fs_create: abi_version mismatch");
  if (c->precision != FS_F32 && c->precision != FS_F64 && c->precision != FS_MIXED && c->precision != FS_F16S)
    return fail(FS_ERR_INVALID, "fs_create: bad precision");
  if (c->precision == FS_F16S && (c->network != FS_NET_MERGE || c->num_vehicles > 64))
    return fail(FS_ERR_UNSUPPORTED, "fs_create: FS_F16S (half state, float32 integrator) is built for FS_NET_MERGE "
                                    "(k_steps_open, <= 64 vehicle slots)");
  if (c->precision == FS_MIXED && (c->network == FS_NET_MERGE || c->network == FS_NET_BOTTLENECK)) {
    // open networks: the float64 kernels with float32 car-following models (k_steps_open<double, ., ., 2>); whatever FS_F64
    // accepts (the lane-drop network beyond 64 slots, k_steps_wide, steps in plain float64)
  } else if (c->precision == FS_MIXED) {
    // float64 state, float32 controllers: k_rollout_pair (all-IDM AccelEnv rollout) and k_ring_pair (IDM + RL vehicles,
    // AccelEnv / WaveAttenuationPOEnv, warm-up steps and masked resets included).  Name the field that does not fit.
    // The figure eight (k_rollout_loop's float64-state instantiation, flowsim_fig8.h): rollouts step there, everything else
    // (resets, masks, single steps of more than 16 vehicles) on the generic float64 kernel -- the reference's arithmetic.
    const bool fig8 = c->network == FS_NET_FIGURE_EIGHT;
    const char* why = nullptr;
    if ((c->network != FS_NET_RING && !fig8) || c->num_lanes > 1) why = "network (single-lane ring or figure eight)";
    else if (c->env != FS_ENV_ACCEL && c->env != FS_ENV_WAVE_ATTENUATION_PO) why = "env (AccelEnv or WaveAttenuationPOEnv)";
    else if (c->evaluate) why = "evaluate";
    else if (c->sims_per_step != 1) why = "sims_per_step (1)";
    else if (c->integrator != FS_EULER) why = "integrator (Euler)";
    else if (c->junction_mode && !fig8) why = "junction_mode";
    else if (c->track_aux && !fig8) why = "track_aux (the scalar Env's previous-speed / acceleration fields: VecFlowEnv(track_aux=False))";
    else if (c->sort_vehicles || c->obs_perm) why = "sort_vehicles / shuffled ids";
    else if (!fig8 && (c->num_vehicles < 2 || c->num_vehicles > 64 || (c->num_vehicles % 2) != 0)) why = "num_vehicles (even, 2..64)";
    else if (!c->vehicles) why = "vehicles (NULL)";
    for (int i = 0; !why && i < c->num_vehicles; ++i) {
      const fs_vehicle_spec& v = c->vehicles[i];
      if (v.controller != FS_CTRL_IDM && v.controller != FS_CTRL_RL && !(fig8 && v.controller == FS_CTRL_SIM))
        why = "vehicles[].controller (IDMController / RLController)";
      else if (v.fail_safe != FS_FAILSAFE_NONE) why = "vehicles[].fail_safe";
    }
    if (why)
      return fail(FS_ERR_UNSUPPORTED, std::string("fs_create: FS_MIXED does not support this configuration: ") + why);
  }
  if (c->network < FS_NET_RING || c->network > FS_NET_BOTTLENECK)
    return fail(FS_ERR_UNSUPPORTED, "fs_create: network not built");
  const bool open_net = c->network == FS_NET_MERGE || c->network == FS_NET_BOTTLENECK;
  const bool merge_env = c->env == FS_ENV_MERGE_PO || c->env == FS_ENV_MERGE_MA;
  const bool bn_env = c->env == FS_ENV_BOTTLENECK_DV || c->env == FS_ENV_BOTTLENECK;
  if ((c->network == FS_NET_MERGE) != merge_env)
    return fail(FS_ERR_INVALID, "fs_create: the merge envs and FS_NET_MERGE go together");
  if (c->network == FS_NET_MERGE && !c->track_followers)
    return fail(FS_ERR_INVALID, "fs_create: the merge observations need track_followers = 1");
  if ((c->network == FS_NET_BOTTLENECK) != bn_env)
    return fail(FS_ERR_INVALID, "fs_create: the bottleneck envs and FS_NET_BOTTLENECK go together");
  if (c->network == FS_NET_BOTTLENECK) {
    if (c->num_vehicles <= 32)
      return fail(FS_ERR_UNSUPPORTED, "fs_create: the bottleneck heads need more than 32 vehicle slots per replica");
    if (!(c->merge1_x <= c->merge2_x) || !(c->merge2_x < c->end_x) || !(c->zipper_distance >= 0))
      return fail(FS_ERR_INVALID, "fs_create: need merge1_x <= merge2_x < end_x and zipper_distance >= 0");
    if (c->junction.enabled) return fail(FS_ERR_INVALID, "fs_create: the priority-junction model is for FS_NET_MERGE");
    if (c->obs_outflow_window < 1 || c->obs_outflow_window > 20 || c->reward_outflow_window < 1 ||
        c->reward_outflow_window > 20)
      return fail(FS_ERR_INVALID, "fs_create: outflow windows must be 1..20 sub-steps");
    if (c->evaluate) return fail(FS_ERR_UNSUPPORTED, "fs_create: evaluate mode of the bottleneck envs is not built");
    if (c->num_paths != 0 && c->num_paths != 4 && c->num_paths != 8)
      return fail(FS_ERR_UNSUPPORTED, "fs_create: num_paths (4 * scaling) must be 4 or 8");
    if (c->num_paths == 8 && c->num_vehicles <= 64)
      return fail(FS_ERR_UNSUPPORTED, "fs_create: num_paths = 8 (scaling 2) is built in the workgroup-per-replica kernel: "
                                      "more than 64 vehicle slots per replica");
    if (c->env == FS_ENV_BOTTLENECK_DV) {
      const int max_cells = c->num_vehicles > 64 ? 128 : 64;
      if (c->num_obs_cells < 1 || c->num_obs_cells > max_cells || !c->obs_cells)
        return fail(FS_ERR_INVALID, "fs_create: 1..64 observation cells (1..128 with more than 64 vehicle slots)");
      for (int k = 0; k < c->num_obs_cells; ++k)
        if (c->obs_cells[k].lane < 0 || c->obs_cells[k].lane > 63) return fail(FS_ERR_INVALID, "fs_create: cell lane");
      for (int k = 0; k < c->num_rl; ++k)
        if (c->act_cells && (c->act_cells[k].lane < 0 || c->act_cells[k].lane > 63))
          return fail(FS_ERR_INVALID, "fs_create: cell lane");
      if (c->num_rl < 0 || c->num_rl > 64 || (c->num_rl > 0 && !c->act_cells))
        return fail(FS_ERR_INVALID, "fs_create: 0..64 action cells");
    }
  }
  if (open_net) {
    if (c->num_lanes > 1) return fail(FS_ERR_UNSUPPORTED, "fs_create: multi-lane merge is not built");
    if (c->num_segments < 2 || c->num_segments > 2 * FS_MAX_SEGMENTS || !c->segments)
      return fail(FS_ERR_INVALID, "fs_create: FS_NET_MERGE needs the segment tables of both routes");
    int per_route[2] = {0, 0};
    for (int k = 0; k < c->num_segments; ++k) {
      const fs_segment& sg = c->segments[k];
      if (sg.route < 0 || sg.route > (c->network == FS_NET_MERGE ? 1 : 0))
        return fail(FS_ERR_INVALID, "fs_create: segment route out of range");
      if (k > 0 && sg.route < c->segments[k - 1].route)
        return fail(FS_ERR_INVALID, "fs_create: segment rows must be grouped by route");
      if (k > 0 && sg.route == c->segments[k - 1].route && !(sg.start > c->segments[k - 1].start))
        return fail(FS_ERR_INVALID, "fs_create: segment starts must increase");
      if (++per_route[sg.route] > FS_MAX_SEGMENTS) return fail(FS_ERR_INVALID, "fs_create: too many segments");
    }
    if (!per_route[0] || (c->network == FS_NET_MERGE && !per_route[1]))
      return fail(FS_ERR_INVALID, "fs_create: a route has no segment");
    if (c->num_inflows < 0 || c->num_inflows > FS_MAX_INFLOWS || (c->num_inflows > 0 && !c->inflows))
      return fail(FS_ERR_INVALID, "fs_create: bad inflow table");
    for (int f = 0; f < c->num_inflows; ++f) {
      const fs_inflow& fl = c->inflows[f];
      if (c->network == FS_NET_MERGE ? (fl.route < 0 || fl.route > 1)
                                     : (fl.route < -1 || fl.route > (c->num_paths == 8 ? 7 : 3)))
        return fail(FS_ERR_INVALID, "fs_create: inflow route out of range");
      if (fl.probability > 1.0) return fail(FS_ERR_INVALID, "fs_create: inflow probability > 1");
      if (!(fl.probability >= 0.0) && !(fl.period > 0)) return fail(FS_ERR_INVALID, "fs_create: inflow period <= 0");
      if (!(fl.depart_speed >= 0) || !(fl.depart_pos >= 0)) return fail(FS_ERR_INVALID, "fs_create: bad inflow departure");
      bool type_ok = false;
      for (int i = 0; i < c->num_vehicles; ++i) type_ok = type_ok || c->vehicles[i].type == fl.type;
      if (!type_ok) return fail(FS_ERR_INVALID, "fs_create: inflow of a vehicle type that has no slot");
    }
    if (!c->init_alive || !c->init_lane) return fail(FS_ERR_INVALID, "fs_create: FS_NET_MERGE needs init_alive and init_lane (routes)");
    if (!(c->box_in < c->merge_x) || !(c->merge_x < c->end_x) || !(c->net_length > 0))
      return fail(FS_ERR_INVALID, "fs_create: need box_in < merge_x < end_x and net_length > 0");
  }
  if (c->network == FS_NET_RING && (c->num_segments != 0 || c->junction.enabled))
    return fail(FS_ERR_INVALID, "fs_create: a ring takes no segment table / junction");
  if (c->network == FS_NET_FIGURE_EIGHT) {
    if (c->num_segments < 1 || c->num_segments > FS_MAX_SEGMENTS || !c->segments)
      return fail(FS_ERR_INVALID, "fs_create: figure eight needs 1..FS_MAX_SEGMENTS segments");
    if (c->num_lanes > 1) return fail(FS_ERR_UNSUPPORTED, "fs_create: multi-lane figure eight is not built");
    if (c->segments[0].start != 0.0) return fail(FS_ERR_INVALID, "fs_create: segment 0 must start at 0");
    for (int k = 1; k < c->num_segments; ++k)
      if (!(c->segments[k].start > c->segments[k - 1].start))
        return fail(FS_ERR_INVALID, "fs_create: segment starts must increase");
  }
  if (c->env < FS_ENV_ACCEL || c->env > FS_ENV_LANE_CHANGE_ACCEL_PO) return fail(FS_ERR_INVALID, "fs_create: bad env");
  const bool lc_env = c->env == FS_ENV_LANE_CHANGE_ACCEL || c->env == FS_ENV_LANE_CHANGE_ACCEL_PO;
  if (lc_env && c->network != FS_NET_RING)
    return fail(FS_ERR_UNSUPPORTED, "fs_create: the lane-change envs are built for rings (k_steps_ml)");
  if (c->env == FS_ENV_LANE_CHANGE_ACCEL_PO && (c->num_rl < 1 || c->obs_perm))
    return fail(FS_ERR_INVALID, "fs_create: LaneChangeAccelPOEnv needs an RL vehicle and takes no observation permutation");
  if (c->env == FS_ENV_WAVE_ATTENUATION_PO_MA || c->env == FS_ENV_ACCEL_PO_MA) {
    if (open_net || c->num_lanes > 1)
      return fail(FS_ERR_UNSUPPORTED, "fs_create: the multi-agent ring heads are built for single-lane closed loops");
    if (c->num_rl < 1) return fail(FS_ERR_INVALID, "fs_create: a multi-agent ring env needs an RL vehicle");
    if (c->sort_vehicles || c->obs_perm) return fail(FS_ERR_INVALID, "fs_create: sort_vehicles / shuffled ids belong to AccelEnv");
  }
  if (c->num_lanes > 1 && c->env == FS_ENV_WAVE_ATTENUATION_PO)
    return fail(FS_ERR_UNSUPPORTED, "fs_create: WaveAttenuationPOEnv on a multi-lane ring is not built");
  if (c->num_lanes > 64) return fail(FS_ERR_INVALID, "fs_create: num_lanes > 64");
  if (c->init_lane && !open_net)
    for (size_t e = 0; e < size_t(c->num_replicas > 0 ? c->num_replicas : 0) * (c->num_vehicles > 0 ? c->num_vehicles : 0); ++e)
      if (c->init_lane[e] < 0 || c->init_lane[e] >= (c->num_lanes < 1 ? 1 : c->num_lanes))
        return fail(FS_ERR_INVALID, "fs_create: init_lane out of range");
  if (c->num_replicas < 1) return fail(FS_ERR_INVALID, "fs_create: num_replicas < 1");
  if (c->replica_offset < 0) return fail(FS_ERR_INVALID, "fs_create: replica_offset < 0");
  if (c->sort_vehicles || c->obs_perm) {
    if (c->env != FS_ENV_ACCEL && !lc_env && c->sort_vehicles)
      return fail(FS_ERR_INVALID, "fs_create: sort_vehicles belongs to AccelEnv / LaneChangeAccelEnv");
    if (c->network == FS_NET_MERGE || c->network == FS_NET_BOTTLENECK ||
        (c->num_lanes > 1 && c->sort_vehicles && !lc_env))
      return fail(FS_ERR_UNSUPPORTED, "fs_create: sort_vehicles / shuffled ids are built for closed loops "
                                      "(sort_vehicles on multi-lane rings: LaneChangeAccelEnv)");
    if (c->obs_perm) {
      unsigned long long seen = 0ull;
      for (int i = 0; i < c->num_vehicles && i < 64; ++i) {
        const int q = c->obs_perm[i];
        if (q < 0 || q >= c->num_vehicles || (seen >> q) & 1ull)
          return fail(FS_ERR_INVALID, "fs_create: obs_perm is not a permutation");
        seen |= 1ull << q;
      }
    }
  }
  if (c->num_vehicles < 1) return fail(FS_ERR_INVALID, "fs_create: num_vehicles < 1");
  if (c->num_vehicles > 64 && c->network != FS_NET_BOTTLENECK)
    return fail(FS_ERR_UNSUPPORTED, "fs_create: more than 64 vehicles per replica is built for FS_NET_BOTTLENECK only");
  if (c->num_vehicles > FS_MAX_SLOTS_WIDE)
    return fail(FS_ERR_UNSUPPORTED, "fs_create: more than 256 vehicle slots per replica is not built");
  if (c->num_rl < 0 || (c->num_rl > c->num_vehicles && c->env != FS_ENV_BOTTLENECK_DV))
    return fail(FS_ERR_INVALID, "fs_create: bad num_rl");
  if (c->sims_per_step < 1) return fail(FS_ERR_INVALID, "fs_create: sims_per_step < 1");
  if (c->warmup_steps < 0) return fail(FS_ERR_INVALID, "fs_create: warmup_steps < 0");
  if (!(c->sim_step > 0)) return fail(FS_ERR_INVALID, "fs_create: sim_step <= 0");
  if (!(c->slowdown_ramp > 0) || c->slowdown_ramp > 1) return fail(FS_ERR_INVALID, "fs_create: slowdown_ramp not in (0,1]");
  if (c->junction_length < 0) return fail(FS_ERR_INVALID, "fs_create: junction_length < 0");
  if (!c->vehicles || (!c->ring_length && !open_net) || !c->init_pos)
    return fail(FS_ERR_INVALID, "fs_create: NULL table pointer");
  int seen_rl = 0;
  unsigned long long rl_mask = 0ull;       // num_vehicles <= 64
  for (int i = 0; i < c->num_vehicles; ++i) {
    const fs_vehicle_spec& v = c->vehicles[i];
    if (v.controller < FS_CTRL_SIM || v.controller > FS_CTRL_USER)
      return fail(FS_ERR_INVALID, "fs_create: unknown controller id");
    if (v.fail_safe < FS_FAILSAFE_NONE || v.fail_safe > FS_FAILSAFE_SAFE_VELOCITY)
      return fail(FS_ERR_INVALID, "fs_create: unknown fail_safe id");
    if (open_net && v.controller == FS_CTRL_PISATURATION)
      return fail(FS_ERR_UNSUPPORTED, "fs_create: PISaturation on an open network is not built");
    if (v.controller == FS_CTRL_RL && (c->env == FS_ENV_MERGE_PO || bn_env)) {
      ++seen_rl;                        // the action column is the place in rl_veh, not rl_index
    } else if (v.controller == FS_CTRL_RL) {
      if (v.rl_index < 0 || v.rl_index >= c->num_rl) return fail(FS_ERR_INVALID, "fs_create: rl_index out of range");
      if (rl_mask & (1ull << v.rl_index)) return fail(FS_ERR_INVALID, "fs_create: rl_index used twice");
      rl_mask |= 1ull << v.rl_index;
      ++seen_rl;
    }
    if (!(v.length > 0)) return fail(FS_ERR_INVALID, "fs_create: vehicle length <= 0");
  }
  if (c->env == FS_ENV_MERGE_PO && c->num_rl > 32)      // the removal pass of rl_veh keeps the list places in one 32-bit mask
    return fail(FS_ERR_UNSUPPORTED, "fs_create: MergePOEnv with more than 32 controlled places (num_rl) is not built");
  if (seen_rl != c->num_rl && c->env != FS_ENV_MERGE_PO && !bn_env)
    return fail(FS_ERR_INVALID, "fs_create: num_rl does not match the RL slots");
  if (open_net) {
    // placement sanity of the initial vehicles: inside their route, alive flags consistent
    for (size_t e = 0; e < size_t(c->num_replicas) * c->num_vehicles; ++e) {
      if (!c->init_alive[e]) continue;
      const int rt = c->init_lane[e];
      if (rt < 0 || rt > (c->network == FS_NET_MERGE ? 1 : (c->num_paths == 8 ? 7 : 3)))
        return fail(FS_ERR_INVALID, "fs_create: initial route out of range");
      const double x = c->init_pos[e];
      if (!(x >= c->route_start[c->network == FS_NET_MERGE ? rt : 0]) || !(x < c->end_x))
        return fail(FS_ERR_INVALID, "fs_create: init_pos outside the route");
    }
    return FS_OK;
  }
  if (c->env == FS_ENV_WAVE_ATTENUATION_PO && c->num_rl < 1)
    return fail(FS_ERR_INVALID, "fs_create: WaveAttenuationPOEnv needs an RL vehicle");
  // placement sanity: every replica's vehicles must fit on its loop (network/base.py:603-605)
  for (int r = 0; r < c->num_replicas; ++r) {
    double need = 0;
    for (int i = 0; i < c->num_vehicles; ++i) need += c->vehicles[i].length;
    const double L = c->ring_length[r] + 4 * c->junction_length;
    if (!(c->ring_length[r] > 0) || need > L * (c->num_lanes < 1 ? 1 : c->num_lanes))
      return fail(FS_ERR_NOSPACE, "fs_create: vehicles do not fit on the ring");
    for (int i = 0; i < c->num_vehicles; ++i) {
      const double x = c->init_pos[size_t(r) * c->num_vehicles + i];
      if (!(x >= 0) || !(x < L)) return fail(FS_ERR_INVALID, "fs_create: init_pos outside [0, length)");
    }
  }
  return FS_OK;
}

template <typename T>
int create_typed(const fs_config* cfg, fs_handle* out) {
  Sim<T>* s = new (std::nothrow) Sim<T>();
  if (!s) return fail(FS_ERR_HIP, "fs_create: out of host memory");
  s->cfg = *cfg;
  s->mixed = cfg->precision == FS_MIXED;
  s->f16s = cfg->precision == FS_F16S;
  s->veh.assign(cfg->vehicles, cfg->vehicles + cfg->num_vehicles);
  if (cfg->num_segments > 0) s->segs.assign(cfg->segments, cfg->segments + cfg->num_segments);
  if (cfg->obs_perm) s->obs_perm.assign(cfg->obs_perm, cfg->obs_perm + cfg->num_vehicles);
  if (cfg->network == FS_NET_MERGE || cfg->network == FS_NET_BOTTLENECK) {
    if (cfg->num_inflows > 0) s->inflows.assign(cfg->inflows, cfg->inflows + cfg->num_inflows);
    s->init_alive.assign(cfg->init_alive, cfg->init_alive + size_t(cfg->num_replicas) * cfg->num_vehicles);
    if (cfg->env == FS_ENV_BOTTLENECK_DV) {
      s->obs_cells.assign(cfg->obs_cells, cfg->obs_cells + cfg->num_obs_cells);
      if (cfg->num_rl > 0) s->act_cells.assign(cfg->act_cells, cfg->act_cells + cfg->num_rl);
    }
  }
  s->obs_dim = (cfg->env == FS_ENV_WAVE_ATTENUATION_PO) ? 3
               : (cfg->env == FS_ENV_WAVE_ATTENUATION_PO_MA) ? 3 * cfg->num_rl
               : (cfg->env == FS_ENV_ACCEL_PO_MA) ? 6 * cfg->num_rl
               : (cfg->env == FS_ENV_BOTTLENECK_DV) ? 4 * cfg->num_obs_cells + 1
               : (cfg->env == FS_ENV_BOTTLENECK) ? 1
               : (cfg->env == FS_ENV_MERGE_PO || cfg->env == FS_ENV_MERGE_MA)
                     ? 5 * cfg->num_rl
                     : (cfg->env == FS_ENV_LANE_CHANGE_ACCEL_PO)
                           ? (4 * (cfg->num_lanes < 1 ? 1 : cfg->num_lanes) + 1) * cfg->num_rl
                           : (cfg->env == FS_ENV_LANE_CHANGE_ACCEL ? 3 : 2) * cfg->num_vehicles;
  if (s->obs_dim < 1) s->obs_dim = 1;      // an env without RL places still gets a (dummy) buffer
  s->act_dim = cfg->num_rl * ((cfg->env == FS_ENV_LANE_CHANGE_ACCEL || cfg->env == FS_ENV_LANE_CHANGE_ACCEL_PO) ? 2 : 1);
  int seg = 8;
  while (seg < cfg->num_vehicles) seg <<= 1;       // 128 / 256: one workgroup of 2 / 4 waves per replica (k_steps_wide)
  s->seg = seg;
  hipError_t e = hipSetDevice(cfg->device);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&s->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete s;
    return fail(FS_ERR_HIP, std::string("fs_create: no usable HIP device: ") + hipGetErrorString(e));
  }
  s->stream = s->own_stream;
  {
    const char* fg = std::getenv("FLOWSIM_FORCE_GENERIC");
    s->force_generic = fg && fg[0] == '1';
    const char* nf = std::getenv("FLOWSIM_NO_FASTDIV");
    s->no_fastdiv = nf && nf[0] == '1';
    const char* nl = std::getenv("FLOWSIM_NO_LOOP_KERNEL");
    s->no_loop_kernel = nl && nl[0] == '1';
    const char* nlf = std::getenv("FLOWSIM_NO_LOOP_FULL");
    s->no_loop_full = nlf && nlf[0] == '1';
    const char* nrr = std::getenv("FLOWSIM_NO_RING_RL");
    s->no_ring_rl = nrr && nrr[0] == '1';
    const char* nq = std::getenv("FLOWSIM_NO_QUEUE");
    s->no_queue = nq && nq[0] == '1';
    const char* np = std::getenv("FLOWSIM_NO_PAIR");
    s->no_pair = np && np[0] == '1';
    const char* pb = std::getenv("FLOWSIM_PAIR_BLOCK");
    if (pb) {
      const int v = std::atoi(pb);
      if (v >= 64 && v <= 256 && v % 64 == 0) s->pair_block = v;
    }
    const char* rb = std::getenv("FLOWSIM_ROLLOUT_BLOCK");
    if (rb) {
      const int v = std::atoi(rb);
      if (v >= 64 && v <= 1024 && v % 64 == 0) s->rollout_block = v;
    }
  }
  int rc = s->init();
  if (rc == FS_OK) {
    hipError_t e2 = hipStreamSynchronize(s->stream);
    if (e2 != hipSuccess) rc = fail(FS_ERR_HIP, std::string("fs_create: ") + hipGetErrorString(e2));
  }
  if (rc != FS_OK) {
    for (void* p : s->allocs) (void)hipFree(p);
    (void)hipStreamDestroy(s->own_stream);
    delete s;
    return rc;
  }
  // the config's table pointers belong to the caller: do not keep them
  s->cfg.vehicles = nullptr;
  s->cfg.ring_length = nullptr;
  s->cfg.init_pos = nullptr;
  s->cfg.init_vel = nullptr;
  s->cfg.init_lane = nullptr;
  s->cfg.segments = nullptr;
  s->cfg.inflows = nullptr;
  s->cfg.init_alive = nullptr;
  s->cfg.obs_cells = nullptr;
  s->cfg.obs_perm = nullptr;
  s->cfg.act_cells = nullptr;
  *out = reinterpret_cast<fs_handle>(static_cast<SimBase*>(s));
  return FS_OK;
}

inline SimBase* S(fs_handle h) { return reinterpret_cast<SimBase*>(h); }

// Every entry point runs on the handle's device whatever the caller's current device is, and leaves the caller's
// current device as it found it (two handles on two GPUs in one process; torch's current device).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int want) {
    if (hipGetDevice(&prev) == hipSuccess && prev != want) switched = hipSetDevice(want) == hipSuccess;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

}  // namespace fsim

using namespace fsim;

extern "C" {

int fs_abi_version(void) { return FS_ABI_VERSION; }

const char* fs_last_error(void) { return g_err.c_str(); }

int fs_create(const fs_config* cfg, fs_handle* out) {
  if (!out) return fail(FS_ERR_INVALID, "fs_create: out is NULL");
  *out = nullptr;
  int rc = validate(cfg);
  if (rc) return rc;
  int prev = -1;
  const bool have_prev = hipGetDevice(&prev) == hipSuccess;
  rc = (cfg->precision == FS_F32 || cfg->precision == FS_F16S) ? create_typed<float>(cfg, out)
                                                                 : create_typed<double>(cfg, out);
  if (have_prev && prev != cfg->device) (void)hipSetDevice(prev);      // the caller's current device is not ours to change
  return rc;
}

void fs_destroy(fs_handle h) {
  if (!h) return;
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  (void)hipStreamSynchronize(s->stream);
  for (void* p : s->allocs) (void)hipFree(p);
  (void)hipStreamDestroy(s->own_stream);
  delete s;
}

int fs_obs_dim(fs_handle h) { return h ? S(h)->obs_dim : fail(FS_ERR_INVALID, "fs_obs_dim: NULL handle"); }

int fs_action_dim(fs_handle h) { return h ? S(h)->act_dim : fail(FS_ERR_INVALID, "fs_action_dim: NULL handle"); }

int fs_set_stream(fs_handle h, void* hip_stream) {
  if (!h) return fail(FS_ERR_INVALID, "fs_set_stream: NULL handle");
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = static_cast<hipStream_t>(hip_stream);
  return FS_OK;
}

int fs_use_own_stream(fs_handle h) {
  if (!h) return fail(FS_ERR_INVALID, "fs_use_own_stream: NULL handle");
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  HIP_TRY(hipStreamSynchronize(s->stream));
  s->stream = s->own_stream;
  return FS_OK;
}

int fs_sync(fs_handle h) {
  if (!h) return fail(FS_ERR_INVALID, "fs_sync: NULL handle");
  DeviceGuard guard(S(h)->cfg.device);
  HIP_TRY(hipStreamSynchronize(S(h)->stream));
  return S(h)->check_qflag();
}

int fs_reset_dev(fs_handle h, const uint8_t* mask_dev, float* obs_dev) {
  if (!h) return fail(FS_ERR_INVALID, "fs_reset_dev: NULL handle");
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  int rc = s->launch_reset(mask_dev);
  if (rc) return rc;
  float* obs = obs_dev ? obs_dev : s->d_obs;
  if (s->cfg.network == FS_NET_MERGE || s->cfg.network == FS_NET_BOTTLENECK) {   // update(reset=True) first
    s->after_reset = 1;
    rc = s->launch_steps(0, mask_dev, nullptr, 0, obs, s->d_rew, s->d_done, 0);
    s->after_reset = 0;
    if (rc) return rc;
    if (s->cfg.warmup_steps > 0)
      return s->launch_steps(s->cfg.warmup_steps, mask_dev, nullptr, 0, obs, s->d_rew, s->d_done, 0);
    return FS_OK;
  }
  if (s->cfg.warmup_steps > 0)   // envs/base.py:554-555: warm-up steps with rl_actions=None
    return s->launch_steps(s->cfg.warmup_steps, mask_dev, nullptr, 0, obs, s->d_rew, s->d_done, 0);
  return s->launch_steps(0, mask_dev, nullptr, 0, obs, s->d_rew, s->d_done, 0);
}

int fs_reset(fs_handle h, const uint8_t* mask, float* obs_out) {
  if (!h) return fail(FS_ERR_INVALID, "fs_reset: NULL handle");
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  const size_t R = size_t(s->cfg.num_replicas);
  const uint8_t* dmask = nullptr;
  if (mask) {
    HIP_TRY(hipMemcpyAsync(s->d_mask, mask, R, hipMemcpyHostToDevice, s->stream));
    dmask = s->d_mask;
  }
  int rc = fs_reset_dev(h, dmask, s->d_obs);
  if (rc) return rc;
  if (obs_out)
    HIP_TRY(hipMemcpyAsync(obs_out, s->d_obs, R * s->obs_dim * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return FS_OK;
}

int fs_step_dev(fs_handle h, const float* actions_dev, float* obs_dev, float* rew_dev, uint8_t* done_dev) {
  if (!h) return fail(FS_ERR_INVALID, "fs_step_dev: NULL handle");
  if (!obs_dev || !rew_dev || !done_dev) return fail(FS_ERR_INVALID, "fs_step_dev: NULL output pointer");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->launch_steps(1, nullptr, actions_dev, 0, obs_dev, rew_dev, done_dev, 0);
}

int fs_step(fs_handle h, const float* actions, float* obs, float* rew, uint8_t* done) {
  if (!h) return fail(FS_ERR_INVALID, "fs_step: NULL handle");
  if (!obs || !rew || !done) return fail(FS_ERR_INVALID, "fs_step: NULL output pointer");
  SimBase* s = S(h);
  DeviceGuard guard(s->cfg.device);
  const size_t R = size_t(s->cfg.num_replicas);
  const float* dact = nullptr;
  if (actions && s->act_dim > 0) {
    HIP_TRY(hipMemcpyAsync(s->d_actions, actions, R * s->act_dim * sizeof(float), hipMemcpyHostToDevice, s->stream));
    dact = s->d_actions;
  }
  int rc = s->launch_steps(1, nullptr, dact, 0, s->d_obs, s->d_rew, s->d_done, 0);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(obs, s->d_obs, R * s->obs_dim * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(rew, s->d_rew, R * sizeof(float), hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipMemcpyAsync(done, s->d_done, R, hipMemcpyDeviceToHost, s->stream));
  HIP_TRY(hipStreamSynchronize(s->stream));
  return s->check_qflag();        // (a queue-order launch that overflowed a path: the rows just copied are not results)
}

int fs_rollout_dev(fs_handle h, int num_steps, const float* actions_dev, size_t action_stride_steps,
                   float* obs_dev, float* rew_dev, uint8_t* done_dev, int obs_every_step) {
  if (!h) return fail(FS_ERR_INVALID, "fs_rollout_dev: NULL handle");
  if (num_steps < 1) return fail(FS_ERR_INVALID, "fs_rollout_dev: num_steps < 1");
  if (!obs_dev || !rew_dev || !done_dev) return fail(FS_ERR_INVALID, "fs_rollout_dev: NULL output pointer");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->launch_steps(num_steps, nullptr, actions_dev, action_stride_steps, obs_dev, rew_dev, done_dev,
                            obs_every_step ? 1 : 0);
}

int fs_get_state(fs_handle h, int field, void* dst, size_t bytes) {
  if (!h || !dst) return fail(FS_ERR_INVALID, "fs_get_state: NULL argument");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->get_state(field, dst, bytes);
}

int fs_set_state(fs_handle h, int field, const void* src, size_t bytes) {
  if (!h || !src) return fail(FS_ERR_INVALID, "fs_set_state: NULL argument");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->set_state(field, src, bytes);
}

int fs_add_vehicle(fs_handle h, int replica, int slot, int route, double x, double speed) {
  if (!h) return fail(FS_ERR_INVALID, "fs_add_vehicle: NULL handle");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->add_vehicle(replica, slot, route, x, speed);
}

int fs_policy_act_dev(fs_handle h, const fs_policy* pol, const float* obs_dev, float* act_dev, float* logp_dev) {
  if (!h || !pol || !obs_dev || !act_dev || !logp_dev) return fail(FS_ERR_INVALID, "fs_policy_act_dev: NULL argument");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->launch_policy(pol, 0, 0, obs_dev, nullptr, act_dev, logp_dev, nullptr, nullptr);
}

int fs_policy_rollout_dev(fs_handle h, const fs_policy* pol, int num_steps, int reset_done, float* obs_dev,
                          float* act_dev, float* logp_dev, float* rew_dev, uint8_t* done_dev) {
  if (!h || !pol || !obs_dev || !act_dev || !logp_dev || !rew_dev || !done_dev)
    return fail(FS_ERR_INVALID, "fs_policy_rollout_dev: NULL argument");
  if (num_steps < 1) return fail(FS_ERR_INVALID, "fs_policy_rollout_dev: num_steps < 1");
  DeviceGuard guard(S(h)->cfg.device);
  return S(h)->launch_policy(pol, num_steps, reset_done ? 1 : 0, nullptr, obs_dev, act_dev, logp_dev, rew_dev, done_dev);
}

const char* fs_last_kernel(fs_handle h) { return h ? reinterpret_cast<SimBase*>(h)->last_kernel : ""; }

int fs_dump_trajectory(fs_handle h, int replica, const char* csv_path) {
  if (!h || !csv_path) return fail(FS_ERR_INVALID, "fs_dump_trajectory: NULL argument");
  SimBase* s = S(h);
  if (replica < 0 || replica >= s->cfg.num_replicas) return fail(FS_ERR_INVALID, "fs_dump_trajectory: replica out of range");
  DeviceGuard guard(s->cfg.device);
  const size_t R = size_t(s->cfg.num_replicas), N = size_t(s->cfg.num_vehicles);
  const bool f32 = s->cfg.precision == FS_F32 || s->cfg.precision == FS_F16S;
  const size_t el = f32 ? sizeof(float) : sizeof(double);
  std::vector<char> pos(R * N * el), vel(R * N * el);
  std::vector<int32_t> tc(R), lane(R * N, 0);
  int rc = s->get_state(FS_FIELD_POS, pos.data(), pos.size());
  if (!rc) rc = s->get_state(FS_FIELD_VEL, vel.data(), vel.size());
  if (!rc) rc = s->get_state(FS_FIELD_TIME, tc.data(), R * sizeof(int32_t));
  const bool open_net = s->cfg.network == FS_NET_MERGE || s->cfg.network == FS_NET_BOTTLENECK;
  if (!rc && (open_net || s->cfg.num_lanes > 1))
    rc = s->get_state(open_net ? FS_FIELD_ROUTE : FS_FIELD_LANE, lane.data(), R * N * sizeof(int32_t));
  if (rc) return rc;
  FILE* probe = std::fopen(csv_path, "r");
  const bool fresh = probe == nullptr;
  if (probe) std::fclose(probe);
  FILE* f = std::fopen(csv_path, "a");
  if (!f) return fail(FS_ERR_INVALID, std::string("fs_dump_trajectory: cannot open ") + csv_path);
  if (fresh) std::fprintf(f, "time,id,x,speed,lane_number\n");
  const double t = double(tc[replica]) * s->cfg.sim_step;
  for (size_t i = 0; i < N; ++i) {
    const size_t e = size_t(replica) * N + i;
    if (open_net && lane[e] < 0) continue;                       // a free slot: no vehicle
    const double x = f32 ? double(reinterpret_cast<const float*>(pos.data())[e]) : reinterpret_cast<const double*>(pos.data())[e];
    const double v = f32 ? double(reinterpret_cast<const float*>(vel.data())[e]) : reinterpret_cast<const double*>(vel.data())[e];
    std::fprintf(f, "%.6f,%zu,%.17g,%.17g,%d\n", t, i, x, v, int(lane[e]));
  }
  std::fclose(f);
  return FS_OK;
}

}  // extern "C"
