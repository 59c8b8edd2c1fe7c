/*
 * flowsim.h -- C ABI of libflowsim.so, the MI355X-native batched traffic
 * micro-simulation step loop.
 *
 * The reference (parthjaggi/flow) has no native boundary on this path: its
 * "FFI" is the TraCI TCP protocol between flow.envs.Env and a SUMO subprocess.
 * Each entry point below replaces the group of reference calls cited next to
 * it (paths relative to the reference root).  Plain pointers and sizes only;
 * no torch / numpy types.  All functions return FS_OK (0) or a negative
 * FS_ERR_* code; fs_last_error() returns the message of the last failure on
 * the calling thread.
 *
 * Threading: a handle is not thread-safe; use one handle per (process, GPU)
 * (reference: one Env + one SUMO process per rollout worker,
 * examples/train.py:149).  All device work of a handle is enqueued on one HIP
 * stream (fs_set_stream) and is asynchronous with respect to the host unless
 * the call copies results to host memory.
 *
 * Ownership: the library owns the simulator state in HBM; the caller owns
 * every buffer it passes in.  "_dev" entry points take device pointers
 * (hipMalloc / torch-ROCm storage); the others take host pointers and copy.
 */
#ifndef FLOWSIM_H
#define FLOWSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FS_ABI_VERSION 8

/* ---- error codes -------------------------------------------------------- */
#define FS_OK 0
#define FS_ERR_INVALID -1       /* bad argument / config (Python: ValueError / KeyError)   */
#define FS_ERR_UNSUPPORTED -2   /* valid in the reference, not built yet (NotImplementedError) */
#define FS_ERR_HIP -3           /* HIP runtime failure (FatalFlowError)                    */
#define FS_ERR_NOSPACE -4       /* vehicles do not fit (network/base.py:603-605)            */

/* ---- enums -------------------------------------------------------------- */
/* FS_F32 / FS_F64: arithmetic and state in that type (FS_F64 = the reference's Python-float arithmetic).
 * FS_MIXED: positions and speeds kept and integrated in float64, the acceleration controller evaluated in
 * float32 on their rounded images -- within 1e-4 of the float64 trajectories over a 1500-step episode at
 * float32 cost.  Built for single-lane rings of IDM and RL vehicles (AccelEnv and WaveAttenuationPOEnv heads, warm-up
 * steps, masked resets, acceleration noise), for the figure eight (rollouts of up to 16 vehicles; what that kernel
 * does not cover steps in float64) and for the open networks up to 64 vehicle slots (the float64 kernel with float32
 * car-following models; beyond 64 slots: plain float64); fs_create names the field of a configuration that does not
 * fit.  State fields are float64 as for FS_F64. */
enum fs_precision { FS_F32 = 0, FS_F64 = 1, FS_MIXED = 2,
                    FS_F16S = 3    /* "fp16 state, fp32 integrator" (BASELINE configs[4]): the positions and speeds a handle
                                      keeps in HBM BETWEEN launches are IEEE half values -- a speed is one half, a position
                                      two (x = hi + lo, 22 significant bits: a single half has a 0.5 m ulp at 700 m) -- and a
                                      launch loads them into float32 registers, steps in float32 exactly as FS_F32 does, and
                                      stores halves again.  6 bytes of state per vehicle instead of 8; every launch boundary
                                      rounds the speeds to 11 bits (<= 0.008 m/s below 32 m/s).  Built for FS_NET_MERGE
                                      (the configuration BASELINE names); fs_get_state / fs_set_state speak float32.
                                      NOTE: a trajectory therefore depends on where the launches are cut -- K fs_step calls
                                      round K times, one fs_rollout_dev of K steps once; runs are reproducible for a fixed
                                      launch pattern only (tests/test_f16s_gpu.py measures the drift between the two). */
};

/* acceleration controllers, flow/controllers/__init__.py */
enum fs_controller {
  FS_CTRL_SIM = 0,              /* SimCarFollowingController: never commanded          car_following_models.py:485-497 */
  FS_CTRL_RL = 1,               /* RLController: commanded by the action vector        rlcontroller.py:6-39 */
  FS_CTRL_IDM = 2,              /* p = {v0, T, a, b, delta, s0}                         car_following_models.py:400-482 */
  FS_CTRL_CFM = 3,              /* p = {k_d, k_v, k_c, d_des, v_des}                    :17-88 */
  FS_CTRL_BCM = 4,              /* p = {k_d, k_v, k_c, d_des, v_des}                    :91-176 */
  FS_CTRL_LAC = 5,              /* p = {k_1, k_2, h, tau}; stateful a                   :179-245 */
  FS_CTRL_OVM = 6,              /* p = {alpha, beta, h_st, h_go, v_max}                 :248-328 */
  FS_CTRL_LINEAR_OVM = 7,       /* p = {v_max, adaptation, h_st}                        :331-397 */
  FS_CTRL_GIPPS = 8,            /* p = {v0, acc, b, b_l, s0, tau}                       :500-582 */
  FS_CTRL_FOLLOWER_STOPPER = 9, /* p = {v_des}                                          velocity_controllers.py:7-116 */
  FS_CTRL_NONLOCAL_FOLLOWER_STOPPER = 10, /* v_des = replica mean speed                 velocity_controllers.py:119-164 */
  FS_CTRL_PISATURATION = 11,    /* no parameters; keeps int(38/sim_step)-1 past speeds  velocity_controllers.py:167-240 */
  FS_CTRL_USER = 12             /* a controller of the USER's (the reference lets user code subclass BaseController and write
                                   get_accel in Python, base_controller.py:42-118): its get_accel is a device function compiled
                                   INTO a copy of this library -- `python -m flow_amd.build` with
                                   -DFS_USER_CONTROLLER_HEADER, see flow_amd.controllers.CompiledController and
                                   flow_amd.build.build_user; p = its (up to 8) parameters.  The stock library refuses a
                                   launch with such a slot (FS_ERR_UNSUPPORTED) */
};

/* BaseController fail-safes, flow/controllers/base_controller.py:113-116 */
enum fs_failsafe { FS_FAILSAFE_NONE = 0, FS_FAILSAFE_INSTANTANEOUS = 1, FS_FAILSAFE_SAFE_VELOCITY = 2 };

/* environments (observation + reward heads) */
enum fs_env {
  FS_ENV_ACCEL = 0,                 /* AccelEnv                 flow/envs/ring/accel.py:25-183 */
  FS_ENV_WAVE_ATTENUATION = 1,      /* WaveAttenuationEnv       flow/envs/ring/wave_attenuation.py:50-210 */
  FS_ENV_WAVE_ATTENUATION_PO = 2,   /* WaveAttenuationPOEnv     flow/envs/ring/wave_attenuation.py:213-276 */
  FS_ENV_LANE_CHANGE_ACCEL = 3,     /* LaneChangeAccelEnv       flow/envs/ring/lane_change_accel.py:28-154;
                                       actions are [acc_0, dir_0, acc_1, dir_1, ...] (2 per RL vehicle) */
  FS_ENV_MERGE_PO = 4,              /* MergePOEnv               flow/envs/merge.py:28-231: num_rl controlled vehicles
                                       (rl_queue / rl_veh slotting), obs 5*num_rl, action column = place in rl_veh */
  FS_ENV_MERGE_MA = 5,              /* MultiAgentMergePOEnv     flow/envs/multiagent/merge.py:19-190: one 5-vector and
                                       one action column per RL slot (column = rl_index), shared reward, no crash */
  FS_ENV_BOTTLENECK_DV = 6,         /* BottleneckDesiredVelocityEnv  flow/envs/bottleneck.py:760-1085: 4 values per
                                       observed lane-segment + outflow; one action per controlled lane-segment (num_rl =
                                       num_act_cells) shifting the maxSpeed of the RL vehicles inside it.  The mean
                                       speed of a lane-segment: FS_F32 handles with more than 64 vehicle slots, and the
                                       queue-order kernel k_drop_queue, add the speeds as integers of 2^-16 m/s (exact,
                                       order-free; oracle/opennet.py cell_sum = 'fixed'); FS_F64 and <= 64 slots add
                                       them as floats in slot order */
  FS_ENV_BOTTLENECK = 7,            /* BottleneckEnv            flow/envs/bottleneck.py:83-483: observation [1], outflow reward */
  FS_ENV_WAVE_ATTENUATION_PO_MA = 8,/* MultiAgentWaveAttenuationPOEnv  flow/envs/multiagent/ring/wave_attenuation.py:128-252:
                                       per RL vehicle (column = rl_index) [v / 15, (v_lead - v) / 15, headway / max_length],
                                       obs 3 * num_rl; WaveAttenuationEnv's reward shared by the agents; crash = 0 */
  FS_ENV_LANE_CHANGE_ACCEL_PO = 10, /* LaneChangeAccelPOEnv     flow/envs/ring/lane_change_accel.py:163-262: LaneChangeAccelEnv's
                                       actions and reward; per RL vehicle (column c = rl_index) and lane q its nearest
                                       leader / follower of that lane as flow/core/kernel/vehicle/traci.py:776-867 finds them:
                                       obs[4 * lanes * c + {0, 1, 2, 3} * lanes + q] = gap to the leader [m], gap to the follower
                                       [m], leader speed / v_max, follower speed / v_max (1000, 1000, 0, 0 for an empty lane; a
                                       vehicle alone in its lane is its own leader and follower, one lap away), then
                                       obs[4 * lanes * num_rl + c] = own speed [m/s]; obs 4 * lanes * num_rl + num_rl */
  FS_ENV_ACCEL_PO_MA = 9            /* MultiAgentAccelPOEnv     flow/envs/multiagent/ring/accel.py:84-227: per RL vehicle
                                       [x / L, v / v_max, (v_lead - v) / v_max, (x_lead - x - len_ego) / L,
                                       (v - v_follow) / v_max, headway(follower) / L], obs 6 * num_rl; desired_velocity
                                       reward shared by the agents; crash = 0 (multiagent/base.py:188-190) */
};

enum fs_network {
  FS_NET_RING = 0,          /* RingNetwork, flow/networks/ring.py (any number of lanes) */
  FS_NET_FIGURE_EIGHT = 1,  /* FigureEightNetwork, flow/networks/figure_eight.py: a closed one-lane loop that
                               crosses itself; described by fs_config.segments + fs_config.junction */
  FS_NET_MERGE = 2,         /* MergeNetwork, flow/networks/merge.py: an OPEN one-lane network, two routes (0 = highway,
                               1 = on-ramp) converging at a priority junction; vehicles enter through fs_config.inflows
                               and leave at end_x.  num_vehicles is the slot CAPACITY of a replica. */
  FS_NET_BOTTLENECK = 3     /* BottleneckNetwork, flow/networks/bottleneck.py (scaling 1 or 2): an OPEN network of
                               num_paths = 4 * scaling entry lanes that join pairwise at two zipper junctions
                               (4 -> 2 -> 1, or 8 -> 4 -> 2); a vehicle's "route" is the entry lane it continues,
                               0..num_paths-1 (the simplified lane changing of M11 moves it to a neighbouring one) */
};

enum fs_integrator { FS_EULER = 0, FS_BALLISTIC = 1 /* SumoParams.use_ballistic, core/params.py:578-602 */ };

/* fields of fs_get_state / fs_set_state (the read accessors of
 * flow/core/kernel/vehicle/base.py:327-672 and the test back-doors of
 * flow/core/kernel/vehicle/traci.py:411-425) */
enum fs_field {
  FS_FIELD_POS = 0,        /* real[R,N]  absolute position along the loop (get_x_by_id) */
  FS_FIELD_VEL = 1,        /* real[R,N]  get_speed                                       */
  FS_FIELD_HEADWAY = 2,    /* real[R,N]  get_headway (derived; read-only)                */
  FS_FIELD_PREV_VEL = 3,   /* real[R,N]  get_previous_speed                              */
  FS_FIELD_ACCEL = 4,      /* real[R,N]  last commanded acceleration (0 if uncommanded)  */
  FS_FIELD_TIME = 5,       /* int32[R]   Env.time_counter                                */
  FS_FIELD_RING_LENGTH = 6,/* real[R]    per-replica ring length (edges only)            */
  FS_FIELD_INIT_POS = 7,   /* real[R,N]  Env.initial_state positions                     */
  FS_FIELD_INIT_VEL = 8,   /* real[R,N]  Env.initial_state speeds                        */
  FS_FIELD_CTRL_STATE = 9, /* real[R,N]  controller state (LAC: self.a)                  */
  FS_FIELD_LANE = 10,      /* int32[R,N] get_lane                                        */
  FS_FIELD_LAST_LC = 11,   /* int32[R,N] time_counter of the last lane change (vehicle/traci.py:205-209) */
  FS_FIELD_LEADER = 12,    /* int32[R,N] slot of the own-lane leader, -1 if none (get_leader; read-only) */
  FS_FIELD_INIT_LANE = 13, /* int32[R,N] Env.initial_state lanes (open networks: routes)  */
  /* open networks only (read-only except ROUTE) */
  FS_FIELD_ROUTE = 14,     /* int32[R,N] route of the vehicle in the slot, -1 = free slot (alias of LANE) */
  FS_FIELD_SEQ = 15,       /* int32[R,N] place in the id list = departure order (vehicle/traci.py:279-280)  */
  FS_FIELD_ORIGIN = 16,    /* int32[R,N] inflow * 2^20 + running number ("flow_<f>.<k>"), or -1-i for initial vehicle i */
  FS_FIELD_FOLLOWER = 17,  /* int32[R,N] get_follower: the sticky follower of vehicle/traci.py:243-250, -1 if none */
  FS_FIELD_CTL_SEQ = 18,   /* int32[R,N] >= 0: the vehicle is in MergePOEnv.rl_veh, value = order of joining */
  FS_FIELD_COUNTERS = 19,  /* int32[R,8] {steps since simulator start, vehicles ever departed (id counter), rl_veh
                              join counter, arrived last sub-step, departed last sub-step, arrived total,
                              departed total, random-lane vehicles dropped at insertion}  (get_num_arrived /
                              get_outflow_rate inputs, vehicle/traci.py:493-533) */
  FS_FIELD_ARRIVED_RL = 20,/* int32[R,N] 1: the RL vehicle of this slot arrived in the last sub-step (get_arrived_rl_ids) */
  FS_FIELD_MAX_SPEED = 21, /* real[R,N]  get_max_speed / set_max_speed: maxSpeed of the SUMO car-following model */
  FS_FIELD_SORT_KEY = 23,  /* real[R,N] AccelEnv.absolute_position as of the last additional_command (sort_vehicles only,
                              flow/envs/ring/accel.py:150-169): the key the kernel ranks observations and actions by */
  FS_FIELD_INIT_RING_LENGTH = 22 /* real[R] the ring length a replica takes at its NEXT reset (WaveAttenuationEnv.reset
                              draws a new length per episode, flow/envs/ring/wave_attenuation.py:157-210): fs_reset[_dev]
                              copies it into FS_FIELD_RING_LENGTH for the replicas it resets, so a reset inside a
                              captured graph can change the length without touching replicas that are mid-episode.
                              Writing FS_FIELD_RING_LENGTH sets both. */
};

#define FS_MAX_CTRL_PARAMS 8
#define FS_MAX_SEGMENTS 16
#define FS_MAX_INFLOWS 8
#define FS_MAX_SLOTS_WIDE 256   /* vehicle slots per replica on FS_NET_BOTTLENECK (64 everywhere else) */

/* One edge of a closed loop in route order (non-ring networks).  `start` is the loop coordinate of the
 * edge's first metre; Flow's own coordinate of a point `x` on it (get_x_by_id, vehicle/traci.py:1011-1017
 * with the network's edge-start table) is flow_start + flow_slope * (x - start); internal edges without
 * a table entry have slope 0 (network/traci.py:280-287). */
typedef struct fs_segment {
  double start;
  double flow_start;
  double flow_slope;
  int32_t internal;                   /* 1: junction-internal edge (':...'): no Flow command with junction_mode */
  int32_t route;                      /* FS_NET_MERGE: the route this row belongs to (rows of a route are contiguous,
                                         starts increasing); 0 otherwise (FS_NET_BOTTLENECK: one table for all lanes) */
} fs_segment;

/* One InFlows.add entry (flow/core/params.py:1080-1213) of an open network. */
typedef struct fs_inflow {
  int32_t type;                       /* vehicle type = fs_vehicle_spec.type of the slots it may occupy */
  int32_t route;                      /* route of its edge (0 highway, 1 on-ramp); FS_NET_BOTTLENECK: entry lane, or -1 for
                                         departLane = "random" (drawn per vehicle from the Philox stream) */
  int32_t number;                     /* total vehicles to create, < 0 = unlimited */
  int32_t reserved;
  double period;                      /* seconds between vehicles: 3600 / vehs_per_hour, or `period` */
  double begin, end;                  /* first departure time / end of the departure interval [s] */
  double depart_speed;                /* departSpeed [m/s] */
  double depart_pos;                  /* front position on the first edge at insertion (SUMO "base": the vehicle length) */
  double probability;                 /* < 0: equally spaced vehicles (`period`).  In [0, 1]: InFlows.add(probability=p),
                                         params.py:1103-1105 -- in every simulation sub-step between begin and end a
                                         vehicle is generated with probability p * sim_step (at most `number` of them);
                                         generated vehicles wait their turn like due ones (M2b, Philox-drawn) */
} fs_inflow;

/* The self-crossing of the figure eight (docs/HISTORY.md S-J; SUMO's junction logic restated, unpinned).
 * Stream a (bottom->top, priority 78) crosses stream b (right->left, priority 46),
 * flow/networks/figure_eight.py:126-154. */
typedef struct fs_junction {
  int32_t enabled;
  int32_t reserved;
  double a_in, a_out;                 /* loop coordinates of the internal edge of stream a */
  double b_in, b_out;                 /* ... of stream b */
  double lookahead;                   /* a vehicle this close to its entry line may have to yield */
  double time_gap;                    /* stream a blocks the box when it reaches a_in within time_gap */
  double za_lo, za_hi, zb_lo, zb_hi;  /* front positions at which a body covers the crossing point */
} fs_junction;

/* One lane-segment of the bottleneck environments (bottleneck.py:796-812): the vehicles on `lane` whose position
 * on the edge lies in (lo, hi]  (np.searchsorted(slices, pos) - 1). */
typedef struct fs_cell {
  double edge_start;                  /* coordinate of the first metre of the edge */
  double lo, hi;                      /* segment boundaries, relative to the edge start */
  int32_t lane;                       /* lane index on that edge */
  int32_t last_segment;               /* 1: last segment of its edge (takes a vehicle standing exactly at pos 0) */
} fs_cell;

/* One vehicle slot; identical for every replica (VehicleParams.add,
 * flow/core/params.py:236-351, expanded per vehicle). */
typedef struct fs_vehicle_spec {
  int32_t controller;                 /* enum fs_controller */
  int32_t fail_safe;                  /* enum fs_failsafe */
  int32_t speed_mode;                 /* SUMO speed-mode bitmask, core/params.py:12-18 */
  int32_t rl_index;                   /* column of the action vector, -1 if not RL */
  int32_t type;                       /* open networks: index of the vehicle type (VehicleParams.add order) */
  int32_t lane_change_mode;           /* SumoLaneChangeParams.lane_change_mode of the type; FS_NET_BOTTLENECK: a strategic /
                                         cooperative / speed-gain / keep-right bit (mode & 0x55) switches the simplified
                                         lane-change model on for this vehicle (docs/HISTORY.md M11) */
  double p[FS_MAX_CTRL_PARAMS];       /* controller parameters, see enum fs_controller */
  double noise;                       /* sigma of the Gaussian acceleration noise */
  double delay;                       /* delay used by the safe_velocity fail-safe */
  double max_accel;                   /* SumoCarFollowingParams accel */
  double max_decel;                   /* |SumoCarFollowingParams decel| */
  double length;                      /* vehicle length [m] */
  double sumo_tau;                    /* SumoCarFollowingParams tau */
  double sumo_min_gap;                /* SumoCarFollowingParams min_gap */
  double sumo_max_speed;              /* SumoCarFollowingParams max_speed */
  double initial_speed;               /* VehicleParams.add(initial_speed=) */
} fs_vehicle_spec;

typedef struct fs_config {
  uint32_t struct_size;               /* sizeof(fs_config), ABI check */
  uint32_t abi_version;               /* FS_ABI_VERSION */
  int32_t precision;                  /* enum fs_precision: arithmetic + state type */
  int32_t network;                    /* enum fs_network */
  int32_t env;                        /* enum fs_env */
  int32_t integrator;                 /* enum fs_integrator */
  int32_t num_replicas;               /* R */
  int32_t num_vehicles;               /* N per replica (<= 64; FS_NET_BOTTLENECK: <= FS_MAX_SLOTS_WIDE); open networks: slot capacity */
  int32_t num_rl;                     /* RL vehicles per replica; FS_ENV_MERGE_PO: env_params 'num_rl' */
  int32_t horizon;                    /* EnvParams.horizon; <0 means inf */
  int32_t warmup_steps;               /* EnvParams.warmup_steps */
  int32_t sims_per_step;              /* EnvParams.sims_per_step */
  int32_t junction_mode;              /* 1: no Flow command while on an internal edge (base_controller.py:98-99) */
  int32_t clip_actions;               /* EnvParams.clip_actions */
  int32_t evaluate;                   /* EnvParams.evaluate */
  int32_t device;                     /* HIP device ordinal */
  int32_t track_aux;                  /* 1: keep FS_FIELD_PREV_VEL / FS_FIELD_ACCEL up to date (get_previous_speed) */
  int32_t num_lanes;                  /* lanes of the ring (net_params 'lanes'); > 1 selects the multi-lane kernel */
  int32_t lane_change_mode;           /* SumoLaneChangeParams.lane_change_mode: 0 = execute every commanded change,
                                         otherwise refuse a change that would overlap a vehicle of the target lane */
  int32_t last_lc_quirk;              /* 1: get_last_lc returns the headway, as this fork does (vehicle/traci.py:604-614) */
  uint64_t seed;                      /* SimParams.seed: key of the per-(replica,vehicle,step) noise stream */
  double sim_step;                    /* SimParams.sim_step */
  double slowdown_ramp;               /* v' = v + (next_vel - v)*ramp; dt/(dt+1e-3) models slowDown(.., 1e-3) */
  double junction_length;             /* length of each internal edge */
  double crash_gap;                   /* crash <=> some headway < crash_gap after the move */
  double max_speed;                   /* k.network.max_speed() */
  double target_velocity;             /* env_params.additional_params['target_velocity'] */
  double action_low, action_high;     /* action_space bounds */
  double po_max_length;               /* WaveAttenuationPOEnv max_length normaliser */
  double lane_change_duration;        /* env_params 'lane_change_duration' (lane_change_accel.py:143-147) */
  const fs_vehicle_spec* vehicles;    /* [N] */
  const double* ring_length;          /* [R] length of each replica's ring (sum of its 4 edges) */
  const double* init_pos;             /* [R,N] initial absolute positions */
  const double* init_vel;             /* [R,N] initial speeds, or NULL -> vehicles[i].initial_speed */
  const int32_t* init_lane;           /* [R,N] initial lanes, or NULL -> lane 0 */
  const fs_segment* segments;         /* [num_segments] edge table of a non-ring loop, or NULL */
  int32_t num_segments;               /* 0 for FS_NET_RING */
  int32_t num_inflows;                /* FS_NET_MERGE: entries of `inflows` (<= FS_MAX_INFLOWS) */
  fs_junction junction;               /* crossing model (figure eight) / merge right of way: for FS_NET_MERGE only
                                         enabled, lookahead and time_gap are read (the box is [box_in, merge_x)) */
  /* ---- open networks (FS_NET_MERGE); ignored otherwise ---- */
  const fs_inflow* inflows;           /* [num_inflows] in InFlows.add order */
  const uint8_t* init_alive;          /* [R,N] 1: the slot holds an initial vehicle (init_pos / init_vel / init_lane =
                                         its coordinate, speed and ROUTE) */
  double route_start[2];              /* coordinate of the first metre of route 0 / 1 */
  double merge_x;                     /* coordinate where the two routes join (start of edge 'center') */
  double box_in;                      /* coordinate of the junction entry lines (merge_x - internal length) */
  double end_x;                       /* vehicles whose front reaches end_x have arrived */
  double net_length;                  /* k.network.length(): the normaliser of MergePOEnv.get_state */
  int32_t ma_apply_actions;           /* FS_ENV_MERGE_MA: 0 = actions are never applied, as this fork ships
                                         (multiagent/merge.py:92-96); 1 = column rl_index commands the slot, NaN = none */
  int32_t num_obs_cells;              /* FS_ENV_BOTTLENECK_DV: observed lane-segments (<= 64; <= 128 with num_vehicles > 64),
                                         obs_dim = 4 * cells + 1 */
  /* ---- FS_NET_BOTTLENECK ---- */
  double merge1_x, merge2_x;          /* coordinates where lanes (2q, 2q+1) join / where the two resulting lanes join */
  double zipper_distance;             /* a vehicle this close to a join follows the nearest vehicle of either joining lane */
  double speed_limit;                 /* edge speed limit: desired speed = min(vehicle maxSpeed, speed_limit); 0 = none */
  double outflow_norm;                /* 2000 * scaling: normaliser of the outflow reward */
  const fs_cell* obs_cells;           /* [num_obs_cells] in observation order (edge, segment, lane) */
  const fs_cell* act_cells;           /* [num_rl] controlled lane-segments, action column order */
  int32_t obs_outflow_window;         /* int(20 * sim_step / sim_step): sub-steps of the observed outflow */
  int32_t reward_outflow_window;      /* int(10 * sim_step / sim_step) */
  int32_t track_followers;            /* open networks: 1 = keep the sticky follower entries (FS_FIELD_FOLLOWER, used by the
                                         merge observations and BCM); 0 = skip them (the bottleneck envs never read them) */
  int32_t num_paths;                  /* FS_NET_BOTTLENECK: entry lanes = 4 * scaling: 4 (0 = 4) or 8 (8 -> 4 -> 2 lanes;
                                         needs num_vehicles > 64, i.e. the workgroup-per-replica kernel) */
  /* ---- simplified lane changing on FS_NET_BOTTLENECK (docs/HISTORY.md M11) ---- */
  int32_t lane_change_cooldown_steps; /* sub-steps a vehicle keeps its lane after a change */
  int32_t reserved6;
  double lane_change_min_gain;        /* leader-gap gain [m] a lane change must bring */
  /* ---- observation order of AccelEnv-style heads on single-lane closed loops ---- */
  int32_t sort_vehicles;              /* env_params 'sort_vehicles' (accel.py:101-169): observation entries and RL action
                                         columns follow the absolute position recorded at the last additional_command */
  int32_t noise_exact;                /* 0: the acceleration noise's Box-Muller uses the hardware's log2 / cos (fast; agrees with
                                         numpy to a few ulp of the draw); 1: log and cos are fixed sequences of float32
                                         multiplications, additions and one division (flowsim_kernels.h bm_ln_exact /
                                         bm_cos_exact; oracle/refsim.py exact_ln_f32 / exact_cos_turns_f32) -- the float32
                                         kernels then reproduce the numpy oracle's noisy runs bit for bit
                                         (SumoParams(noise_math='exact'); base_controller.py:109-110) */
  const int32_t* obs_perm;            /* [N] place of slot i's vehicle in get_ids() when InitialConfig.shuffle assigned the
                                         start positions in shuffled id order (envs/base.py:268-292); NULL = identity */
  /* ---- sharding ---- */
  int64_t replica_offset;             /* global index of this handle's replica 0 (one handle per GPU holds a contiguous
                                         block of the job's replicas): the Philox streams (acceleration noise, random
                                         entry lanes) are keyed by offset + local index, so a replica's trajectory does
                                         not depend on how the job is sharded */
} fs_config;

typedef struct fs_sim* fs_handle;

/* ---- life cycle ----------------------------------------------------------
 * fs_create replaces Env.__init__ -> Kernel / generate_network /
 * start_simulation / setup_initial_state (flow/envs/base.py:102-229,
 * 268-292; flow/core/kernel/simulation/traci.py:70-174).
 * fs_destroy replaces Env.terminate (flow/envs/base.py:680-703). */
int fs_create(const fs_config* cfg, fs_handle* out);
void fs_destroy(fs_handle h);
const char* fs_last_error(void);
int fs_abi_version(void);

/* observation / action width of the configured env (observation_space.shape[0], action_space.shape[0]) */
int fs_obs_dim(fs_handle h);
int fs_action_dim(fs_handle h);

/* Enqueue all later work of this handle on `hip_stream` (a hipStream_t; NULL is
 * HIP's default stream).  fs_use_own_stream goes back to the non-blocking stream
 * the handle created for itself. */
int fs_set_stream(fs_handle h, void* hip_stream);
int fs_use_own_stream(fs_handle h);
/* Block until the handle's stream is idle. */
int fs_sync(fs_handle h);

/* ---- reset ---------------------------------------------------------------
 * Env.reset (flow/envs/base.py:414-560): re-place the vehicles of the
 * replicas selected by mask (uint8[R], NULL = all) at their initial state,
 * zero their time counters, then run warmup_steps steps with no RL action.
 * obs_out (real32[R,obs_dim], may be NULL) receives the observation of every
 * replica after the call. */
int fs_reset(fs_handle h, const uint8_t* mask, float* obs_out);
int fs_reset_dev(fs_handle h, const uint8_t* mask_dev, float* obs_dev);

/* ---- step ----------------------------------------------------------------
 * Env.step (flow/envs/base.py:294-412) for all R replicas: sims_per_step
 * sub-steps of {controllers -> fail-safes -> apply_acceleration
 * (vehicle/traci.py:952-963) -> simulation_step (simulation/traci.py:54-56)
 * -> vehicle update (vehicle/traci.py:119-259) -> check_collision}, then
 * get_state / compute_reward / done.
 *   actions  real32[R,A] or NULL (reference: rl_actions=None); A = num_rl, or 2*num_rl for
 *            FS_ENV_LANE_CHANGE_ACCEL (fs_action_dim)
 *   obs      real32[R,obs_dim]     rew  real32[R]     done  uint8[R]
 * Observations are float32 as in the reference's Box(dtype=np.float32). */
int fs_step(fs_handle h, const float* actions, float* obs, float* rew, uint8_t* done);
int fs_step_dev(fs_handle h, const float* actions_dev, float* obs_dev, float* rew_dev, uint8_t* done_dev);

/* K consecutive Env.step calls in one launch (state stays in registers).
 *   actions_dev  real32[K,R,num_rl] (action_stride_steps = R*num_rl),
 *                or one real32[R,num_rl] reused every step (stride 0), or NULL
 *   obs_dev      real32[K,R,obs_dim] if obs_every_step, else real32[R,obs_dim] (last step)
 *   rew_dev      real32[K,R] / real32[R]      done_dev uint8[K,R] / uint8[R]
 *   done: 0 = the episode goes on; non-zero = done, bit 0: the horizon was reached (time_counter >= sims_per_step *
 *   (warmup_steps + horizon)), bit 1: a collision ended it (envs/base.py:381-400)
 * A replica that is done keeps stepping (the caller resets it, as
 * Experiment.run / RLlib do after `done`, flow/core/experiment.py:144-161). */
int fs_rollout_dev(fs_handle h, int num_steps, const float* actions_dev, size_t action_stride_steps,
                   float* obs_dev, float* rew_dev, uint8_t* done_dev, int obs_every_step);

/* ---- state access --------------------------------------------------------
 * dst/src are host buffers of `bytes` bytes holding the field in the
 * handle's precision (float or double) or int32 for FS_FIELD_TIME. */
int fs_get_state(fs_handle h, int field, void* dst, size_t bytes);
int fs_set_state(fs_handle h, int field, const void* src, size_t bytes);

/* k.vehicle.add(veh_id, type_id, edge, pos, lane, speed) (flow/core/kernel/vehicle/traci.py:1089-1122 -> TraCI vehicle.add)
 * for a vehicle that HAS a slot and is not in the network: BottleneckAccelEnv.additional_command re-inserts an RL vehicle
 * that left (flow/envs/bottleneck.py:733-757, add_rl_if_exit).  Open networks only.  The vehicle of `slot` (which must be
 * empty: FS_ERR_INVALID otherwise) is placed at route coordinate `x` [m] on entry lane / route `route` with `speed`
 * [m/s], joins the END of the id list (FS_FIELD_SEQ), with its type's maxSpeed and no lane-change history; the
 * neighbour fields (FS_FIELD_LEADER / HEADWAY) are refreshed.  Whether the vehicle fits is the caller's test (the
 * reference lets TraCI raise and ignores it).  Host call between launches: synchronises the handle's stream. */
int fs_add_vehicle(fs_handle h, int replica, int slot, int route, double x, double speed);

/* Family of the step kernel the handle's last fs_step / fs_rollout / fs_policy_rollout launch chose ("k_rollout_pair" with
 * "+speed_mode" and / or "+noise", "k_rollout_idm", "k_ring_pair<Accel | PO | POMA | AccelMA>", "k_rollout_loop",
 * "k_rollout_loop<FULL>", "k_rollout_loop<AccelMA>", "k_rollout_loop<FULL,AccelMA>", "k_ring_policy", "k_loop_policy",
 * "k_steps<FAST>", "k_steps<CSET>", "k_steps", "k_steps_ml", "k_steps_open" (also "<mixed>"), "k_steps_wide",
 * "k_merge_queue", "k_drop_queue", "k_obs_mixed"; "" before the first launch).  Diagnostics for tests and bench.py: which
 * configuration class a workload landed in (no reference counterpart).  The string is static.
 * The queue-order kernel of the lane-drop network (k_drop_queue) reports a replica that outgrew its layout (more than
 * 64 vehicles on one entry path, more than 8 arrivals in one sub-step) through fs_sync /
 * fs_get_state: FS_ERR_UNSUPPORTED with fs_last_error naming it; FLOWSIM_NO_QUEUE=1 in the environment of fs_create
 * keeps a handle on the slot-order kernels. */
const char* fs_last_kernel(fs_handle h);

/* ---- policy in the loop ------------------------------------------------------
 * What examples/train.py:110-212 runs per rollout worker -- policy forward pass, Env.step, reset of a finished episode,
 * one Python call and N socket round trips per step -- as ONE launch per fragment of K steps.  Built for the
 * reference's RL ring experiments (examples/exp_configs/rl/singleagent/singleagent_ring.py: IDM vehicles + ONE RL
 * vehicle, WaveAttenuationPOEnv: observation 3, action 1), for its figure-eight experiment
 * (singleagent_figure_eight.py: FS_NET_FIGURE_EIGHT, 13 IDM + ONE RL vehicle, AccelEnv: observation 2 N = 28, or the
 * WaveAttenuationPOEnv head: observation 3) and its default model class: a fully connected network of 1..3
 * hidden layers of 32 tanh units (examples/train.py:152 fcnet_hiddens [32, 32, 32]) with a diagonal Gaussian head.
 *   weights_dev  float32, device: per layer W [out][in] row-major then b [out]; the last layer has 2 outputs (mean,
 *                log std: RLlib's DiagGaussian) or, with log_std_dev != NULL, 1 output and a free log std [1]
 *   seed         keys the action sampling streams (Philox, per global replica, continued from fragment to fragment)
 * The arithmetic (fma order, hardware exp2 / rcp) is defined by flow_amd/csrc/flowsim_policy.h; fs_policy_act_dev is
 * the SAME evaluation as a call of its own, so fs_policy_rollout_dev equals K x (fs_policy_act_dev, fs_step_dev
 * [, fs_reset_dev(done)]) bit for bit.  Other environments / models: FS_ERR_UNSUPPORTED (capture K single steps
 * around any policy instead: flow_amd.envs.VecFlowEnv.capture). */
typedef struct fs_policy {
  uint32_t struct_size;               /* sizeof(fs_policy) */
  int32_t obs_dim;                    /* must equal fs_obs_dim (3) */
  int32_t num_hidden;                 /* 1..3 hidden layers ... */
  int32_t hidden_width;               /* ... of 32 units each */
  int32_t activation;                 /* 0 = tanh */
  const float* weights_dev;
  const float* log_std_dev;           /* NULL: the network's second output is the log std */
  uint64_t seed;
} fs_policy;

/* actions [R] and log-probabilities [R] for the observations obs_dev [R, obs_dim]; advances the sampling streams */
int fs_policy_act_dev(fs_handle h, const fs_policy* pol, const float* obs_dev, float* act_dev, float* logp_dev);
/* K x (policy -> action -> Env.step), with reset_done != 0 followed by Env.reset of the replicas whose episode ended
 * (placement, FS_FIELD_INIT_RING_LENGTH, warm-up steps).  obs_dev [K+1, R, obs_dim]: obs[0] = observation of the state
 * the fragment starts from (written by the call), obs[k+1] = observation after step k (after the reset, if one
 * happened); act_dev [K, R], logp_dev [K, R], rew_dev [K, R], done_dev [K, R] (flags as in fs_rollout_dev). */
int fs_policy_rollout_dev(fs_handle h, const fs_policy* pol, int num_steps, int reset_done, float* obs_dev,
                          float* act_dev, float* logp_dev, float* rew_dev, uint8_t* done_dev);

/* Append the CURRENT state of one replica to a CSV file (header `time,id,x,speed,lane_number` when the file is new;
 * one row per vehicle, id = slot index, free slots of open networks skipped): the trajectory ("emission") dump of
 * flow/core/kernel/simulation/traci.py:95-101 (SUMO's --emission-output) for callers without the Python layer;
 * flow_amd.core.util.TrajectoryRecorder writes the reference's full column set (flow/core/util.py:57-99). */
int fs_dump_trajectory(fs_handle h, int replica, const char* csv_path);

#ifdef __cplusplus
}
#endif
#endif /* FLOWSIM_H */
