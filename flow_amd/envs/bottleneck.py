"""Bottleneck environments (flow/envs/bottleneck.py) on the lane-drop network.

Built: ``BottleneckEnv`` (the base: observation [1], outflow reward) and ``BottleneckDesiredVelocityEnv`` (variable
speed limits per lane-segment for the RL vehicles) with the toll booth and the ramp meter switched OFF, as every
shipped bottleneck experiment runs them (examples/exp_configs/rl/singleagent/singleagent_bottleneck.py:24-25).
Observation, maxSpeed updates and reward are computed in the HIP step kernel (heads FS_ENV_BOTTLENECK /
FS_ENV_BOTTLENECK_DV).  ``BottleneckAccelEnv`` steps on the device too (per-vehicle RL accelerations); its
observation, lane-change commands and the re-insertion of RL vehicles are host code over the device state, as they are
Python around the simulator in the reference.  Not built: the toll-booth / ALINEA ramp-meter logic (traffic lights)
and the ``evaluate`` reward; they raise NotImplementedError at construction."""
from copy import deepcopy

import numpy as np

from flow_amd import _lib as L
from flow_amd.envs.base import Env
from flow_amd.utils.spaces import Box

MAX_LANES = 4  # base number of largest number of lanes in the network
EDGE_LIST = ["1", "2", "3", "4", "5"]  # Edge 1 is before the toll booth
NUM_TOLL_LANES = MAX_LANES
MEAN_NUM_SECONDS_WAIT_AT_TOLL = 15  # Average waiting time at toll
BOTTLE_NECK_LEN = 280  # Length of bottleneck
NUM_VEHICLE_NORM = 20

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration for autonomous vehicles, in m/s^2
    "max_accel": 3,
    # maximum deceleration for autonomous vehicles, in m/s^2
    "max_decel": 3,
    # lane change duration for autonomous vehicles, in s. Autonomous vehicles
    # reject new lane changing commands for this duration after successfully
    # changing lanes.
    "lane_change_duration": 5,
    # whether the toll booth should be active
    "disable_tb": True,
    # whether the ramp meter is active
    "disable_ramp_metering": True,
}

# Keys for RL experiments
ADDITIONAL_RL_ENV_PARAMS = {
    # velocity to use in reward functions
    "target_velocity": 30,
    # if an RL vehicle exits, place it back at the front
    "add_rl_if_exit": True,
}

# Keys for VSL style experiments
ADDITIONAL_VSL_ENV_PARAMS = {
    # number of controlled regions for velocity bottleneck controller
    "controlled_segments": [("1", 1, True), ("2", 1, True), ("3", 1, True), ("4", 1, True), ("5", 1, True)],
    # whether lanes in a segment have the same action or not
    "symmetric": False,
    # which edges are observed
    "observed_segments": [("1", 1), ("2", 1), ("3", 1), ("4", 1), ("5", 1)],
    # whether the inflow should be reset on each rollout
    "reset_inflow": False,
    # the range of inflows to reset on
    "inflow_range": [1000, 2000]
}


class BottleneckEnv(Env):
    """flow/envs/bottleneck.py:83-483 with ``disable_tb`` and ``disable_ramp_metering`` (the shipped setting)."""

    FS_ENV = L.FS_ENV_BOTTLENECK

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        add = env_params.additional_params
        if not add["disable_tb"] or not add["disable_ramp_metering"]:
            raise NotImplementedError("the toll booth / ramp meter (traffic lights, ALINEA) are not built")
        net = scenario if scenario is not None else network
        self.scaling = net.net_params.additional_params.get("scaling", 1)
        self.edge_dict = dict()
        # drawn (and never used while the toll booth is off) by the reference's constructor (:203-205); kept so that
        # numpy's global stream stands where it stands there when reset() draws a new inflow rate
        self.toll_wait_time = np.abs(np.random.normal(MEAN_NUM_SECONDS_WAIT_AT_TOLL / sim_params.sim_step,
                                                      4 / sim_params.sim_step, NUM_TOLL_LANES * self.scaling))
        self.smoothed_num = np.zeros(10)          # vehicles on edge '4' over the last 10 steps (:266-268)
        self.outflow_index = 0
        super().__init__(env_params, sim_params, network, simulator, scenario)

    def _fs_cells(self, tables):
        """(observation cells, action cells) for the kernel: none for the base environment."""
        return [], []

    @property
    def action_space(self):
        return Box(low=-float("inf"), high=float("inf"), shape=(1,), dtype=np.float32)

    @property
    def observation_space(self):
        return Box(low=-float("inf"), high=float("inf"), shape=(1,), dtype=np.float32)

    def _action_vector(self):
        return None

    def additional_command(self):
        """bottleneck.py:234-268: the per-edge / per-lane vehicle lists and the smoothed count on edge 4."""
        super().additional_command()
        empty_edge = [[] for _ in range(MAX_LANES * self.scaling)]
        self.edge_dict = {k: deepcopy(empty_edge) for k in EDGE_LIST}
        veh = self.k.vehicle
        for veh_id in veh.get_ids():
            edge = veh.get_edge(veh_id)
            if edge not in self.edge_dict:
                self.edge_dict[edge] = deepcopy(empty_edge)
            self.edge_dict[edge][veh.get_lane(veh_id)].append((veh_id, veh.get_position(veh_id)))
        self.smoothed_num[self.outflow_index] = len(veh.get_ids_by_edge('4'))
        self.outflow_index = (self.outflow_index + 1) % self.smoothed_num.shape[0]

    def get_bottleneck_density(self, lanes=None):
        """bottleneck.py:439-454: vehicles on edges 3 and 4 per metre of bottleneck."""
        bottleneck_ids = self.k.vehicle.get_ids_by_edge(['3', '4'])
        if lanes:
            veh_ids = [v for v in bottleneck_ids
                       if str(self.k.vehicle.get_edge(v)) + "_" + str(self.k.vehicle.get_lane(v)) in lanes]
        else:
            veh_ids = bottleneck_ids
        return len(veh_ids) / BOTTLE_NECK_LEN

    def compute_reward(self, rl_actions, **kwargs):
        """Outflow rate over the last ten steps normalised to a maximum of 1 (:474-478), from the kernel."""
        return self._last_reward

    def get_state(self):
        return np.asarray([1])


class BottleneckAccelEnv(BottleneckEnv):
    """flow/envs/bottleneck.py:486-757: RL vehicles that accelerate and change lanes on the lane-drop network.

    The step loop runs on the device (head FS_ENV_BOTTLENECK with one acceleration column per RL slot, NaN = no command);
    what the reference does in Python around the simulator step is host code here too, over the device state:

    * ``_apply_rl_actions`` (:662-690): actions pair up with the RL vehicles sorted by position; directions are rounded,
      blocked while ``time_counter <= lane_change_duration + get_last_lc`` (this fork's ``get_last_lc`` returns the
      headway, vehicle/traci.py:604-614), clipped to the lanes of the vehicle's edge (vehicle/traci.py:965-997);
    * the lane change itself is SUMO's: restated as in the multi-lane ring (ML3) -- executed with the step's move, refused
      when ``lane_change_mode != 0`` and the vehicle would overlap one on the target lane; the vehicle continues on the
      lowest entry lane of its new lane (M11).  Applied through ``fs_set_state(FS_FIELD_ROUTE)`` after the launch;
    * ``additional_command`` (:733-757): an RL vehicle that left is handed back to the simulator (``fs_add_vehicle``:
      edge "1", position 0, lane ``index % MAX_LANES * scaling``, speed "max"); as with TraCI's ``vehicle.add`` it enters
      when there is room (the insertion test M3 of the inflows), at the end of a simulation step;
    * ``get_state`` (:539-640) and ``compute_reward`` (:642-649) literally, over ``k.vehicle.get_lane_leaders`` /
      ``followers`` / ``headways`` / ``tailways`` (vehicle/traci.py:776-950; the lane connections across the drops are
      flow/networks/bottleneck.py:179-201, the internal lanes netconvert adds are one per connection -- stated in
      ``NetworkKernel.next_edge`` / ``prev_edge``, unpinned).
    SUMO-side statements (lane-change execution, insertion) are unpinned like every M-rule.
    The reference's own test of the class has no RL vehicles (tests/fast_tests/test_environments.py:813-878): then the
    observation is the per-edge block only and the action space is empty."""

    HOST_HEADS = True
    APPLY_ENUMERATE_QUIRK = False            # the kernel's action columns are the RL slots (spec: ma_apply_actions)
    PER_VEHICLE_ACTIONS = True

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_RL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        net = network if network is not None else scenario
        self.num_rl = int(net.vehicles.num_rl_vehicles)     # the spaces are read while the simulator is being set up
        if self.num_rl > 0 and env_params.sims_per_step != 1:
            raise NotImplementedError("BottleneckAccelEnv with RL vehicles is built for sims_per_step = 1")
        self._readd, self._lc_targets = [], {}
        super().__init__(env_params, sim_params, network, simulator, scenario)
        self.add_rl_if_exit = env_params.get_additional_param("add_rl_if_exit")
        self.num_rl = deepcopy(self.initial_vehicles.num_rl_vehicles)
        self.rl_id_list = deepcopy(self.initial_vehicles.get_rl_ids())
        self.max_speed = self.k.network.max_speed()

    @property
    def observation_space(self):
        num_edges = len(self.k.network.get_edge_list())
        num_obs = 2 * num_edges + 4 * MAX_LANES * self.scaling * self.num_rl + 4 * self.num_rl
        return Box(low=0, high=1, shape=(num_obs, ), dtype=np.float32)

    @property
    def action_space(self):
        max_decel = self.env_params.additional_params["max_decel"]
        max_accel = self.env_params.additional_params["max_accel"]
        lb = [-abs(max_decel), -1] * self.num_rl
        ub = [max_accel, 1] * self.num_rl
        return Box(np.array(lb, dtype=np.float32), np.array(ub, dtype=np.float32), dtype=np.float32)

    def get_state(self):
        """:539-640, statement by statement."""
        veh, net = self.k.vehicle, self.k.network
        headway_scale = 1000
        rl_ids = veh.get_rl_ids()
        rl_obs = np.empty(0)
        id_counter = 0
        for veh_id in rl_ids:
            rl_id_num = self.rl_id_list.index(veh_id)
            if rl_id_num != id_counter:                      # a vehicle before this one is missing: pad its place
                rl_obs = np.concatenate((rl_obs, np.zeros(4 * (rl_id_num - id_counter))))
                id_counter = rl_id_num + 1
            else:
                id_counter += 1
            edge_num = veh.get_edge(veh_id)
            if edge_num is None or edge_num == '' or edge_num[0] == ':':
                edge_num = -1
            else:
                edge_num = int(edge_num) / 6
            rl_obs = np.concatenate((rl_obs, [veh.get_x_by_id(veh_id) / 1000, veh.get_speed(veh_id) / self.max_speed,
                                              veh.get_lane(veh_id) / MAX_LANES, edge_num]))
        diff = self.num_rl - int(rl_obs.shape[0] / 4)
        if diff > 0:
            rl_obs = np.concatenate((rl_obs, np.zeros(4 * diff)))

        relative_obs = np.empty(0)
        id_counter = 0
        for veh_id in rl_ids:
            rl_id_num = self.rl_id_list.index(veh_id)
            if rl_id_num != id_counter:
                relative_obs = np.concatenate((relative_obs,
                                               np.zeros(4 * MAX_LANES * self.scaling * (rl_id_num - id_counter))))
                id_counter = rl_id_num + 1
            else:
                id_counter += 1
            num_lanes = MAX_LANES * self.scaling
            headway = np.asarray([1000] * num_lanes) / headway_scale
            tailway = np.asarray([1000] * num_lanes) / headway_scale
            vel_in_front = np.asarray([0] * num_lanes) / self.max_speed
            vel_behind = np.asarray([0] * num_lanes) / self.max_speed
            table = veh.lane_neighbour_table(veh_id)         # [(leader, headway, follower, tailway)] per lane
            headway[0:len(table)] = np.asarray([t[1] for t in table]) / headway_scale
            tailway[0:len(table)] = np.asarray([t[3] for t in table]) / headway_scale
            for i, t in enumerate(table):
                if t[0] != '':
                    vel_in_front[i] = veh.get_speed(t[0]) / self.max_speed
            for i, t in enumerate(table):                    # (:608: the test is on the LIST, so it always holds -- an
                vel_behind[i] = veh.get_speed(t[2]) / self.max_speed     # empty lane reads get_speed('') = -1001)
            relative_obs = np.concatenate((relative_obs, headway, tailway, vel_in_front, vel_behind))
        diff = self.num_rl - int(relative_obs.shape[0] / (4 * MAX_LANES))         # (:616-619: without the scaling)
        if diff > 0:
            relative_obs = np.concatenate((relative_obs, np.zeros(4 * MAX_LANES * diff)))

        ids = veh.get_ids()
        on_edge = np.asarray(veh.get_edge(ids), dtype=object)
        speed = veh.speeds(ids)
        edge_obs = []
        for edge in net.get_edge_list():
            here = on_edge == edge
            n = int(here.sum())
            edge_obs += [(sum(speed[here].tolist()) / n) / self.max_speed, n / net.edge_length(edge)] if n else [0, 0]
        return np.concatenate((rl_obs, relative_obs, edge_obs))

    def compute_reward(self, rl_actions, **kwargs):
        """:642-649."""
        from flow_amd.core import rewards
        num_rl = self.k.vehicle.num_rl_vehicles
        acts = np.zeros(0) if rl_actions is None else np.asarray(rl_actions, dtype=np.float64).reshape(-1)
        lane_change_acts = np.abs(np.round(acts[1::2])[:num_rl])
        return rewards.desired_velocity(self) + rewards.rl_forward_progress(self, gain=0.1) - \
            rewards.boolean_action_penalty(lane_change_acts, gain=1.0)

    def _apply_rl_actions(self, actions):
        """:662-690."""
        veh = self.k.vehicle
        num_rl = veh.num_rl_vehicles
        if num_rl == 0:
            return
        actions = np.asarray(actions, dtype=np.float64).reshape(-1)
        acceleration = actions[::2][:num_rl]
        direction = np.round(actions[1::2])[:num_rl]
        sorted_rl_ids = sorted(veh.get_rl_ids(), key=veh.get_x_by_id)
        non_lane_changing_veh = [
            self.time_counter <= self.env_params.additional_params['lane_change_duration'] + veh.get_last_lc(veh_id)
            for veh_id in sorted_rl_ids]
        direction[non_lane_changing_veh] = np.array([0] * sum(non_lane_changing_veh))
        veh.apply_acceleration(sorted_rl_ids, acc=acceleration)
        veh.apply_lane_change(sorted_rl_ids, direction=[int(d) for d in direction])
        # vehicle/traci.py:978-990: the target lane of a commanded change, from the lane and edge of NOW
        net = self.k.network
        for veh_id, d in zip(sorted_rl_ids, direction):
            lane = veh.get_lane(veh_id)
            target = min(max(lane + int(d), 0), net.num_lanes(veh.get_edge(veh_id)) - 1)
            if target != lane:
                self._lc_targets[veh_id] = int(target)

    def _action_vector(self):
        """[1, RL slots] accelerations by slot column (NaN: the slot's vehicle got no command this step)."""
        pend = self.k.vehicle._pending
        if not pend or self.sim.act_dim == 0:
            return None
        row = np.full((1, self.sim.act_dim), np.nan, dtype=np.float32)
        for veh_id, a in pend.items():
            slot = self.k.vehicle._slot.get(veh_id)
            if slot is not None:
                row[0, self._spec["vehicles"][slot]["rl_index"]] = a
        return row

    def additional_command(self):
        """:733-757."""
        super().additional_command()
        num_rl = self.k.vehicle.num_rl_vehicles
        if num_rl != len(self.rl_id_list) and self.add_rl_if_exit:
            here = set(self.k.vehicle.get_rl_ids())
            for rl_id in self.rl_id_list:                    # (the reference walks a set difference: any order)
                if rl_id not in here and rl_id not in self._readd:
                    self._readd.append(rl_id)                # k.vehicle.add: SUMO inserts it when there is room

    def _after_sim_step(self):
        if not self._lc_targets and not self._readd:
            return
        sim, spec = self.sim, self._spec
        x = sim.get_state(L.FS_FIELD_POS).astype(np.float64)
        v = sim.get_state(L.FS_FIELD_VEL).astype(np.float64)
        route = sim.get_state(L.FS_FIELD_ROUTE).copy()
        m1, m2, zip_d = float(spec["merge1_x"]), float(spec["merge2_x"]), float(spec.get("zipper_distance", 50.0))
        length = np.array([s_["length"] for s_ in spec["vehicles"]], dtype=np.float64)

        def shift(xx):
            return (np.asarray(xx) >= m1).astype(np.int64) + (np.asarray(xx) >= m2).astype(np.int64)

        # ---- commanded lane changes (ML3's statement on the lane-drop network), with this step's move
        changed = False
        for veh_id, target in self._lc_targets.items():
            i = self.k.vehicle._slot.get(veh_id)
            if i is None or route[0, i] < 0:                 # left the network in this step
                continue
            g = int(shift(x[0, i]))
            if (int(route[0, i]) >> g) == target or target >= (int(spec["num_paths"]) >> g):
                continue                                     # (a join passed in between has changed the lanes)
            if int(spec["vehicles"][i].get("lane_change_mode", 0)) != 0:
                alive = route[0] >= 0
                alive[i] = False
                on_target = alive & ((np.maximum(route[0], 0) >> g) == target)
                d = x[0] - x[0, i]
                if (on_target & (((d >= 0) & (d < length)) | ((d < 0) & (-d < length[i])))).any():
                    continue
            route[0, i] = target << g
            changed = True
        self._lc_targets = {}
        if changed:
            sim.set_state(L.FS_FIELD_ROUTE, route)
        # ---- re-insertion of the RL vehicles that left (M3's test at position 0 of edge "1", departSpeed "max")
        waiting = []
        for rl_id in self._readd:
            slot = self._spec["init_slot"][rl_id]
            if route[0, slot] >= 0:
                continue
            lane = self.rl_id_list.index(rl_id) % MAX_LANES * self.scaling
            vs = spec["vehicles"][slot]
            v_dep = min(float(vs.get("sumo_max_speed", 30.0)), float(spec["speed_limit"]))
            alive = route[0] >= 0
            sh = shift(np.maximum(x[0], 0.0 + zip_d))
            cand = alive & ((np.maximum(route[0], 0) >> sh) == (lane >> sh))
            ok = True
            if cand.any():
                j = int(np.flatnonzero(cand)[np.argmin(x[0][cand])])
                gap = (x[0, j] - length[j]) - 0.0
                need = float(vs.get("sumo_min_gap", 2.5)) + max(
                    0.0, v_dep * float(vs.get("sumo_tau", 1.0)) + v_dep * (v_dep - v[0, j]) /
                    (2.0 * np.sqrt(float(vs["max_accel"]) * float(vs["max_decel"]))))
                ok = gap >= need
            if ok:
                sim.add_vehicle(slot, lane, 0.0, v_dep)
                route[0, slot], x[0, slot], v[0, slot] = lane, 0.0, v_dep
            else:
                waiting.append(rl_id)
        self._readd = waiting

    def reset(self):
        self._readd, self._lc_targets = [], {}
        return super().reset()


class BottleneckDesiredVelocityEnv(BottleneckEnv):
    """flow/envs/bottleneck.py:760-1085.

    States: for every observed lane-segment the number of human vehicles / 20, the number of RL vehicles / 20, their
    mean speeds / 50, and the outflow of the last 20 steps / 2000.  Actions: one value per controlled lane-segment,
    added to the maxSpeed of the RL vehicles inside it, clipped to [0.01, 23]; RL vehicles elsewhere get 23.
    Reward: outflow of the last 10 steps / (2000 * scaling)."""

    FS_ENV = L.FS_ENV_BOTTLENECK_DV

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_VSL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        add = env_params.additional_params
        default = [(str(i), 1, True) for i in range(1, 6)]
        self.segments = add.get("controlled_segments", default)
        self.symmetric = add.get("symmetric")
        self.obs_segments = add.get("observed_segments", [])
        self.num_obs_segments = [segment[1] for segment in self.obs_segments]
        self.is_controlled = [segment[2] for segment in self.segments]
        self.num_controlled_segments = [segment[1] for segment in self.segments if segment[2]]
        self.total_segments = int(np.sum([segment[1] for segment in self.segments]))
        self.total_controlled_segments = int(np.sum([segment[1] for segment in self.segments if segment[2]]))
        self.controlled_edges = [segment[0] for segment in self.segments if segment[2]]
        self._dv_actions = None
        super().__init__(env_params, sim_params, network, simulator, scenario)
        self.slices = {edge: np.linspace(0, self.k.network.edge_length(edge), n + 1) for edge, n, _ in self.segments}
        self.obs_slices = {edge: np.linspace(0, self.k.network.edge_length(edge), n + 1)
                           for edge, n in self.obs_segments}
        # offset of every controlled edge in the action vector (bottleneck.py:826-844); symmetric: the offsets advance by
        # ONE per controlled edge whatever its number of segments (:829-831 adds `controlled`), kept as written
        self.action_index, offset, per_lane = {}, 0, 0
        self._sym_map = []                                   # kernel column (edge, segment, lane) -> symmetric action index
        for edge, n, controlled in self.segments:
            if controlled:
                lanes = self.k.network.num_lanes(edge)
                self.action_index[edge] = [offset if self.symmetric else per_lane]
                self._sym_map += [bucket + offset for bucket in range(n) for _ in range(lanes)]
                offset += 1
                per_lane += n * lanes
        self._sym_map = np.asarray(self._sym_map, dtype=np.int64) if self.symmetric else None

    def _fs_cells(self, tables):
        """Lane-segments in the order get_state / the action vector walk them: edge, segment, lane."""
        net = self.k.network
        starts = dict(net._open_starts[0])

        def cells(segments):
            out = []
            for edge, n in segments:
                bounds = np.linspace(0, net.edge_length(edge), n + 1)
                for k in range(n):
                    for lane in range(net.num_lanes(edge)):
                        out.append((starts[edge], float(bounds[k]), float(bounds[k + 1]), lane, k == n - 1))
            return out
        if [e for e, _ in self.obs_segments] != EDGE_LIST[:len(self.obs_segments)]:
            raise NotImplementedError("observed_segments must list the edges '1'..'5' in order (bottleneck.py:884-886 "
                                      "pairs them with EDGE_LIST by position)")
        obs = cells(self.obs_segments)
        act = cells([(e, n) for e, n, c in self.segments if c])
        if len(obs) > 128 or len(act) > 64:
            raise NotImplementedError("more than 128 observed / 64 controlled lane-segments is not built")
        return obs, act

    @property
    def observation_space(self):
        num_obs = 0
        for segment in self.obs_segments:
            num_obs += 4 * segment[1] * self.k.network.num_lanes(segment[0])
        num_obs += 1
        return Box(low=0.0, high=1.0, shape=(num_obs,), dtype=np.float32)

    @property
    def action_space(self):
        action_size = 0
        for segment in self.segments:
            if segment[2]:
                action_size += self.k.network.num_lanes(segment[0]) * segment[1]
        if self.symmetric:                                   # :850-851: one action per segment for all its lanes
            action_size = self.total_controlled_segments
        add_params = self.env_params.additional_params
        return Box(low=-add_params.get("max_decel") * self.sim_step, high=add_params.get("max_accel") * self.sim_step,
                   shape=(int(action_size),), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        """bottleneck.py:926-969: handed to the kernel, which finds each RL vehicle's lane-segment itself.  symmetric
        (:946-949): the action of a lane-segment is ``rl_actions[bucket + action_index[edge]]`` -- gathered here into the
        kernel's one-column-per-lane-segment row."""
        a = np.asarray(rl_actions, dtype=np.float32).reshape(-1)
        self._dv_actions = (a[self._sym_map] if self.symmetric else a).reshape(1, -1)

    def compute_reward(self, rl_actions, **kwargs):
        """:971-981; evaluate: the outflow over the last 500 s at the horizon, 0 before."""
        if self.env_params.evaluate:
            if self.time_counter == self.env_params.horizon:
                return self.k.vehicle.get_outflow_rate(500)
            return 0
        return self._last_reward

    def _action_vector(self):
        a, self._dv_actions = self._dv_actions, None
        return a

    def get_state(self):
        return np.asarray(self._last_obs, dtype=np.float32).copy()

    def reset(self):
        """bottleneck.py:983-1085: with ``reset_inflow`` a new inflow rate is drawn from ``inflow_range`` and the
        network (hence the simulator) is rebuilt; otherwise the generic reset."""
        add_params = self.env_params.additional_params
        if add_params.get("reset_inflow"):
            from flow_amd.core.params import InFlows, NetParams
            inflow_range = add_params.get("inflow_range")
            flow_rate = np.random.uniform(min(inflow_range), max(inflow_range)) * self.scaling
            inflow = InFlows()
            inflow.add(veh_type="followerstopper", edge="1", vehs_per_hour=flow_rate * .1, departLane="random",
                       departSpeed=10)
            inflow.add(veh_type="human", edge="1", vehs_per_hour=flow_rate * .9, departLane="random", departSpeed=10)
            net_params = NetParams(inflows=inflow, additional_params={
                "scaling": self.scaling, "speed_limit": self.net_params.additional_params['speed_limit']})
            self.network = self.network.__class__(name=self.network.orig_name, vehicles=self.network.vehicles,
                                                  net_params=net_params, initial_config=self.initial_config,
                                                  traffic_lights=self.network.traffic_lights)
            self.net_params = net_params
            self.restart_simulation(self.sim_params)
        observation = super().reset()
        self.time_counter = 0
        return observation
