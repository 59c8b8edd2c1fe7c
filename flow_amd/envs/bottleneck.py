"""Bottleneck environments (flow/envs/bottleneck.py) on the lane-drop network.

Built: ``BottleneckEnv`` (the base: observation [1], outflow reward) and ``BottleneckDesiredVelocityEnv`` (variable
speed limits per lane-segment for the RL vehicles) with the toll booth and the ramp meter switched OFF, as every
shipped bottleneck experiment runs them (examples/exp_configs/rl/singleagent/singleagent_bottleneck.py:24-25).
Observation, maxSpeed updates and reward are computed in the HIP step kernel (heads FS_ENV_BOTTLENECK /
FS_ENV_BOTTLENECK_DV).  ``BottleneckAccelEnv`` is built for networks without RL vehicles (what the reference's own
test exercises).  Not built: the toll-booth / ALINEA ramp-meter logic (traffic lights), ``BottleneckAccelEnv`` with RL
vehicles (per-vehicle lane changes) and the ``evaluate`` reward; they raise NotImplementedError at construction."""
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
        if env_params.evaluate:
            raise NotImplementedError("the evaluate reward of the bottleneck envs (outflow over 500 s) is not built")
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
    """flow/envs/bottleneck.py:486-757.

    Built for the case the reference itself tests (tests/fast_tests/test_environments.py:813-878): NO RL vehicles in
    the network.  The observation is then the per-edge block only (mean speed / max speed and vehicles per metre for
    every edge of ``get_edge_list()``, the rendering-only "fake_edge" included: 12 numbers), the action space is empty,
    and the reward is ``rewards.desired_velocity`` (the forward-progress and lane-change terms are sums over the RL
    vehicles).  Both are host heads over the device state (``HOST_HEADS``): masked reductions over the replica's rows.  With RL vehicles the env needs per-vehicle lane-change commands on the lane-drop network and the
    re-insertion of exited RL vehicles (:733-757): not built, raises at construction."""

    HOST_HEADS = True

    def __init__(self, env_params, sim_params, network=None, simulator='traci', scenario=None):
        for p in ADDITIONAL_RL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        net = network if network is not None else scenario
        if net.vehicles.num_rl_vehicles > 0:
            raise NotImplementedError("BottleneckAccelEnv with RL vehicles needs per-vehicle lane-change commands and "
                                      "re-insertion on the lane-drop network: not built")
        self.num_rl = 0                              # the spaces are read while the simulator is being set up
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
        """The per-edge block of :539-640 (with no RL vehicle the rl / relative blocks are empty): for every edge of
        ``get_edge_list()`` the mean speed over ``max_speed`` and the vehicles per metre -- two masked reductions over
        the replica's speed row per edge."""
        veh, net = self.k.vehicle, self.k.network
        ids = veh.get_ids()
        on_edge = np.asarray(veh.get_edge(ids), dtype=object)
        speed = veh.speeds(ids)
        out = []
        for edge in net.get_edge_list():
            here = on_edge == edge
            n = int(here.sum())
            out += [speed[here].sum() / n / self.max_speed, n / net.edge_length(edge)] if n else [0., 0.]
        return np.asarray(out, dtype=np.float64)

    def compute_reward(self, rl_actions, **kwargs):
        """:642-649; no RL vehicles: the lane-change penalty and the forward-progress term are empty sums."""
        from flow_amd.core import rewards
        return rewards.desired_velocity(self) + rewards.rl_forward_progress(self, gain=0.1) - \
            rewards.boolean_action_penalty(np.zeros(0), gain=1.0)

    def _apply_rl_actions(self, actions):
        return None


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
        if self.symmetric:
            raise NotImplementedError("symmetric=True (one action per segment for all lanes) is not built")
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
        # offset of every controlled edge in the action vector (bottleneck.py:826-844)
        self.action_index, offset = {}, 0
        for edge, n, controlled in self.segments:
            if controlled:
                self.action_index[edge] = [offset]
                offset += n * self.k.network.num_lanes(edge)

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
        add_params = self.env_params.additional_params
        return Box(low=-add_params.get("max_decel") * self.sim_step, high=add_params.get("max_accel") * self.sim_step,
                   shape=(int(action_size),), dtype=np.float32)

    def _apply_rl_actions(self, rl_actions):
        """bottleneck.py:926-969: handed to the kernel, which finds each RL vehicle's lane-segment itself."""
        self._dv_actions = np.asarray(rl_actions, dtype=np.float32).reshape(1, -1)

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
