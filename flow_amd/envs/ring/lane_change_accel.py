"""LaneChangeAccelEnv (flow/envs/ring/lane_change_accel.py:28-154) on the multi-lane step kernel."""
import numpy as np

from flow_amd import _lib as L
from flow_amd.envs.ring.accel import AccelEnv
from flow_amd.utils.spaces import Box

ADDITIONAL_ENV_PARAMS = {
    # maximum acceleration for autonomous vehicles, in m/s^2
    "max_accel": 3,
    # maximum deceleration for autonomous vehicles, in m/s^2
    "max_decel": 3,
    # lane change duration for autonomous vehicles, in s
    "lane_change_duration": 5,
    # desired velocity for all vehicles in the network, in m/s
    "target_velocity": 10,
    # whether vehicles are sorted by position in the observation
    'sort_vehicles': False
}


class LaneChangeAccelEnv(AccelEnv):
    """States: speeds, absolute positions and lane indices of all vehicles; actions: per RL vehicle an
    acceleration and a lane-change direction in {-1, 0, 1}; reward: desired_velocity minus 0.1 per RL lane
    change (lane_change_accel.py:86-98).  ``LAST_LC_QUIRK``: this fork's get_last_lc returns the headway
    (vehicle/traci.py:604-614), which is what rate-limits lane changes here too; set it to False on a
    subclass for the upstream meaning (time of the last lane change)."""

    FS_ENV = L.FS_ENV_LANE_CHANGE_ACCEL
    LAST_LC_QUIRK = True

    def __init__(self, env_params, sim_params, network, simulator='traci'):
        for p in ADDITIONAL_ENV_PARAMS.keys():
            if p not in env_params.additional_params:
                raise KeyError('Environment parameter "{}" not supplied'.format(p))
        super().__init__(env_params, sim_params, network, simulator)

    @property
    def action_space(self):
        max_decel = self.env_params.additional_params["max_decel"]
        max_accel = self.env_params.additional_params["max_accel"]
        lb = [-abs(max_decel), -1] * self.initial_vehicles.num_rl_vehicles
        ub = [max_accel, 1] * self.initial_vehicles.num_rl_vehicles
        return Box(np.array(lb), np.array(ub), dtype=np.float32)

    @property
    def observation_space(self):
        return Box(low=0, high=1, shape=(3 * self.initial_vehicles.num_vehicles, ), dtype=np.float32)

    def _apply_rl_actions(self, actions):
        """lane_change_accel.py:132-154: the rate limit itself is evaluated in the kernel (it reads the
        same time counter and get_last_lc value the reference reads)."""
        actions = np.asarray(actions, dtype=np.float64)
        acceleration = actions[::2]
        direction = actions[1::2]
        sorted_rl_ids = [veh_id for veh_id in self.sorted_ids if veh_id in self.k.vehicle.get_rl_ids()]
        self.k.vehicle.apply_acceleration(sorted_rl_ids, acc=acceleration)
        self.k.vehicle.apply_lane_change(sorted_rl_ids, direction=[int(d) if float(d).is_integer() else d
                                                                   for d in direction])
        if self.env_params.additional_params['sort_vehicles']:
            # pair k commands the k-th RL vehicle in sorted order: the kernel resolves that order itself
            self._sorted_actions = actions.astype(np.float32).reshape(1, -1)


class LaneChangeAccelPOEnv(LaneChangeAccelEnv):
    """POMDP version (flow/envs/ring/lane_change_accel.py:163-262): for every RL vehicle the gap to and the speed of
    its nearest leader and follower in EVERY lane, then the RL vehicles' own speeds.

    A kernel head (``FS_ENV_LANE_CHANGE_ACCEL_PO``, written by ``k_steps_ml``): the per-lane neighbour search of
    flow/core/kernel/vehicle/traci.py:776-867 runs on the device for the RL vehicles, the observation row arrives with
    the step.  What the reference's vector holds, kept as is: gaps in METRES with 1000 for an empty lane (its
    normalisation divides a list it has already copied from, :236-247), neighbour speeds over ``max_speed`` with 0 for an
    empty lane, the ego speed in m/s.  The reference's ``return`` sits inside its loop over the RL vehicles (:262), so
    only the FIRST RL vehicle's block is filled and ONE ego speed appended -- ``RETURN_IN_LOOP_QUIRK``, applied here by
    slicing the kernel's row (False: every RL vehicle, the evident intent and what ``VecFlowEnv`` hands out)."""

    FS_ENV = L.FS_ENV_LANE_CHANGE_ACCEL_PO
    RETURN_IN_LOOP_QUIRK = True

    def __init__(self, env_params, sim_params, network, simulator='traci'):
        super().__init__(env_params, sim_params, network, simulator)
        self.num_lanes = max(self.k.network.num_lanes(edge) for edge in self.k.network.get_edge_list())

    @property
    def observation_space(self):
        n_rl = self.initial_vehicles.num_rl_vehicles
        return Box(low=0, high=1, shape=(4 * n_rl * self.num_lanes + n_rl, ), dtype=np.float32)

    def get_state(self):
        row = np.array(self._last_obs, dtype=np.float64)
        if not self.RETURN_IN_LOOP_QUIRK:
            return row
        n_rl, block = self.k.vehicle.num_rl_vehicles, 4 * self.num_lanes
        out = np.zeros(block * n_rl + 1)
        out[:block] = row[:block]
        out[-1] = row[block * n_rl]
        return out

    @property
    def visible(self):
        """The vehicles that entered the observation (lane_change_accel.py:240-254), for rendering: read back from the
        device state on demand."""
        veh = self.k.vehicle
        shown = veh.get_rl_ids()[:1] if self.RETURN_IN_LOOP_QUIRK else veh.get_rl_ids()
        return [v for rl_id in shown for t in veh.lane_neighbour_table(rl_id) for v in (t[0], t[2]) if v != '']

    def additional_command(self):
        """lane_change_accel.py:257-262: the neighbours that entered the observation are the observed vehicles."""
        for veh_id in self.visible:
            self.k.vehicle.set_observed(veh_id)
