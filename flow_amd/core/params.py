"""Experiment meta-parameters: the same objects, field names and defaults as
flow/core/params.py, so that a reference ``flow_params`` dict builds unchanged.

Only what the GPU step loop consumes carries behaviour; renderer / SUMO-process
options are accepted and stored so existing configs keep working.
"""
import collections
import logging

from flow_amd.controllers.car_following_models import SimCarFollowingController
from flow_amd.controllers.lane_change_controllers import SimLaneChangeController
from flow_amd.controllers.rlcontroller import RLController

# flow/core/params.py:12-20
SPEED_MODES = {"aggressive": 0, "obey_safe_speed": 1, "no_collide": 7, "right_of_way": 25, "all_checks": 31}
LC_MODES = {"aggressive": 0, "no_lat_collide": 512, "strategic": 1621}


class TrafficLightParams:
    """Traffic lights (flow/core/params.py:29-196).  The ring / figure-eight hot path has
    none; the object exists so network constructors keep their signature."""

    def __init__(self, baseline=False):
        self.baseline = baseline
        self._tls = dict()

    def add(self, node_id, **kwargs):
        raise NotImplementedError("traffic lights are outside the GPU hot path (SURVEY.md 8: out of scope)")

    def get_properties(self):
        return self._tls


class SumoCarFollowingParams:
    """flow/core/params.py:785-901."""

    def __init__(self, speed_mode='right_of_way', accel=2.6, decel=4.5, sigma=0.5, tau=1.0, min_gap=2.5,
                 max_speed=30, speed_factor=1.0, speed_dev=0.1, impatience=0.5, car_follow_model="IDM",
                 **kwargs):
        renamed = {"minGap": "min_gap", "maxSpeed": "max_speed", "speedFactor": "speed_factor",
                   "speedDev": "speed_dev", "carFollowModel": "car_follow_model"}
        local = dict(min_gap=min_gap, max_speed=max_speed, speed_factor=speed_factor, speed_dev=speed_dev,
                     car_follow_model=car_follow_model)
        for old, new in renamed.items():
            if old in kwargs:
                logging.warning("%s is deprecated, use %s", old, new)
                local[new] = kwargs[old]
        self.controller_params = {
            "accel": accel, "decel": decel, "sigma": sigma, "tau": tau, "minGap": local["min_gap"],
            "maxSpeed": local["max_speed"], "speedFactor": local["speed_factor"],
            "speedDev": local["speed_dev"], "impatience": impatience,
            "carFollowModel": local["car_follow_model"],
        }
        if isinstance(speed_mode, str) and speed_mode in SPEED_MODES:
            speed_mode = SPEED_MODES[speed_mode]
        elif not isinstance(speed_mode, (int, float)):
            logging.error("Setting speed mode of to default.")
            speed_mode = SPEED_MODES["obey_safe_speed"]
        self.speed_mode = speed_mode


class SumoLaneChangeParams:
    """flow/core/params.py:904-1067."""

    def __init__(self, lane_change_mode="no_lat_collide", model="LC2013", lc_strategic=1.0, lc_cooperative=1.0,
                 lc_speed_gain=1.0, lc_keep_right=1.0, lc_look_ahead_left=2.0, lc_speed_gain_right=1.0,
                 lc_sublane=1.0, lc_pushy=0, lc_pushy_gap=0.6, lc_assertive=1, lc_accel_lat=1.0, **kwargs):
        if model not in ["LC2013", "SL2015"]:
            logging.error("Invalid lane change model! Defaulting to LC2013")
            model = "LC2013"
        self.controller_params = {
            "laneChangeModel": model, "lcStrategic": str(kwargs.get("lcStrategic", lc_strategic)),
            "lcCooperative": str(kwargs.get("lcCooperative", lc_cooperative)),
            "lcSpeedGain": str(kwargs.get("lcSpeedGain", lc_speed_gain)),
            "lcKeepRight": str(kwargs.get("lcKeepRight", lc_keep_right)),
        }
        if model == "SL2015":
            self.controller_params.update({
                "lcLookaheadLeft": str(lc_look_ahead_left), "lcSpeedGainRight": str(lc_speed_gain_right),
                "lcSublane": str(lc_sublane), "lcPushy": str(lc_pushy), "lcPushyGap": str(lc_pushy_gap),
                "lcAssertive": str(lc_assertive), "lcAccelLat": str(lc_accel_lat)})
        if isinstance(lane_change_mode, str) and lane_change_mode in LC_MODES:
            lane_change_mode = LC_MODES[lane_change_mode]
        elif not isinstance(lane_change_mode, (int, float)):
            logging.error("Setting lane change mode to default.")
            lane_change_mode = LC_MODES["no_lat_collide"]
        self.lane_change_mode = lane_change_mode


class VehicleParams:
    """The vehicles of an experiment (flow/core/params.py:199-361): ``add`` appends
    ``num_vehicles`` vehicles named ``{veh_id}_{k}``; order of insertion is slot order."""

    def __init__(self):
        self.ids = []
        self.__vehicles = collections.OrderedDict()
        self.num_vehicles = 0
        self.num_rl_vehicles = 0
        self.num_types = 0
        self.types = []
        self.type_parameters = dict()
        self.minGap = dict()
        self.initial = []

    def add(self, veh_id, length=None, acceleration_controller=(SimCarFollowingController, {}),
            lane_change_controller=(SimLaneChangeController, {}), routing_controller=None, initial_speed=0,
            num_vehicles=0, car_following_params=None, lane_change_params=None, color=None):
        if car_following_params is None:
            car_following_params = SumoCarFollowingParams()
        if lane_change_params is None:
            lane_change_params = SumoLaneChangeParams()
        type_params = {}
        type_params.update(car_following_params.controller_params)
        type_params.update(lane_change_params.controller_params)
        self.type_parameters[veh_id] = {
            "acceleration_controller": acceleration_controller,
            "lane_change_controller": lane_change_controller,
            "routing_controller": routing_controller,
            "initial_speed": initial_speed,
            "car_following_params": car_following_params,
            "lane_change_params": lane_change_params}
        if length:
            type_params['length'] = length
            self.type_parameters[veh_id]['length'] = length
        if color:
            type_params['color'] = color
            self.type_parameters[veh_id]['color'] = color
        self.initial.append({
            "veh_id": veh_id, "acceleration_controller": acceleration_controller,
            "lane_change_controller": lane_change_controller, "routing_controller": routing_controller,
            "initial_speed": initial_speed, "num_vehicles": num_vehicles,
            "car_following_params": car_following_params, "lane_change_params": lane_change_params})
        self.minGap[veh_id] = type_params["minGap"]
        for i in range(num_vehicles):
            v_id = veh_id + '_%d' % i
            self.ids.append(v_id)
            self.__vehicles[v_id] = {"type": veh_id}
            self.num_vehicles += 1
            if acceleration_controller[0] == RLController:
                self.num_rl_vehicles += 1
        self.num_types += 1
        self.types.append({"veh_id": veh_id, "type_params": type_params})

    def get_type(self, veh_id):
        return self.__vehicles[veh_id]["type"]


class SimParams(object):
    """flow/core/params.py:364-422."""

    def __init__(self, sim_step=0.1, render=False, restart_instance=False, emission_path=None,
                 save_render=False, sight_radius=25, show_radius=False, pxpm=2, force_color_update=False):
        self.sim_step = sim_step
        self.render = render
        self.restart_instance = restart_instance
        self.emission_path = emission_path
        self.save_render = save_render
        self.sight_radius = sight_radius
        self.pxpm = pxpm
        self.show_radius = show_radius
        self.force_color_update = force_color_update


class SumoParams(SimParams):
    """flow/core/params.py:510-617.  Extra, GPU-simulator-only keywords (all optional):

    slowdown_ramp   None -> dt/(dt+1e-3): the ramp of TraCI slowDown(v, 1e-3) (docs/HISTORY.md S6); 1.0 = exact
    junction_mode   1 -> vehicles on internal edges get no Flow command (base_controller.py:98-99);
                    None -> 0 on a ring (0.1 m junctions, treated as seamless), 1 on a figure eight
    center_length   length of the ':center_*' internal edges of a figure eight (None -> 9.4, the
                    netconvert value in the reference's fixture)
    crossing_time_gap  right-of-way model of the figure-eight crossing / the merge junction (docs/HISTORY.md S-J, M6):
                    None -> 3.0 s on a figure eight, 1.0 s on a merge
    max_vehicles    open networks: vehicle slots per replica (<= 64; <= 256 on BottleneckNetwork), shared out over the vehicle types
    slot_capacity   open networks: {vehicle type: slots}, overrides the default share-out
    merge_right_of_way  open networks: False switches the junction priority model off
    zipper_distance lane-drop networks: distance before a zipper junction from which a vehicle follows the nearest
                    vehicle of either joining lane (docs/HISTORY.md M8)
    lane_change_cooldown / lane_change_min_gain  lane-drop networks: seconds a vehicle keeps its lane after a change /
                    leader-gap gain [m] a change must bring, for vehicle types whose lane_change_mode lets SUMO change
                    lanes (the simplified model M11, NOT LC2013)
    junction_length length of each internal edge (netconvert output in the reference)
    crash_gap       a replica crashes when a bumper gap falls below this after a move
    noise_math      'hw': the acceleration noise's Box-Muller transform uses the GPU's log2 / cos instructions (default);
                    'exact': fixed float32 operation sequences instead, so that noisy float32 runs equal the numpy oracle
                    bit for bit (a few VALU instructions more per draw)
    precision       'f32' | 'f64' arithmetic and state type of the kernels; 'f16s' (merge network: half state in HBM
                    between launches, float32 integrator -- include/flowsim.h FS_F16S); 'mixed' (float64 state, float32
                    controller: all-IDM single-lane rings, holds 1e-4 of the float64 trajectories at f32 cost)
    """

    def __init__(self, port=None, sim_step=0.1, emission_path=None, lateral_resolution=None, no_step_log=True,
                 render=False, save_render=False, sight_radius=25, show_radius=False, pxpm=2,
                 force_color_update=False, overtake_right=False, seed=None, restart_instance=False,
                 print_warnings=True, start_at_load=True, teleport_time=-1, num_clients=1, color_by_speed=False,
                 use_ballistic=False, slowdown_ramp=None, junction_mode=None, junction_length=0.1, crash_gap=0.0,
                 precision="f32", center_length=None, crossing_time_gap=None, max_vehicles=64, slot_capacity=None,
                 merge_right_of_way=True, zipper_distance=50.0, lane_change_cooldown=5.0, lane_change_min_gain=10.0,
                 noise_math="hw"):
        super(SumoParams, self).__init__(sim_step, render, restart_instance, emission_path, save_render,
                                         sight_radius, show_radius, pxpm, force_color_update)
        self.port = port
        self.lateral_resolution = lateral_resolution
        self.no_step_log = no_step_log
        self.seed = seed
        self.overtake_right = overtake_right
        self.print_warnings = print_warnings
        self.start_at_load = start_at_load
        self.teleport_time = teleport_time
        self.num_clients = num_clients
        self.color_by_speed = color_by_speed
        self.use_ballistic = use_ballistic
        self.slowdown_ramp = slowdown_ramp
        self.junction_mode = junction_mode
        self.junction_length = junction_length
        self.crash_gap = crash_gap
        self.precision = precision
        if noise_math not in ("hw", "exact"):
            raise ValueError("noise_math must be 'hw' or 'exact'")
        self.noise_math = noise_math
        self.center_length = center_length
        self.crossing_time_gap = crossing_time_gap
        self.max_vehicles = max_vehicles
        self.slot_capacity = slot_capacity
        self.merge_right_of_way = merge_right_of_way
        self.zipper_distance = zipper_distance
        self.lane_change_cooldown = lane_change_cooldown
        self.lane_change_min_gain = lane_change_min_gain


class EnvParams:
    """flow/core/params.py:620-671."""

    def __init__(self, additional_params=None, horizon=float('inf'), warmup_steps=0, sims_per_step=1,
                 evaluate=False, clip_actions=True):
        self.additional_params = additional_params if additional_params is not None else {}
        self.horizon = horizon
        self.warmup_steps = warmup_steps
        self.sims_per_step = sims_per_step
        self.evaluate = evaluate
        self.clip_actions = clip_actions

    def get_additional_param(self, key):
        return self.additional_params[key]


class InFlows:
    """flow/core/params.py:1070-1220: ``add`` validates and stores; the open-network simulator reads
    ``vehsPerHour`` / ``period``, ``departSpeed``, ``begin``, ``end``, ``number``."""

    def __init__(self):
        self.__flows = []

    def add(self, edge, veh_type, vehs_per_hour=None, probability=None, period=None, depart_lane="first",
            depart_speed=0, name="flow", begin=1, end=86400, number=None, **kwargs):
        if "vehsPerHour" in kwargs:                                  # deprecated spellings, params.py:1167-1178
            vehs_per_hour = kwargs.pop("vehsPerHour")
        if "departLane" in kwargs:
            depart_lane = kwargs.pop("departLane")
        if "departSpeed" in kwargs:
            depart_speed = kwargs.pop("departSpeed")
        new = {"name": "%s_%d" % (name, len(self.__flows)), "vtype": veh_type, "edge": edge,
               "departLane": depart_lane, "departSpeed": depart_speed, "begin": begin, "end": end}
        new.update(kwargs)
        given = [x is not None for x in (vehs_per_hour, probability, period)]
        if sum(given) != 1:                                         # params.py:1188-1194
            raise ValueError("Exactly one among the three parameters 'vehs_per_hour', 'probability' and "
                             "'period' must be specified in InFlows.add. {} were specified.".format(sum(given)))
        if probability is not None and not (0 <= probability <= 1):
            raise ValueError("Inflow.add called with parameter 'probability' set to {}, but probability should "
                             "be between 0 and 1.".format(probability))
        if begin is not None and begin < 1:                          # params.py:1198-1200
            raise ValueError("Inflow.add called with parameter 'begin' set to {}, but begin should be greater "
                             "or equal than 1 second.".format(begin))
        if number is not None:
            del new["end"]
            new["number"] = number
        if vehs_per_hour is not None:
            new["vehsPerHour"] = vehs_per_hour
        if probability is not None:
            new["probability"] = probability
        if period is not None:
            new["period"] = period
        self.__flows.append(new)

    def sort(self, key):
        self.__flows.sort(key=key)

    def get(self):
        return self.__flows


class NetParams:
    """flow/core/params.py:674-711."""

    def __init__(self, inflows=None, osm_path=None, template=None, additional_params=None):
        self.inflows = inflows or InFlows()
        self.osm_path = osm_path
        self.template = template
        self.additional_params = additional_params or {}


class InitialConfig:
    """flow/core/params.py:714-782."""

    def __init__(self, shuffle=False, spacing="uniform", min_gap=0, perturbation=0.0, x0=0, bunching=0,
                 lanes_distribution=float("inf"), edges_distribution="all", additional_params=None):
        self.shuffle = shuffle
        self.spacing = spacing
        self.min_gap = min_gap
        self.perturbation = perturbation
        self.x0 = x0
        self.bunching = bunching
        self.lanes_distribution = lanes_distribution
        self.edges_distribution = edges_distribution
        self.additional_params = additional_params or dict()
