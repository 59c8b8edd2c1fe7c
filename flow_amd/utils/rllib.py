"""flow_params <-> JSON, as flow/utils/rllib.py:22-192 does for RLlib checkpoints and the visualizers.

``get_flow_params`` also reads files written by the reference itself (class paths under ``flow.`` are
resolved to the same names under ``flow_amd.``), so a stored ``flow_params.json`` replays on the GPU."""
import inspect
import json
from copy import deepcopy

from flow_amd.core.params import (EnvParams, InFlows, InitialConfig, NetParams, SumoCarFollowingParams,
                                  SumoLaneChangeParams, SumoParams, TrafficLightParams, VehicleParams)
from flow_amd.envs.base import Env
from flow_amd.networks.base import Network


class FlowParamsEncoder(json.JSONEncoder):
    """json.JSONEncoder that understands VehicleParams, env / network classes and the param objects
    (flow/utils/rllib.py:22-58)."""

    def default(self, obj):
        if isinstance(obj, VehicleParams):
            res = deepcopy(obj.initial)
            for res_i in res:
                for key in ("acceleration_controller", "lane_change_controller"):
                    res_i[key] = (res_i[key][0].__name__, res_i[key][1])
                if res_i["routing_controller"] is not None:
                    res_i["routing_controller"] = (res_i["routing_controller"][0].__name__,
                                                   res_i["routing_controller"][1])
            return res
        if inspect.isclass(obj):
            if issubclass(obj, Env) or issubclass(obj, Network):
                return "{}.{}".format(obj.__module__, obj.__name__)
        if hasattr(obj, '__name__'):
            return obj.__name__
        if hasattr(obj, '__dict__'):
            return obj.__dict__
        return json.JSONEncoder.default(self, obj)


def _resolve(path, default_package):
    """Class named ``path`` ('pkg.mod.Name' or bare 'Name'); ``flow.*`` maps to ``flow_amd.*``."""
    if "." not in path:
        module_name, name = default_package, path
    else:
        module_name, name = ".".join(path.split(".")[:-1]), path.split(".")[-1]
    if module_name == "flow" or module_name.startswith("flow."):
        module_name = "flow_amd" + module_name[len("flow"):]
    module = __import__(module_name, fromlist=[name])
    if not hasattr(module, name):                # e.g. flow.envs.ring.wave_attenuation moved: fall back to the package
        module = __import__(default_package, fromlist=[name])
    return getattr(module, name)


def get_flow_params(config):
    """flow/utils/rllib.py:61-192: ``config`` is an RLlib config dict (``config['env_config']['flow_params']``)
    or the path of a flow_params json file."""
    if isinstance(config, dict):
        flow_params = json.loads(config['env_config']['flow_params'])
    else:
        with open(config, 'r') as f:
            flow_params = json.load(f)

    veh = VehicleParams()
    for veh_params in flow_params["veh"]:
        veh_params = dict(veh_params)
        acc = veh_params.pop('acceleration_controller')
        lc = veh_params.pop('lane_change_controller')
        rt = veh_params.pop('routing_controller')
        cf = SumoCarFollowingParams()
        cf.__dict__ = veh_params.pop("car_following_params")
        lcp = SumoLaneChangeParams()
        lcp.__dict__ = veh_params.pop("lane_change_params")
        veh.add(acceleration_controller=(_resolve(acc[0], "flow_amd.controllers"), acc[1]),
                lane_change_controller=(_resolve(lc[0], "flow_amd.controllers"), lc[1]),
                routing_controller=None if rt is None else (_resolve(rt[0], "flow_amd.controllers"), rt[1]),
                car_following_params=cf, lane_change_params=lcp, **veh_params)

    sim = SumoParams()
    sim.__dict__.update(flow_params["sim"])
    net = NetParams()
    net.__dict__ = dict(flow_params["net"])
    inflows = InFlows()
    stored = flow_params["net"].get("inflows")
    if stored:
        flows = stored.get("_InFlows__flows", []) if isinstance(stored, dict) else []
        inflows._InFlows__flows = list(flows)
    net.inflows = inflows
    env = EnvParams()
    env.__dict__ = dict(flow_params["env"])
    initial = InitialConfig()
    if "initial" in flow_params:
        initial.__dict__ = dict(flow_params["initial"])
    tls = TrafficLightParams()

    flow_params['env_name'] = _resolve(flow_params['env_name'], "flow_amd.envs")
    flow_params['network'] = _resolve(flow_params['network'], "flow_amd.networks")
    flow_params["sim"], flow_params["env"], flow_params["initial"] = sim, env, initial
    flow_params["net"], flow_params["veh"], flow_params["tls"] = net, veh, tls
    return flow_params
