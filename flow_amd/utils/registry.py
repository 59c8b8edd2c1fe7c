"""make_create_env (flow/utils/registry.py:13-134): turn a ``flow_params`` dict into an
environment constructor.  Registers with gym when gym is installed; works without it."""
from copy import deepcopy

from flow_amd.core.params import InitialConfig, TrafficLightParams

_REGISTERED = set()


def make_create_env(params, version=0, render=None):
    """Same contract as the reference: returns ``(create_env, env_name)``.  ``detector_params``
    is optional here (the fork hard-requires it although the in-scope networks do not accept
    it, registry.py:87, 101)."""
    exp_tag = params["exp_tag"]
    if isinstance(params["env_name"], str):
        import flow_amd.envs as envs
        import flow_amd.envs.multiagent as ma_envs
        env_class = getattr(envs, params["env_name"], None) or getattr(ma_envs, params["env_name"])
    else:
        env_class = params["env_name"]
    base_env_name = env_class.__name__
    while "{}-v{}".format(base_env_name, version) in _REGISTERED:
        version += 1
    env_name = "{}-v{}".format(base_env_name, version)
    _REGISTERED.add(env_name)

    if isinstance(params["network"], str):
        import flow_amd.networks as networks
        network_class = getattr(networks, params["network"])
    else:
        network_class = params["network"]

    env_params = params['env']
    net_params = params['net']
    initial_config = params.get('initial', InitialConfig())
    traffic_lights = params.get("tls", TrafficLightParams())

    def create_env(*_):
        sim_params = deepcopy(params['sim'])
        vehicles = deepcopy(params['veh'])
        network = network_class(name=exp_tag, vehicles=vehicles, net_params=net_params,
                                initial_config=initial_config, traffic_lights=traffic_lights)
        sim_params.render = render or sim_params.render
        kwargs = {"env_params": env_params, "sim_params": sim_params, "network": network,
                  "simulator": params.get('simulator', 'traci')}
        try:                                                  # pragma: no cover - gym absent in this image
            import gym
            from gym.envs.registration import register
            register(id=env_name, entry_point=env_class.__module__ + ':' + env_class.__name__, kwargs=kwargs)
            return gym.envs.make(env_name)
        except ImportError:
            return env_class(**kwargs)

    return create_env, env_name


def env_constructor(params, version=0, render=None):
    create_env, env_name = make_create_env(params, version, render)
    return create_env
