"""22 IDM cars on a 230 m ring creating stop-and-go waves (BASELINE configs[0]; the experiment of the
reference's examples/exp_configs/non_rl/ring.py, written against the ``flow`` names so that it runs on
either package: ``flow_amd.install_as_flow()`` maps them to the GPU implementation)."""
from flow.controllers import IDMController, ContinuousRouter
from flow.core.params import SumoParams, EnvParams, InitialConfig, NetParams
from flow.core.params import VehicleParams
from flow.envs.ring.accel import AccelEnv, ADDITIONAL_ENV_PARAMS
from flow.networks.ring import RingNetwork, ADDITIONAL_NET_PARAMS

vehicles = VehicleParams()
vehicles.add(veh_id="idm", acceleration_controller=(IDMController, {}), routing_controller=(ContinuousRouter, {}),
             num_vehicles=22)

flow_params = dict(
    exp_tag='ring', env_name=AccelEnv, network=RingNetwork, simulator='traci',
    sim=SumoParams(render=False, sim_step=0.1),
    env=EnvParams(horizon=1500, additional_params=ADDITIONAL_ENV_PARAMS),
    net=NetParams(additional_params=ADDITIONAL_NET_PARAMS.copy()),
    veh=vehicles,
    initial=InitialConfig(bunching=20),
)
