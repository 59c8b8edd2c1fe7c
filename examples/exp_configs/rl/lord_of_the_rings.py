"""Several separate rings, one RL vehicle on each (the flow_params of the reference's
examples/exp_configs/rl/multiagent/lord_of_the_rings.py, without the RLlib policy boilerplate): every ring is an agent
of MultiWaveAttenuationPOEnv with a 3-value observation and its own desired-velocity reward.  On the GPU step loop
ring r runs as replica r of the ring kernel."""
from flow.controllers import ContinuousRouter, IDMController, RLController
from flow.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
from flow.envs.multiagent import MultiWaveAttenuationPOEnv
from flow.networks import MultiRingNetwork

HORIZON = 3000
NUM_RINGS = 7

vehicles = VehicleParams()
for i in range(NUM_RINGS):
    vehicles.add(veh_id='human_{}'.format(i), acceleration_controller=(IDMController, {'noise': 0.2}),
                 routing_controller=(ContinuousRouter, {}), num_vehicles=21)
    vehicles.add(veh_id='rl_{}'.format(i), acceleration_controller=(RLController, {}),
                 routing_controller=(ContinuousRouter, {}), num_vehicles=1)

flow_params = dict(
    exp_tag='lord_of_numrings{}'.format(NUM_RINGS),
    env_name=MultiWaveAttenuationPOEnv,
    network=MultiRingNetwork,
    simulator='traci',
    sim=SumoParams(sim_step=0.1, render=False),
    env=EnvParams(horizon=HORIZON, warmup_steps=750,
                  additional_params={'max_accel': 1, 'max_decel': 1, 'ring_length': [230, 230], 'target_velocity': 4}),
    net=NetParams(additional_params={'length': 230, 'lanes': 1, 'speed_limit': 30, 'resolution': 40,
                                     'num_rings': NUM_RINGS}),
    veh=vehicles,
    initial=InitialConfig(bunching=20.0, spacing='custom'),
)
