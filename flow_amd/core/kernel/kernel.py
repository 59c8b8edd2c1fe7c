"""Kernel facade (flow/core/kernel/kernel.py:48-107): bundles network, vehicle, simulation views."""
from flow_amd.core.kernel.network import NetworkKernel
from flow_amd.core.kernel.vehicle import VehicleKernel


class SimulationKernel(object):
    """k.simulation: no subprocess and no socket stand behind it (flow/core/kernel/simulation/traci.py) -- a step is
    one kernel launch made by Env.step; what remains is the collision flag of the last step (traci.py:66-68)."""

    def __init__(self, master_kernel):
        self.master_kernel = master_kernel
        self.crashed = False

    def check_collision(self):
        return self.crashed

    def update(self, reset):
        pass

    def close(self):
        pass


class Kernel(object):
    def __init__(self, simulator, sim_params):
        if simulator not in ("traci", "hip"):
            raise ValueError('Simulator type "{}" is not valid.'.format(simulator))
        self.kernel_api = None
        self.sim_params = sim_params
        self.network = None                      # set by Env: NetworkKernel(network)
        self.vehicle = VehicleKernel(self, sim_params)
        self.simulation = SimulationKernel(self)
        self.traffic_light = None
        self.detector = None

    def generate_network(self, network):
        self.network = NetworkKernel(network, junction_length=getattr(self.sim_params, "junction_length", 0.1),
                                     center_length=getattr(self.sim_params, "center_length", None))
        return self.network

    def pass_api(self, kernel_api):
        self.kernel_api = kernel_api

    def update(self, reset):
        """kernel.py:89-107 order: vehicle, (traffic light), network, simulation."""
        self.vehicle.update(reset)
        self.network.update(reset)
        self.simulation.update(reset)

    def close(self):
        self.network.close()
        self.simulation.close()
