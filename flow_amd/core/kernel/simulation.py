"""k.simulation: the stepping facade (stands where flow/core/kernel/simulation/traci.py stood:
no subprocess, no socket -- one kernel launch per Env.step)."""


class SimulationKernel(object):
    def __init__(self, master_kernel):
        self.master_kernel = master_kernel
        self.crashed = False

    def start_simulation(self, network, sim_params):
        return None

    def simulation_step(self):
        raise NotImplementedError("the GPU simulator advances inside Env.step (one fused launch)")

    def update(self, reset):
        pass

    def check_collision(self):
        """simulation/traci.py:66-68: did the last step crash?"""
        return self.crashed

    def close(self):
        pass
