"""A user-defined controller (the reference's extension point: subclass BaseController, write get_accel in Python --
flow/controllers/base_controller.py:42-118; here: flow_amd.controllers.CompiledController, get_accel as a device
function compiled into a copy of the library, FS_CTRL_USER): the generic step kernel with the user's law against the oracle
running the user's numpy restatement of it, bit for bit in float32; noise and the safe_velocity fail-safe on top; the stock
library refuses such a population loudly."""
import numpy as np
import pytest

from helpers import idm_vehicle, ring_spec
from oracle import refsim as S

pytestmark = pytest.mark.gpu

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
from compiled_controller import TimeGap, time_gap_numpy  # noqa: E402

BODY = TimeGap.SOURCE


def user_spec(R=7, N=22, noise=0.0, fail_safe=0):
    spec = ring_spec(R=R, N=N, junction_length=0.1, horizon=150, seed=3, noise_math="exact")
    rng = np.random.default_rng(1)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.4, (R, N)))
    veh = []
    for i in range(N):
        d = idm_vehicle(noise=noise, speed_mode=25, fail_safe=fail_safe)
        if i % 3 != 2:                                  # two of three vehicles run the user's law, the others IDM
            d.update(controller=S.CTRL_USER, p=[1.0 + 0.1 * (i % 4), 2.0, 0.3, 0.6, 0.05, 0.0, 0.0, 0.0])
        veh.append(d)
    spec["vehicles"] = veh
    spec["user_controller_source"] = BODY
    spec["user_controller_numpy"] = time_gap_numpy
    return spec


@pytest.mark.parametrize("noise,fail_safe", [(0.0, 0), (0.2, 2)])
def test_user_controller_in_the_generic_kernel_equals_the_oracle(noise, fail_safe):
    from flow_amd.sim import FlowSim
    spec = user_spec(noise=noise, fail_safe=fail_safe)
    sim, ora = FlowSim(spec, "f32"), S.RingOracle(spec, np.float32)
    assert "_user" in sim.lib._name
    np.testing.assert_array_equal(sim.reset(), ora.reset().astype(np.float32))
    for k in range(150):
        o, r, d = sim.step(None)
        o_ref, r_ref, d_ref = ora.step(None)
        np.testing.assert_array_equal(o, o_ref.astype(np.float32), err_msg="obs, step %d" % k)
        np.testing.assert_array_equal(r, r_ref.astype(np.float32))
        np.testing.assert_array_equal(d, d_ref)
    np.testing.assert_array_equal(sim.pos, ora.x)
    assert sim.last_kernel.startswith("k_steps") and sim.vel.max() > 2.0
    sim.close()


def test_user_controller_in_float64():
    """T = double: the same body in the float64 kernels (the reference's arithmetic type), against the float64 oracle."""
    from flow_amd.sim import FlowSim
    spec = user_spec(R=3)
    sim, ora = FlowSim(spec, "f64"), S.RingOracle(spec, np.float64)
    sim.reset(), ora.reset()
    for k in range(120):
        o, r, d = sim.step(None)
        o_ref, r_ref, d_ref = ora.step(None)
        np.testing.assert_allclose(o, o_ref.astype(np.float32), rtol=0, atol=1e-6)
    np.testing.assert_allclose(sim.pos, ora.x, rtol=0, atol=1e-9)
    np.testing.assert_allclose(sim.vel, ora.v, rtol=0, atol=1e-9)
    sim.close()


def test_the_stock_library_refuses_a_user_controller_and_the_class_builds_through_vehicle_params():
    from flow_amd import _lib as L
    from flow_amd.controllers import ContinuousRouter
    from flow_amd.core.params import EnvParams, InitialConfig, NetParams, SumoParams, VehicleParams
    from flow_amd.envs import AccelEnv
    from flow_amd.networks import RingNetwork
    from flow_amd.sim import FlowSim
    spec = dict(user_spec(R=2), user_controller_source=None)
    stock = FlowSim(spec, "f32")
    with pytest.raises(NotImplementedError, match="FS_CTRL_USER"):
        stock.reset()
    stock.close()

    veh = VehicleParams()
    veh.add("gap", acceleration_controller=(TimeGap, {"t_gap": 1.0}), routing_controller=(ContinuousRouter, {}),
            num_vehicles=14)
    net = RingNetwork("ring", veh, NetParams(additional_params={"length": 230, "lanes": 1, "speed_limit": 30, "resolution": 40}),
                      InitialConfig(bunching=20))
    env = AccelEnv(EnvParams(horizon=50, additional_params={"max_accel": 1, "max_decel": 1, "target_velocity": 10,
                                                            "sort_vehicles": False}), SumoParams(sim_step=0.1), net)
    assert env._spec["vehicles"][0]["controller"] == L.FS_CTRL_USER and env._spec["user_controller_source"] == BODY.strip()
    env.reset()
    for _ in range(50):
        obs, rew, done, _ = env.step(None)
    assert done and np.isfinite(obs).all() and env.k.vehicle.get_speed("gap_0") > 0.5
    env.terminate()
