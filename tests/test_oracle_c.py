"""The C restatement (oracle/csim) must agree with the numpy oracle: bit-identical in
float32 (same operation order), and to rounding in float64."""
import numpy as np

from helpers import ring_spec
from oracle import cbuild
from oracle import refsim as S


def perturbed(R, N, seed, **kw):
    spec = ring_spec(R=R, N=N, junction_length=0.1, **kw)
    rng = np.random.default_rng(seed)
    spec["init_pos"] = np.asarray(spec["init_pos"]) + np.abs(rng.normal(0, 0.5, (R, N)))
    return spec


def test_c_twin_bit_identical_to_numpy_oracle_f32():
    spec = perturbed(6, 22, 0, horizon=250)
    ora = S.RingOracle(spec, np.float32)
    c = cbuild.CRingIDM(spec, np.float32)
    ora.reset()
    obs_c, rew_c, done_c = c.rollout(250, obs_every_step=True)
    for k in range(250):
        o, r, d = ora.step(None)
        np.testing.assert_array_equal(obs_c[k], o.astype(np.float32))
        np.testing.assert_array_equal(rew_c[k], r.astype(np.float32))
        np.testing.assert_array_equal(done_c[k], d)
    np.testing.assert_array_equal(c.x, ora.x)
    np.testing.assert_array_equal(c.v, ora.v)


def test_c_twin_matches_numpy_oracle_f64_and_threads_do_not_matter():
    spec = perturbed(9, 14, 1, length=200.0, bunching=10, horizon=400)
    ora = S.RingOracle(spec, np.float64)
    ora.reset()
    for _ in range(400):
        ora.step(None)
    a = cbuild.CRingIDM(spec, np.float64, threads=1)
    b = cbuild.CRingIDM(spec, np.float64, threads=4)
    a.rollout(400), b.rollout(400)
    np.testing.assert_array_equal(a.x, b.x)
    np.testing.assert_allclose(a.x, ora.x, rtol=0, atol=1e-10)
    np.testing.assert_allclose(a.v, ora.v, rtol=0, atol=1e-10)
