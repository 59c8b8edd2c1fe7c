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


# ------------------------------------------------------------------ FS_MIXED twin and the float64 division route
def mixed_numpy(spec, steps):
    """The FS_MIXED arithmetic in numpy (float64 state, float32 controller): second statement of the C twin."""
    f32 = np.float32
    R, N = spec["num_replicas"], spec["num_vehicles"]
    x = np.asarray(spec["init_pos"], np.float64).reshape(R, N).copy()
    v = np.zeros((R, N))
    p = np.array([vs["p"][:6] for vs in spec["vehicles"]], np.float64).T.astype(f32)        # [6, N]
    ln = np.array([vs.get("length", 5.0) for vs in spec["vehicles"]], np.float64)
    L = float(np.asarray(spec["ring_length"]).reshape(-1)[0]) + 4.0 * spec["junction_length"]
    dt = float(spec["sim_step"])
    ramp = dt / (dt + 1e-3)
    for _ in range(steps):
        d = np.roll(x, -1, 1) - x
        d = np.where(d < 0, d + L, d)
        h = (d - np.roll(ln, -1)[None, :]).astype(f32)
        vi, vl = v.astype(f32), np.roll(v, -1, 1).astype(f32)
        hh = np.where(np.abs(h) < f32(1e-3), f32(1e-3), h)
        tsab = f32(2) * np.sqrt(p[2] * p[3])
        dyn = vi * p[1] + vi * (vi - vl) / tsab
        s_star = p[5] + np.maximum(dyn, f32(0))
        q = s_star / hh
        r2 = (vi / p[0]) * (vi / p[0])
        acc = p[2] * (f32(1) - r2 * r2 - q * q)
        nv = np.maximum(v + acc.astype(np.float64) * dt, 0.0)
        v = v + (nv - v) * ramp
        xn = x + v * dt
        x = np.where(xn >= L, xn - L, xn)
    return x, v


def test_mixed_c_twin_bit_identical_to_its_numpy_statement():
    spec = perturbed(5, 22, 2, horizon=300)
    c = cbuild.CRingIDMMixed(spec)
    c.rollout(300)
    x, v = mixed_numpy(spec, 300)
    np.testing.assert_array_equal(c.x, x)
    np.testing.assert_array_equal(c.v, v)


def test_mixed_arithmetic_holds_1e4_against_float64_over_1500_steps_where_float32_does_not():
    # the north-star bar (trajectories within 1e-4 of the reference's float64 arithmetic) on the C2 start state
    spec = perturbed(256, 22, 1000, horizon=1500, bunching=20.0)
    ref = cbuild.CRingIDM(spec, np.float64, threads=4)
    mix = cbuild.CRingIDMMixed(spec, threads=4)
    f32 = cbuild.CRingIDM(spec, np.float32, threads=4)
    ref.rollout(1500), mix.rollout(1500), f32.rollout(1500)
    L = 230.4

    def dist(a, b):
        d = np.abs(np.asarray(a, np.float64) - b)
        return np.minimum(d, L - d).max()
    assert dist(mix.x, ref.x) < 1e-4 and np.abs(mix.v - ref.v).max() < 1e-4
    assert dist(f32.x, ref.x) > 1e-4                       # why FS_MIXED exists (DESIGN.md "Precision")
    assert ref.v.max() > 1.0


def test_div_via_f64_equals_ieee_float32_division_for_every_float():
    # flowsim_pair.h div_via_f64 (speed observation): all 2^31 non-negative floats for the benchmark's divisor,
    # the whole denormal-result range and a stride of the rest for other divisors
    assert cbuild.div_via_f64_mismatches(30.0, 0, 0x7F800000) == 0
    for c in (23.0, 15.0, 40.0, 7.3, 1e-3, 123456.7):
        assert cbuild.div_via_f64_mismatches(c, 0, 0x10000000) == 0
        assert cbuild.div_via_f64_mismatches(c, 0x3F000000, 0x40800000) == 0
