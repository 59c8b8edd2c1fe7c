"""Pin the oracle's controller / fail-safe restatement against
(a) the reference's own known-answer tests (tests/fast_tests/test_controllers.py)
and (b) golden vectors produced by importing the reference (tests/golden/gen_golden.py)."""
import json
import os

import numpy as np
import pytest

from oracle import controllers as C

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
DOC = json.load(open(os.path.join(GOLDEN, "controllers.json")))
FS = json.load(open(os.path.join(GOLDEN, "failsafes.json")))
DT = 0.1


def ring_inputs(speeds, headways):
    v = np.asarray(speeds, dtype=np.float64)
    h = np.asarray(headways, dtype=np.float64)
    n = len(v)
    has = np.full(n, n > 1)
    return dict(v=v, h=h, has=has, v_lead=np.roll(v, -1), v_follow=np.roll(v, 1),
                h_follow=np.roll(h, 1), n=n)


def oracle_eval(name, speeds, headways):
    s = ring_inputs(speeds, headways)
    v, vl, h, has = s["v"], s["v_lead"], s["h"], s["has"]
    if name == "CFM_test":
        return C.cfm(v, vl, h, has, 20, 1, 1, 1, 1, 8)
    if name == "BCM_test":
        return C.bcm(v, vl, h, has, s["v_follow"], s["h_follow"], 15, 1, 1, 1, 1, 8)
    if name == "OVM_test":
        return C.ovm(v, vl, h, has, 15, 1, 1, 2, 15, 30)
    if name == "LinearOVM_test":
        return C.linear_ovm(v, h, 30, 0.65, 5)
    if name in ("IDM_test", "IDM_default"):
        return C.idm(v, vl, h, has, 30, 1, 1, 1.5, 4, 2)
    if name == "FollowerStopper_test":
        a = C.follower_stopper(v, vl, h, has, DT, 7.5)
        return C.failsafe_safe_velocity(a, v, vl, h, DT, 1.0, s["n"])      # velocity_controllers.py:31-33
    if name == "NonLocalFollowerStopper_test":
        a = C.follower_stopper(v, vl, h, has, DT, np.mean(v))
        return C.failsafe_safe_velocity(a, v, vl, h, DT, 1.0, s["n"])
    if name == "LAC_test":
        return C.lac(v, vl, h, 5.0, np.zeros_like(v), DT, 0.3, 0.4, 1, 0.1)
    if name == "Gipps_test":
        return C.gipps(v, vl, h, DT, 30, 1.5, -1, -1, 2, 1)
    raise KeyError(name)


@pytest.mark.parametrize("name", [k for k in DOC["known"] if k != "PISaturation_test"])
def test_known_answers_of_reference_tests(name):
    case = DOC["known"][name]
    got = oracle_eval(name, case["speeds"], case["headways"])
    # the literal arrays in the reference's tests (6 decimals, assert_array_almost_equal)
    np.testing.assert_array_almost_equal(got, case["reference_test_expected"], decimal=6)
    # and the imported reference itself, to float64 rounding
    np.testing.assert_allclose(got, case["reference_output"], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("name", list(DOC["random"]))
def test_random_grids_vs_imported_reference(name):
    for case in DOC["random"][name]:
        got = oracle_eval(name, case["speeds"], case["headways"])
        ref = np.array([np.nan if o is None else o for o in case["out"]], dtype=np.float64)
        np.testing.assert_allclose(got, ref, rtol=2e-12, atol=1e-12, equal_nan=True)


def test_idm_zero_headway_does_not_raise():
    # tests/fast_tests/test_controllers.py:276-285
    got = oracle_eval("IDM_test", [0] * 5, [0] * 5)
    assert np.all(np.isfinite(got))


def test_pisaturation_trajectory():
    st = C.PISaturationState((5,))
    for k, step in enumerate(DOC["pisaturation_traj"]):
        s = ring_inputs(step["speeds"], step["headways"])
        got = C.pi_saturation(st, s["v"], s["v_lead"], s["h"], DT, 20)
        np.testing.assert_allclose(got, step["out"], rtol=1e-12, atol=1e-12)
        if k == 0:
            np.testing.assert_array_almost_equal(
                got, DOC["known"]["PISaturation_test"]["reference_test_expected"], decimal=6)


def test_lac_trajectory():
    a = np.zeros(5)
    for step in DOC["lac_traj"]:
        s = ring_inputs(step["speeds"], step["headways"])
        a = C.lac(s["v"], s["v_lead"], s["h"], 5.0, a, DT, 0.3, 0.4, 1, 0.1)
        np.testing.assert_allclose(a, step["out"], rtol=1e-12, atol=1e-12)


def test_failsafes_vs_imported_reference():
    for case in FS["cases"]:
        s = ring_inputs(case["speeds"], case["headways"])
        raw = C.idm(s["v"], s["v_lead"], s["h"], s["has"])
        np.testing.assert_allclose(raw, case["raw_idm"], rtol=1e-12, atol=1e-12)
        if case["fail_safe"] == "instantaneous":
            got = C.failsafe_instantaneous(raw, s["v"], s["h"], s["has"], DT, s["n"])
        else:
            got = C.failsafe_safe_velocity(raw, s["v"], s["v_lead"], s["h"], DT, case["delay"], s["n"])
        np.testing.assert_allclose(got, case["out"], rtol=1e-12, atol=1e-12)


def test_float32_twin_close_to_float64():
    rng = np.random.default_rng(0)
    v = rng.uniform(0, 20, 256)
    vl = rng.uniform(0, 20, 256)
    h = rng.uniform(0.5, 50, 256)
    has = np.ones(256, bool)
    a64 = C.idm(v, vl, h, has)
    a32 = C.idm(v.astype(np.float32), vl.astype(np.float32), h.astype(np.float32), has)
    assert a32.dtype == np.float32
    np.testing.assert_allclose(a32, a64, rtol=5e-5, atol=5e-5)


def test_exact_box_muller_functions_are_accurate_and_pure_float32():
    """oracle/refsim.py exact_ln_f32 / exact_cos_turns_f32 (the fixed float32 sequences the kernels run with
    noise_math='exact'): float32 in, float32 out, within 1e-6 (on values up to 16.6) / 1.2e-7 of the float64 functions over the whole input grid
    (u = m / 2^24), and the Gaussian draws they give keep mean 0 and variance 1."""
    from oracle import refsim as S
    u = (np.arange(1, 1 << 24, 251, dtype=np.float64) / 16777216.0).astype(np.float32)
    ln = S.exact_ln_f32(u)
    assert ln.dtype == np.float32 and np.abs(ln - np.log(u.astype(np.float64))).max() < 1e-6
    assert S.exact_ln_f32(np.float32(1.0)) == 0.0
    t = np.concatenate([np.arange(0, 1 << 24, 257) / 16777216.0, np.arange(0, 1 << 24, 263) / 16777216.0 - 0.25]).astype(np.float32)
    c = S.exact_cos_turns_f32(t)
    assert c.dtype == np.float32 and np.abs(c - np.cos(2 * np.pi * t.astype(np.float64))).max() < 1.2e-7
    n = 200000
    g = S.gaussian_noise(7, np.arange(n, dtype=np.uint32) % 1000, np.arange(n, dtype=np.uint32) // 1000,
                         np.arange(n, dtype=np.uint32) % 37, np.float32, exact=True)
    ref = S.gaussian_noise(7, np.arange(n, dtype=np.uint32) % 1000, np.arange(n, dtype=np.uint32) // 1000,
                           np.arange(n, dtype=np.uint32) % 37, np.float64)
    assert g.dtype == np.float32 and np.abs(g - ref).max() < 2e-6
    assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01
