"""CPU: pin the plain-C MSDA oracle (oracle/msda_ref.c) against the golden vectors produced by the
reference's python path (tests/golden/make_golden.py; reference test: ops/test.py:35-89)."""
import numpy as np
import pytest

from conftest import MSDA_CFG, MSDA_TESTPY, load_msda_fixture, smooth_points


def _check_grad_value(z, gv, rtol, atol):
    if "grad_value" in z:
        np.testing.assert_allclose(gv, z["grad_value"], rtol=rtol, atol=atol)
    else:
        np.testing.assert_allclose(gv.reshape(-1)[::7], z["grad_value_stride7"], rtol=rtol, atol=atol)
        np.testing.assert_allclose(gv.sum(), z["grad_value_sum"], rtol=1e-9)


@pytest.mark.parametrize("name", MSDA_TESTPY + MSDA_CFG)
def test_oracle_f64_matches_reference(oracle_msda, name):
    z = load_msda_fixture(name)
    v = z["value"].astype(np.float64)
    loc = z["loc"].astype(np.float64)
    attn = z["attn"].astype(np.float64)
    g = z["grad_out"].astype(np.float64)
    out = oracle_msda.msda_forward(v, z["shapes"], z["level_start"], loc, attn)
    # fp64 tolerance of the reference's own test (test.py:43: torch.allclose defaults)
    np.testing.assert_allclose(out, z["out"], rtol=1e-5, atol=1e-8)
    # ... and much tighter, since both sides are the same fp64 function
    np.testing.assert_allclose(out, z["out"], rtol=1e-10, atol=1e-13)
    gv, gl, ga = oracle_msda.msda_backward(v, z["shapes"], z["level_start"], loc, attn, g)
    # fp32-stored grad_value in the cfg fixtures => 1e-6 relative
    _check_grad_value(z, gv, rtol=2e-6 if name in MSDA_CFG else 1e-10, atol=1e-6 if name in MSDA_CFG else 1e-13)
    np.testing.assert_allclose(gl, z["grad_loc"], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(ga, z["grad_attn"], rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("name", ["msda_testpy_float"] + MSDA_CFG)
def test_oracle_f32_within_reference_float_tolerance(oracle_msda, name):
    z = load_msda_fixture(name)
    f = np.float32
    out = oracle_msda.msda_forward(z["value"].astype(f), z["shapes"], z["level_start"],
                                   z["loc"].astype(f), z["attn"].astype(f))
    assert out.dtype == np.float32
    # fp32 tolerance of the reference's own test (test.py:59)
    np.testing.assert_allclose(out, z["out"], rtol=1e-2, atol=1e-3)
    if "out_f32" in z:  # the reference's own fp32 run
        np.testing.assert_allclose(out, z["out_f32"], rtol=1e-5, atol=1e-7)
    gv, gl, ga = oracle_msda.msda_backward(z["value"].astype(f), z["shapes"], z["level_start"],
                                           z["loc"].astype(f), z["attn"].astype(f), z["grad_out"].astype(f))
    np.testing.assert_allclose(gv, z["grad_value"], rtol=1e-2, atol=1e-3)
    ok = smooth_points(z)
    np.testing.assert_allclose(gl[ok], z["grad_loc"][ok], rtol=1e-2, atol=2e-3)
    np.testing.assert_allclose(ga, z["grad_attn"], rtol=1e-2, atol=1e-3)


def test_oracle_linearity_in_value(oracle_msda):
    """size-independent property: out is linear in value and in attn."""
    z = load_msda_fixture("msda_cfg_E_wide")
    v = z["value"].astype(np.float64)
    rng = np.random.default_rng(0)
    v2 = rng.standard_normal(v.shape)
    a = (z["shapes"], z["level_start"], z["loc"].astype(np.float64), z["attn"].astype(np.float64))
    o1 = oracle_msda.msda_forward(v, *a)
    o2 = oracle_msda.msda_forward(v2, *a)
    o3 = oracle_msda.msda_forward(2.0 * v - 3.0 * v2, *a)
    np.testing.assert_allclose(o3, 2.0 * o1 - 3.0 * o2, rtol=1e-10, atol=1e-11)
