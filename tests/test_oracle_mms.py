"""CPU: the reference's manufactured-solution test test/swe2d/test_steady_state_basin_mms.py (setups 7-9, element family
dg-dg) with the oracle and SSPRK33 - second-order convergence of elevation and velocity to the ANALYTIC steady state pins
every term of the operator (pressure gradient, HUDiv, advection, Coriolis, sources, SIPG viscosity with the grad-div and
grad-depth terms, and all Function-valued boundary kinds) against truth, not against another restatement."""
import pytest

import mms_basin


@pytest.mark.parametrize('name', ['setup7', 'setup8', 'setup9'])
def test_steady_state_basin_convergence_oracle(name):
    refs = [1, 2]                          # the reference uses [1, 2, 4, 6] (the GPU test does too); kept short for the CPU suite
    errs = [mms_basin.run_oracle(name, r) for r in refs]
    slope_e, slope_u = mms_basin.convergence_rates(errs, refs)
    # test_steady_state_basin_mms.py:277-278: |slope - (order+1)|/(order+1) < 0.2
    assert abs(slope_e - 2.0)/2.0 < 0.2, (errs, slope_e)
    assert abs(slope_u - 2.0)/2.0 < 0.2, (errs, slope_u)
