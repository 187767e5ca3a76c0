"""The reference's Wilson HMC test again, with every pseudofermion solve done by the mixed-precision CG (tunable
mixed_action_solver): the stopping rule is enforced on the true fp64 residual, so the trajectory statistics are unchanged."""
import numpy as np
import pytest

from test_gpu_md import BETA, KAPPA, REF_PLAQ_WILSON_HMC, DeviceHMC, _fixture

pytestmark = pytest.mark.gpu


def test_hmc_with_mixed_precision_action_solver(lq, orc):
    assert lq.lib.device_count() > 0
    L, Uh, U = _fixture(lq)
    U.lattice.set_param("mixed_action_solver", 1)
    h = DeviceHMC(lq, U, KAPPA, BETA, dtau=0.05, mdsteps=20, nsw=10, seed=111)
    for _ in range(10):
        h.update()
    plaq = lq.calculate_Plaquette(U)
    print("mixed-solver HMC: dH =", ["%.3f" % d for d in h.dH], "accepted", sum(h.accepted), "/ 10, plaquette", plaq)
    assert abs(plaq - REF_PLAQ_WILSON_HMC) / REF_PLAQ_WILSON_HMC < 0.1
    assert sum(h.accepted) >= 6 and np.abs(np.array(h.dH)).max() < 2.0
    # same seeds, same integrator: the fp64-solver run of test_gpu_md.py gives dH within solver tolerance of these
    U2 = _fixture(lq)[2]
    h2 = DeviceHMC(lq, U2, KAPPA, BETA, dtau=0.05, mdsteps=20, nsw=10, seed=111)
    h2.update()
    assert abs(h2.dH[0] - h.dH[0]) < 1e-6


@pytest.mark.parametrize("mode", [1, 2])
def test_rhmc_trajectories_with_mixed_precision_pole_solves(lq, orc, mode):
    """The reference's Nf = 3 staggered RHMC stream (test/test_Nf3.toml) with every pole solved in mixed precision -- mode 1: one
    mixed-precision CG per pole; mode 2: the mixed-precision multi-shift CG (one fp32 pass for all poles + fp64 defect correction per
    pole).  The stopping rule holds for the true fp64 residuals, so dH and the plaquettes of the stream are those of the fp64 run."""
    from hmc_harness import run_stream
    assert lq.lib.device_count() > 0
    ref = run_stream(lq, "staggered_nf3", 2, seed=5)
    mix = run_stream(lq, "staggered_nf3", 2, seed=5, params={"mixed_action_solver": mode})
    assert np.array_equal(ref["accepted"], mix["accepted"])
    assert np.abs(ref["dH"] - mix["dH"]).max() < 1e-6
    assert np.abs(ref["plaq"] - mix["plaq"]).max() < 1e-9
