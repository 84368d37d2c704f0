"""First-collected GPU canary: the engine round 1's driver-side run aborted in (`rbl_engine_create`, 1 lane, 1dx4f,
DCFR) is created, stepped and destroyed in-process before any parity test; a throwing constructor (bad device id)
leaves the runtime usable (streams / events are released)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_canary_one_lane_engine_lifecycle():
    from rebel_amd import capi

    assert capi.device_count() >= 1
    p = capi.make_params(num_iters=64, max_depth=2, use_cfr=True, dcfr=True, dcfr_alpha=1.5, dcfr_beta=0.5,
                         dcfr_gamma=2.0)
    for _ in range(3):
        e = capi.Engine(1, 4, p, max_lanes=1)
        e.set_net_synthetic()
        e.reset([-1], [0], np.full((1, 2, 4), 0.25))
        e.multistep()
        assert np.isfinite(e.get(0, capi.GET_REGRETS)).all()
        e.close()


def test_throwing_constructor_releases_handles():
    from rebel_amd import capi

    p = capi.make_params(num_iters=4, max_depth=2, use_cfr=True)
    for _ in range(64):  # leaked streams would exhaust the runtime's queue pool long before this
        with pytest.raises(capi.RebelError, match="no such HIP device"):
            capi.Engine(1, 4, p, max_lanes=1, device=4096)
    with pytest.raises(capi.RebelError, match="exclusive"):
        capi.Engine(1, 4, capi.make_params(num_iters=4, use_cfr=True, linear_update=True, dcfr=True), max_lanes=1)
    e = capi.Engine(1, 4, p, max_lanes=1)
    e.close()


def test_null_handles_are_refused():
    from rebel_amd import capi

    L = capi.lib()
    assert L.rbl_solver_num_lanes(None) == -1 and L.rbl_solver_total_rows(None) == -1
    assert L.rbl_engine_stream(None) is None
    assert L.rbl_solver_step(None, 0) != 0 and b"null engine" in L.rbl_last_error()
