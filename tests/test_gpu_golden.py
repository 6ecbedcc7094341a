"""The reference's recorded jest snapshots (tests/golden) replayed on the HIP engine through
the same scenarios as tests/test_oracle_golden.py; tolerance 1e-6 abs (BASELINE.json north_star)."""
import numpy as np
import pytest

import test_oracle_golden as G
from elementary_amd.offline import OfflineRenderer

TOL = 1e-6


@pytest.fixture
def core_factory(gpu_required):
    from elementary_amd.runtime import Runtime

    def make(**kw):
        c = OfflineRenderer(lambda sr, bs: Runtime(sr, bs, device=0))
        c.initialize(**kw)
        return c

    def same(got, want):
        want = np.asarray(want, np.float64)
        assert float(np.abs(np.asarray(got, np.float64) - want).max()) <= TOL * max(1.0, float(np.abs(want).max()))
    make.same = same
    return make


def _replay(fn):
    def test(core_factory):
        fn(core_factory)
    test.__name__ = fn.__name__
    test.__doc__ = fn.__doc__
    return pytest.mark.gpu(test)


for _name in dir(G):
    if _name.startswith("test_"):
        globals()[_name] = _replay(getattr(G, _name))
