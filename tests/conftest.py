import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _fixed_forward_walk():
    """Tests run with a FIXED walk of the forward compositing (four waves per quadrant) so that two runs of a frame are
    comparable bit for bit; the trainer's own choice (fused._FwdWalkTuner alternates the two walks on early frames and
    keeps the faster) is exercised by the tests that ask for ``free_forward_walk``, the one-wave walk by ``fwd_walk``."""
    try:
        from fusionsense_amd import fused
    except Exception:  # (the package needs torch; CPU-only helpers do not)
        yield
        return
    t = fused.FWD_WALK
    saved = (t.forced, t.forced_walk)
    t.forced, t.forced_walk = True, 0
    yield
    t.forced, t.forced_walk = saved
    t.state.clear()


@pytest.fixture(params=[0, 1], ids=["four_waves", "one_wave"])
def fwd_walk(request, _fixed_forward_walk):
    from fusionsense_amd import fused
    fused.FWD_WALK.forced, fused.FWD_WALK.forced_walk = True, int(request.param)
    return int(request.param)


@pytest.fixture
def free_forward_walk(_fixed_forward_walk):
    from fusionsense_amd import fused
    fused.FWD_WALK.forced = False
    fused.FWD_WALK.state.clear()
    return fused.FWD_WALK
