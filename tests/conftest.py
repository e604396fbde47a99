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
def _default_forward_walk():
    """Every test starts from (and leaves behind) the product's forward walk: one wave per quadrant with long walks
    handed off to the second pass (fused.FWD_WALK).  The walk is fixed — no measurement decides it any more — so two runs
    of a frame are comparable bit for bit; ``fwd_walk`` runs a test once per walk."""
    try:
        from fusionsense_amd import fused
    except Exception:  # (the package needs torch; CPU-only helpers do not)
        yield
        return
    t = fused.FWD_WALK
    saved = (t.forced, t.forced_walk, t.handoff_records, t.max_items, t.handoff_rel_len, t.handoff_gate_len)
    yield
    t.forced, t.forced_walk, t.handoff_records, t.max_items, t.handoff_rel_len, t.handoff_gate_len = saved


# (id, walk, handoff_records): the four-wave walk of rounds 2-3, the one-wave walk without a hand-off (round 4), and the
# product's walk with the hand-off forced EARLY (after 64 records: at test sizes most quadrants with more than one chunk
# of entries then go through the second pass) — the default (128) is what every test without this fixture runs
_WALKS = [("four_waves", 0, 0), ("one_wave", 1, 0), ("handoff", 1, 64)]


@pytest.fixture(params=_WALKS, ids=[w[0] for w in _WALKS])
def fwd_walk(request, _default_forward_walk):
    from fusionsense_amd import fused
    _, walk, handoff = request.param
    fused.FWD_WALK.forced, fused.FWD_WALK.forced_walk, fused.FWD_WALK.handoff_records = True, walk, handoff
    # (every quadrant that reaches the threshold hands off, whatever its list's length and the frame's longest list)
    fused.FWD_WALK.handoff_rel_len, fused.FWD_WALK.handoff_gate_len = 0, 0
    return request.param[0]
