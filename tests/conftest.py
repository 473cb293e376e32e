import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "oracle", ROOT / "tools", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import gs_oracle
    return gs_oracle


@pytest.fixture(scope="session")
def gs():
    import gsdeblur_amd
    # test infrastructure: the Python orchestration twin of the frame pipeline and its A/B switches (the "plain path" of
    # the equivalence tests) plugs into ops.frame_backend; with every switch at its default the product path runs
    import python_frame_path
    python_frame_path.install()
    # the slice budget adapts across frames by default; the tests compare paths at the budget they set themselves
    # (test_adaptive_slice_budget switches it back on)
    from gsdeblur_amd import ops
    ops.SLICE_ADAPT = 0
    return gsdeblur_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
