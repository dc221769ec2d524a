import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    import torch
    # the CPU oracle runs many small ops: more threads than ~16 only adds contention
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no ROCm GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_options_back_to_defaults():
    """Tests flip library options (anyloc_set_option) to compare kernel variants: every test starts from the defaults
    (+ ANYLOC_OPTIONS) again."""
    yield
    from anyloc_amd import _lib
    if _lib._lib is not None:
        _lib._lib.anyloc_reset_options()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
