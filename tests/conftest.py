import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_settings():
    """160x120, 3 levels (40x30 coarsest) -- the size the oracle chews in milliseconds."""
    from revo_amd.settings import ImgPyramidSettings
    return ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
