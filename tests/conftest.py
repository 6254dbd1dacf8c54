import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """libvlg.so is built in-tree (hipcc, gfx950) if it is missing or older than its sources; the tests never fall back."""
    import video_llamagen_amd  # noqa: F401
    from video_llamagen_amd import build
    build.build(verbose=False)


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    class G:
        def __init__(self):
            self._c = {}

        def __call__(self, part):
            if part not in self._c:
                self._c[part] = dict(np.load(os.path.join(ROOT, "tests", "golden", part + ".npz")))
            return self._c[part]

    return G()
