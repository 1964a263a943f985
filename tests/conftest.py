import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell_flat():
    import vkrt_amd  # noqa: F401
    from vkrt_amd.flat_scene import FlatScene

    return FlatScene.load_npz(os.path.join(GOLDEN, "cornell_flat.npz"))


@pytest.fixture(scope="session")
def cornell_oracle(cornell_flat):
    import oracle_py

    return oracle_py.OracleScene(cornell_flat)


def default_camera(width, height, **kw):
    import camera_np
    from vkrt_amd.flat_scene import uniforms_from_matrices

    return uniforms_from_matrices(*camera_np.global_uniforms(width=width, height=height, **kw))
