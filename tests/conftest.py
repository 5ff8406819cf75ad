import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tests.paths import DATA, DATA_ROOT  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cornell():
    from oracle import formats
    return formats.load_scene(os.path.join(DATA, "cornell_box.vks"), os.path.join(DATA, "cornell_box_textures"))


@pytest.fixture(scope="session")
def cornell_lights():
    from oracle import formats
    return formats.load_lights(os.path.join(DATA, "cornell_box.lights"))


@pytest.fixture(scope="session")
def cornell_spec():
    from oracle import formats
    return formats.load_rt_save(os.path.join(DATA, "saves", "cornell_box", "default.rt_save"))


@pytest.fixture(scope="session")
def oracle_cornell(cornell):
    import oracle
    return oracle.OracleScene(cornell)
