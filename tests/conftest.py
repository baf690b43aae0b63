import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')


AB_LIB = os.path.join(ROOT, 'nele_gan_amd', 'libnele_hip_ab.so')


def ab_env(**flags):
    """Environment of an A/B child process: the TEST library (libnele_hip_ab.so: same sources and ABI as the product library, built
    with -DNELE_AB, in which the superseded kernel variants exist and the NELE_* switches are read) plus the switches.  The product
    library ignores every NELE_* switch - it has one path per operation."""
    assert os.path.exists(AB_LIB), "build first (make -C nele_gan_amd/csrc builds libnele_hip.so and libnele_hip_ab.so)"
    return dict(os.environ, NELE_LIB=AB_LIB, **{k: str(v) for k, v in flags.items()})
