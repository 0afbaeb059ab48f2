import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'slow: CPU test that takes more than a few seconds')


@pytest.fixture(scope='session', autouse=True)
def _oracle_threads():
    """the CPU oracle runs beside the GPU tests: keep torch's CPU backend off os.cpu_count() threads on GPU boxes."""
    from oracle import hostcpu
    return hostcpu.configure()


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
