import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
GOLDEN = ROOT / 'tests' / 'golden'


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a ROCm GPU (MI355X); run with -m gpu on the GPU box')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def built_lib():
    """The in-tree HIP library (built here by hipcc cross-compilation; it travels to the GPU box)."""
    from tedm_amd import build
    return build.build()
