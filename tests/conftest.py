import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# Collection order of the GPU suite (the driver runs it with -x): what is held to the ORACLE and to the golden / reference-run
# fixtures goes first, then the stress and surface suites, then property tests that compare the device path with itself
# (run twice, adapter vs torch-built batch), the multi-process RCCL test last.  Files not listed keep their alphabetical place
# in the middle.
_ORDER = ["test_gpu_parity.py", "test_gpu_reference_pins.py", "test_gpu_ard.py", "test_gpu_gnn.py", "test_evaluate.py",
          "test_gpu_stress.py", "test_gpu_surface.py"]
_LAST = ["test_gpu_properties.py", "test_gpu_f3.py", "test_gpu_determinism.py", "test_zz_gpu_nccl.py"]


def pytest_collection_modifyitems(session, config, items):
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name in _ORDER:
            return (0, _ORDER.index(name))
        if name in _LAST:
            return (2, _LAST.index(name))
        return (1, 0)
    items.sort(key=rank)   # stable: the order inside a file is kept
