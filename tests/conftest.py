import os
import sys

import pytest

os.environ.setdefault('CASV_FAULT_INJECTION', '1')     # the library's give-up paths are forced by tests only (casv_set_option)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_sessionstart(session):
    """A fresh checkout has no built library (it is git-ignored): build it once (hipcc cross-compiles without a GPU).
    The tests themselves still fail loudly if the library cannot be loaded."""
    lib = os.path.join(ROOT, 'cor_asv_ann_amd', 'lib', 'libcor_asv_ann_hip.so')
    if not os.path.exists(lib):
        try:
            import __graft_entry__
            __graft_entry__.build()
        except Exception as err:      # reported by the first test that loads the library
            sys.stderr.write('could not build the HIP library: %s\n' % err)


def pytest_collection_modifyitems(config, items):
    """A box without a GPU skips the gpu-marked tests (a missing or unloadable library is NOT a reason to skip: then
    they run and fail loudly)."""
    gpu_items = [it for it in items if it.get_closest_marker('gpu')]
    if not gpu_items:
        return
    try:
        import cor_asv_ann_amd._native as nv
        ndev = nv.load().casv_device_count()
    except Exception:
        return
    if ndev == 0:
        skip = pytest.mark.skip(reason='no HIP device on this box')
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
