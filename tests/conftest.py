import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    if os.environ.get('E2T_TEST_SEGV_BT'):
        # diagnostics: native stack of a crashing thread (tests/native/segv_bt.c), installed over Python's faulthandler
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, 'tests', 'native', 'libe2t_test_segv_bt.so')).e2t_test_install_segv_bt()


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped automatically where no GPU is visible."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_objects(request):
    """(diagnostics: E2T_TEST_SEGV_BT=1 re-installs the native-backtrace handler before every test.)  After every GPU test: collect the engines of the test (they hold captured graphs, streams and workspaces in reference
    cycles, so they outlive the test until the cyclic collector happens to run) and wait for the device.  Without this the
    graph executables of dozens of tests pile up in one process."""
    if os.environ.get('E2T_TEST_SEGV_BT'):
        import ctypes
        ctypes.CDLL(os.path.join(ROOT, 'tests', 'native', 'libe2t_test_segv_bt.so')).e2t_test_install_segv_bt()
    yield
    if 'gpu' in request.keywords:
        import gc
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
            if os.environ.get('E2T_TEST_COUNT_GRAPHS'):
                n = sum(1 for o in gc.get_objects() if type(o).__name__ == 'CUDAGraph')
                free, total = torch.cuda.mem_get_info()
                print('\n[graphs alive %d, streams?, device memory used %.1f GB] %s' % (n, (total - free) / 1e9, request.node.name), flush=True)
