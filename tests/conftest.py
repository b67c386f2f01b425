import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# Collection order under `pytest -x`: kernels first, then end-to-end parity, graph / trainer semantics, and the
# multi-process data-parallel tests LAST, so that an infrastructure problem in a process-spawning test can never
# gate a kernel or parity test (round-1 driver run: the alphabetically-first DP test hung and nothing else ran).
_FILE_RANK = {
    'test_cabi_and_host.py': 0,
    'test_oracle_golden.py': 1,
    'test_hip_kernels.py': 2,
    'test_hip_fp32.py': 3,
    'test_hip_parity.py': 4,
    'test_baseline_configs.py': 5,
    'test_audio_frontend_gpu.py': 7,
    'test_dp_gloo.py': 8,
    'test_dp_rccl_gpu.py': 9,
}
DEFAULT_TIMEOUT_S = 420            # hard per-test limit (pytest-timeout); spawning tests set their own, shorter, limits


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'fresh_process: starts another interpreter (a second `import torch`: seconds on most boxes, a minute on some); never skipped')
    config.addinivalue_line('markers', 'timeout: per-test limit (pytest-timeout)')


def pytest_collection_modifyitems(config, items):
    order = {id(it): i for i, it in enumerate(items)}
    items.sort(key=lambda it: (_FILE_RANK.get(os.path.basename(str(it.fspath)), 6), order[id(it)]))
    have_timeout = config.pluginmanager.hasplugin('timeout')
    for it in items:
        if have_timeout and it.get_closest_marker('timeout') is None:
            it.add_marker(pytest.mark.timeout(DEFAULT_TIMEOUT_S))


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)
        return cache[name]
    return load


@pytest.fixture(autouse=True)
def _collect_between_tests():
    """Captured steps (hipGraphs + their private memory pools) of a finished test are destroyed HERE, at the test boundary — not by a
    cyclic-GC pass that happens to run in the middle of a later test's graph capture or replay (DAV_TEST_GC: off = never collect,
    debugging aid)."""
    import gc
    mode = os.environ.get('DAV_TEST_GC', 'collect')
    if mode == 'off':
        gc.disable()
    yield
    if mode == 'collect':
        gc.collect()
