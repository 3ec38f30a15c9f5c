import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'trtllm-llama_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    # a -m gpu run on a box without a GPU must fail loudly, not skip silently: the gpu-marked tests that were SELECTED need one
    expr = config.getoption('-m') or ''
    if 'gpu' not in expr or 'not gpu' in expr:
        return
    if not any(it.get_closest_marker('gpu') for it in items):
        return
    import torch
    if not torch.cuda.is_available():
        pytest.exit('pytest -m gpu: no GPU is visible to torch - the gpu-marked tests cannot run here (they do not skip)', returncode=3)


@pytest.fixture(scope='session')
def lib():
    from tensorrt_llm.plugin import capi
    return capi.load_library()
