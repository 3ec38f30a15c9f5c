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
    # a -m gpu run on a box without a GPU must fail loudly, not skip silently
    pass


@pytest.fixture(scope='session')
def lib():
    from tensorrt_llm.plugin import capi
    return capi.load_library()
