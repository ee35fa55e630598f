import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def vkn():
    """The product package (directory `video-k-net_amd/`, imported as `video_k_net_amd`)."""
    import vkn_import
    mod = vkn_import.load()
    if not os.path.exists(mod._lib.LIBPATH) or os.environ.get('VKN_REBUILD') == '1':
        mod.build(force=True)          # hipcc cross-compiles gfx950 without a GPU (same recipe as __graft_entry__.build)
    if os.environ.get('VKN_EXPECT_MM') == '1':      # tests/test_mm_registry_branch.py re-runs parts of the suite with mmcv / mmdet importable
        assert mod.registry.HAVE_MM, 'VKN_EXPECT_MM=1 but mmcv / mmdet did not import: the plug-in branch of registry.py was not taken'
    return mod
