"""Import helper: the product package lives in the directory `video-k-net_amd/` (not a valid Python identifier);
`load()` imports it under the module name `video_k_net_amd`."""
import importlib.util
import os
import sys

_NAME = 'video_k_net_amd'
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'video-k-net_amd')


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, '__init__.py'),
                                                  submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    try:
        spec.loader.exec_module(mod)
    except BaseException:
        sys.modules.pop(_NAME, None)
        raise
    return mod
