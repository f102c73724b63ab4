"""MI355X-native path-tracing hot path behind the C ABI of include/rtx_abi.h.

The directory name follows the reference repository (`ray-tracing-series-rust` + `_amd`) and
is therefore not a Python identifier; import it with

    import importlib
    rtsr = importlib.import_module("ray-tracing-series-rust_amd")

(tests/conftest.py does this once and exposes it as the `rtsr` fixture).
"""
from .api import *  # noqa: F401,F403
from .api import ABI, LIB_PATH, lib  # noqa: F401
