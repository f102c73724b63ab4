import importlib
import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Load order matters when torch and librtx_hip.so share a process: torch bundles its own
    # libamdhip64.so.7, librtx_hip.so links the system one (same soname).  Whichever loads first is the
    # one both use; torch only finds the GPU through its own copy, so torch goes first (bench.py does
    # the same).  Device pointers are then interchangeable between torch and the library.
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass


def _build_mod():
    spec = importlib.util.spec_from_file_location("rtsr_build", os.path.join(ROOT, "ray-tracing-series-rust_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def rtsr():
    """The product package (dlopens lib/librtx_hip.so; builds it first if it is missing or stale)."""
    b = _build_mod()
    if os.path.isdir("/opt/rocm") or os.environ.get("ROCM_PATH"):
        b.build_library(verbose=False)
    return importlib.import_module("ray-tracing-series-rust_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure)."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    import oracle_py
    oracle_py.load()
    return oracle_py
