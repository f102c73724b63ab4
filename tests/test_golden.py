"""Committed golden fixtures (tests/golden, produced by tests/golden/make_golden.py with oracle O1).

CPU: both oracles must reproduce them bit for bit.  GPU (-m gpu): so must the HIP path.
These fixtures pin this build's own restatement; they are not outputs of the reference binary
(which cannot be built here) -- "parity unpinned" against reference output.
"""
import ctypes as C
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(GOLD, "make_golden.py"))
make_golden = importlib.util.module_from_spec(spec)
spec.loader.exec_module(make_golden)
MANIFEST = json.load(open(os.path.join(GOLD, "manifest.json")))
NAMES = sorted(make_golden.CASES)


def _golden(name):
    accum = np.load(os.path.join(GOLD, name + ".accum.npy"), allow_pickle=False)
    ppm = open(os.path.join(GOLD, name + ".ppm"), "rb").read()
    assert hashlib.sha256(accum.tobytes()).hexdigest() == MANIFEST[name]["accum_sha256"]
    assert hashlib.sha256(ppm).hexdigest() == MANIFEST[name]["ppm_sha256"]
    return accum, ppm


def _ppm_bytes(rtsr, tmp_path, w, h, rgb8):
    p = tmp_path / "img.ppm"
    rtsr.Screen(w, h, rgb8).write_to_ppm_file(str(p))
    return p.read_bytes()


@pytest.mark.parametrize("name", NAMES)
def test_oracles_reproduce_golden(rtsr, orc, tmp_path, name):
    accum, ppm = _golden(name)
    b, world, cam, cfg, h = make_golden.build_case(rtsr, name)
    a1, r1 = orc.o1_render(b.graph_ptr(), world, cam, cfg, h, threads=8)
    assert np.array_equal(a1, accum)
    assert _ppm_bytes(rtsr, tmp_path, cfg.image_width, h, r1) == ppm
    flat = b.flatten(world)
    a2, r2 = orc.o2_render(flat.arrays_ptr(), cam, cfg, h, threads=8)
    assert np.array_equal(a2, accum) and np.array_equal(r2, r1)


def test_golden_streams(orc):
    for key, want in MANIFEST["_streams"].items():
        seed, pixel, sample = (int(x) for x in key.split(","))
        out = np.empty(8)
        orc.load().oracle_sample_stream(seed, pixel, sample, 8, out.ctypes.data_as(C.POINTER(C.c_double)))
        assert [float(x).hex() for x in out] == want


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_reproduces_golden(rtsr, tmp_path, name):
    accum, ppm = _golden(name)
    b, world, cam, cfg, h = make_golden.build_case(rtsr, name)
    screen = b.flatten(world).upload().render(cam, cfg)
    assert np.array_equal(screen.accum, accum)
    p = tmp_path / "gpu.ppm"
    screen.write_to_ppm_file(str(p))
    assert p.read_bytes() == ppm
