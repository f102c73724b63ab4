"""AddressSanitizer + UndefinedBehaviorSanitizer over everything that runs on the CPU: the product's host code (scene graph,
catalogue, flattener, BVH builders, time-aware boxes) and both CPU checkers, as one g++ program (oracle/sanitize_main.cpp,
`make -C oracle sanitize`).  GPU sanitizers are not available on this pool; the device code is covered by bit-exact parity instead."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_and_oracles_are_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    env.pop("LD_PRELOAD", None)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_build", "sanitize_check")], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "sanitizer run clean" in out.stdout and "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    assert out.stdout.count("O1 == O2") == 18  # 15 catalogue scenes + 3 with the reference's BVH rule
