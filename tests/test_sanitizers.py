"""AddressSanitizer + UBSan on everything that runs on the host: the oracle and the host half of librtamd (builders,
JSON/YAML/OBJ readers, flattener, accel builder, PNG writer) linked against a test-only device stub, then the CPU
tests re-run on those builds (sanitizers on the CPU build; GPU ASan is not available on this pool)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


def test_host_code_and_oracle_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run([os.path.join(ROOT, "tests", "asan", "run_host_asan.sh")], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    if "cannot find -lasan" in tail or "libasan" in tail and "No such file" in tail:
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in tail and "runtime error:" not in tail, tail


def test_multi_gpu_fan_out_under_thread_sanitizer():
    """SURVEY s5 "race detection": rt_render_multi's host threads (one per rank) on the device stub's fake devices, driven from two caller
    threads at once by tests/asan/tsan_fanout.cpp, built with -fsanitize=thread: no report, frames identical for every device list."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run([os.path.join(ROOT, "tests", "asan", "run_host_tsan.sh")], capture_output=True, text=True, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    if "cannot find -ltsan" in tail or ("libtsan" in tail and "No such file" in tail):
        pytest.skip("sanitizer runtime not installed")
    assert r.returncode == 0 and "ThreadSanitizer" not in tail and "tsan fan-out driver: ok" in tail, tail
