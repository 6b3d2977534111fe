"""The library's host-side sources (no GPU calls in them) under AddressSanitizer and
UndefinedBehaviorSanitizer, driven by tests/tools/host_sanitize.cpp with random and
edge-case inputs.  The reference has no sanitizer runs (its memory safety rests on
convention, e.g. the USEDMALLOC hook of include/vectors.h:6-8)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    src = [os.path.join(ROOT, "tests", "tools", "host_sanitize.cpp"),
           os.path.join(ROOT, "presto_amd", "csrc", "accel_host.cpp"),
           os.path.join(ROOT, "presto_amd", "csrc", "accel_output.cpp")]
    subprocess.check_call([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "presto_amd", "csrc")] + src + ["-pthread", "-o", exe], timeout=300)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, timeout=300, env=env)
    assert r.returncode == 0, (r.stdout.decode()[-2000:], r.stderr.decode()[-4000:])
    assert "host sanitizer run ok" in r.stdout.decode()
