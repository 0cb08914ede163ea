"""Host-side sanitizers (CPU container only; SURVEY.md section 5 -- the GPU pool allows none, so these never run with `-m gpu`):
  * AddressSanitizer + UndefinedBehaviorSanitizer over the CPU oracle (oracle/sanitize_driver.c, `make -C oracle sanitize`);
  * ThreadSanitizer over the C-ABI's host code: minsnap_capi.hip compiled host-only with -fsanitize=thread and driven from
    eight threads (cs-pathplan_amd/host/tsan_driver.cpp): validation, kernel naming, thread-local error text, the sharded
    entry's device enumeration and the staging-arena pool."""
import importlib.util
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _have_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_oracle_under_asan_ubsan():
    if _have_gpu():
        pytest.skip("sanitizers run on the CPU build only")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "sanitize"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "sanitize_driver: ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_cabi_host_code_under_tsan():
    if _have_gpu():
        pytest.skip("sanitizers run on the CPU build only")
    spec = importlib.util.spec_from_file_location("csp_build", os.path.join(ROOT, "cs-pathplan_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    exe = b.build_tsan_check()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-6000:])
    assert "tsan_driver: ok" in r.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-6000:]
