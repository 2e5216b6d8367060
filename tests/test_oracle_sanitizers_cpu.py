"""The CPU oracle is the checker every GPU parity test trusts, so its C code is itself run under
AddressSanitizer + UndefinedBehaviorSanitizer ("race detection / sanitizers": the reference has
none) on ragged, empty, capped and degenerate inputs -- oracle/sanitize_run.py in a child process with libasan
preloaded."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_oracle_c_code_is_clean_under_asan_and_ubsan():
    from oracle import build

    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    build.build_oracle_sanitized()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "sanitize_run.py")], env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "sanitize_run: ok" in out.stdout
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
