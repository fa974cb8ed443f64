"""The host team of the host-buffer path (quantumcollocation.jl_amd/csrc/qc_host_team.h: worker pool, landing watch, pinned-ring re-arm,
deadline) on the CPU under the sanitizers: tests/host_team_test.cpp drives it with a thread standing in for the GPU's copy engine
(in address order, in random order, stalling for ever), built with -fsanitize=thread and with -fsanitize=address,undefined."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_host_team_under_thread_and_address_sanitizers():
    r = subprocess.run(["bash", os.path.join(ROOT, "tests", "run_sanitizers_host.sh"), "5"], capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "thread sanitizer: clean" in out and "address + undefined-behaviour sanitizers: clean" in out, out[-2000:]
    assert out.count("host team test: 0 failure(s)") == 2, out[-2000:]
    assert "ThreadSanitizer" not in out and "AddressSanitizer" not in out and "runtime error" not in out, out[-3000:]
