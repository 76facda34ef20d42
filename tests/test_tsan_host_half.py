"""CPU suite: the host half of the LM step under ThreadSanitizer (`make -C eventcalib_amd/csrc tsan`): the solver's worker pool
(HostPool: epochs, nudges, spin-then-sleep), the partitioned banded-arrow solves on it and the host tasks of the streamed
evaluation (arrow_streamed_tasks: flags polled with acquire loads, interiors factorised as they arrive, separators behind them)
— the product's own headers (eventcalib_amd/csrc/arrow_host.hpp, arrow_host_parts.hpp) compiled with g++ -fsanitize=thread
into tests/cpp/tsan_host_half.cpp, where a host thread plays the kernel (ne_publish_progress in ecal_solver.hip).  No GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "eventcalib_amd", "csrc")


@pytest.mark.timeout(600)
def test_host_half_is_clean_under_thread_sanitizer():
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("no g++ / make")
    build = subprocess.run(["make", "-s", "-C", CSRC, "tsan"], capture_output=True, text=True)
    assert build.returncode == 0, build.stdout + build.stderr
    exe = os.path.join(CSRC, "build", "tsan_host_half")
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1")
    run = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=540)
    if "FATAL: ThreadSanitizer" in run.stderr and "unexpected memory mapping" in run.stderr:
        pytest.skip("ThreadSanitizer cannot map its shadow memory on this kernel (ASLR entropy)")
    assert "ThreadSanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.returncode == 0 and "tsan_host_half: ok" in run.stdout, run.stdout[-2000:] + run.stderr[-2000:]
