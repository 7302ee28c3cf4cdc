"""The host-only units of the product (thread pool + host lane, control plane, work-list build, closed-form FIFO, modulator) compiled
with g++ -fsanitize=thread and -fsanitize=address,undefined and run (tests/host_sanitize/host_units.cpp).  CPU build only: GPU
sanitizers are not available on this pool, and none of these units makes a GPU call.  SURVEY.md section 5, "Race detection /
sanitizers" (the reference has none; its own race is dab2eti.c:117-130)."""
import os
import subprocess

import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_sanitize")


@pytest.mark.parametrize("flavour,env", [("tsan", {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"}),
                                          ("asan", {"ASAN_OPTIONS": "detect_leaks=1 abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})])
def test_host_units_under_sanitizer(flavour, env):
    build = subprocess.run(["make", "-C", HERE, flavour], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert build.returncode == 0, build.stdout[-3000:]
    exe = os.path.join(HERE, "build", "host_units_" + flavour)
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, **env), timeout=600)
    assert run.returncode == 0, (run.stdout + run.stderr)[-4000:]
    assert "Sanitizer" not in run.stderr, run.stderr[-4000:]
    assert run.stdout.split() == ["ok", "pool", "ok", "worklist", "ok", "fifo", "ok", "synth"]
