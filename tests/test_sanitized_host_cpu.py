"""CPU: the C++ host layer (csrc/swp_sched.cpp, product source) under AddressSanitizer + UndefinedBehaviorSanitizer: the event scripts of
tests/test_sched_volumes_cpu.py (volume bookkeeping, placements with attachments, freeVolumes) and a few twin scripts of
tests/test_sched_cpu.py, and the container tests of tests/test_host_json_cpu.py (malformed documents, the task table closing its holes,
the decision log over many ticks) run in a child process over a sanitized build of the scripted engine double + host layer. A finding aborts the
child; its report is the assertion message."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    return p if os.path.isabs(p) and os.path.exists(p) else None


def test_host_layer_event_scripts_under_the_sanitizers():
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes next to this gcc")
    env = dict(os.environ, SWP_FAKE_SANITIZE="1", LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", SWP_TWIN_SEEDS="3", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(HERE, "test_sched_volumes_cpu.py"),
                        os.path.join(HERE, "test_sched_cpu.py"), os.path.join(HERE, "test_host_json_cpu.py"),
                        "-k", "volume or attachments or books or start or twin or refused or coerced or survives or decisions or escape or nesting or repeated"], capture_output=True, text=True, timeout=900, env=env, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.parametrize("seed", [11, 12])
def test_malformed_documents_at_the_boundary_under_the_sanitizers(seed):
    """tools/host_fuzz.py: structurally random variants of real documents (members dropped, replaced by values of any type, text cut, bytes
    flipped) through every entry point of include/swp_sched.h — a return code every time, valid JSON whenever the layer answers, and no
    finding of AddressSanitizer / UBSan (this is how a signed overflow on a reservation of INT64_MIN and a pass-through of strings that
    are not UTF-8 were found)."""
    asan, ubsan = _runtime("libasan.so"), _runtime("libubsan.so")
    if not asan or not ubsan:
        pytest.skip("no sanitizer runtimes next to this gcc")
    env = dict(os.environ, SWP_FAKE_SANITIZE="1", LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "host_fuzz.py"), str(seed), "1500"], capture_output=True, text=True, timeout=900, env=env,
                       cwd=os.path.dirname(HERE))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "return codes" in r.stdout
