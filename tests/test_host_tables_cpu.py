"""CPU: the host layer's id-keyed containers (swarmkit_amd/csrc/swp_tables.hpp) against std::map / a plain vector under long random
operation sequences (tests/cxx/tables_model_test.cpp) — directly, not through the C boundary: every answer, the iteration orders, the
closing of an IdTable's holes and the growth of its index. Also once under AddressSanitizer + UBSan."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "cxx", "tables_model_test.cpp")
DEPS = [SRC, os.path.join(HERE, "..", "swarmkit_amd", "csrc", "swp_tables.hpp"), os.path.join(HERE, "..", "swarmkit_amd", "csrc", "swp_json.hpp")]


def build(name, flags):
    out = os.path.join(HERE, "_build", name)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in DEPS):
        tmp = out + ".%d.tmp" % os.getpid()
        subprocess.run(["g++", "-std=c++17", "-Wall"] + flags + ["-o", tmp, SRC], check=True)
        os.replace(tmp, out)
    return out


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_tables_match_their_models(seed):
    r = subprocess.run([build("tables_model_test", ["-O2"]), str(seed), "300000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "-> OK" in r.stderr, r.stderr[-2000:]


def test_tables_under_the_sanitizers():
    r = subprocess.run([build("tables_model_test_san", ["-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer"]), "7", "120000"], capture_output=True, text=True,
                       timeout=900, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and "-> OK" in r.stderr, r.stderr[-3000:]
