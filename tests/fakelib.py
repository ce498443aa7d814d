"""Builds tests/_build/libswpfake.so: the C++ host layer (swarmkit_amd/csrc/swp_sched.cpp, the product source) linked
against tests/fake_swp.cpp, a scripted TEST DOUBLE of the engine ABI. CPU-only tests of the host layer use it; it is
never loaded by the product (swarmkit_amd/ does not know it exists)."""
import ctypes
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "_build", "libswpfake.so")
SRCS = [os.path.join(ROOT, "tests", "fake_swp.cpp"), os.path.join(ROOT, "swarmkit_amd", "csrc", "swp_sched.cpp")]
DEPS = SRCS + [os.path.join(ROOT, "swarmkit_amd", "csrc", "swp_json.hpp"), os.path.join(ROOT, "swarmkit_amd", "csrc", "swp_tables.hpp"), os.path.join(ROOT, "swarmkit_amd", "csrc", "swp_generic.hpp"), os.path.join(ROOT, "include", "swp.h"), os.path.join(ROOT, "include", "swp_sched.h")]


def build():
    # SWP_FAKE_SANITIZE=1: the same library under AddressSanitizer + UBSan (tests/test_sanitized_host_cpu.py runs the host-layer tests
    # against it in a child process that preloads the sanitizer runtimes)
    # SWP_FAKE_O3=1: the host layer at the product's optimisation level (csrc/Makefile: -O3) — what tools/host_layer_bench.py times; the tests
    # take -O1 (a third of the compile time)
    san = os.environ.get("SWP_FAKE_SANITIZE") == "1"
    if os.environ.get("SWP_FAKE_O3") == "1" and not san:
        return _build(OUT.replace(".so", "_O3.so"), ["-O3"])
    out = OUT.replace(".so", "_san.so") if san else OUT
    return _build(out, ["-O0", "-fsanitize=address,undefined,float-cast-overflow", "-fno-omit-frame-pointer"] if san else [])   # (-O0: a fifth of the compile time)


def _build(OUT, extra):
    if os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in DEPS):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    tmp = "%s.%d.tmp" % (OUT, os.getpid())   # parallel test workers: build privately, publish atomically
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-fPIC", "-shared"] + extra + ["-o", tmp] + SRCS, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libswpfake.so build failed:\n" + r.stdout + r.stderr)
    os.replace(tmp, OUT)
    return OUT


def take_log(engine):
    """The fake engine's call log since the last take (ids resolved to strings)."""
    fn = engine.L.swp_fake_take_log
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = ctypes.c_char_p
    return fn(engine.h).decode().splitlines()


def script_hist(engine, service, hist):
    """REPLAY mode: the next task of `service` finds no node; `hist` = the eight Pipeline counters its explanation is made of."""
    fn = engine.L.swp_fake_script_hist
    fn.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)]
    fn.restype = ctypes.c_int
    assert fn(engine.h, service.encode(), (ctypes.c_uint32 * 8)(*hist)) == 0


def script(engine, service, node, volumes=()):
    """REPLAY mode of the double: the next task of `service` that reaches it is answered with `node` ("" / None: no suitable node) and, for
    its cluster mounts, `volumes` (ids, mount order; empty: assigned without attachments)."""
    fn = engine.L.swp_fake_script
    fn.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_char_p), ctypes.c_uint32]
    fn.restype = ctypes.c_int
    arr = (ctypes.c_char_p * max(len(volumes), 1))(*[v.encode() for v in volumes])
    assert fn(engine.h, service.encode(), (node or "").encode(), arr, len(volumes)) == 0
