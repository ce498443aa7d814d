"""CPU: static checks over the device code the product ships (tools/check_kernels.py): every instance of the hand-scheduled matcher in
every kernel has its 64 bodies at the stride its computed jump assumes; no 1024-thread kernel creeps into the last registers of the
file, spills vector registers or grows a scratch segment unless it is on the tool's list with its reason."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shipped_kernels_pass_the_static_checks():
    sys.path.insert(0, ROOT)
    from swarmkit_amd import abi
    abi.build_library()   # (a no-op when libswp.so is up to date: the objects the tool reads are make's)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_kernels.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "instances of match_seq64 with 64 bodies" in r.stdout
