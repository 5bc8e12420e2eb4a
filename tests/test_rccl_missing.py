"""when no RCCL library can be loaded the multi-GPU entry points return an error status with a message (they used to
dereference a null dlerror() string).  Runs in a child process with NX_RCCL_LIB pointing nowhere; no GPU needed."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from nexus_amd import capi
L = capi.lib()
buf = (C.c_ubyte * 128)()
rc = L.nxhip_mgpu_unique_id(buf)
msg = L.nxhip_last_error().decode()
print("RC", rc)
print("MSG", msg)
"""


def test_missing_rccl_is_an_error_status_not_a_crash():
    env = dict(os.environ, NX_RCCL_LIB="/nonexistent/librccl.so.1")
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RC 1" in r.stdout, r.stdout
    assert "cannot load RCCL" in r.stdout and "NX_RCCL_LIB" in r.stdout and "/nonexistent/librccl.so.1" in r.stdout
