"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares (no compute calls)."""
import os
import re

import pytest

from nexus_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"^static inline .*?^}", "", text, flags=re.S | re.M)  # header-only helpers (nxhip_header_abi_stamp) are not exports
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, text)))


def test_library_exports_every_declared_symbol():
    lib = capi.lib()
    hip = _declared("nexus_hip.h", "nxhip_")
    host = sorted(_declared("nexus_host.h", "nxh_") + _declared("nexus_host.h", "nxs_"))
    assert len(hip) >= 40 and len(host) >= 10
    for name in hip + host:
        assert hasattr(lib, name), "libnexus_amd.so does not export %s" % name
    assert sorted(capi.HIP_SYMBOLS) == hip, "capi.HIP_SYMBOLS is out of date with include/nexus_hip.h"
    assert sorted(capi.HOST_SYMBOLS) == host
    assert lib.nxhip_has_gfx950_code() == 1


def test_no_device_fails_loudly_not_silently():
    """Without a GPU the product must refuse, never fall back to a CPU path."""
    import pytest

    if capi.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(capi.NexusError):
        capi.Context(16, 16)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "nexus_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower() or f in ("nx_math.h",), "%s mentions the oracle" % os.path.join(dirpath, f)


def test_a_stale_or_foreign_library_is_an_error_status_not_a_device_fault():
    """a library whose idea of the structs differs from its caller's (round 3: a variant .so driven through
    NEXUS_AMD_LIB faulted the GPU in its first launch, gpurun_out/r3_07) must be refused before anything is launched.  The
    bindings hash the size and key offsets of every struct that crosses the boundary (capi.abi_words, mirroring
    nxhip_header_abi_stamp in include/nexus_hip.h); the library compares and also checks that all of its own translation
    units were compiled with the same device-side layouts."""
    import ctypes as C

    import pytest

    lib = capi.lib()
    assert lib.nxhip_abi_stamp() == capi.abi_stamp()          # the Python mirrors describe the structs the library was built with
    assert lib.nxhip_check_library(capi.abi_stamp()) == 0      # ... and the library's translation units agree among themselves
    # a caller with another idea of ONE struct (a 64-byte material instead of 60): refused, with a message that says what to do
    words = capi.abi_words()
    words[words.index(60)] = 64
    assert lib.nxhip_check_library(capi.abi_stamp(words)) == 5  # NXHIP_ERR_ABI
    msg = lib.nxhip_last_error().decode()
    assert "ABI mismatch" in msg and "stale or foreign" in msg
    with pytest.raises(capi.NexusError, match="ABI mismatch"):
        capi.check_library(lib, stamp=capi.abi_stamp(words))
    # another API version alone is enough
    assert lib.nxhip_check_library(capi.abi_stamp([capi.API_VERSION + 1] + capi.abi_words()[1:])) == 5
    # a library without the entry point (built before round 4) is refused by the bindings as well
    class Old:
        pass
    with pytest.raises(capi.NexusError, match="predates"):
        capi.check_library(Old())
    assert C.sizeof(C.c_uint64) == 8


def test_a_release_library_keeps_the_symbols_and_refuses_the_test_hooks(tmp_path):
    """`make release` (VERDICT r5 housekeeping): the same code with NX_NO_DEBUG_HOOKS — every nxhip_debug_* symbol is still exported (the
    header and the ABI stamp do not change), each refuses with a message, and nxhip_build_info() says so.  No GPU needed: the hooks
    refuse before they look at their context."""
    import ctypes as C
    import shutil
    import subprocess

    if not shutil.which("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    out = os.path.join(ROOT, "nexus_amd", "lib", "release", "libnexus_amd.so")
    subprocess.run(["make", "-C", ROOT, "-j", "8", "release"], check=True, capture_output=True, timeout=900)
    L = C.CDLL(out)
    L.nxhip_last_error.restype = C.c_char_p
    info = L.nxhip_build_info()
    assert info & 1 and not info & 4, info
    from nexus_amd import capi
    L.nxhip_abi_stamp.restype = C.c_uint64
    assert L.nxhip_abi_stamp() == capi.lib().nxhip_abi_stamp(), "same ABI as the default build"
    hooks = [s for s in capi.HIP_SYMBOLS if s.startswith("nxhip_debug_")]
    assert len(hooks) >= 6
    for name in hooks:
        fn = getattr(L, name)  # (exported)
        fn.restype = C.c_int
        rc = fn(None, 0, 0, 0)
        assert rc != 0 and b"built without the test hooks" in L.nxhip_last_error(), name
    assert capi.lib().nxhip_build_info() & 4, "the default build has them"
