"""The C-ABI library loads on a machine without a GPU and exports every symbol include/*.h declares (no compute calls)."""
import os
import re

from nexus_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
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
