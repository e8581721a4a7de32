"""The C-ABI library loads without a GPU and exports every symbol include/*.h declares."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b(dna[a-z]*_[a-z0-9_]+)\s*\(", text)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(built):
    lib = C.CDLL(os.path.join(ROOT, "dynadjust_amd", "libdnagpu.so"))
    from dynadjust_amd import _lib
    for header, listed in (("dnagpu.h", _lib.EXPORTED_DNAGPU), ("dnaadjust_c.h", _lib.EXPORTED_DNAADJ)):
        declared = declared_functions(header)
        assert declared, header
        for name in declared:
            assert hasattr(lib, name), f"{name} declared in include/{header} but not exported"
        assert sorted(listed) == declared, (header, set(listed) ^ set(declared))


def test_no_device_is_reported_loudly(built):
    """without a GPU the product refuses to run: no CPU fallback behind the boundary"""
    if built.dnagpu_device_count() > 0:
        import pytest
        pytest.skip("a GPU is visible")
    h = C.c_void_p()
    assert built.dnagpu_create(0, C.byref(h)) == -5   # DNAGPU_ENODEVICE
    from dynadjust_amd import adjust
    import pytest
    a = adjust.DnaAdjust()
    p = adjust.ProjectSettings("tiny_net", os.path.join(ROOT, "tests", "golden"))
    with pytest.raises(adjust.NetAdjustException) as e:
        a.PrepareAdjustment(p)
    assert "no MI355X device" in str(e.value)
    a.close()


def test_product_does_not_reference_the_oracle():
    """the oracle is test infrastructure: nothing under dynadjust_amd/ may include or load it"""
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "dynadjust_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) and f != "smoke.py":
                t = open(os.path.join(dp, f), errors="replace").read()
                if "liboracle" in t or "dna_oracle" in t or "tests.oracle" in t or "from tests" in t:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
