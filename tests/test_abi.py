"""The C ABI library loads on a box without a GPU and exports every symbol include/mmtta.h declares;
the ctypes table of the Python binding covers exactly that set."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mmtta.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mmtta_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from multimodal_tta_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/mmtta.h but not exported by libmmtta.so"
    assert lib.mmtta_abi_version() == 2


def test_binding_table_matches_header():
    from multimodal_tta_amd import _lib
    assert sorted(_lib.exported_names()) == declared_functions()
    _lib.load()


def test_argument_validation_without_a_gpu():
    """Entry points reject bad arguments before touching the device."""
    from multimodal_tta_amd import _lib
    lib = _lib.load()
    d = _lib.ConvDesc(0, 5, 1, 4, 4, 0)     # ksize 5: unsupported
    assert lib.mmtta_conv_packed_bytes(ctypes.byref(d)) == -1
    assert b"ksize" in lib.mmtta_last_error()
    d = _lib.ConvDesc(0, 3, 1, 4, 32, 0)
    assert lib.mmtta_conv_packed_bytes(ctypes.byref(d)) == 27 * 32 * 32 * 4
    assert lib.mmtta_copy_strided(None, None, None) == -1
