"""CPU-only: the C-ABI library loads and exports every symbol include/pyvb_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(REPO, "include", "pyvb_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pyvb_[A-Za-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported_and_bound():
    from pyvb_amd import _capi
    names = _declared()
    assert len(names) >= 25
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libpyvb_hip.so does not export %s" % n
        assert n in _capi.SIGNATURES, "pyvb_amd._capi does not bind %s" % n
    assert sorted(_capi.SIGNATURES) == names


def test_error_string_and_argument_checks_without_gpu():
    from pyvb_amd import _capi
    assert _capi.lib.pyvb_version() >= 100
    h = ctypes.c_void_p()
    rc = _capi.lib.pyvb_lds_create(ctypes.byref(h), 0, 1, 1, 4, 4, 0)     # T = 1 is refused before any HIP call
    assert rc == _capi.E_ARG
    assert b"T must be" in _capi.lib.pyvb_last_error()
    rc = _capi.lib.pyvb_lds_create(ctypes.byref(h), 0, 1, 10, 129, 4, 0)        # D, K <= 128
    assert rc == _capi.E_ARG
    rc = _capi.lib.pyvb_lds_create(ctypes.byref(h), 0, 1, 10, 4, 129, _capi.NOISE_WISHART)    # the same with Wishart noise (<= 64 until round 4)
    assert rc == _capi.E_ARG


def test_library_has_no_unresolved_symbols_of_its_own():
    """A shared library links with undefined symbols silently; every pyvb_* symbol one translation unit calls must be
    defined by another (with the same linkage)."""
    import subprocess
    from pyvb_amd import _capi
    out = subprocess.run(["nm", "-D", "--undefined-only", _capi.LIB_PATH], capture_output=True, text=True).stdout
    bad = [l for l in out.splitlines() if "pyvb" in l or "launch_" in l]
    assert not bad, bad


def test_shipped_library_is_the_build_of_the_sources_beside_it():
    """The .so is not in git (it travels with the push): its embedded source hash must equal the hash of pyvb_amd/csrc as
    it stands, so a stale library cannot pass for the sources."""
    from pyvb_amd import _capi
    want = _capi.source_id()
    assert want is not None and len(want) == 32
    assert _capi.lib.pyvb_build_id().decode() == want, "rebuild: make -C pyvb_amd/csrc"
