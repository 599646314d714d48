"""The C-ABI library builds, loads without a GPU, and exports every symbol include/psvo_hip.h
declares; argument validation returns status codes before anything touches a device."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "psvo_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(psvo_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound(built_lib):
    from psvo_amd import _lib
    lib = ctypes.CDLL(built_lib)
    names = _declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), "libpsvo_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "psvo_amd/_lib.py does not bind %s" % n
    for n in _lib.SIGNATURES:
        assert n in names, "%s is bound but not declared in include/psvo_hip.h" % n


def test_version_and_status_strings(built_lib):
    from psvo_amd import _lib
    lib = _lib.load()
    assert lib.psvo_abi_version() == 4      # (4: psvo_desc.layers, psvo_mlp.Wh / bh, psvo_mlp2_wgrad)
    assert lib.psvo_status_string(0) == b"ok"
    assert b"unsupported" in lib.psvo_status_string(_lib.PSVO_ERR_UNSUPPORTED)
    assert lib.psvo_filter_acc_size(2, 1) == 21 and lib.psvo_bsim_acc_size(3, 2) == 23
    import ctypes
    from psvo_amd import ops
    blocks = lambda B, T, N, M, H, Dx, Dy=1: lib.psvo_bsim_blocks(ctypes.byref(ops.make_desc(B, T, N, M, Dx, Dy, H)))
    assert lib.psvo_get_tuning(_lib.PSVO_TUNE_BSIM_BWD) == -1
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, 0) == 0           # v1 geometry: lane = (chain, half, m)
    assert blocks(32, 200, 128, 16, 32, 2) == 16 and blocks(32, 200, 128, 16, 32, 3) == 8 and blocks(2, 6, 8, 4, 16, 2) == 1
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, 1) == 0           # v2: 8 chains per 256-thread workgroup at M = 16
    assert blocks(32, 200, 128, 16, 32, 2) == 16 and blocks(32, 200, 128, 16, 32, 3) == 16 and blocks(2, 6, 8, 4, 16, 2) == 1
    assert blocks(64, 1000, 512, 16, 32, 4) == 32                         # (4 GiB per array: v1 geometry)
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, 7) == _lib.PSVO_ERR_INVALID
    assert lib.psvo_set_tuning(_lib.PSVO_TUNE_BSIM_BWD, -1) == 0
    assert lib.psvo_mlp_wgrad_blocks(10) == 1 and lib.psvo_mlp_wgrad_blocks(10 ** 9) == 1024


def test_invalid_arguments_are_rejected_without_a_device(built_lib):
    from psvo_amd import _lib
    lib = _lib.load()
    d = _lib.psvo_desc()
    d.B, d.T, d.N, d.M, d.Dx, d.Dy, d.H = 2, 4, 8, 4, 2, 1, 32
    nul = [None] * 20
    st = lib.psvo_filter_forward(ctypes.byref(d), None, None, None, *nul, None)
    assert st == _lib.PSVO_ERR_INVALID
    st = lib.psvo_elbo_filter(ctypes.byref(d), None, None, None)
    assert st == _lib.PSVO_ERR_INVALID
    with pytest.raises(ValueError):
        _lib.check(_lib.PSVO_ERR_UNSUPPORTED, "x")
    with pytest.raises(_lib.PsvoHipError):
        _lib.check(_lib.PSVO_ERR_HIP, "x")


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from psvo_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.PsvoHipError):
        _lib.load()


@pytest.mark.timeout(900)
def test_no_register_copies_ahead_of_an_exec_restore(built_lib):
    """ISA check of every kernel of the library (tools/exec_restore_check.py): in the kernels compiled with hipcc's default
    (greedy, live-range-splitting) register allocator no VGPR <-> AGPR move or scratch access sits between the skip target of a
    divergent `if` and the `s_or_b64 exec` that ends it.  hipcc 7.2 placed the copies of a live-range split there under
    register pressure; they then run under the `if`'s partial EXEC mask and the lanes that skipped the `if` read a stale copy
    back later -- a two-hidden-layer PSVOwR kernel lost part of one scale gradient that way.  The
    two-hidden-layer units (namespace psvo::l2) are compiled with the basic allocator instead, which never splits: there a
    store after each definition and a reload before each use are expected inside divergent regions, and lane-exact.
    Parity tests sample the instantiated (Dx, Dy, H, M, threads) combinations; this covers all of them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("exec_restore_check", os.path.join(ROOT, "tools", "exec_restore_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    n, rows = chk.scan([built_lib], verbose=False)
    assert n > 1000, "expected the whole kernel set, scanned %d" % n
    class_a = [(k, a) for _, k, a, _ in rows if a and "4psvo2l2" not in k]
    assert not class_a, "register copies ahead of an EXEC restore in: %s" % class_a[:10]
    from psvo_amd import build
    assert build.L2_FLAGS == ["-mllvm", "-vgpr-regalloc=basic"]     # (what the exemption of psvo::l2 rests on)
