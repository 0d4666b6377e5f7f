"""CPU suite: the C-ABI library builds, loads and exports every symbol include/dcf_hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re

from _util import M64, ROOT, pkg, rand32


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dcf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dcf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    H = pkg("_hip")
    if not os.path.exists(H.LIB_PATH):
        H.build()
    lib = ctypes.CDLL(H.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_table_matches_header():
    H = pkg("_hip")
    assert sorted(H.SIGNATURES.keys()) == declared_symbols()


def test_version_and_error_text_without_gpu():
    H = pkg("_hip")
    L = H.lib()
    assert L.dcf_version() >= 100
    assert isinstance(L.dcf_last_error(), bytes)
    assert ctypes.sizeof(H.ConvParam) == 10 * 8 + 8 * 4


def test_argument_validation_is_host_side():
    """Bad arguments are rejected before anything touches a device."""
    H = pkg("_hip")
    L = H.lib()
    rc = L.dcf_knn_bev(None, None, 0, 9, 4, 4, 2, 1.0, 0.0, 1.0, 0.0, -1.0, None, None, None)
    assert rc == -1 and b"dcf_knn_bev" in L.dcf_last_error()
    rc = L.dcf_conv2d_fwd(0, 1, 1, None, None, 1, 1, 8, 8, 7, 8, 8, 32, 3, 3, 1, 1, 0, None)
    assert rc == -1 and b"Cin" in L.dcf_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    H = pkg("_hip")
    monkeypatch.setattr(H, "_LIB", None)
    monkeypatch.setattr(H, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        H.lib()
        raise AssertionError("expected DcfError")
    except H.DcfError as e:
        assert "no CPU fallback" in str(e)


def test_sampler_hash_restatement_equals_the_library():
    """dcf_loss_sample_rand (host-callable; the device sampler of loss_sampling: device uses the same function) against the Python
    restatement the GPU tests build their expected lists from."""
    lib = pkg("_hip").lib()
    for args in ((0, 0, 1, 0, 0), (12345678901234567, 3, 2, 511, 7), (M64, 15, 1, 1023, 0), (42, 1, 2, 128, 1000), (2 ** 63 + 5, 7, 2, 99999, 3)):
        assert lib.dcf_loss_sample_rand(*args) == rand32(*args)
