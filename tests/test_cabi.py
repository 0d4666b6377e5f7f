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


def test_no_raw_barrier_is_reached_with_lds_reads_in_flight():
    """The LDS-DMA ring kernels hand slots back to the DMA behind raw s_barrier instructions: every fragment read a wave issued must
    have returned when it arrives there (Weak #2).  tools/audit_barrier_lds.py walks the disassembly of the built
    objects; round 5's conv_rs.o had 885 such barriers (the compiler sank an MFMA and its lgkmcnt wait below the barrier), the
    explicit waits of round 6 leave none."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "deep_continuous_fusion_for_multi-sensor_3d_object_detection_amd", "csrc")
    objs = [os.path.join(csrc, f) for f in ("conv_rs.o", "conv_chain.o", "conv_wgs.o", "conv_wg1.o", "conv_sp.o", "conv_lc.o", "conv_wgv.o", "conv.o")]
    pkg("_hip").build()                                   # (no-op when the library is up to date)
    assert all(os.path.exists(o) for o in objs)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit_barrier_lds.py")] + objs, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert p.stdout.count(" 0 with LDS reads possibly in flight") == len(objs), p.stdout
