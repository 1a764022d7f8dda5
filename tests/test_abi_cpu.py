"""CPU-side checks of the C-ABI library: it builds for gfx950, loads without a GPU, exports every
symbol include/dtp.h declares, and its host-only logic (DDIM tables) matches the reference fixtures.
No compute entry point is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from diffusiontexturepainting_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


def test_header_and_binding_agree(lib):
    from diffusiontexturepainting_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "dtp.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(dtp_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), f"libdtp.so does not export {name}"
    assert lib.dtp_abi_version() == 3


def test_ddim_tables_match_reference_fixture(lib, golden_dir):
    g = np.load(os.path.join(golden_dir, "ddim.npz"))
    for n in (4, 8, 20, 50):
        ts = (C.c_int64 * n)()
        al = (C.c_float * n)()
        fin = C.c_float()
        assert lib.dtp_ddim_tables(n, ts, al, C.byref(fin)) == 0
        assert list(ts) == g[f"timesteps_{n}"].tolist()
        np.testing.assert_allclose(np.array(al, dtype=np.float32), g[f"alphas_{n}"], rtol=2e-7, atol=0)
        assert abs(fin.value - float(g[f"final_alpha_{n}"])) <= 1e-7
    assert lib.dtp_ddim_tables(0, None, None, None) != 0
    assert b"steps" in lib.dtp_last_error()
    # steps = 1000 would gather alphas_cumprod[1000] (IndexError in the reference): rejected, 999 is the largest schedule
    assert lib.dtp_ddim_tables(1000, None, None, None) != 0
    ts = (C.c_int64 * 999)()
    assert lib.dtp_ddim_tables(999, ts, None, None) == 0 and max(ts) == 999 and min(ts) == 1


def test_missing_library_is_loud(monkeypatch):
    from diffusiontexturepainting_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libdtp.so")
    with pytest.raises(_lib.DtpError):
        _lib.load()


def test_inpainter_requires_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from diffusiontexturepainting_amd._lib import DtpError
    from diffusiontexturepainting_amd.inpainter import MI355ConditionalInpainter
    with pytest.raises(DtpError):
        MI355ConditionalInpainter(256)
