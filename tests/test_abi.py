"""CPU-side checks of the C-ABI library: it loads, exports every declared symbol,
its host-only entry points work, and argument validation fails loudly before any launch."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

import oracle
from pose_estimation_amd import _ffi

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "pe_api.h"


def declared_symbols():
    text = HEADER.read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pe_[a-z0-9_]+)\s*\(", text)))


def test_library_is_built():
    assert _ffi.LIB_PATH.exists(), "run python -m pose_estimation_amd.build"


def test_every_declared_symbol_is_exported_and_bound():
    syms = declared_symbols()
    assert len(syms) >= 10
    handle = ctypes.CDLL(str(_ffi.LIB_PATH))
    for s in syms:
        assert hasattr(handle, s), f"{s} declared in pe_api.h but not exported"
        assert s in _ffi.SIGNATURES, f"{s} has no ctypes signature in _ffi.py"
    for s in _ffi.SIGNATURES:
        assert s in syms, f"{s} bound in _ffi.py but not declared in pe_api.h"


def test_version_and_error_string():
    lib = _ffi.lib()
    assert lib.pe_version() >= 1000
    assert isinstance(lib.pe_last_error(), bytes)


@pytest.mark.parametrize("sigma,radius", [(1.0, 1), (2.0, 6), (3.0, 9), (0.75, 3), (10.0, 31)])
def test_gaussian_table_matches_oracle_bit_for_bit(sigma, radius):
    from pose_estimation_amd import gaussian_table

    got = gaussian_table(sigma, radius)
    want = oracle.gaussian_table(sigma, radius)
    assert got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert got[0] == 1.0


def test_gaussian_table_rejects_bad_arguments():
    lib = _ffi.lib()
    buf = np.zeros(64, dtype=np.float32)
    assert lib.pe_gaussian_table_f32(2.0, 32, buf.ctypes.data) == _ffi.PE_E_UNSUPPORTED
    assert b"radius" in lib.pe_last_error()
    assert lib.pe_gaussian_table_f32(0.0, 3, buf.ctypes.data) == _ffi.PE_E_INVALID
    assert lib.pe_gaussian_table_f32(2.0, 3, None) == _ffi.PE_E_INVALID
    with pytest.raises(_ffi.PeError):
        _ffi.check(lib.pe_gaussian_table_f32(-1.0, 3, buf.ctypes.data))


def test_validation_fails_before_any_launch():
    """NULL pointers and bad sizes are rejected on the host (no GPU is touched)."""
    lib = _ffi.lib()
    E = _ffi.PE_E_INVALID
    assert lib.pe_decode_f32(None, 2, 3, 8, 8, 192, 64, 8, 1, 1, None, None, None, None) == E
    assert lib.pe_decode_f32(None, 2, 3, 0, 8, 0, 0, 8, 1, 1, None, None, None, None) == E
    assert lib.pe_decode_f32(None, 0, 3, 8, 8, 192, 64, 8, 1, 1, None, None, None, None) == 0
    assert lib.pe_render_targets_f32(None, 2, 3, 8, 8, 4.0, 2.0, 6, None, 192, 64, 8, 1,
                                     None, None) == E
    assert lib.pe_render_targets_f32(None, 2, 3, 8, 8, 4.0, 2.0, 99, None, 192, 64, 8, 1,
                                     None, None) == _ffi.PE_E_UNSUPPORTED
    assert lib.pe_mse_fwd_bwd(None, 0, None, None, None, None, 2, 3, 8, 8, 4.0, 2.0, 6,
                              None, None, None, None, None) == E
    assert lib.pe_mse_fwd_bwd(None, 7, None, None, None, None, 2, 3, 8, 8, 4.0, 2.0, 6,
                              None, None, None, None, None) == E
    assert lib.pe_adam_step(None, None, None, None, None, 16, 0.9, 0.999, 1e-8, 1e-3, 1.0, 1.0,
                            None) == E
    assert lib.pe_adam_step(None, None, None, None, None, 0, 0.9, 0.999, 1e-8, 1e-3, 1.0, 1.0,
                            None) == 0
    assert lib.pe_sgd_step(None, None, None, None, -1, 0.1, 0.9, 1.0, None) == E
    assert lib.pe_mse_workspace_bytes(64, 17) >= 64 * 17 * 4


def test_ops_refuse_cpu_tensors():
    import torch

    import pose_estimation_amd as pe

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pe.decode(torch.zeros(1, 1, 8, 8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pe.render_targets(torch.zeros(1, 1, 3), 8, 8, 4.0, 2.0)
