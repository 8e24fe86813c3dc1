"""pe_mse_fwd_bwd_static (the entry point that allocates nothing) against pe_mse_fwd_bwd, both
called through the C ABI on the same device buffers: loss, per-joint losses, weights and the
gradient must agree bit for bit, on two calls in a row of which the second is smaller (the
library's device scratch still holds the first call's partial sums), and a size beyond the
static scratch must be refused before anything is launched."""
import numpy as np
import pytest
import torch

import oracle
from pose_estimation_amd import _ffi, synthetic

pytestmark = pytest.mark.gpu

H, W = 64, 48


def _ptr(t):
    return None if t is None else t.data_ptr()


def _call_both(B, K, dtype, fused, masked, seed):
    dev = torch.device("cuda", 0)
    lib = _ffi.lib()
    rng = np.random.default_rng(seed)
    pred_np = rng.normal(size=(B, K, H, W)).astype(np.float32)
    j_np = synthetic.joints(B, K, 256, 192, seed=seed + 1)
    pred = torch.from_numpy(pred_np).to(dev).to(dtype)
    joints = torch.from_numpy(j_np).to(dev)
    t_np, w_np = oracle.render_targets(j_np, H, W, 4.0, 2.0, 6)
    target = None if fused else torch.from_numpy(t_np).to(dev)
    mask = torch.from_numpy(synthetic.ignore_mask(B, H, W, seed=seed + 2)).to(dev) if masked else None
    dt = _ffi.PE_DTYPE_F32 if dtype == torch.float32 else _ffi.PE_DTYPE_BF16
    stream = torch.cuda.current_stream(dev).cuda_stream

    results = []
    for static in (False, True):
        loss = torch.full((1,), float("nan"), device=dev)
        per_joint = torch.full((K,), float("nan"), device=dev)
        grad = torch.full_like(pred, float("nan"))
        weight = torch.from_numpy(w_np).to(dev) if not fused else torch.full((B, K), -1.0, device=dev)
        args = [pred.data_ptr(), dt, _ptr(target), joints.data_ptr() if fused else None,
                weight.data_ptr(), _ptr(mask), B, K, H, W, 4.0, 2.0, 6, loss.data_ptr(),
                per_joint.data_ptr(), grad.data_ptr()]
        if static:
            _ffi.check(lib.pe_mse_fwd_bwd_static(*args, stream))
        else:
            ws = torch.empty(lib.pe_mse_workspace_bytes(B, K), dtype=torch.uint8, device=dev)
            _ffi.check(lib.pe_mse_fwd_bwd(*args, ws.data_ptr(), stream))
        torch.cuda.synchronize(dev)
        results.append((loss.cpu(), per_joint.cpu(), grad.float().cpu(), weight.cpu()))
    return results, (pred_np, t_np, w_np, mask)


def _assert_identical(results):
    (l0, p0, g0, w0), (l1, p1, g1, w1) = results
    assert torch.isfinite(l1).all() and torch.isfinite(g1).all()
    assert torch.equal(l0.view(torch.int32), l1.view(torch.int32))
    assert torch.equal(p0.view(torch.int32), p1.view(torch.int32))
    assert torch.equal(g0.view(torch.int32), g1.view(torch.int32))
    assert torch.equal(w0, w1)


@pytest.mark.parametrize("dtype,fused,masked", [
    (torch.float32, False, True),
    (torch.float32, True, False),
    (torch.bfloat16, True, True),
])
def test_static_entry_point_equals_the_workspace_one_on_two_calls_in_a_row(dtype, fused, masked):
    big, _ = _call_both(48, 17, dtype, fused, masked, seed=11)
    _assert_identical(big)
    # second call, smaller: maps 5*17 .. 48*17 of the scratch still hold the first call's sums
    small, (pred_np, t_np, w_np, mask) = _call_both(5, 17, dtype, fused, masked, seed=23)
    _assert_identical(small)
    if dtype == torch.float32:
        m_np = None if mask is None else mask.cpu().numpy()
        oloss, opj, og = oracle.mse_fwd_bwd(pred_np, t_np, w_np, m_np)
        loss, pj, g, _ = small[1]
        assert np.array_equal(g.numpy(), og)
        assert abs(float(loss) - float(oloss)) <= 2e-6 * max(1.0, abs(float(oloss)))
        assert np.allclose(pj.numpy(), opj, rtol=2e-6, atol=0)


def test_static_entry_point_at_its_maximum_and_beyond():
    dev = torch.device("cuda", 0)
    lib = _ffi.lib()
    K = 16
    B = _ffi.PE_MSE_STATIC_MAX_MAPS // K          # exactly the maximum number of maps
    h, w = 8, 8
    pred = torch.randn(B, K, h, w, device=dev)
    target = torch.zeros(B, K, h, w, device=dev)
    out = []
    for static in (False, True):
        loss = torch.empty(1, device=dev)
        grad = torch.empty_like(pred)
        args = [pred.data_ptr(), _ffi.PE_DTYPE_F32, target.data_ptr(), None, None, None, B, K, h, w,
                4.0, 2.0, 6, loss.data_ptr(), None, grad.data_ptr()]
        if static:
            _ffi.check(lib.pe_mse_fwd_bwd_static(*args, None))
        else:
            ws = torch.empty(lib.pe_mse_workspace_bytes(B, K), dtype=torch.uint8, device=dev)
            _ffi.check(lib.pe_mse_fwd_bwd(*args, ws.data_ptr(), None))
        torch.cuda.synchronize(dev)
        out.append((loss.cpu(), grad.cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    # one image more does not fit: refused on the host, nothing launched, loss untouched
    loss = torch.full((1,), 123.0, device=dev)
    rc = lib.pe_mse_fwd_bwd_static(pred.data_ptr(), _ffi.PE_DTYPE_F32, target.data_ptr(), None,
                                   None, None, B + 1, K, h, w, 4.0, 2.0, 6, loss.data_ptr(), None,
                                   None, None)
    assert rc == _ffi.PE_E_UNSUPPORTED
    assert b"static scratch" in lib.pe_last_error()
    torch.cuda.synchronize(dev)
    assert float(loss) == 123.0
