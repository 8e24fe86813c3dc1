"""GPU parity of the backward kernels (backward-data on the implicit-GEMM engine, the
backward-weight GEMM, batch-norm forward/backward, max-pool backward, loss-gradient hand-over)
against torch-CPU float32/float64 autograd of the same layers.

Integer-operand cases are exact in float32 whatever the summation order, so they are compared
bit for bit (after the one bf16 rounding of a bf16 output); random cases use
|got - ref| <= 4e-3 * max|ref| for bf16 outputs and 1e-4 * max|ref| for float32 outputs.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_oracle as to
from pose_estimation_amd import nn as pnn

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def ints(shape, g, lo, hi):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def rnd(shape, g, s=1.0):
    return to.bf16r(torch.randn(shape, generator=g) * s)


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def close(got, ref, tol):
    err = (got - ref).abs().max().item()
    m = ref.abs().max().item()
    assert err <= tol * m + 1e-7, f"max err {err} vs max|ref| {m}"


def conv_wt(w_oihw):
    """[Cout,Cin,k,k] -> device w_t [Cin, k*k, Cout] through the library's own transpose."""
    wp = pnn.pack_conv_weight(w_oihw).cuda()
    cout, k, _, cin = wp.shape
    out = torch.zeros(cin, k * k, cout, dtype=BF16, device="cuda")
    return pnn.transpose_weights(wp.view(cout, k * k, cin), out)


BWD_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 12, 64, 64, 1, 1, 0),
    (2, 16, 12, 256, 64, 1, 1, 0),
    (2, 16, 12, 64, 128, 3, 1, 1),
    (3, 16, 12, 128, 128, 3, 2, 1),
    (2, 16, 12, 256, 512, 1, 2, 0),
    (1, 8, 6, 512, 256, 3, 1, 1),
    (2, 10, 14, 64, 192, 3, 2, 1),
    (2, 16, 12, 96, 24, 1, 1, 0),      # channel counts that are multiples of 8, not of 64
    (2, 16, 12, 24, 144, 1, 1, 0),
    (3, 10, 10, 160, 960, 1, 1, 0),
    (2, 9, 7, 96, 40, 3, 1, 1),
    (2, 12, 10, 144, 72, 3, 2, 1),
]


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_bwd_data(case):
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(17 + sum(case))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    for integer in (True, False):
        if integer:
            dy = ints((B, Ho, Wo, Cout), g, -2, 2)
            w = ints((Cout, Cin, k, k), g, -1, 1)
        else:
            dy = rnd((B, Ho, Wo, Cout), g)
            w = rnd((Cout, Cin, k, k), g, (1.0 / (Cout * k * k)) ** 0.5)
        ref = nhwc(torch.nn.grad.conv2d_input((B, Cin, H, W), w, nchw(dy), stride=s, padding=p))
        got = pnn.conv2d_bwd_data(dy.to(BF16).cuda(), conv_wt(w), (H, W), k, s, p).float().cpu()
        if integer:
            assert torch.equal(got, to.bf16r(ref))
        else:
            close(got, ref, 4e-3)
    # with a residual (only where every input parity has a tap)
    if not (k == 1 and s == 2):
        res = rnd((B, H, W, Cin), g)
        got = pnn.conv2d_bwd_data(dy.to(BF16).cuda(), conv_wt(w), (H, W), k, s, p,
                                  residual=res.to(BF16).cuda()).float().cpu()
        close(got, ref + res, 4e-3)


@pytest.mark.parametrize("case", BWD_CASES)
def test_conv_bwd_weight(case):
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(23 + sum(case))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    for integer in (True, False):
        if integer:
            x = ints((B, H, W, Cin), g, -2, 2)
            dy = ints((B, Ho, Wo, Cout), g, -1, 1)
        else:
            x = rnd((B, H, W, Cin), g)
            dy = rnd((B, Ho, Wo, Cout), g)
        ref = torch.nn.grad.conv2d_weight(nchw(x), (Cout, Cin, k, k), nchw(dy), stride=s, padding=p)
        ref = ref.permute(0, 2, 3, 1).contiguous()  # packed layout [Cout,k,k,Cin]
        dw = torch.zeros(Cout, k, k, Cin, device="cuda")
        pnn.conv2d_bwd_weight(x.to(BF16).cuda(), dy.to(BF16).cuda(), dw, k, s, p)
        if integer:
            assert torch.equal(dw.cpu(), ref)
        else:
            close(dw.cpu(), ref, 1e-4)
    # accumulation: a second call adds on top
    pnn.conv2d_bwd_weight(x.to(BF16).cuda(), dy.to(BF16).cuda(), dw, k, s, p)
    close(dw.cpu(), 2 * ref, 1e-4)


def test_conv_bwd_weight_batched_equals_the_single_calls():
    """Every case of the list above (1x1 / 3x3 / strided / narrow inputs) as jobs of batched
    launches: integer operands make every partial sum exact, so the batched result must equal
    torch's weight gradient bit for bit whatever the order of the pixel splits; with random
    operands it must match the single calls to float32 rounding.  A second batched call adds on
    top, and a batch of more than PE_WGRAD_MAX_BATCH jobs is cut into several launches."""
    g = torch.Generator().manual_seed(77)
    jobs, refs, singles = [], [], []
    for integer in (True, False):
        for case in BWD_CASES + BWD_CASES[:3]:
            B, H, W, Cin, Cout, k, s, p = case
            Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
            if integer:
                x, dy = ints((B, H, W, Cin), g, -2, 2), ints((B, Ho, Wo, Cout), g, -1, 1)
            else:
                x, dy = rnd((B, H, W, Cin), g), rnd((B, Ho, Wo, Cout), g)
            ref = torch.nn.grad.conv2d_weight(nchw(x), (Cout, Cin, k, k), nchw(dy), stride=s, padding=p)
            refs.append((integer, ref.permute(0, 2, 3, 1).contiguous()))
            xg, dyg = x.to(BF16).cuda(), dy.to(BF16).cuda()
            jobs.append((xg, dyg, torch.zeros(Cout, k, k, Cin, device="cuda"), k, s, p, None))
            one = torch.zeros(Cout, k, k, Cin, device="cuda")
            pnn.conv2d_bwd_weight(xg, dyg, one, k, s, p)
            singles.append(one)
    assert len(jobs) > pnn.PE_WGRAD_MAX_BATCH
    pnn.conv2d_bwd_weight_batched(jobs)
    for (integer, ref), job, one in zip(refs, jobs, singles):
        if integer:
            assert torch.equal(job[2].cpu(), ref)
        else:
            close(job[2].cpu(), ref, 1e-4)
            close(job[2].cpu(), one.cpu(), 1e-5)
    pnn.conv2d_bwd_weight_batched(jobs[:3])
    for (integer, ref), job in list(zip(refs, jobs))[:3]:
        assert torch.equal(job[2].cpu(), 2 * ref)
    with pytest.raises(ValueError, match="packed layout"):
        pnn.conv2d_bwd_weight_batched([jobs[0][:2] + (torch.zeros(3, device="cuda"),) + jobs[0][3:]])


def test_bwd_weight_large_reduction_and_padded_dy():
    """Final layer shape: 256 -> 17 channels with dy zero-padded to 64, M = B*64*48 pixels."""
    g = torch.Generator().manual_seed(31)
    B, H, W, Cin, K, Kpad = 4, 64, 48, 256, 17, 64
    x = ints((B, H, W, Cin), g, -2, 2)
    gr = ints((B, K, H, W), g, -1, 1)
    dbias = torch.zeros(K, device="cuda")
    dyp = pnn.final_grad_prepare(gr.cuda(), Kpad, dbias)
    assert dyp.shape == (B, H, W, Kpad)
    assert torch.equal(dyp[..., :K].float().cpu(), gr.permute(0, 2, 3, 1))
    assert not dyp[..., K:].any()
    assert torch.equal(dbias.cpu(), gr.sum(dim=(0, 2, 3)))
    dw = torch.zeros(K, 1, 1, Cin, device="cuda")
    pnn.conv2d_bwd_weight(x.to(BF16).cuda(), dyp, dw, 1, 1, 0, cout=K)
    ref = torch.nn.grad.conv2d_weight(nchw(x), (K, Cin, 1, 1), gr, stride=1, padding=0)
    assert torch.equal(dw.cpu().view(K, Cin), ref.view(K, Cin))
    # backward-data through the padded gradient: w_t [Cin][1][Kpad] with zero columns past K
    w = ints((K, Cin, 1, 1), g, -1, 1)
    wt = torch.zeros(Cin, 1, Kpad, dtype=BF16, device="cuda")
    pnn.transpose_weights(pnn.pack_conv_weight(w).cuda().view(K, 1, Cin), wt)
    dx = pnn.conv2d_bwd_data(dyp, wt, (H, W), 1, 1, 0)
    refx = nhwc(torch.nn.grad.conv2d_input((B, Cin, H, W), w, gr, stride=1, padding=0))
    assert torch.equal(dx.float().cpu(), to.bf16r(refx))


@pytest.mark.parametrize("case", [(2, 8, 6, 64, 64), (2, 8, 6, 512, 256), (1, 16, 12, 256, 256)])
def test_deconv_backward(case):
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(37 + Cin)
    x = ints((B, H, W, Cin), g, -2, 2)
    w = ints((Cin, Cout, 4, 4), g, -1, 1)
    dy = ints((B, 2 * H, 2 * W, Cout), g, -1, 1)
    xt = nchw(x).requires_grad_(True)
    wt_ = w.clone().requires_grad_(True)
    y = F.conv_transpose2d(xt, wt_, None, stride=2, padding=1)
    y.backward(nchw(dy))
    wp = pnn.pack_deconv_weight(w).cuda()                       # [4,Cout,4,Cin]
    w_t = torch.zeros(Cin, 16, Cout, dtype=BF16, device="cuda")
    for ph in range(4):
        pnn.transpose_weights(wp[ph], w_t, t0=4 * ph)
    dx = pnn.deconv4x4s2_bwd_data(dy.to(BF16).cuda(), w_t)
    assert torch.equal(dx.float().cpu(), to.bf16r(nhwc(xt.grad)))
    dw = torch.zeros(4, Cout, 4, Cin, device="cuda")
    pnn.deconv4x4s2_bwd_weight(x.to(BF16).cuda(), dy.to(BF16).cuda(), dw)
    assert torch.equal(dw.cpu(), pnn.pack_deconv_weight(wt_.grad).float())


def test_stem_bwd_weight():
    g = torch.Generator().manual_seed(41)
    B, H, W = 2, 32, 24
    x3 = ints((B, H, W, 3), g, -2, 2)
    dy = ints((B, H // 2, W // 2, 64), g, -1, 1)
    ref = torch.nn.grad.conv2d_weight(nchw(x3), (64, 3, 7, 7), nchw(dy), stride=2, padding=3)
    xs = torch.from_numpy(to.space_to_depth16(x3.numpy())).to(BF16).cuda()
    dw = torch.zeros(64, 4, 64, device="cuda")
    pnn.stem7x7s2_bwd_weight(xs, dy.to(BF16).cuda(), dw)
    mask = pnn.pack_stem_weight(torch.ones(64, 3, 7, 7)).float()
    got = dw.cpu() * mask  # structural zeros of the packing carry no parameter
    assert torch.equal(got, pnn.pack_stem_weight(ref).float())


def test_transpose_weights():
    g = torch.Generator().manual_seed(43)
    for (A, T, C) in [(64, 9, 128), (17, 1, 256), (100, 4, 72)]:
        w = torch.randn(A, T, C, generator=g).to(BF16)
        out = torch.zeros(C, T + 1, A + 3, dtype=BF16, device="cuda")
        pnn.transpose_weights(w.cuda(), out, t0=1)
        ref = torch.zeros(C, T + 1, A + 3, dtype=BF16)
        ref[:, 1:, :A] = w.permute(2, 1, 0)
        assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("shape", [(4, 16, 12, 64), (2, 8, 6, 2048), (3, 10, 7, 256), (64, 64, 48, 64)])
def test_batch_norm_forward_backward(shape):
    B, H, W, C = shape
    g = torch.Generator().manual_seed(47 + C)
    z = to.bf16r(rnd(shape, g) * 2 + 0.5)
    res = rnd(shape, g)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.2
    da = rnd(shape, g)
    rm0, rv0 = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    M = B * H * W
    # float64 reference through autograd
    zd = nchw(z).double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    rm, rv = rm0.double().clone(), rv0.double().clone()
    y = F.batch_norm(zd, rm, rv, gd, bd, training=True, momentum=0.1, eps=1e-5)
    a_ref = torch.relu(y + nchw(res).double())
    # the kernels mask with the bf16-rounded output they themselves saved
    ws = pnn.bn_workspace(M, C, "cuda")
    rmg, rvg = rm0.cuda(), rv0.cuda()
    zg = z.to(BF16).cuda()
    mean, rstd, scale, shift = pnn.bn_stats(zg, gamma.cuda(), beta.cuda(), 1e-5, 0.1, rmg, rvg, ws)
    a = pnn.bn_apply(zg, scale, shift, res.to(BF16).cuda(), relu=True)
    close(a.float().cpu(), nhwc(a_ref).float(), 4e-3)
    assert torch.allclose(mean.cpu(), nchw(z).double().mean(dim=(0, 2, 3)).float(), atol=1e-5)
    assert torch.allclose(rmg.cpu(), rm.float(), atol=1e-5)
    assert torch.allclose(rvg.cpu(), rv.float(), rtol=1e-4, atol=1e-5)
    mask = (a.float().cpu() > 0).double()
    (y * nchw(mask) * nchw(da).double()).sum().backward()  # == relu backward with that mask
    dgamma = torch.empty(C, device="cuda")
    dbeta = torch.empty(C, device="cuda")
    dz, gm = pnn.bn_bwd(da.to(BF16).cuda(), a, zg, gamma.cuda(), mean, rstd, dgamma, dbeta, ws,
                        want_masked=True)
    close(dz.float().cpu(), nhwc(zd.grad).float(), 6e-3)
    assert torch.allclose(dgamma.cpu(), gd.grad.float(), rtol=2e-3, atol=2e-3 * gd.grad.abs().max().item())
    assert torch.allclose(dbeta.cpu(), bd.grad.float(), rtol=2e-3, atol=2e-3 * bd.grad.abs().max().item())
    assert torch.equal(gm.float().cpu(), (da * mask.float()))
    # run-to-run bit stability of the ordered reductions
    dz2, _ = pnn.bn_bwd(da.to(BF16).cuda(), a, zg, gamma.cuda(), mean, rstd, dgamma, dbeta, ws)
    assert torch.equal(dz, dz2)


@pytest.mark.parametrize("shape", [(3, 10, 10, 96), (2, 12, 8, 24), (2, 6, 5, 1000)])
def test_batch_norm_relu6_any_channel_count(shape):
    """ReLU6 masks (0 < a < 6) and channel counts that are multiples of 8 but not 8 * 2^n."""
    B, H, W, C = shape
    g = torch.Generator().manual_seed(53 + C)
    z = to.bf16r(rnd(shape, g) * 3 + 2.0)
    gamma = torch.rand(C, generator=g) * 2 + 1.0
    beta = torch.randn(C, generator=g) + 2.0
    da = rnd(shape, g)
    M = B * H * W
    zd = nchw(z).double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    y = F.batch_norm(zd, None, None, gd, bd, training=True, eps=1e-5)
    a_ref = torch.clamp(y, 0.0, 6.0)
    ws = pnn.bn_workspace(M, C, "cuda")
    zg = z.to(BF16).cuda()
    mean, rstd, scale, shift = pnn.bn_stats(zg, gamma.cuda(), beta.cuda(), 1e-5, 0.1, None, None, ws)
    a = pnn.bn_apply(zg, scale, shift, None, relu=2)
    close(a.float().cpu(), nhwc(a_ref).float(), 4e-3)
    af = a.float().cpu()
    assert af.max().item() == 6.0 and af.min().item() == 0.0   # both clamps are exercised
    mask = ((af > 0) & (af < 6)).double()
    (y * nchw(mask) * nchw(da).double()).sum().backward()
    dgamma = torch.empty(C, device="cuda")
    dbeta = torch.empty(C, device="cuda")
    dz, gm = pnn.bn_bwd(da.to(BF16).cuda(), a, zg, gamma.cuda(), mean, rstd, dgamma, dbeta, ws,
                        want_masked=True, relu=2)
    close(dz.float().cpu(), nhwc(zd.grad).float(), 6e-3)
    assert torch.allclose(dgamma.cpu(), gd.grad.float(), rtol=2e-3, atol=2e-3 * gd.grad.abs().max().item())
    assert torch.allclose(dbeta.cpu(), bd.grad.float(), rtol=2e-3, atol=2e-3 * bd.grad.abs().max().item())
    assert torch.equal(gm.float().cpu(), (da * mask.float()))
    # the mask rebuilt from z (no read of the stored activation) gives the same bits, for ReLU6
    # and for ReLU
    for relu in (2, 1):
        a_r = pnn.bn_apply(zg, scale, shift, None, relu=relu)
        dg1, db1 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dg2, db2 = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dz1, g1 = pnn.bn_bwd(da.to(BF16).cuda(), a_r, zg, gamma.cuda(), mean, rstd, dg1, db1, ws,
                             want_masked=True, relu=relu)
        dz1 = dz1.clone()
        dz2, g2 = pnn.bn_bwd(da.to(BF16).cuda(), None, zg, gamma.cuda(), mean, rstd, dg2, db2, ws,
                             want_masked=True, relu=relu, act_scale=scale, act_shift=shift)
        assert torch.equal(dz1, dz2) and torch.equal(g1, g2)
        assert torch.equal(dg1, dg2) and torch.equal(db1, db2)
        assert (g2 == 0).any() and (g2 != 0).any()


@pytest.mark.parametrize("cfg", [
    # kind, B, H, W, Cin, Cout, k, stride, pad
    ("conv", 4, 16, 12, 64, 64, 1, 1, 0),
    ("conv", 3, 17, 13, 64, 128, 3, 2, 1),      # ragged last tile
    ("conv", 64, 16, 12, 256, 512, 1, 1, 0),    # 128-pixel tiles
    ("conv", 2, 8, 6, 512, 2048, 1, 1, 0),
    ("conv", 32, 64, 48, 64, 64, 3, 1, 1),
    ("deconv", 3, 8, 6, 64, 256, 4, 2, 1),
    ("stem", 4, 32, 24, 16, 64, 7, 2, 3),
])
def test_conv_with_fused_batch_statistics(cfg):
    """The *_bnstats convs store the same z as the plain convs and their partial rows finalize
    to the statistics pe_bn_stats_bf16 computes from z (the two sum in different orders)."""
    kind, B, H, W, Cin, Cout, k, stride, pad = cfg
    g = torch.Generator().manual_seed(61 + Cout + H)
    x = (rnd((B, H, W, Cin), g)).to(BF16).cuda()
    gamma = (torch.rand(Cout, generator=g) + 0.5).cuda()
    beta = (torch.randn(Cout, generator=g) * 0.2).cuda()
    if kind == "conv":
        w = (rnd((Cout, k, k, Cin), g) * (2.0 / (k * k * Cin)) ** 0.5).to(BF16).cuda()
        z_ref = pnn.conv2d(x, w, stride=stride, pad=pad)
        M = z_ref.numel() // Cout
        ws = pnn.bn_workspace(M, Cout, "cuda")
        z, rows = pnn.conv2d_bnstats(x, w, ws, stride, pad)
    elif kind == "deconv":
        w = (rnd((4, Cout, 4, Cin), g) * (2.0 / (4 * Cin)) ** 0.5).to(BF16).cuda()
        z_ref = pnn.deconv4x4s2(x, w)
        M = z_ref.numel() // Cout
        ws = pnn.bn_workspace(M, Cout, "cuda")
        z, rows = pnn.deconv4x4s2_bnstats(x, w, ws)
    else:
        w = (rnd((Cout, 4, 64), g) * 0.1).to(BF16).cuda()
        z_ref = pnn.stem7x7s2(x, w, None, None, relu=False)
        M = z_ref.numel() // Cout
        ws = pnn.bn_workspace(M, Cout, "cuda")
        z, rows = pnn.stem7x7s2_bnstats(x, w, ws)
    assert torch.equal(z, z_ref) and rows > 0
    rm1, rv1 = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    rm2, rv2 = rm1.clone(), rv1.clone()
    got = pnn.bn_finalize(ws, rows, M, Cout, gamma, beta, 1e-5, 0.1, rm1, rv1)
    got = [t.clone() for t in got]
    ws2 = pnn.bn_workspace(M, Cout, "cuda")
    want = pnn.bn_stats(z_ref, gamma, beta, 1e-5, 0.1, rm2, rv2, ws2)
    zf = z_ref.float().reshape(M, Cout).double()
    assert torch.allclose(got[0].cpu().double(), zf.mean(0).cpu(), atol=2e-6, rtol=1e-5)
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-6)
    assert torch.allclose(rm1, rm2, rtol=1e-4, atol=1e-6) and torch.allclose(rv1, rv2, rtol=2e-4)
    # ordered sums: a second run gives the same bits
    if kind == "conv":
        pnn.conv2d_bnstats(x, w, ws, stride, pad)
        again = pnn.bn_finalize(ws, rows, M, Cout, gamma, beta, 1e-5, 0.1, None, None)
        assert all(torch.equal(a, b) for a, b in zip(again, got))


def test_fused_statistics_reject_an_epilogue_and_a_small_workspace():
    x = torch.zeros((2, 8, 8, 64), dtype=BF16, device="cuda")
    w = torch.zeros((64, 1, 1, 64), dtype=BF16, device="cuda")
    tiny = torch.empty(64, dtype=torch.uint8, device="cuda")
    with pytest.raises(Exception, match="workspace|capacity|rows"):
        pnn.conv2d_bnstats(x, w, tiny)


def test_maxpool_backward_with_ties():
    g = torch.Generator().manual_seed(53)
    x = torch.relu(rnd((3, 16, 12, 64), g))  # about half the cells are exactly 0: many ties
    dy = rnd((3, 8, 6, 64), g)
    xt = nchw(x).requires_grad_(True)
    F.max_pool2d(xt, 3, 2, 1).backward(nchw(dy))
    y, idx = pnn.maxpool3x3s2(x.to(BF16).cuda(), want_index=True)
    assert torch.equal(y.float().cpu(), nhwc(F.max_pool2d(nchw(x), 3, 2, 1)))
    assert idx.dtype == torch.uint8 and int(idx.max()) <= 8
    got = pnn.maxpool3x3s2_bwd(idx, dy.to(BF16).cuda())
    close(got.float().cpu(), nhwc(xt.grad), 4e-3)  # sums of up to 4 bf16 values, rounded once


# ------------------------------------------------------------------ narrow channel counts
def packed_fwd_weight(w_oihw):
    cin = w_oihw.shape[1]
    return (pnn.pack_conv_weight_narrow(w_oihw) if cin in (8, 16, 32)
            else pnn.pack_conv_weight(w_oihw)).cuda()


def planned_wt(w_oihw, k, s, p):
    """Backward-data weights built from the forward packing through a TransposePlan."""
    cout, cin = w_oihw.shape[:2]
    wt = torch.zeros(pnn.bwd_data_weight_shape(cin, cout, k, s, p), dtype=BF16, device="cuda")
    plan = pnn.TransposePlan(torch.device("cuda:0"))
    wp = packed_fwd_weight(w_oihw)
    pnn.plan_bwd_data_weight(plan, wp, wt, cin, cout, k, s, p)
    plan.run()
    torch.cuda.synchronize()
    return wt


NARROW_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad
    (2, 16, 12, 32, 32, 3, 1, 1),
    (3, 16, 12, 32, 32, 3, 2, 1),
    (2, 17, 13, 32, 64, 3, 2, 1),     # odd map: ragged parity classes
    (2, 16, 12, 64, 32, 1, 1, 0),
    (2, 16, 12, 256, 32, 3, 1, 1),
    (2, 8, 12, 32, 128, 3, 2, 1),
    (2, 8, 6, 16, 16, 3, 1, 1),
    (2, 8, 6, 32, 17, 1, 1, 0),       # head: 17 joints, gradient padded to 64 channels
    (2, 16, 12, 128, 128, 3, 1, 1),   # regular on both sides, through the same plan
]


@pytest.mark.parametrize("case", NARROW_CASES)
def test_narrow_conv_backward(case):
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(71 + sum(case))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = ints((B, H, W, Cin), g, -2, 2)
    dy = ints((B, Ho, Wo, Cout), g, -1, 1)
    w = ints((Cout, Cin, k, k), g, -1, 1)
    # backward-data (the 17-channel head pads its gradient and uses the regular path)
    ref = nhwc(torch.nn.grad.conv2d_input((B, Cin, H, W), w, nchw(dy), stride=s, padding=p))
    if Cout == 17:
        dyp = torch.zeros(B, Ho, Wo, 64)
        dyp[..., :17] = dy
        wpad = torch.zeros(64, Cin, k, k)
        wpad[:17] = w
        got = pnn.conv2d_bwd_data(dyp.to(BF16).cuda(), planned_wt(wpad, k, s, p), (H, W), k, s, p)
    else:
        got = pnn.conv2d_bwd_data(dy.to(BF16).cuda(), planned_wt(w, k, s, p), (H, W), k, s, p)
        dyp = dy
    assert torch.equal(got.float().cpu(), to.bf16r(ref))
    if not (k == 1 and s == 2) and Cout != 17:
        res = ints((B, H, W, Cin), g, -3, 3)
        got = pnn.conv2d_bwd_data(dy.to(BF16).cuda(), planned_wt(w, k, s, p), (H, W), k, s, p,
                                  residual=res.to(BF16).cuda())
        assert torch.equal(got.float().cpu(), to.bf16r(ref + res))
    # backward-weight, in the forward weight's own packing
    refw = torch.nn.grad.conv2d_weight(nchw(x), (Cout, Cin, k, k), nchw(dy), stride=s, padding=p)
    wp = packed_fwd_weight(w)
    dw = torch.zeros(wp.shape, device="cuda")
    pnn.conv2d_bwd_weight(x.to(BF16).cuda(), dyp.to(BF16).cuda(), dw, k, s, p, cout=Cout)
    if Cin in (8, 16, 32):
        expect = pnn.pack_conv_weight_narrow(refw).float()
        got = dw.cpu() * pnn.narrow_weight_mask(Cout, Cin, k)
    else:
        expect = pnn.pack_conv_weight(refw).float()
        got = dw.cpu()
    assert torch.equal(got, expect)
    # forward with fused statistics on the narrow packing stores the plain conv's output
    if Cout % 8 == 0:
        ws = pnn.bn_workspace(B * Ho * Wo, Cout, "cuda")
        kw = {"kw": k} if Cin in (8, 16, 32) else {}
        z, rows = pnn.conv2d_bnstats(x.to(BF16).cuda(), wp.to(BF16), ws, s, p, **kw)
        refz = nhwc(F.conv2d(nchw(x), w, None, stride=s, padding=p))
        assert torch.equal(z.float().cpu(), to.bf16r(refz))
        assert rows >= 1


def test_stem3x3_training_entries():
    g = torch.Generator().manual_seed(83)
    B, H, W, Cout = 2, 32, 24, 64
    x3 = ints((B, H, W, 3), g, -2, 2)
    w = ints((Cout, 3, 3, 3), g, -1, 1)
    dy = ints((B, H // 2, W // 2, Cout), g, -1, 1)
    xs = torch.from_numpy(to.space_to_depth16(x3.numpy())).to(BF16).cuda()
    wp = pnn.pack_stem3x3_weight(w).cuda()
    ws = pnn.bn_workspace(B * (H // 2) * (W // 2), Cout, "cuda")
    z, rows = pnn.stem3x3s2_bnstats(xs, wp, ws)
    refz = nhwc(F.conv2d(nchw(x3), w, None, stride=2, padding=1))
    assert torch.equal(z.float().cpu(), to.bf16r(refz)) and rows >= 1
    gamma, beta = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
    mean = pnn.bn_finalize(ws, rows, z.numel() // Cout, Cout, gamma, beta, 1e-5, 0.1, None, None)[0]
    assert torch.allclose(mean.cpu(), refz.mean(dim=(0, 1, 2)), atol=1e-5)
    ref = torch.nn.grad.conv2d_weight(nchw(x3), (Cout, 3, 3, 3), nchw(dy), stride=2, padding=1)
    dw = torch.zeros(Cout, 2, 64, device="cuda")
    pnn.stem3x3s2_bwd_weight(xs, dy.to(BF16).cuda(), dw)
    mask = pnn.pack_stem3x3_weight(torch.ones(Cout, 3, 3, 3)).float()
    assert torch.equal(dw.cpu() * mask, pnn.pack_stem3x3_weight(ref).float())


@pytest.mark.parametrize("relu", [True, False])
def test_fuse_upsample_add_backward(relu):
    g = torch.Generator().manual_seed(89)
    B, H, W, C = 2, 16, 8, 32
    base = rnd((B, H, W, C), g)
    lows = [rnd((B, H >> f, W >> f, C), g) for f in (1, 2, 3)]
    dy = rnd((B, H, W, C), g)
    tb = nchw(base).requires_grad_(True)
    tl = [nchw(t).requires_grad_(True) for t in lows]
    s = tb
    for t, f in zip(tl, (1, 2, 3)):
        s = s + t.repeat_interleave(1 << f, dim=2).repeat_interleave(1 << f, dim=3)
    y = pnn.fuse_upsample_add(base.to(BF16).cuda(), [t.to(BF16).cuda() for t in lows], [1, 2, 3],
                              relu=relu)
    # mask with the kernel's own (bf16-rounded) output, as the training graph does
    yf = nchw(y.float().cpu())
    out = s * (yf > 0).float() if relu else s
    out.backward(nchw(dy))
    dbase, dlows = pnn.fuse_upsample_add_bwd(dy.to(BF16).cuda(), y if relu else None, [1, 2, 3])
    assert torch.equal(dbase.float().cpu(), to.bf16r(nhwc(tb.grad)))
    for got, t in zip(dlows, tl):
        close(got.float().cpu(), nhwc(t.grad), 4e-3)
    a, b = rnd((3, 5, 7, 8), g), rnd((3, 5, 7, 8), g)
    assert torch.equal(pnn.add_bf16(a.to(BF16).cuda(), b.to(BF16).cuda()).float().cpu(), to.bf16r(a + b))


@pytest.mark.parametrize("case", [
    # B, H, W, C, stride, dilation
    (2, 12, 10, 64, 1, 1),
    (2, 13, 11, 192, 2, 1),
    (3, 10, 10, 96, 1, 2),
    (2, 8, 8, 960, 1, 4),
    (1, 16, 12, 24, 2, 1),
])
def test_depthwise_backward(case):
    B, H, W, C, s, d = case
    g = torch.Generator().manual_seed(97 + sum(case))
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = ints((B, H, W, C), g, -2, 2)
    w = ints((C, 1, 3, 3), g, -1, 1)
    dy = ints((B, Ho, Wo, C), g, -1, 1)
    xt = nchw(x).requires_grad_(True)
    wt_ = w.clone().requires_grad_(True)
    y = F.conv2d(xt, wt_, None, stride=s, padding=d, dilation=d, groups=C)
    assert tuple(y.shape[2:]) == (Ho, Wo)
    y.backward(nchw(dy))
    w9c = w.reshape(C, 9).t().contiguous().to(BF16).cuda()
    dx = pnn.depthwise3x3_bwd_data(dy.to(BF16).cuda(), w9c, (H, W), s, d)
    assert torch.equal(dx.float().cpu(), to.bf16r(nhwc(xt.grad)))
    dw = torch.zeros(9, C, device="cuda")
    pnn.depthwise3x3_bwd_weight(x.to(BF16).cuda(), dy.to(BF16).cuda(), dw, s, d)
    assert torch.equal(dw.cpu(), wt_.grad.reshape(C, 9).t())
    pnn.depthwise3x3_bwd_weight(x.to(BF16).cuda(), dy.to(BF16).cuda(), dw, s, d)   # accumulates
    assert torch.equal(dw.cpu(), 2 * wt_.grad.reshape(C, 9).t())


# ------------------------------------------------------------- fused batch-norm backward (mask + sums)
def _bn_rows(z, C, g):
    """Per-channel rows a forward pass would have left: mean, rstd, scale, shift."""
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.3).cuda()
    ws = pnn.bn_workspace(z.numel() // C, C, "cuda")
    mean, rstd, scale, shift = pnn.bn_stats(z, gamma, beta, 1e-5, 0.1, None, None, ws)
    return gamma, mean, rstd, scale, shift, ws


def _check_fused(g_f, dz_f, dg_f, db_f, g_r, dz_r, dg_r, db_r):
    assert torch.equal(g_f, g_r)                         # the masked gradient, bit for bit
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()  # noqa: E731
    assert rel(dg_f, dg_r) < 1e-5 and rel(db_f, db_r) < 1e-5          # summation order only
    d = (dz_f.float() - dz_r.float()).abs()
    # coefficients differ in their last bits, so a few dz land on the neighbouring bf16 value
    assert d.max().item() <= 8e-3 * dz_r.float().abs().max().item()
    assert (d > 0).float().mean().item() < 0.02


FUSED_BWD_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, relu
    (2, 16, 12, 64, 256, 1, 1, 0, 1),      # conv3 of a bottleneck -> conv2's batch norm
    (3, 16, 12, 128, 128, 3, 1, 1, 1),     # conv2 -> conv1's batch norm
    (3, 16, 12, 128, 128, 3, 2, 1, 1),     # stride 2: four input-parity phases
    (2, 20, 20, 192, 64, 1, 1, 0, 2),      # ReLU6 (pointwise projection -> depthwise batch norm)
    (2, 24, 20, 256, 32, 1, 1, 0, 1),      # narrow gradient operand (final conv -> last deconv)
    (5, 64, 48, 64, 64, 3, 1, 1, 1),       # many tiles
]


@pytest.mark.parametrize("case", FUSED_BWD_CASES)
def test_bwd_data_with_fused_mask_and_sums_equals_the_separate_passes(case):
    B, H, W, Cin, Cout, k, s, p, relu = case
    g = torch.Generator().manual_seed(61 + sum(case))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    z = rnd((B, H, W, Cin), g).to(BF16).cuda()
    dy = rnd((B, Ho, Wo, Cout), g).to(BF16).cuda()
    w = rnd((Cout, Cin, k, k), g, (1.0 / (Cout * k * k)) ** 0.5)
    if Cout in (8, 16, 32):
        w_t = planned_wt(w, k, s, p)
    else:
        w_t = conv_wt(w)
    gamma, mean, rstd, scale, shift, ws = _bn_rows(z, Cin, g)
    if relu == 2:
        scale, shift = scale * 3.0, shift + 3.0      # spreads a over (0, 6) and beyond
    da = pnn.conv2d_bwd_data(dy, w_t, (H, W), k, s, p)
    dg_r, db_r = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    dz_r, g_r = pnn.bn_bwd(da, None, z, gamma, mean, rstd, dg_r, db_r, ws, want_masked=True,
                           relu=relu, act_scale=scale, act_shift=shift)
    g_r, dz_r = g_r.clone(), dz_r.clone()
    frac = (g_r == 0).float().mean().item()
    assert 0.15 < frac < 0.9                              # the mask does something
    ws2 = pnn.bn_workspace(B * H * W, Cin, "cuda")
    ws2.fill_(0xFF)                                       # NaN patterns: every row must be written
    dg_f, db_f = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    g_f, rows = pnn.conv2d_bwd_data_bnmask(dy, w_t, (H, W), k, s, p, z, scale, shift, ws2, relu)
    assert rows >= 1
    dz_f = pnn.bn_bwd_from_partials(g_f, z, rows, gamma, mean, rstd, dg_f, db_f, ws2)
    _check_fused(g_f, dz_f, dg_f, db_f, g_r, dz_r, dg_r, db_r)


RESMASK_CASES = [
    # B, H, W, Cin, Cout, k, stride, pad, with residual
    (2, 16, 12, 256, 64, 1, 1, 0, True),      # conv1 of a bottleneck -> the block below's conv3
    (3, 16, 12, 512, 128, 1, 1, 0, True),
    (2, 8, 6, 2048, 512, 1, 1, 0, True),      # few tiles: 64 x 64 double buffer
    (3, 16, 12, 64, 64, 3, 1, 1, True),       # basic block: 3x3 conv1
    (2, 24, 20, 32, 32, 3, 1, 1, True),       # narrow branch of the multi-branch net
    (3, 16, 12, 128, 128, 3, 2, 1, False),    # stride 2, no residual
    (5, 64, 48, 256, 64, 1, 1, 0, True),      # many tiles
]


@pytest.mark.parametrize("case", RESMASK_CASES)
def test_bwd_data_with_residual_and_stored_activation_mask(case):
    """The gradient of a residual layer's output act = relu(bn(z) + identity), made by the last of
    its consumers: backward-data + residual, masked by act > 0, with batch-norm backward's sums,
    against pe_conv2d_bwd_data_bf16(residual) followed by pe_bn_bwd_bf16(a = act)."""
    B, H, W, Cin, Cout, k, s, p, with_res = case
    g = torch.Generator().manual_seed(67 + sum(case[:8]))
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    z = rnd((B, H, W, Cin), g).to(BF16).cuda()
    ident = rnd((B, H, W, Cin), g).to(BF16).cuda()
    res = rnd((B, H, W, Cin), g).to(BF16).cuda() if with_res else None
    dy = rnd((B, Ho, Wo, Cout), g).to(BF16).cuda()
    w = rnd((Cout, Cin, k, k), g, (1.0 / (Cout * k * k)) ** 0.5)
    w_t = planned_wt(w, k, s, p) if Cout in (8, 16, 32) else conv_wt(w)
    gamma, mean, rstd, scale, shift, ws = _bn_rows(z, Cin, g)
    act = pnn.bn_apply(z, scale, shift, ident, relu=1)
    da = pnn.conv2d_bwd_data(dy, w_t, (H, W), k, s, p, residual=res)
    dg_r, db_r = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    dz_r, g_r = pnn.bn_bwd(da, act, z, gamma, mean, rstd, dg_r, db_r, ws, want_masked=True)
    g_r, dz_r = g_r.clone(), dz_r.clone()
    frac = (g_r == 0).float().mean().item()
    assert 0.15 < frac < 0.9
    ws2 = pnn.bn_workspace(B * H * W, Cin, "cuda")
    ws2.fill_(0xFF)
    dg_f, db_f = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    g_f, rows = pnn.conv2d_bwd_data_resmask(dy, w_t, (H, W), k, s, p, res, act, z, ws2)
    assert rows >= 1
    dz_f = pnn.bn_bwd_from_partials(g_f, z, rows, gamma, mean, rstd, dg_f, db_f, ws2)
    _check_fused(g_f, dz_f, dg_f, db_f, g_r, dz_r, dg_r, db_r)


@pytest.mark.parametrize("case", [(2, 8, 6, 256, 256), (3, 16, 12, 512, 256)])
def test_deconv_bwd_data_with_fused_mask_and_sums(case):
    B, H, W, Cin, Cout = case
    g = torch.Generator().manual_seed(71 + Cin)
    z = rnd((B, H, W, Cin), g).to(BF16).cuda()
    dy = rnd((B, 2 * H, 2 * W, Cout), g).to(BF16).cuda()
    w = rnd((Cin, Cout, 4, 4), g, (1.0 / (Cout * 4)) ** 0.5)
    wp = pnn.pack_deconv_weight(w).cuda()
    w_t = torch.zeros(Cin, 16, Cout, dtype=BF16, device="cuda")
    for ph in range(4):
        pnn.transpose_weights(wp[ph], w_t, t0=4 * ph)
    gamma, mean, rstd, scale, shift, ws = _bn_rows(z, Cin, g)
    da = pnn.deconv4x4s2_bwd_data(dy, w_t)
    dg_r, db_r = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    dz_r, g_r = pnn.bn_bwd(da, None, z, gamma, mean, rstd, dg_r, db_r, ws, want_masked=True,
                           act_scale=scale, act_shift=shift)
    g_r, dz_r = g_r.clone(), dz_r.clone()
    ws2 = pnn.bn_workspace(B * H * W, Cin, "cuda")
    ws2.fill_(0xFF)
    dg_f, db_f = torch.empty(Cin, device="cuda"), torch.empty(Cin, device="cuda")
    g_f, rows = pnn.deconv4x4s2_bwd_data_bnmask(dy, w_t, z, scale, shift, ws2)
    dz_f = pnn.bn_bwd_from_partials(g_f, z, rows, gamma, mean, rstd, dg_f, db_f, ws2)
    _check_fused(g_f, dz_f, dg_f, db_f, g_r, dz_r, dg_r, db_r)


def test_fused_mask_argument_errors():
    z = torch.zeros(1, 8, 8, 64, dtype=BF16, device="cuda")
    dy = torch.zeros(1, 4, 4, 64, dtype=BF16, device="cuda")
    w_t = torch.zeros(64, 1, 64, dtype=BF16, device="cuda")
    sc = torch.ones(64, device="cuda")
    ws = pnn.bn_workspace(64, 64, "cuda")
    with pytest.raises(Exception):     # 1x1 stride 2: three input parities have no tap
        pnn.conv2d_bwd_data_bnmask(dy, w_t, (8, 8), 1, 2, 0, z, sc, sc, ws)
    with pytest.raises(Exception):     # relu must say which activation
        pnn.conv2d_bwd_data_bnmask(torch.zeros_like(z), w_t, (8, 8), 1, 1, 0, z, sc, sc, ws, relu=0)


@pytest.mark.parametrize("shape,relu", [((4, 16, 12, 64), 1), ((2, 8, 6, 2048), 1),
                                        ((3, 10, 10, 96), 2), ((64, 64, 48, 64), 1),
                                        ((2, 6, 5, 1000), 2), ((1, 4, 4, 8), 0)])
def test_one_launch_batch_norm_equals_the_separate_launches(shape, relu):
    """Finalize folded into the element-wise pass (an in-launch hand-off between workgroups)
    against the separate finalize launch: every output bit for bit, forward and backward, with
    and without a residual / stored activation, repeated back to back so that the hand-off words
    are reused by consecutive launches; they must be zero afterwards and no wait may time out."""
    B, H, W, C = shape
    g = torch.Generator().manual_seed(59 + C)
    M = B * H * W
    z = (rnd(shape, g) * 2 + 0.5).to(BF16).cuda()
    res = rnd(shape, g).to(BF16).cuda()
    da = rnd(shape, g).to(BF16).cuda()
    gamma = (torch.rand(C, generator=g) + 0.5).cuda()
    beta = (torch.randn(C, generator=g) * 0.2).cuda()
    rm0, rv0 = (torch.randn(C, generator=g) * 0.1).cuda(), (torch.rand(C, generator=g) + 0.5).cuda()
    ws = pnn.bn_workspace(M, C, "cuda")
    sync = pnn.bn_sync("cuda")

    def separate(residual):
        rm, rv = rm0.clone(), rv0.clone()
        mean, rstd, scale, shift = pnn.bn_stats(z, gamma, beta, 1e-5, 0.1, rm, rv, ws)
        a = pnn.bn_apply(z, scale, shift, residual, relu=relu)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        if residual is not None or relu == 0:
            dz, gm = pnn.bn_bwd(da, a if relu else None, z, gamma, mean, rstd, dg, db, ws,
                                want_masked=True, relu=relu or 1)
        else:
            dz, gm = pnn.bn_bwd(da, None, z, gamma, mean, rstd, dg, db, ws, want_masked=True,
                                relu=relu, act_scale=scale, act_shift=shift)
        return [t.clone() for t in (mean, rstd, scale, shift, rm, rv, a, dz, gm, dg, db)]

    def fused(residual):
        rm, rv = rm0.clone(), rv0.clone()
        rows = torch.empty(4, C, device="cuda")
        a = pnn.bn_stats_apply(z, gamma, beta, 1e-5, 0.1, rm, rv, ws, rows, sync, residual, relu)
        mean, rstd, scale, shift = rows.unbind(0)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        if residual is not None or relu == 0:
            dz, gm = pnn.bn_bwd(da, a if relu else None, z, gamma, mean, rstd, dg, db, ws,
                                want_masked=True, relu=relu or 1, sync=sync)
        else:
            dz, gm = pnn.bn_bwd(da, None, z, gamma, mean, rstd, dg, db, ws, want_masked=True,
                                relu=relu, act_scale=scale, act_shift=shift, sync=sync)
        return [t.clone() for t in (mean, rstd, scale, shift, rm, rv, a, dz, gm, dg, db)]

    names = "mean rstd scale shift running_mean running_var a dz masked dgamma dbeta".split()
    for residual in (None, res):
        want = separate(residual)
        for rep in range(6):
            got = fused(residual)
            for nm, x, y in zip(names, got, want):
                assert torch.equal(x, y), (nm, rep)
    torch.cuda.synchronize()
    assert int(sync[512]) == 0, "a wait for the fold workgroups timed out"
