"""The shape-generic backward kernels (csrc/gen_bwd.hip, the gradient variant of the generic 1x1 kernel in csrc/ofa.hip)
one by one through the C-ABI against float64 torch autograd of the same operation."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def rel(got, want):
    return float((got.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))


def lazy_grad(g, npix, C, affine):
    dz = torch.randn(npix, C, generator=g).to(DEV)
    if not affine:
        return (dz, None, None), dz.double()
    y = torch.randn(npix, C, generator=g).to(DEV)
    c3 = torch.randn(6, C, generator=g)
    c3[2] *= 1e-7
    c3[5] *= 1e-7
    c3 = c3.to(DEV).contiguous()
    d = c3.double()
    dy = d[0] * ((dz.double() - d[1]) - d[2]) + d[3] * ((y.double() - d[4]) - d[5])
    return (dz, y, c3.view(-1)), dy


@pytest.mark.parametrize("npix,Cout,Cin,affine,relu6", [(1000, 576, 96, True, True), (777, 16, 32, False, False),
                                                        (4100, 128, 576, True, True), (333, 20, 12, True, False),
                                                        (64, 384, 128, False, True),
                                                        # every blocking of the weight-gradient kernel (which operand the
                                                        # waves split, 1..16 channel tiles of the other) and of the 1x1 kernel
                                                        (1500, 16, 144, True, True), (1500, 144, 16, True, False),
                                                        (2000, 96, 576, True, True), (900, 128, 768, False, True),
                                                        (900, 768, 128, True, True), (3000, 24, 24, True, True),
                                                        (700, 4, 8, True, True), (2100, 36, 20, False, False),
                                                        (5000, 64, 384, True, True), (31, 192, 32, True, True)])
def test_pw_dgrad_and_wgrad(npix, Cout, Cin, affine, relu6):
    from noise_amd import generic_train as gt
    g = torch.Generator().manual_seed(npix)
    grad, dy = lazy_grad(g, npix, Cout, affine)
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(DEV)
    a_y = torch.randn(npix, Cin, generator=g).to(DEV)
    a_coef = torch.cat([1 + 0.2 * torch.randn(Cin, generator=g), 2 + torch.randn(Cin, generator=g), torch.zeros(3 * Cin)]).to(DEV)
    a = a_y.double() * a_coef[:Cin].double() + a_coef[Cin:2 * Cin].double()
    if relu6:
        a = a.clamp(0, 6)
    assert rel(gt.pw_dgrad(grad, w, npix, Cout, Cin), dy @ w.double()) < 1e-5
    assert rel(gt.pw_wgrad(grad, (a_y, a_coef, relu6), npix, Cout, Cin), dy.t() @ a) < 1e-5
    assert rel(gt.pw_wgrad(grad, (a_y, None, False), npix, Cout, Cin), dy.t() @ a_y.double()) < 1e-5


def test_small_pieces():
    from noise_amd import generic_train as gt
    g = torch.Generator().manual_seed(1)
    for M, N in [(5000, 128), (300, 1), (70000, 1), (999, 384), (10, 3)]:
        x = torch.randn(M, N, generator=g).to(DEV)
        assert rel(gt.colsum(x, N), x.double().sum(0)) < 1e-5
    gr, h = torch.randn(1000, 128, generator=g).to(DEV), torch.randn(1000, 128, generator=g).to(DEV)
    assert torch.equal(gt.relu_bwd(gr, h), gr * (h > 0))
    dst, src = torch.randn(3, 7, 40, 16, generator=g).to(DEV), torch.randn(3, 5, 33, 16, generator=g).to(DEV)
    want = dst.clone()
    want[:, 2:, 7:] += src
    gt.add_cropped(dst, src)
    assert torch.equal(dst, want)


@pytest.mark.parametrize("M,D", [(1000, 128), (37, 32), (5, 321), (4096, 1024)])
def test_ln_bwd(M, D):
    from noise_amd import generic_train as gt
    g = torch.Generator().manual_seed(M + D)
    r = (torch.randn(M, D, generator=g) * 2 + 0.5).to(DEV)
    gr = torch.randn(M, D, generator=g).to(DEV)
    add = torch.randn(M, D, generator=g).to(DEV)
    ln = torch.nn.LayerNorm(D, eps=1e-6).to(DEV)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.3 * torch.randn(D, generator=g))
        ln.bias.copy_(torch.randn(D, generator=g))
    r64 = r.double().requires_grad_(True)
    w64, b64 = ln.weight.double().detach().requires_grad_(True), ln.bias.double().detach().requires_grad_(True)
    F.layer_norm(r64, (D,), w64, b64, 1e-6).backward(gr.double())
    out, dgm, dbt = gt.ln_bwd(gr, r, ln, add)
    assert rel(out, r64.grad + add.double()) < 2e-5
    assert rel(dgm, w64.grad) < 2e-5 and rel(dbt, b64.grad) < 2e-5
    out2, _, _ = gt.ln_bwd(gr, r, ln)
    assert rel(out2, r64.grad) < 2e-5


@pytest.mark.parametrize("B,n,H,dk,p", [(2, 250, 6, 21, 0.1), (1, 300, 4, 8, 0.0), (3, 37, 3, 32, 0.25), (1, 260, 2, 40, 0.1),
                                         (2, 251, 6, 21, 0.0), (1, 515, 2, 16, 0.2), (1, 65, 1, 5, 0.3), (2, 16, 2, 24, 0.1),
                                         (1, 1031, 2, 10, 0.1)])
def test_attention_bwd(B, n, H, dk, p):
    from noise_amd import _lib, generic, generic_train  # noqa: F401  (register the entry points)
    g = torch.Generator().manual_seed(n + dk)
    HD = H * dk
    HDp = (HD + 3) // 4 * 4
    ld = 3 * HDp
    qkv = torch.zeros(B * n, ld)
    for i in range(3):
        qkv[:, i * HDp:i * HDp + HD] = torch.randn(B * n, HD, generator=g)
    qkv = qkv.to(DEV)
    keep = (torch.rand(B, H, n, n, generator=g) >= p).to(torch.uint8).to(DEV) if p > 0 else None
    ks = 1.0 / (1.0 - p)
    o = torch.zeros(B * n, HDp, device=DEV)
    base = qkv.data_ptr()
    q_, k_, v_ = (ctypes.c_void_p(base + 4 * i * HDp) for i in range(3))
    _lib.call("noise_tok_attention", q_, k_, v_, ld, P(keep), ks, P(o), HDp, B, n, H, dk, float(dk) ** 0.5, _lib.stream())
    go = torch.zeros(B * n, HDp)
    go[:, :HD] = torch.randn(B * n, HD, generator=g)
    go = go.to(DEV)
    dqkv = torch.zeros_like(qkv)
    dbase = dqkv.data_ptr()
    stats = torch.empty(B * H * n * 3, device=DEV)
    _lib.call("noise_gen_attn_bwd", q_, k_, v_, ld, P(keep), ks, P(o), P(go), HDp, ctypes.c_void_p(dbase),
              ctypes.c_void_p(dbase + 4 * HDp), ctypes.c_void_p(dbase + 8 * HDp), P(stats), B, n, H, dk, float(dk) ** 0.5,
              _lib.stream())
    x = qkv.double().view(B, n, ld).requires_grad_(True)
    q, k, v = [x[..., i * HDp:i * HDp + HD].reshape(B, n, H, dk).transpose(1, 2) for i in range(3)]
    a = torch.softmax(q / dk ** 0.5 @ k.transpose(2, 3), -1)
    if keep is not None:
        a = a * keep.double() * ks
    out = (a @ v).transpose(1, 2).reshape(B * n, HD)
    assert rel(o[:, :HD], out.detach()) < 1e-5
    out.backward(go[:, :HD].double())
    assert rel(dqkv, x.grad.view(B * n, ld)) < 2e-5


def test_vad_bwd():
    from noise_amd import _lib, generic_train  # noqa: F401
    g = torch.Generator().manual_seed(9)
    B, n, D = 37, 11, 128
    x = torch.randn(B, n, D, generator=g).to(DEV)
    gx = torch.randn(B, n, D, generator=g).to(DEV)
    w, dvad = torch.randn(D, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    want = gx.clone()
    want[:, 0, :] += dvad[:, None] * w[None, :]
    dw, db = torch.empty(D, device=DEV), torch.empty(1, device=DEV)
    _lib.call("noise_gen_vad_bwd", P(x), P(gx), D, n, P(w), P(dvad), P(dw), P(db), B, D, _lib.stream())
    assert (gx - want).abs().max() < 1e-6
    assert rel(dw, (dvad.double()[:, None] * x[:, 0, :].double()).sum(0)) < 1e-5
    assert rel(db, dvad.double().sum().view(1)) < 1e-5


@pytest.mark.parametrize("B,Hi,Wi,C", [(3, 1, 500, 32), (2, 9, 13, 16), (1, 4, 4, 256), (2, 3, 70, 24)])
def test_last_convT_bwd(B, Hi, Wi, C):
    from noise_amd import _lib, generic_train  # noqa: F401
    g = torch.Generator().manual_seed(C + Wi)
    a = torch.randn(B, Hi, Wi, C, generator=g).to(DEV)
    w = torch.randn(C, 1, 3, 3, generator=g).to(DEV)
    Hy, Wy = 2 * Hi - 1, 2 * Wi - 1
    dout = torch.randn(B, Hy, Wy, generator=g).to(DEV)
    gin, dw = torch.empty_like(a), torch.empty_like(w)
    slab = torch.empty(_lib.lib.noise_gen_last_bwd_rows() * 9 * C, device=DEV)
    _lib.call("noise_gen_last_convT_bwd", P(dout), P(a), P(w), P(gin), P(dw), P(slab), B, Hi, Wi, C, _lib.stream())
    a64 = a.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    w64 = w.double().requires_grad_(True)
    F.conv_transpose2d(a64, w64, None, 2, 1).backward(dout.double().view(B, 1, Hy, Wy))
    assert rel(gin, a64.grad.permute(0, 2, 3, 1)) < 1e-5
    assert rel(dw, w64.grad) < 1e-5


@pytest.mark.parametrize("npix,Cout,Cin,relu6", [(1000, 16, 96, True), (4100, 128, 576, True), (333, 20, 12, False)])
def test_pw_dgrad_epilogue(npix, Cout, Cin, relu6):
    """Input gradient masked by the producer's ReLU6 + that BatchNorm's two backward sums, as the epilogue leaves them."""
    from noise_amd import _lib, generic_train as gt
    g = torch.Generator().manual_seed(npix + 1)
    grad, dy = lazy_grad(g, npix, Cout, True)
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).to(DEV)
    ey = torch.randn(npix, Cin, generator=g).to(DEV)
    scale, shift = 1 + 0.2 * torch.randn(Cin, generator=g), 2 + torch.randn(Cin, generator=g)
    mean, istd = 0.1 * torch.randn(Cin, generator=g), 0.5 + torch.rand(Cin, generator=g)
    coef = torch.cat([scale, shift, mean, istd, 1e-8 * torch.randn(Cin, generator=g)]).to(DEV)
    slab = torch.empty(_lib.lib.noise_ofa_max_rows() * 2 * Cin, device=DEV)
    got, rows = gt.pw_dgrad(grad, w, npix, Cout, Cin, epi=(ey, coef, relu6), slab=slab)
    want = dy @ w.double()
    c = coef.double()
    if relu6:
        v = ey.double() * c[:Cin] + c[Cin:2 * Cin]
        want = want * ((v > 0) & (v < 6))
    assert rel(got, want) < 1e-5
    xh = ((ey.double() - c[2 * Cin:3 * Cin]) - c[4 * Cin:]) * c[3 * Cin:4 * Cin]
    sums = slab[:rows * 2 * Cin].view(rows, 2 * Cin).double().sum(0)
    assert rel(sums[:Cin], want.sum(0)) < 2e-5 and rel(sums[Cin:], (want * xh).sum(0)) < 2e-5
