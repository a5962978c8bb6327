"""noise_bn_bwd_static (the entry point that allocates nothing: its partial slab is a static device array)
against noise_bn_bwd with a caller workspace, through the C ABI.  Two calls in a row, the second with a smaller
input, so that stale rows of the shared array would show."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

_P, _I, _L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long


def _bind():
    from noise_amd import _lib, ops  # noqa: F401  (ops binds noise_bn_bwd and sizes its workspace)
    lib = _lib.lib
    lib.noise_bn_bwd_static.argtypes = [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _L, _I, _P]
    lib.noise_bn_bwd_static.restype = ctypes.c_int
    return _lib, ops


def _case(dev, npix, C, relu6, second, seed):
    gen = torch.Generator(device="cpu").manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=gen).to(dev)
    y = r(npix, C)
    g1 = r(npix, C)
    g2 = r(npix, C) if second else None
    mean, var = y.mean(0), y.var(0, unbiased=False)
    gamma = (1.0 + 0.1 * r(C)).contiguous()
    istd = (var + 1e-5).rsqrt()
    coef = torch.cat([gamma * istd, 0.2 * r(C) - mean * gamma * istd, mean, istd, torch.zeros(C, device=dev)]).contiguous()
    return y, g1, g2, gamma, coef


def _run(_lib, ops, static, y, g1, g2, gamma, coef, relu6):
    p = _lib.ptr
    npix, C = y.shape
    dev = y.device
    out = dict(dz=torch.empty_like(y), dgamma=torch.empty(C, device=dev), dbeta=torch.empty(C, device=dev),
               coef3=torch.empty(6 * C, device=dev))
    st = _lib.stream()
    if static:
        rc = _lib.lib.noise_bn_bwd_static(p(g1), p(g2), p(y), p(coef), int(relu6), p(gamma), p(out["dz"]), p(out["dgamma"]),
                                          p(out["dbeta"]), p(out["coef3"]), npix, C, st)
        assert rc == 0, _lib.lib.noise_last_error().decode()
    else:
        ws = torch.empty(ops.CONV_MAX_BLOCKS * 2 * C, device=dev)
        ops.bn_bwd(g1, g2, y, coef, relu6, gamma, out["dz"], out["dgamma"], out["dbeta"], out["coef3"], ws, npix, C)
    torch.cuda.synchronize()
    return out


def test_bn_bwd_static_matches_workspace_variant():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    dev = torch.device("cuda:0")
    _lib, ops = _bind()
    # large first (fills all 1024 slab rows), then smaller inputs that use fewer rows and other widths
    for npix, C, relu6, second, seed in [(200000, 192, True, True, 1), (3000, 64, False, False, 2), (517, 16, True, True, 3)]:
        args = _case(dev, npix, C, relu6, second, seed)
        a = _run(_lib, ops, True, *args, relu6)
        b = _run(_lib, ops, False, *args, relu6)
        for k in a:
            assert torch.equal(a[k], b[k]), (npix, C, k)      # same kernels, same order of summation: bit-identical
    # sizes beyond the static workspace are refused with a clear message, nothing is launched
    y, g1, g2, gamma, coef = _case(dev, 64, 260, False, False, 4)
    p = _lib.ptr
    rc = _lib.lib.noise_bn_bwd_static(p(g1), None, p(y), p(coef), 0, p(gamma), p(torch.empty_like(y)), p(gamma.clone()),
                                      p(gamma.clone()), p(torch.empty(6 * 260, device=dev)), 64, 260, _lib.stream())
    assert rc != 0 and b"exceeds the static workspace" in _lib.lib.noise_last_error()
