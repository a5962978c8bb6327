"""Phase timing of noise_pw_bwd (debug build with -DPW_STAMPS): cycles block 0 / thread 0 spends in each phase."""
import ctypes
import os
import sys

import torch

lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), sys.argv[1] if len(sys.argv) > 1 else "libpw_dbg.so"))
lib.noise_last_error.restype = ctypes.c_char_p
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_long
lib.noise_pw_bwd.argtypes = [P, P, P, P, P, I, P, P, I, P, P, P, P, P, P, I, P, P, L, I, I, P]
lib.noise_pw_bwd_max_rows.argtypes = [I, I]
lib.noise_debug_pw_stamps.argtypes = [P, I]
dev = torch.device("cuda:0")


def run(npix, Cout, Cin, has_b):
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g)
    dz, y, a = r(npix, Cout), r(npix, Cout), r(npix, Cin)
    b = r(npix, Cin) if has_b else None
    coef3 = r(6 * Cout)
    acoef = torch.cat([torch.ones(Cin, device=dev), torch.zeros(3 * Cin, device=dev), torch.zeros(Cin, device=dev)])
    w = r(Cout, Cin)
    g_in = torch.empty(npix, Cin, device=dev)
    rows = lib.noise_pw_bwd_max_rows(Cout, Cin)
    scratch = torch.empty(rows * Cout * Cin, device=dev)
    stats = torch.empty(rows * 2 * Cin, device=dev)
    nrows = torch.zeros(1, dtype=torch.int32)
    pp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        rc = lib.noise_pw_bwd(pp(dz), pp(y), pp(coef3), pp(a), pp(acoef), 1, pp(b), pp(acoef) if has_b else None, 1, pp(w),
                              pp(g_in), None, pp(scratch), None, pp(acoef), 1, pp(stats), ctypes.c_void_p(nrows.data_ptr()),
                              npix, Cout, Cin, st)
        assert rc == 0, lib.noise_last_error()
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.noise_debug_pw_stamps(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 10
    for _ in range(n):
        call()
    e1.record()
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 16)()
    lib.noise_debug_pw_stamps(out, 0)
    ms = e0.elapsed_time(e1) / n
    nb = 4 * npix * (2 * Cout + Cin * (3 if has_b else 2))
    names = ["prologue", "wait-prev(barrier)", "stage+commit", "barrier", "dgrad", "wgrad", "epilogue"]
    ntiles = (npix + 63) // 64
    print(f"Cout={Cout} Cin={Cin} B={has_b} npix={npix}: {ms:.3f} ms  {nb / ms / 1e6:.0f} GB/s  rows={int(nrows[0])} tiles/WG={ntiles / max(1, int(nrows[0])):.1f}")
    tot = sum(out[i] for i in range(7))
    for i, nm in enumerate(names):
        print(f"   {nm:20s} {out[i] / n:12.0f} ticks/launch  {100.0 * out[i] / tot:5.1f} %")


if __name__ == "__main__":
    run(1280 * 21 * 9, 32, 192, False)
    run(1280 * 21 * 9, 192, 64, True)
    run(1280 * 81 * 26, 16, 72, False)
