"""Weight-gradient blocking sweep (diagnostic): time of noise_gen_pw_wgrad per forced plan, one process per plan
(NOISE_WGRAD_PLAN is read once).  python scripts/dbg/wgrad_sweep.py  -> table on stdout"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [(384, 64, 1281280), (64, 384, 1281280), (576, 96, 641280), (96, 576, 641280), (768, 128, 321280),
          (128, 768, 321280), (144, 24, 2561280), (24, 144, 2561280), (96, 16, 5121280), (16, 144, 5121280),
          (192, 32, 1281280), (32, 192, 1281280), (128, 128, 321280), (384, 128, 321280), (16, 32, 5121280)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch
    from noise_amd import generic_train as gt
    dev = torch.device("cuda:0")
    out = []
    for Cout, Cin, npix in SHAPES:
        dz = torch.randn(npix, Cout, device=dev); y = torch.randn(npix, Cout, device=dev)
        coef3 = torch.randn(6 * Cout, device=dev); a = torch.randn(npix, Cin, device=dev)
        ac = torch.randn(5 * Cin, device=dev)
        for _ in range(2):
            gt.pw_wgrad((dz, y, coef3), (a, ac, True), npix, Cout, Cin)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            gt.pw_wgrad((dz, y, coef3), (a, ac, True), npix, Cout, Cin)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 5)
        del dz, y, a
    print(" ".join(f"{t:.3f}" for t in out))
    sys.exit(0)
print("plan      " + " ".join(f"{co}x{ci}" for co, ci, _ in SHAPES))
for plan in ["auto", "01", "02", "04", "08", "016", "11", "12", "14", "18", "116"]:
    env = dict(os.environ)
    if plan != "auto":
        env["NOISE_WGRAD_PLAN"] = plan
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True, timeout=280)
    print(f"{plan:9s} {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}", flush=True)
