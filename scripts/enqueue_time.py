import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from noise_amd.data import synth_batch
dev = torch.device("cuda:0")
m = bench.build_model(dev); opt = m.configure_optimizers()
noisy, clean, _ = synth_batch(64, 160000, dev, seed=1)
for _ in range(3): m.train_step_fused(noisy, clean, opt, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): m.train_step_fused(noisy, clean, opt, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1000*(t1-t0)/10:.2f} ms/step, total {1000*(t2-t0)/10:.2f} ms/step")
