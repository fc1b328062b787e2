import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pragma_dsp_amd.batch import BatchedFft
n = int(sys.argv[1]); frames = int(sys.argv[2]); dt = torch.float64 if sys.argv[3] == "f64" else torch.float32
plan = BatchedFft(n, "cuda:0", dtype=dt)
x = torch.randn((frames, n), device="cuda", dtype=dt)
amp = torch.empty((frames, n // 2 + 1), device="cuda", dtype=dt)
for _ in range(4):
    plan.spectrum(x, "hann", "one", out=amp)
torch.cuda.synchronize()
