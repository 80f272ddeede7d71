"""Keeps most of the GPU busy with FP64 matmuls for a few seconds (clock experiment: does a lone latency-bound workgroup run
faster when the rest of the chip is loaded?)."""
import sys, time, torch
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
a = torch.randn(6144, 6144, device="cuda", dtype=torch.float64)
t0 = time.time()
while time.time() - t0 < secs:
    for _ in range(10):
        b = a @ a
    torch.cuda.synchronize()
