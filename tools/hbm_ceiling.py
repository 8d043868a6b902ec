"""Practical HBM ceilings on this box: pure 16-B streaming writes (fill) and copy, 2.2 GB like one step launch."""
import torch
n = 2218262528 // 4
x = torch.empty(n, dtype=torch.float32, device='cuda')
y = torch.empty(n, dtype=torch.float32, device='cuda')
def t(f, reps=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(reps): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
tf = t(lambda: x.fill_(1.5))
tc = t(lambda: y.copy_(x))
tz = t(lambda: x.zero_())
print("fill  %.1f us  %.2f TB/s written" % (tf * 1e6, n * 4 / tf / 1e12))
print("zero  %.1f us  %.2f TB/s written" % (tz * 1e6, n * 4 / tz / 1e12))
print("copy  %.1f us  %.2f TB/s read+written" % (tc * 1e6, 2 * n * 4 / tc / 1e12))
