"""Is the per-allocation speed difference visible to a plain streaming fill / read?"""
import torch, time
x = torch.empty(1 << 28, device='cuda'); t0 = time.time()
while time.time() - t0 < 2.0:
    x.fill_(1.0); torch.cuda.synchronize()
del x
n = 65536 * 26800 // 4
keep = []
for a in range(8):
    t = torch.empty(n, dtype=torch.float32, device='cuda')
    keep.append(t)
    def tm(f, reps=10):
        f(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    tf = tm(lambda: t.fill_(2.0))
    tr = tm(lambda: t.sum())
    print("alloc %d %#x  fill %.2f TB/s  read(sum) %.2f TB/s" % (a, t.data_ptr(), n * 4 / tf / 1e12, n * 4 / tr / 1e12))
