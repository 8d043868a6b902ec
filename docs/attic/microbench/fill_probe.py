import torch, time
for mb in (64, 128, 192, 248, 300, 512, 1024, 2048):
    x = torch.empty(mb * 1000 * 1000 // 4, dtype=torch.float32, device='cuda')
    for _ in range(5): x.fill_(1.0)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): x.fill_(1.0)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1000 / 50
    print("fill %5d MB: %7.1f us  %.2f TB/s" % (mb, us, mb * 1e6 / us / 1e6))
