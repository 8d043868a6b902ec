"""Does the per-allocation speed class of the obs buffer show up in a translation-bound access pattern?

For several candidate allocations: time sgx_observe (the env's store pattern), a sequential fill, and a random 4-byte scatter
of 32 M elements (page-translation / DRAM-row bound).  Correlated observe/scatter times across candidates point at address
translation (fragment size of the mapping) rather than at the kernel's store pattern.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from stratego_env_amd.vec_env import VecStrategoEnv  # noqa: E402


def timed(fn, n):
    fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    x = torch.empty(1 << 28, device='cuda')
    t0 = time.time()
    while time.time() - t0 < 2.0:
        x.fill_(1.0)
        torch.cuda.synchronize()
    del x
    env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
    env.reset()
    n_el = env.obs.numel()
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    n_cand = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    cands = [env.obs] + [torch.empty_like(env.obs) for _ in range(n_cand - 1)]
    # random scatters confined to windows of 2 MiB / 32 MiB / 512 MiB / everything: if the slow class disappears for small
    # windows the cost is address translation (TLB reach), not DRAM rows
    wins = [1 << 19, 1 << 23, 1 << 27, n_el]
    widx = [torch.randint(0, w, (1 << 24,), device='cuda', generator=g) for w in wins]
    print("%-3s %-16s %10s %8s | scatter_us for windows 2MiB 32MiB 512MiB all" % ("i", "data_ptr", "observe_us", "fill_us"))
    for i, c in enumerate(cands):
        env.obs = c
        t_obs = timed(env.observe, 6)
        pass  # (the linear-map experiment knob SGX_MAP_MODE was removed from the library after this study)
        t_lin = timed(env.observe, 6)
        pass
        flat = c.view(-1)
        t_fill = timed(lambda: flat.fill_(0.5), 4)
        ts = [timed(lambda: flat.index_fill_(0, ix, 1.0), 3) for ix in widx]
        print("%-3d 0x%014x %10.1f (linear map %6.1f) %8.1f | %s" % (i, c.data_ptr(), t_obs, t_lin, t_fill, " ".join("%8.1f" % t for t in ts)), flush=True)


def slabs():
    """Is one huge allocation (large buddy blocks -> large mapping fragments?) in a better class than obs-sized ones?"""
    env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
    env.reset()
    n_el = env.obs.numel()
    shape = tuple(env.obs.shape)
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    idx = torch.randint(0, n_el, (1 << 24,), device='cuda', generator=g)
    keep = []
    print("%-22s %-16s %10s %10s" % ("allocation", "data_ptr", "observe_us", "scatter_us"))
    for label, gib in (("obs-sized #0", 0), ("obs-sized #1", 0), ("obs-sized #2", 0), ("slab 4 GiB", 4), ("slab 16 GiB", 16),
                       ("slab 64 GiB", 64), ("slab 128 GiB", 128), ("obs-sized #3", 0), ("obs-sized #4", 0)):
        if gib == 0:
            bufs = [(label, torch.empty(shape, dtype=torch.float32, device='cuda'))]
        else:
            slab = torch.empty(gib << 30, dtype=torch.uint8, device='cuda')
            keep.append(slab)
            bufs = []
            for off_gib in sorted(set([0, gib // 2, max(0, gib - 2)])):
                v = slab[off_gib << 30:(off_gib << 30) + n_el * 4].view(torch.float32).view(shape)
                bufs.append(("%s +%d GiB" % (label, off_gib), v))
        for name, b in bufs:
            env.obs = b
            t_obs = timed(env.observe, 6)
            flat = b.view(-1)
            t_sc = timed(lambda: flat.index_fill_(0, idx, 1.0), 3)
            print("%-22s 0x%014x %10.1f %10.1f" % (name, b.data_ptr(), t_obs, t_sc), flush=True)
            keep.append(b)


def regions():
    """Where inside an allocation does the slow class come from?  Random scatters into 256 MiB windows at several positions,
    into pairs of far-apart windows, and into strided page subsets."""
    env = VecStrategoEnv('barrage', 65536, seed=1, auto_reset=True)
    env.reset()
    n_el = env.obs.numel()
    g = torch.Generator(device='cuda')
    g.manual_seed(1)
    W = 1 << 26                                    # 256 MiB of float32
    base = torch.randint(0, W, (1 << 24,), device='cuda', generator=g)
    offs = [0, W, 2 * W, 3 * W, 4 * W, 5 * W, n_el - W]
    half = torch.randint(0, W // 2, (1 << 24,), device='cuda', generator=g)
    sel = torch.randint(0, 2, (1 << 24,), device='cuda', generator=g)
    pair_near = half + sel * (W // 2)              # two adjacent 128 MiB windows
    pair_far = half + sel * (n_el - W // 2)        # two 128 MiB windows at both ends
    allidx = torch.randint(0, n_el, (1 << 24,), device='cuda', generator=g)
    # every 8th 2-MiB page over the whole buffer: same footprint as a 1/8 window, full span
    pg = torch.randint(0, n_el // (1 << 19) // 8, (1 << 24,), device='cuda', generator=g)
    strided = pg * (8 << 19) + torch.randint(0, 1 << 19, (1 << 24,), device='cuda', generator=g)
    cands = [env.obs] + [torch.empty_like(env.obs) for _ in range(9)]
    print("cand observe_us | 256MiB windows at 0..5, last | pair near, pair far | every 8th 2MiB page | all")
    for i, c in enumerate(cands):
        env.obs = c
        t_obs = timed(env.observe, 6)
        flat = c.view(-1)
        tw = [timed(lambda: flat.index_fill_(0, base + o, 1.0), 3) for o in offs]
        tn = timed(lambda: flat.index_fill_(0, pair_near, 1.0), 3)
        tf = timed(lambda: flat.index_fill_(0, pair_far, 1.0), 3)
        tst = timed(lambda: flat.index_fill_(0, strided, 1.0), 3)
        ta = timed(lambda: flat.index_fill_(0, allidx, 1.0), 3)
        print("%-2d %8.1f | %s | %6.1f %6.1f | %6.1f | %6.1f" % (i, t_obs, " ".join("%6.1f" % t for t in tw), tn, tf, tst, ta), flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'regions':
        regions()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'slabs':
        slabs()
        sys.exit(0)
    main()
