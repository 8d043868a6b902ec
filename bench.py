"""bench.py -- env steps/sec of the batched env.step() hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 65,536 concurrent Barrage games PER GPU, synthetic
random-valid-action rollout with auto-reset, setups from the Gravon table, everything keyed by the counter RNG on
(seed, global env id, game, turn).  One "step" = one batched env.step() over all of the rank's games: action in,
move/combat/capture applied, win/draw detection, next mover's valid-actions mask (uint8 [R,C,K]) and normalised
partial observation (float32 [R,C,67]) written to HBM, plus the next random valid action.  Inputs are resident in
HBM when the timed region starts.  Multi-GPU: env ids are sharded contiguously across ranks, no collective on the
data path (one barrier + one MAX all-reduce of the elapsed time for reporting) => "scaling": "weak".

Prints ONE JSON line (rank 0) with `roofline` (HBM, algorithmic bytes = B_alg x games per launch / measured launch
time via HIP events on the launch stream) and `cpu_baseline` (the CPU oracle, a port of the reference's algorithm,
timed on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASE_SEED = 0x5712A7E60
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


def b_alg(rows, cols):
    """Algorithmic bytes per env step (SURVEY 8d / BASELINE.md): state read + compulsory write-back + action +
    float32 obs + uint8 mask + result record."""
    rc = rows * cols
    k = 2 * (rows - 1) + 2 * (cols - 1) + 1
    return (32 * rc + 16) + (rc + 16) + 4 + 4 * 67 * rc + rc * k + 12


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU boxes show
    256 logical CPUs but run under a 16-CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = max(1, min(n, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def measured_traffic(version, n_envs):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        e = t.get(version)
        if e and e['games_per_launch'] == n_envs:
            return e['hbm_bytes_per_launch']
    except Exception:
        pass
    return None


def cpu_baseline(version, seed, target_seconds):
    """Time the oracle's rollout harness (same workload rule) on the host cores; bounded to ~target_seconds."""
    from oracle import oracle as orc   # checker / baseline only
    from stratego_env_amd import setups as S
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    table = S.load_setup_table(v.human_inits) if v.human_inits else None
    cv = orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                           v.initial_state_usable_rows, setups=table)
    cores = usable_cores()
    n_probe, t_probe = 16 * cores, 256
    orc.rollout(cv, seed, 0, n_probe, t_probe, threads=cores)    # untimed: starts the OpenMP team, pages everything in
    t0 = time.perf_counter()
    total, _, _ = orc.rollout(cv, seed, 0, n_probe, t_probe, threads=cores)
    rate = total / max(time.perf_counter() - t0, 1e-6)
    n_steps = 512
    n_envs = int(max(cores, min(65536, rate * target_seconds / n_steps)))
    n_envs = (n_envs // cores) * cores or cores
    t0 = time.perf_counter()
    total, _, _ = orc.rollout(cv, seed, 0, n_envs, n_steps, threads=cores)
    dt = time.perf_counter() - t0
    # SURVEY 8d also asks for the one-thread figure: ~2 s of the same workload on one core
    n1 = max(8, int(rate / cores * 2.0 / n_steps))
    t1 = time.perf_counter()
    total1, _, _ = orc.rollout(cv, seed, 0, n1, n_steps, threads=1)
    one_thread = total1 / (time.perf_counter() - t1)
    return {"value": total / dt, "unit": "env steps/s", "cores": cores, "kind": "port", "value_1_thread": one_thread,
            "sample": "%d %s games x %d steps (envs 0..%d of the same seeded workload), oracle C port with OpenMP over "
                      "games on %d threads, %.1f s" % (n_envs, version, n_steps, n_envs - 1, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--envs', type=int, default=65536, help='games per GPU (weak scaling: fixed as --gpus grows)')
    ap.add_argument('--total-envs', type=int, default=0,
                    help='strong scaling instead (SURVEY 8d config 5): this many games in total, split evenly over the ranks')
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--unfused', action='store_true', help='sample actions with the standalone sampler kernel')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--traffic-bytes', type=float, default=None, help='HBM bytes per launch from a rocprofv3 --pmc pass')
    ap.add_argument('--placement-trials', type=int, default=96,
                    help='candidate allocations of the output tensors tried by VecStrategoEnv.tune_placement (1 = off)')
    ap.add_argument('--wake-seconds', type=float, default=2.0,
                    help='untimed GPU wake-up before the warmup steps (a fresh box runs its first ~second at idle clocks)')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or 'TORCHELASTIC_RUN_ID' in os.environ:   # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    if args.wake_seconds > 0:   # bring the GPU out of its idle power state; touches no env state
        scratch = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < args.wake_seconds:
            scratch.fill_(1.0)
            torch.cuda.synchronize()
        del scratch

    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[args.version]
    n = args.envs if not args.total_envs else args.total_envs // world    # rank r owns global ids [r*n, (r+1)*n)
    env = VecStrategoEnv(args.version, n, device=local_rank, seed=BASE_SEED, env_id_offset=rank * n, auto_reset=True)
    env.reset()
    placement_us = env.tune_placement(args.placement_trials, max_memory_fraction=0.5) if args.placement_trials > 1 else None
    env.sample_valid_actions()

    def one_step():
        if args.unfused:
            env.step(env.next_actions, want_next_actions=False)
            env.sample_valid_actions()
        else:
            env.rollout_step()

    for _ in range(args.warmup):
        one_step()
    games_before = env.env_info()[:, 1].to(torch.int64).sum()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    if args.unfused:
        for _ in range(args.steps):
            one_step()
    else:
        env.rollout_steps(args.steps)    # the same K batched steps, enqueued by one C-ABI call (sgx_step_n)
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    games = int(env.env_info()[:, 1].to(torch.int64).sum() - games_before)
    invalid = int(env.invalid_action.sum())
    if dist:
        t = torch.tensor([elapsed, dev_ms], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, dev_ms = float(t[0]), float(t[1])
        c = torch.tensor([games, invalid], dtype=torch.int64, device='cuda')
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        games, invalid = int(c[0]), int(c[1])
    assert invalid == 0, "rollout produced invalid actions"

    if rank == 0:
        total_steps = world * n * args.steps
        launch_s = dev_ms / 1e3 / args.steps                 # average device time per batched step (HIP events)
        bytes_per_launch = b_alg(v.rows, v.columns) * n
        achieved = bytes_per_launch / launch_s / 1e9
        out = {
            "metric": "env steps/sec", "value": total_steps / elapsed, "unit": "env steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_envs else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d concurrent %s games per GPU (%dx%d), random-valid-action rollout with auto-reset, "
                                   "%s step+sample" % (n, args.version, v.rows, v.columns,
                                                       "separate" if args.unfused else "fused"),
                       "games_per_gpu": n, "version": args.version, "seed": BASE_SEED,
                       "games_finished_in_timed_region": games, "b_alg_bytes_per_step": b_alg(v.rows, v.columns),
                       # per-candidate sgx_observe times of the start-up placement trial: the fastest is kept;
                       # "first" is the allocation the env would have used without the trial
                       "placement_trial_us": ({k: {"candidates": len(t), "first": round(t[0], 1), "min": round(min(t), 1),
                                                   "median": round(sorted(t)[len(t) // 2], 1), "max": round(max(t), 1)}
                                               for k, t in placement_us.items()} if placement_us else None)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": args.traffic_bytes if args.traffic_bytes is not None else measured_traffic(args.version, n),
                         "kernel": "step_kernel<%d,%d>" % (v.rows, v.columns),
                         "launch_us": launch_s * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:            # the CPU leg is timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.version, BASE_SEED, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    env.close()
    if dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
