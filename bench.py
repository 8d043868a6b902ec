"""bench.py -- env steps/sec of the batched env.step() hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (BASELINE.json configs[1]): 65,536 concurrent Barrage games PER GPU, synthetic
random-valid-action rollout with auto-reset, setups from the Gravon table, everything keyed by the counter RNG on
(seed, global env id, game, turn).  One "step" = one batched env.step() over all of the rank's games: action in,
move/combat/capture applied, win/draw detection, next mover's valid-actions mask (uint8 [R,C,K]) and normalised
partial observation (float32 [R,C,67]) written to HBM, plus the next random valid action.  Inputs are resident in
HBM when the timed region starts.

Multi-GPU: one process per GPU.  Under a launcher (RANK / WORLD_SIZE set) this process is one rank; run directly with
--gpus N > 1 it starts the N ranks itself as fresh child processes BEFORE anything touches the GPU (a process that has
initialised HIP is never re-executed) and relays rank 0's line.  Global env ids are sharded contiguously across the ranks
(stratego_env_amd.sharding.shard_range), no collective on the data path: one barrier on each side of the timed region and
one MAX / SUM all-reduce for reporting => "scaling": "weak" (fixed games per GPU) or "strong" (--total-envs).

Prints ONE JSON line (rank 0) with `roofline` (HBM; algorithmic bytes = B_alg x games per launch / measured launch time via
HIP events on the launch stream; also the fraction on an untuned output allocation and the fraction by measured HBM traffic),
`config.other_workloads` (BASELINE configs 3 and 4 timed after the headline, 1-GPU run only) and `cpu_baseline` (the CPU
oracle, a port of the reference's algorithm, timed on this box's host cores on a bounded sample of the same workload).

`--dry-run` is the launcher's self-test: same process / rendezvous / sharding / reduction code over gloo with a stub in
place of the env, no GPU, no measurement (`value` is null) -- what tests/test_bench_launcher_cpu.py runs.
"""
import argparse
import json
import os
import socket
import subprocess
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=512)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--envs', type=int, default=65536, help='games per GPU (weak scaling: fixed as --gpus grows)')
    ap.add_argument('--total-envs', type=int, default=0,
                    help='strong scaling instead (SURVEY 8d config 5): this many games in total, split over the ranks')
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--unfused', action='store_true', help='sample actions with the standalone sampler kernel')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the BASELINE config 3 / 4 legs (262,144 Standard games, 65,536 Micro games) after the headline')
    ap.add_argument('--traffic-bytes', type=float, default=None, help='HBM bytes per launch from a rocprofv3 --pmc pass')
    ap.add_argument('--placement-trials', type=int, default=None,
                    help='candidate allocations of the output tensors tried by VecStrategoEnv.tune_placement (1 = off; '
                         'default: as many as --placement-gb allows)')
    ap.add_argument('--placement-gb', type=float, default=8.0,
                    help='most extra device memory the placement trial may hold at any time')
    ap.add_argument('--wake-seconds', type=float, default=2.0,
                    help='untimed GPU wake-up before the warmup steps (a fresh box runs its first ~second at idle clocks)')
    ap.add_argument('--chains', type=int, default=1,
                    help='sgx_rollout: split the batch into this many ranges of games whose launches overlap on streams of their own '
                         '(1 = sgx_step_n, one launch per step: what the headline uses, so that the per-launch figures are per step)')
    ap.add_argument('--dry-run', action='store_true',
                    help="launcher self-test on CPU: gloo, stub env, no measurement (value is null)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# Launcher: `python bench.py --gpus N` outside any launcher starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args, argv):
    """Start one fresh `python bench.py` process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's
    stdout, return the first non-zero exit code (the other ranks are then terminated).  This process never initialises HIP:
    torch.cuda.device_count() does not, and nothing else here touches torch.cuda."""
    n = args.gpus
    if not args.dry_run:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            print("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SGX_BENCH_LAUNCHER='bench.py')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in sorted(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print("bench.py: rank %d exited with code %d; stopping the other ranks" % (r, code), file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------------------------------------------
# One rank
# ---------------------------------------------------------------------------------------------------------------
class Rank:
    """This process's place in the job, from the launcher's environment.  world must equal --gpus: a launcher that
    silently started fewer ranks is an error, not a smaller run."""

    def __init__(self, gpus, backend, use_cuda):
        import torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
        if self.world != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (gpus, self.world))
        if not 0 <= self.rank < self.world:
            raise SystemExit("bench.py: RANK=%d outside WORLD_SIZE=%d" % (self.rank, self.world))
        self.use_cuda = use_cuda
        self.dist = None
        if use_cuda:
            torch.cuda.set_device(self.local_rank)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            kw = {'device_id': torch.device('cuda', self.local_rank)} if use_cuda else {}
            dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            if dist.get_world_size() != gpus:
                raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), gpus))
            self.dist = dist
        self.device = 'cuda' if use_cuda else 'cpu'

    def sync(self):
        if self.use_cuda:
            import torch
            torch.cuda.synchronize()

    def barrier(self):
        """barrier + device synchronize on both sides (the bench contract's bracket of the timed region)."""
        self.sync()
        if self.dist:
            self.dist.barrier()
        self.sync()

    def reduce(self, maxes, sums):
        """MAX over ranks of the float list `maxes`, SUM over ranks of the int list `sums`; the only collectives of the run."""
        if not self.dist:
            return list(maxes), list(sums)
        import torch
        t = torch.tensor(list(maxes), dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        c = torch.tensor(list(sums), dtype=torch.int64, device=self.device)
        self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM)
        return [float(x) for x in t], [int(x) for x in c]

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


def shard_of(args, rk):
    """(first global env id, games) of this rank: --envs games per GPU (weak) or --total-envs split over the ranks (strong)."""
    from stratego_env_amd.sharding import shard_range
    total = args.total_envs if args.total_envs else args.envs * rk.world
    return shard_range(total, rk.rank, rk.world) + (total,)


def timed_steps(rk, run_warmup, run_timed, counters):
    """Warm up, then time run_timed() between barrier + synchronize brackets.  Returns (elapsed seconds MAX over ranks,
    device ms MAX over ranks or None, summed counter deltas)."""
    run_warmup()
    before = counters()
    ev = None
    if rk.use_cuda:
        import torch
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    rk.barrier()
    t0 = time.perf_counter()
    if ev:
        ev[0].record()
    run_timed()
    if ev:
        ev[1].record()
    rk.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev[0].elapsed_time(ev[1]) if ev else 0.0
    after = counters()
    (elapsed, dev_ms), sums = rk.reduce([elapsed, dev_ms], [a - b for a, b in zip(after, before)] + [1])
    assert sums[-1] == rk.world, "reduction covered %d ranks of %d" % (sums[-1], rk.world)
    return elapsed, (dev_ms if ev else None), sums[:-1]


class _StubEnv:
    """--dry-run only: stands in for VecStrategoEnv so that the launcher, rendezvous, sharding and reductions can be
    exercised without a GPU.  It plays no game; the run reports value null."""

    def __init__(self, first, n):
        self.first, self.n, self.steps_done = first, n, 0

    def rollout_steps(self, k):
        time.sleep(0.002 * k)
        self.steps_done += k

    def counters(self):
        return [self.steps_done * self.n, 0]


def make_env(version, n, first, local_rank):
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv(version, n, device=local_rank, seed=BASE_SEED, env_id_offset=first, auto_reset=True)
    env.reset()
    return env


def time_workload(rk, env, steps, warmup, unfused=False, chains=1):
    """(elapsed s, device ms, games finished, invalid actions) of `steps` batched steps on `env`, all MAX / SUM over ranks."""
    import torch

    def one_step():
        if unfused:
            env.step(env.next_actions, want_next_actions=False)
            env.sample_valid_actions()
        else:
            env.rollout_step()

    def run_warmup():
        env.sample_valid_actions()
        if chains > 1 and not unfused:       # (the chains' streams are created on first use: not inside the timed region)
            env.rollout_steps(warmup, chains=chains)
            return
        for _ in range(warmup):
            one_step()

    def run_timed():
        if unfused:
            for _ in range(steps):
                one_step()
        else:
            env.rollout_steps(steps, chains=chains)    # the same K batched steps, enqueued by one C-ABI call (sgx_step_n / sgx_rollout)

    def counters():
        return [int(env.env_info()[:, 1].to(torch.int64).sum()), 0]

    elapsed, dev_ms, (games, _) = timed_steps(rk, run_warmup, run_timed, counters)
    _, (invalid,) = rk.reduce([], [int(env.invalid_action.sum())])
    return elapsed, dev_ms, games, invalid


def other_workload(rk, version, n, seconds=1.0, extra_bytes=0, chains=1):
    """One of the other BASELINE configs on this GPU, about `seconds` of timed steps; output buffers from the same bounded
    placement trial as the headline (extra_bytes = 0: plain first allocation)."""
    import torch
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    env = make_env(version, n, 0, rk.local_rank)
    try:
        trial = env.tune_placement(max_extra_bytes=extra_bytes) if extra_bytes else None
        _, probe_ms, _, _ = time_workload(rk, env, 8, 8)
        steps = int(max(16, min(4096, seconds * 1e3 / max(probe_ms / 8, 1e-3))))
        elapsed, dev_ms, games, invalid = time_workload(rk, env, steps, 4)
        two = None
        if chains > 1:                     # the same steps with the batch split over concurrent chains of launches (sgx_rollout)
            e2, d2, _, inv2 = time_workload(rk, env, steps, 4, chains=chains)
            assert inv2 == 0
            two = {"chains": chains, "value": n * steps / e2, "us_per_step": d2 / steps * 1e3,
                   "frac": b_alg(v.rows, v.columns) * n / (d2 / 1e3 / steps) / 1e9 / HBM_PEAK_GBS}
        assert invalid == 0
        launch_s = dev_ms / 1e3 / steps
        bpl = b_alg(v.rows, v.columns) * n
        return {"workload": "%d concurrent %s games (%dx%d), same rollout" % (n, version, v.rows, v.columns),
                "value": n * steps / elapsed, "unit": "env steps/s", "steps": steps, "launch_us": launch_s * 1e6,
                "frac": bpl / launch_s / 1e9 / HBM_PEAK_GBS, "b_alg_bytes_per_step": b_alg(v.rows, v.columns),
                "kernel": "step_kernel<%d,%d>" % (v.rows, v.columns), "games_finished_in_timed_region": games,
                "traffic": measured_traffic(version, n), "concurrent_chains": two,
                "placement_trial_us": ({"candidates": len(trial['obs']), "first": round(trial['obs'][0], 1), "min": round(min(trial['obs']), 1)}
                                       if trial and trial['obs'] else None)}
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()


def run_rank(args):
    dry = args.dry_run
    if not dry:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    rk = Rank(args.gpus, 'gloo' if dry else 'nccl', use_cuda=not dry)
    first, n, total = shard_of(args, rk)
    if n <= 0:
        raise SystemExit("bench.py: rank %d got no games (%d games over %d ranks)" % (rk.rank, total, rk.world))

    if dry:
        env = _StubEnv(first, n)
        elapsed, _, (steps_x_games, _) = timed_steps(rk, lambda: env.rollout_steps(args.warmup),
                                                     lambda: env.rollout_steps(args.steps), env.counters)
        _, (covered, lo_gap) = rk.reduce([], [n, first if rk.rank == 0 else 0])
        if rk.rank == 0:
            print(json.dumps({"metric": "env steps/sec", "value": None, "unit": "env steps/s", "n_gpus": rk.world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                              "dry_run": True, "data": "none (launcher self-test)",
                              "launched_by": os.environ.get('SGX_BENCH_LAUNCHER', 'external'),
                              "scaling": "strong" if args.total_envs else "weak",
                              "config": {"total_games": total, "games_covered_by_ranks": covered,
                                         "stub_steps_x_games": steps_x_games}}), flush=True)
        rk.close()
        return

    import torch
    if args.wake_seconds > 0:   # bring the GPU out of its idle power state; touches no env state
        scratch = torch.empty(1 << 28, dtype=torch.float32, device='cuda')
        t_wake = time.perf_counter()
        while time.perf_counter() - t_wake < args.wake_seconds:
            scratch.fill_(1.0)
            torch.cuda.synchronize()
        del scratch

    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[args.version]
    env = make_env(args.version, n, first, rk.local_rank)
    placement_us = None
    if args.placement_trials is None or args.placement_trials > 1:
        placement_us = env.tune_placement(args.placement_trials, max_extra_bytes=int(args.placement_gb * (1 << 30)))
    elapsed, dev_ms, games, invalid = time_workload(rk, env, args.steps, args.warmup, args.unfused, args.chains)
    assert invalid == 0, "rollout produced invalid actions"
    two_chains = None
    if rk.world == 1 and args.chains == 1 and not args.unfused:
        # The same K steps with the batch split into two ranges of games whose launches overlap (sgx_rollout, chains = 2): reported
        # next to the headline, which stays one launch per step so that its per-launch figures can be checked against a kernel trace.
        e2, d2, _, inv2 = time_workload(rk, env, args.steps, args.warmup, False, 2)
        assert inv2 == 0
        two_chains = {"chains": 2, "value": total * args.steps / e2, "us_per_step": d2 / args.steps * 1e3,
                      "frac": b_alg(v.rows, v.columns) * n / (d2 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS}

    out = None
    if rk.rank == 0:
        total_steps = total * args.steps
        launch_s = dev_ms / 1e3 / args.steps                 # average device time per batched step (HIP events)
        bytes_per_launch = b_alg(v.rows, v.columns) * n
        achieved = bytes_per_launch / launch_s / 1e9
        traffic = args.traffic_bytes if args.traffic_bytes is not None else measured_traffic(args.version, n)
        first_us = placement_us['obs'][0] if placement_us and placement_us['obs'] else None
        out = {
            "metric": "env steps/sec", "value": total_steps / elapsed, "unit": "env steps/s",
            "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_envs else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d concurrent %s games per GPU (%dx%d), random-valid-action rollout with auto-reset, "
                                   "%s step+sample" % (n, args.version, v.rows, v.columns,
                                                       "separate" if args.unfused else "fused"),
                       "games_per_gpu": n, "total_games": total, "version": args.version, "seed": BASE_SEED,
                       "games_finished_in_timed_region": games, "b_alg_bytes_per_step": b_alg(v.rows, v.columns),
                       "concurrent_chains": args.chains, "two_chains": two_chains,
                       "launched_by": os.environ.get('SGX_BENCH_LAUNCHER', 'external' if rk.world > 1 else 'direct'),
                       # per-candidate sgx_observe times of the start-up placement trial: the fastest is kept;
                       # "first" is the allocation the env would have used without the trial
                       "placement_trial_us": ({k: {"candidates": len(t), "first": round(t[0], 1), "min": round(min(t), 1),
                                                   "median": round(sorted(t)[len(t) // 2], 1), "max": round(max(t), 1)}
                                               for k, t in placement_us.items() if t} if placement_us else None),
                       # most device memory the trial held beyond the buffers it kept (budget: --placement-gb)
                       "placement_peak_extra_gb": round(getattr(env, 'placement_peak_extra_bytes', 0) / 2.0 ** 30, 2)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "step_kernel<%d,%d>" % (v.rows, v.columns),
                         "launch_us": launch_s * 1e6, "algorithmic_bytes_per_launch": bytes_per_launch,
                         # the same kernel on the allocation the process got first (observe-only launch of the placement trial,
                         # algorithmic bytes over its time) and by measured HBM traffic instead of algorithmic bytes
                         "frac_untuned": (bytes_per_launch / (first_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if first_us else None,
                         "frac_traffic": (traffic / launch_s / 1e9 / HBM_PEAK_GBS) if traffic else None},
        }
    env.close()
    del env
    torch.cuda.empty_cache()
    if rk.rank == 0:
        out["config"]["other_workloads"] = None
        if rk.world == 1 and not args.no_other_workloads and args.version == 'barrage':
            extra = int(args.placement_gb * (1 << 30)) if placement_us else 0
            out["config"]["other_workloads"] = [other_workload(rk, 'standard', 262144, extra_bytes=extra, chains=2),
                                                other_workload(rk, 'micro', 65536, extra_bytes=extra, chains=2)]
        if not args.no_cpu_baseline and rk.world == 1:         # the CPU leg is timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.version, BASE_SEED, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    rk.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))     # nothing in this process has touched the GPU
    run_rank(args)


if __name__ == '__main__':
    main()
