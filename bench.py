"""bench.py -- env steps/sec of the batched env.step() hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload.  BASELINE.json configs[1] on EVERY GPU: 65,536 concurrent Barrage games per GPU, synthetic
random-valid-action rollout with auto-reset, setups from the Gravon table, everything keyed by the counter RNG on (seed, global env id,
game, turn) -- so `value` over --gpus 1 / 2 / 4 / 8 is the weak-scaling curve of ONE workload ("scaling": "weak": per-GPU work fixed).
Every multi-GPU line is self-anchoring: before the ranks run a workload side by side, rank 0 times the same per-GPU workload ALONE (the
other ranks parked at a host-side barrier; same process, env object and buffers), and the line reports config.solo, config.scaling_x =
value / solo value and config.per_gpu_value_min_over_solo.  BASELINE configs[4] (262,144 games per GPU, 2,097,152 on
8 GPUs) is config.scaling_legs[0] of every multi-GPU line (with its own solo anchor and scaling_x; the strong split of 2,097,152 games
follows where it is a different size) and config.other_workloads[3] of the 1-GPU line; --envs / --total-envs override.
One "step" = one batched env.step() over all of the rank's games: action in, move/combat/capture applied, win/draw detection,
next mover's valid-actions mask (uint8 [R,C,K]) and normalised partial observation (float32 [R,C,67]) written to HBM, plus the
next random valid action.  Inputs are resident in HBM when the timed region starts.  On one GPU the K timed steps write their outputs
round-robin into THREE output sets (sgx_step_ring: a trajectory buffer of the last three steps): what a learner that stores every step
does, and the rate a policy that reads the outputs between steps sees (config.consumer_in_loop) -- nothing a launch writes can still be
cached when its addresses are written again, so `value` and `roofline` are DRAM-side figures.  The in-place variant (one set of
tensors rewritten every step: the headline of rounds 1-3, 8-10 % faster because rewriting the same 1-2 GB back to back is) follows as
config.in_place; config.no_settle is the headline's K steps once more without the untimed gpu_settle launches.  --output-sets 1 makes
in place the headline again.

Multi-GPU: one process per GPU.  Under a launcher (RANK / WORLD_SIZE set) this process is one rank; run directly with
--gpus N > 1 it starts the N ranks itself as fresh child processes BEFORE anything touches the GPU (a process that has
initialised HIP is never re-executed; the parent does not even import torch) and relays rank 0's line.  Global env ids are sharded
contiguously across the ranks (stratego_env_amd.sharding.shard_range), no collective on the data path: one barrier on each side
of the timed region (host-side, gloo) and one MAX / SUM all-reduce for reporting (RCCL when EVERY rank brought it up, else gloo for all:
bench_launcher.Rank).  The legs other than the headline live in bench_legs.py.

Prints ONE JSON line (rank 0) with
  `roofline`      HBM.  `achieved` = B_min x games per launch / launch time (HIP events on the launch stream over the timed region),
                  B_min = the packed layout's own byte minimum per env step (record in + record out + action in + next action out +
                  float32 observation + uint8 mask + results: 31,544 B for Barrage) -- what the kernel cannot avoid
                  moving; `frac` = achieved / 8 TB/s, and with the ring headline that is a DRAM fraction (`frac_dram` repeats it).
                  `in_place_rate_over_spec_peak` = the same bytes over the in-place leg's launch time: a memory-side rate that can touch
                  the spec peak, deliberately not called a fraction.  `traffic` = counter bytes per launch from the committed rocprofv3
                  --pmc passes (profiles/traffic.json; `traffic_source` says which binary they were measured on: "static" unless the
                  build id matches); `frac_untuned` = B_min over the observe launch of the process's plain first allocation;
                  `survey_8d` = SURVEY 8d's 33,848 B per step x games / launch time, a labelled comparison only (the kernel moves
                  fewer bytes than that formula assumes, so it is not a fraction of anything);
  `verified_envs` sampled envs of the very env object that was timed, checked after the timed region against the CPU oracle
                  replaying the same number of steps (outputs of the last step, turn and game counters);
  `config.in_place`   the in-place leg (+ `config.two_chains`); `config.consumer_in_loop`  sgx_step alternating with a device policy that READS
                  the observation and the mask (examples/batched_policy_loop.py), non-temporal against plain stores under that reader;
  `config.other_workloads`  BASELINE configs 3 and 4 and the reference's default BOTH_OBSERVATIONS mode (1-GPU run only);
  `cpu_baseline`  the CPU oracle (a port of the reference's algorithm) timed on this box's host cores on a bounded sample.

`--dry-run` is the launcher's self-test: same process / rendezvous / sharding / reduction code over gloo with a stub in
place of the env, no GPU, no measurement (`value` is null) -- what tests/test_bench_launcher_cpu.py runs.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASE_SEED = 0x5712A7E60
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
HBM_PIN_RATE_GBS = 8192.0   # 8 stacks x 1,024 pins x 8 Gbit/s: what the "8 TB/s" of the data sheet rounds


def hbm_mclk_mhz():
    """The memory clock the driver reports for this rank's device (sysfs pp_dpm_mclk, the level marked '*'); 2,000 MHz x 4 transfers = the
    8 Gbit/s per pin behind HBM_PIN_RATE_GBS.  None where the file is not readable."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_mclk')):
        try:
            for line in open(f):
                m = re.match(r'\s*\d+:\s*(\d+)\s*Mhz\s*\*', line, re.I)
                if m:
                    best = max(best or 0, int(m.group(1)))
        except OSError:
            pass
    return best
GAMES_1GPU, GAMES_PER_GPU_MULTI, STRONG_TOTAL = 65536, 262144, 2097152


def b_alg(rows, cols, full_obs=False):
    """Algorithmic bytes per env step (SURVEY 8d / BASELINE.md): state read + compulsory write-back + action +
    float32 obs + uint8 mask + result record; BOTH_OBSERVATIONS adds the 79-channel observation."""
    rc = rows * cols
    k = 2 * (rows - 1) + 2 * (cols - 1) + 1
    return (32 * rc + 16) + (rc + 16) + 4 + 4 * 67 * rc + rc * k + 12 + (4 * 79 * rc if full_obs else 0)


def record_bytes(v):
    """Bytes of one game's packed record in HBM -- the arithmetic of sgx_create (sgx_record_bytes(h) returns the same for a live
    handle; tests/test_gpu_parity.py compares the two): 4 dense int8 boards, two never-moved bitmaps, 32 B of scalars, the
    capture-event list (2 x pieces entries, at most one per cell), rounded up to whole 128-byte lines."""
    rc = v.rows * v.columns
    s4 = (rc + 3) & ~3
    st_off = (4 * s4 + 15) & ~15
    sb = (((rc + 7) // 8) + 15) & ~15
    pieces = max(sum(v.piece_counts), int(getattr(v, 'capture_capacity', 0)))
    max_events = min(2 * pieces, rc)
    return (st_off + 2 * sb + 32 + (4 if rc > 256 else 2) * max_events + 127) & ~127


def b_min(v, full_obs=False, rec_bytes=None, fused_steps=1):
    """The packed layout's own byte minimum of one env step -- what `roofline` is priced on: the record read
    once and written once, the action read (4) and the next action written (4), the float32 observation(s), the uint8 mask and
    12 B of results (two float32 rewards, done, invalid_action, ending_invalid, player).  Barrage 31,544 B, Standard 31,800 B,
    Micro 3,624 B.  fused_steps = K > 1: the K steps of one multi-step launch (boards of at most 16 cells: the games stay in registers,
    DESIGN.md section 3) move the record and the action once per LAUNCH: per step (2 x record + 8) / K + outputs -- Micro 3,368 B + 264 / K."""
    rc = v.rows * v.columns
    k = 2 * (v.rows - 1) + 2 * (v.columns - 1) + 1
    rb = rec_bytes if rec_bytes else record_bytes(v)
    per_launch = 2 * rb + 4 + 4
    outputs = 4 * 67 * rc + rc * k + 12 + (4 * 79 * rc if full_obs else 0)
    if fused_steps and fused_steps > 1:
        return outputs + per_launch / float(fused_steps)
    return per_launch + outputs


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


def measured_traffic(key, n_envs, build_id=None):
    """(counter bytes per launch, source) from the committed rocprofv3 --pmc passes (profiles/traffic.json), or (None, None).
    The kernel moves the same bytes for every game, so an entry measured at another batch size is scaled by the game count
    (and labelled as such).  The file is STATIC: bench.py cannot run itself under the profiler; the source string says whether
    the entry was measured on the binary that is running now (same build id) or on an earlier one."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json')))
        e = t.get(key)
        if not e:
            return None, None
        same = bool(build_id) and e.get('build_id') == build_id
        src = "%s: profiles/traffic.json[%s], rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE of %s (build id %s)" % (
            "static, measured on this binary" if same else "static, measured on an EARLIER binary", key,
            e.get('source', 'the committed pmc summary'), e.get('build_id', 'not recorded'))
        if e['games_per_launch'] == n_envs:
            return e['hbm_bytes_per_launch'], src
        return e['hbm_bytes_per_launch'] * (n_envs / e['games_per_launch']), src + " (measured at %d games per launch, scaled per game)" % e['games_per_launch']
    except Exception:
        return None, None


def live_traffic(version, n_envs, output_sets, full_obs=False, timeout=240):
    """HBM bytes per launch of the headline's kernel measured NOW, on this box and this binary: two short child processes of this
    script (`--traffic-probe`: the same env, 2 + 6 rollout steps into the same number of output sets) under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and `... --pmc WRITE_SIZE` -- separate passes, run from /tmp with TMPDIR=/tmp, the program
    itself after `--`: MI355X_MICROARCH.md's recipe -- and the guide's gfx950 correction: bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024,
    mean over the step kernel's dispatches.  The children start after every timed region of this process is over.
    -> (bytes per launch, source) or (None, why not)."""
    import csv
    import shutil
    import tempfile
    exe = shutil.which('rocprofv3') or '/opt/rocm/bin/rocprofv3'
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if 'rocprof' in os.environ.get('LD_PRELOAD', '').lower() or any(k.startswith(('ROCPROF', 'ROCP_')) for k in os.environ):
        return None, "this run is itself under a profiler: no nested rocprofv3"
    vals = {}
    probe_steps = 2 + 6                  # what traffic_probe plays
    try:
        for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
            out = tempfile.mkdtemp(prefix='sgx_traffic_', dir='/tmp')
            cmd = [exe, '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.abspath(__file__),
                   '--traffic-probe', '--version', version, '--envs', str(n_envs), '--output-sets', str(output_sets)] + (['--full-obs'] if full_obs else [])
            env = dict(os.environ, TMPDIR='/tmp')
            p = subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=timeout)
            rows = []
            for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
                rows += [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter and 'observe_kernel' not in r['Kernel_Name'] and
                         any(k in r['Kernel_Name'] for k in ('step_kernel<', 'steps_kernel<'))]
            shutil.rmtree(out, ignore_errors=True)
            if p.returncode != 0 or not rows:
                return None, "rocprofv3 --pmc %s over the probe gave no step kernel rows (rc %d: %s)" % (counter, p.returncode, p.stderr.decode('utf-8', 'replace')[-200:].replace('\n', ' '))
            multi = any('steps_kernel<' in r['Kernel_Name'] for r in rows)          # multi-step launches: counters per STEP = sum over the launches / steps played
            total = sum(float(r['Counter_Value']) for r in rows)
            vals[counter] = (total / probe_steps if multi else total / len(rows), probe_steps if multi else len(rows))
    except Exception as e:          # noqa: BLE001 -- a profiler that is not usable here must not cost the run its line
        return None, "%s: %s" % (type(e).__name__, str(e)[:160])
    byts = (2.0 * vals['FETCH_SIZE'][0] + vals['WRITE_SIZE'][0]) * 1024.0
    return byts, ("live: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (two passes) over a child of this run on this box and binary "
                  "(the same env, %d + %d steps averaged), (2 x FETCH_SIZE %.0f KiB + WRITE_SIZE %.0f KiB) x 1024 per step" %
                  (vals['FETCH_SIZE'][1], vals['WRITE_SIZE'][1], vals['FETCH_SIZE'][0], vals['WRITE_SIZE'][0]))


def traffic_probe(args):
    """--traffic-probe (the child of live_traffic, under rocprofv3): the headline's env, 2 + 6 rollout steps, nothing printed."""
    import torch
    env = make_env(args.version, args.envs, 0, 0, full_obs=args.full_obs)
    if args.output_sets >= 2:
        env.alloc_output_ring(args.output_sets)
    env.rollout_steps(2, ring=args.output_sets >= 2)
    env.rollout_steps(6, ring=args.output_sets >= 2)
    torch.cuda.synchronize()
    env.close()


def oracle_variant(version):
    from oracle import oracle as orc   # checker / baseline only
    from stratego_env_amd import setups as S
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    table = S.load_setup_table(v.human_inits) if v.human_inits else None
    return orc, orc.make_cvariant(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts,
                                  v.initial_state_usable_rows, setups=table)


def cpu_baseline(version, seed, target_seconds):
    """Time the oracle's rollout harness (same workload rule) on the host cores; bounded to ~target_seconds."""
    orc, cv = oracle_variant(version)
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
    ap.add_argument('--envs', type=int, default=None,
                    help='games per GPU (weak scaling: fixed as --gpus grows); default 65,536 (BASELINE config 2) on ANY number of GPUs, so '
                         'that `value` over --gpus 1 / 2 / 4 / 8 is a curve of one workload; config 5 (262,144 per GPU) is a leg of every line')
    ap.add_argument('--total-envs', type=int, default=0,
                    help='strong scaling instead (SURVEY 8d config 5): this many games in total, split over the ranks')
    ap.add_argument('--strong-total', type=int, default=None,
                    help='games of the strong-scaling LEG reported next to the weak headline (default 2,097,152 when --gpus > 1, '
                         '0 = skip; pass it explicitly to get the leg on one GPU)')
    ap.add_argument('--leg-envs', type=int, default=None,
                    help="games per GPU of the multi-GPU line's first scaling leg (default 262,144 = BASELINE config 5 when --envs is "
                         "defaulted, else no leg; 0 = skip)")
    ap.add_argument('--version', default='barrage')
    ap.add_argument('--unfused', action='store_true', help='sample actions with the standalone sampler kernel')
    ap.add_argument('--full-obs', action='store_true',
                    help="BOTH_OBSERVATIONS (the reference's default observation mode): the 79-channel fully-observable observation is "
                         "written next to the partial one (profiling runs of that workload; the headline is PARTIALLY_OBSERVABLE)")
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the legs after the headline (262,144 Standard games, 65,536 Micro games, BOTH_OBSERVATIONS; the '
                         'other per-GPU sizes of a multi-GPU run)')
    ap.add_argument('--traffic-bytes', type=float, default=None, help='HBM bytes per launch from a rocprofv3 --pmc pass')
    ap.add_argument('--placement', default='trial', choices=('trial', 'plain'),
                    help="output buffers: 'trial' = library-owned, picked by sgx_alloc_outputs' bounded placement trial; "
                         "'plain' = the torch.empty tensors of VecStrategoEnv")
    ap.add_argument('--placement-trials', type=int, default=None, help='most candidate allocations (default: what --placement-gb allows)')
    ap.add_argument('--placement-wide-gb', type=float, default=64.0,
                    help='budget of the second, wide placement pass that runs when the first found no fast candidate (0: no second pass)')
    ap.add_argument('--placement-free-fraction', type=float, default=0.4,
                    help='cap of both placement budgets: this fraction of the FREE device memory, divided by the ranks sharing the device')
    ap.add_argument('--placement-gb', type=float, default=8.0,
                    help='most extra device memory the placement trial may hold at any time')
    ap.add_argument('--settle-seconds', type=float, default=0.1,
                    help='untimed, state-preserving sgx_observe launches directly before the timed bracket (gpu_settle); 0 = none')
    ap.add_argument('--wake-seconds', type=float, default=2.0,
                    help='untimed GPU wake-up before the warmup steps (a fresh box runs its first ~second at idle clocks)')
    ap.add_argument('--output-sets', type=int, default=None,
                    help='output sets the HEADLINE writes round-robin (sgx_step_ring: a trajectory buffer of the last R steps).  Default: 3 '
                         '-- no line a launch writes can still be cached when it is written again, so value / roofline are DRAM-side '
                         'figures, the rate a learner that stores every step, or a policy that reads the outputs, sees; 1 = in place')
    ap.add_argument('--no-in-place-leg', action='store_true',
                    help='skip the in-place leg after a ring headline (the same K steps into ONE set of tensors, and its two-chains variant)')
    ap.add_argument('--no-consumer-leg', action='store_true', help='skip the consumer-in-the-loop leg (sgx_step alternating with a device policy)')
    ap.add_argument('--no-scaling-legs', action='store_true', help="skip config.scaling_legs (a multi-GPU line's other games-per-GPU sizes)")
    ap.add_argument('--no-settle-leg', action='store_true', help='skip config.no_settle (the headline steps once more without gpu_settle)')
    ap.add_argument('--chains', type=int, default=1,
                    help='sgx_rollout: split the batch into this many ranges of games whose launches overlap on streams of their own '
                         '(1 = sgx_step_n, one launch per step: what the headline uses, so that the per-launch figures are per step)')
    ap.add_argument('--no-two-chains', action='store_true', help='skip the two-concurrent-chains leg after the headline (profiling runs: '
                    'keeps the kernel statistics to one launch shape)')
    ap.add_argument('--verify-envs', type=int, default=32,
                    help='envs per rank checked against the CPU oracle after the timed region (0 = off)')
    ap.add_argument('--devices', default=None,
                    help='comma-separated device index per local rank (default: LOCAL_RANK); "0,0" runs two ranks on one GPU '
                         '(tests/test_gpu_two_ranks.py)')
    ap.add_argument('--backend', default=None, choices=('nccl', 'gloo'),
                    help='process-group backend of the reporting reductions (default nccl = RCCL; gloo when ranks share a GPU)')
    ap.add_argument('--no-live-traffic', action='store_true',
                    help="roofline.traffic from profiles/traffic.json only (static); default on one GPU: measured now by two short children of this "
                         "run under rocprofv3 --pmc (about 20 s), the static entry kept as roofline.traffic_static")
    ap.add_argument('--no-store-probe', action='store_true',
                    help='skip roofline.store_probe (the store-only kernel on the headline ring buffers: three payloads per set and one long launch)')
    ap.add_argument('--no-trajectory-leg', action='store_true', help='skip config.trajectory (the rollout into a 64-slot trajectory buffer, sgx_step_traj)')
    ap.add_argument('--trajectory-slots', type=int, default=64)
    ap.add_argument('--no-facade-leg', action='store_true', help='skip config.facade_n1 (BASELINE config 1: one game behind the dict API)')
    ap.add_argument('--traffic-probe', action='store_true', help=argparse.SUPPRESS)
    ap.add_argument('--dry-run', action='store_true',
                    help="launcher self-test on CPU: gloo, stub env, no measurement (value is null)")
    args = ap.parse_args(argv)
    if args.envs is None:
        # the SAME per-GPU workload on any number of GPUs (BASELINE config 2 on every GPU): `value` over --gpus 1 / 2 / 4 / 8 is a weak-scaling
        # curve of one workload.  BASELINE config 5 (262,144 games per GPU, 2,097,152 on 8) is config.scaling_legs[0] of every multi-GPU
        # line and config.other_workloads[3] of the 1-GPU line, each with its own anchor.
        args.envs = GAMES_1GPU
        args.envs_defaulted = True
    else:
        args.envs_defaulted = False
    if args.output_sets is None:
        args.output_sets = 3 if (not args.unfused and args.chains == 1) else 1
    if args.output_sets < 1:
        ap.error("--output-sets must be >= 1")
    args.rotate_sets = 3           # the ring of the Micro leg (other_workloads)
    if args.leg_envs is None:
        args.leg_envs = GAMES_PER_GPU_MULTI if (args.gpus > 1 and not args.total_envs and args.envs_defaulted) else 0
    if args.strong_total is None:
        args.strong_total = STRONG_TOTAL if (args.gpus > 1 and not args.total_envs and args.envs_defaulted) else 0
    return args


import bench_legs  # noqa: E402  (the legs other than the headline; every function takes this module as its first argument)
from bench_launcher import Rank, launch_ranks, visible_gpus  # noqa: E402,F401  (process plumbing: bench_launcher.py)


def legs_for(rank, world, args, dry=False):
    """Which of the line's optional legs THIS rank runs (a pure function of its arguments: tests/test_bench_launcher_cpu.py checks every
    rank of an 8-rank job).  The CPU baseline, the live counter passes (children of rank 0 under rocprofv3), the store probe, the facade
    and the other workloads belong to rank 0 of a ONE-GPU run only; a multi-GPU line spends its time on the scaling legs, which every
    rank takes part in."""
    legs = set()
    # (the scaling legs: every rank takes part, same barriers)
    if not args.no_scaling_legs and ((world > 1 and not args.total_envs and args.leg_envs) or args.strong_total):
        legs.add('scaling_legs')
    if dry:
        return legs
    if SETTLE_SECONDS > 0 and not args.no_settle_leg:
        legs.update(('no_settle', 'one_launch_per_step'))
    if world > 1:
        return legs
    if rank != 0:
        return legs
    headline_ring = args.output_sets >= 2 and not args.unfused and args.chains == 1
    if headline_ring and not args.no_in_place_leg:
        legs.add('in_place')
    if args.chains == 1 and not args.unfused and not args.no_two_chains and not (headline_ring and args.no_in_place_leg):
        legs.add('two_chains')
    if not args.no_store_probe and not args.unfused and not args.full_obs:
        legs.add('store_probe')
    if not args.no_other_workloads and args.version == 'barrage':
        legs.add('other_workloads')
        if not args.no_consumer_leg:
            legs.add('consumer_in_loop')
        if not args.no_trajectory_leg:
            legs.add('trajectory')
    if not args.no_facade_leg:
        legs.add('facade_n1')
    if not args.no_live_traffic and not args.unfused:
        legs.add('live_traffic')
    if not args.no_cpu_baseline:
        legs.add('cpu_baseline')
    return legs


def shard_of(rk, per_gpu, total_envs):
    """(first global env id, games, total) of this rank: per_gpu games per GPU (weak) or total_envs split over the ranks (strong)."""
    from stratego_env_amd.sharding import shard_range
    total = total_envs if total_envs else per_gpu * rk.world
    return shard_range(total, rk.rank, rk.world) + (total,)


def timed_steps(rk, run_warmup, run_timed, counters, settle=None):
    """Warm up, then time run_timed() between barrier + synchronize brackets.  Returns (elapsed seconds MAX over ranks,
    device ms MAX over ranks or None, elapsed seconds MIN over ranks, summed counter deltas).  settle(): untimed, state-preserving GPU
    work that directly precedes the bracket (gpu_settle)."""
    run_warmup()
    before = counters()
    if settle:
        if rk.world > 1:
            rk.barrier()        # the ranks settle side by side, so that none of them idles at the bracket's barrier for long
        settle()
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
    rk.sync()
    own = time.perf_counter() - t0            # this rank's own K steps (no waiting for the others)
    rk.barrier()
    elapsed = time.perf_counter() - t0
    dev_ms = ev[0].elapsed_time(ev[1]) if ev else 0.0
    after = counters()
    (elapsed, dev_ms, neg_own, max_own), sums = rk.reduce([elapsed, dev_ms, -own, own], [a - b for a, b in zip(after, before)] + [1])
    assert sums[-1] == rk.world, "reduction covered %d ranks of %d" % (sums[-1], rk.world)
    return elapsed, (dev_ms if ev else None), (-neg_own, max_own), sums[:-1]


class _StubEnv:
    """--dry-run only: stands in for VecStrategoEnv so that the launcher, rendezvous, sharding, reductions, solo anchors and the leg
    structure of a multi-GPU run can be exercised without a GPU (tests/test_bench_launcher_cpu.py).  It plays no game; the run reports
    value null."""

    class _Zero:
        @staticmethod
        def sum():
            return 0

    def __init__(self, first, n):
        self.first, self.num_envs, self.steps_done, self.bench_steps_played = first, n, 0, 0
        self.invalid_action, self.ring_sets = self._Zero(), 1
        self.build_id, self.record_bytes = 'dry-run', 0

    def sample_valid_actions(self):
        pass

    def rollout_steps(self, k, chains=1, ring=False):
        time.sleep(0.0005 * k)
        self.steps_done += k

    def rollout_step(self):
        self.rollout_steps(1)

    def counters(self):
        return [self.steps_done * self.num_envs, 0]

    def snapshot(self):
        snap = _StubEnv(self.first, self.num_envs)
        snap.steps_done = self.steps_done
        return snap

    def restore(self, snap):
        self.steps_done = snap.steps_done

    def close(self):
        pass


DRY_RUN = False


def make_env(version, n, first, device_index, full_obs=False, compact=False):
    if DRY_RUN:
        return _StubEnv(first, n)
    from stratego_env_amd.vec_env import VecStrategoEnv
    # (placement='plain': the line reports the plain first allocation as frac_untuned and runs its own, larger search -- place_outputs)
    env = VecStrategoEnv(version, n, device=device_index, seed=BASE_SEED, env_id_offset=first, auto_reset=True, full_obs=full_obs,
                         compact_outputs=compact, placement='plain')
    env.reset()
    env.bench_steps_played = 0               # rollout steps since reset(): what the oracle replays in verify_against_oracle
    return env


SETTLE_SECONDS = 0.1


def gpu_settle(env, seconds):
    """Untimed and state-preserving: `seconds` of back-to-back sgx_observe launches (the step kernel's own stores into the same buffers;
    nothing is played) directly before the barrier + synchronize bracket of the timed region.  A short run does not reach the steady
    rate otherwise: after 5 to 20 warm-up steps (and the few small kernels and the device-to-host copy of the counters read) the next 20
    launches take 277-297 us instead of the 259-262 us of launches 64 ... 512 -- a burst of step launches that starts from a lightly
    loaded GPU runs ~10 launches fast, ~15-40 slow, then settles (kernel trace: tools/trace_series.py; idle time before a burst:
    tools/idle_burst.py, <= 3 ms harmless, >= 10 ms +5 %; tools/early_burst.py: the same 20 launches at turns 6-25 take 260 us directly
    after a series of observe launches, 286 us after 5 step launches).  With this, `--steps 20 --warmup 5` reads 263.0 us per launch
    (247 M steps/s; 277-286 us = 227-236 M without) and the default 512 / 64 259.5 us (260.5-262.1 us without): the steady rate either
    way.  The timed region is still exactly K steps after W warm-up steps."""
    import torch
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            env.observe()
        torch.cuda.synchronize()


MULTI_STEP_TALLY = {"launches": 0, "steps": 0}      # multi-step launches of this process and the steps they played (config.multi_step_tally)


def _tally(env, k):
    """After a rollout call of k steps: count it if it went out as multi-step launches (a kernel trace's total duration of steps_kernel /
    lane_steps_kernel divided by the tally's steps is the per-step time the line reports)."""
    if DRY_RUN or k <= 0 or not fused_launch(env):
        return
    MULTI_STEP_TALLY["launches"] += -(-k // WSTEPS_MAX_PER_LAUNCH)         # (both multi-step kernels go out in launches of at most 256 steps)
    MULTI_STEP_TALLY["steps"] += k


def time_workload(rk, env, steps, warmup, unfused=False, chains=1, ring=False, settle=True):     # noqa: C901
    """(elapsed s, device ms, (min, max) of the ranks' own seconds, games finished, invalid actions) of `steps` batched steps on
    `env`, MAX / SUM over ranks.  ring: the steps write the env's ring of output sets in turn (env.alloc_output_ring).
    settle=False: no gpu_settle between the warm-up steps and the bracket (config.no_settle)."""
    import torch

    def one_step():
        if unfused:
            env.step(env.next_actions, want_next_actions=False)
            env.sample_valid_actions()
        else:
            env.rollout_step()

    def run_warmup():
        env.sample_valid_actions()
        if ring:
            env.rollout_steps(warmup, ring=True)
            _tally(env, warmup)
        elif not unfused:                      # (the chains' streams are created on first use: not inside the timed region)
            env.rollout_steps(warmup, chains=chains)
            _tally(env, warmup)
        else:
            for _ in range(warmup):
                one_step()

    def run_timed():
        if unfused:
            for _ in range(steps):
                one_step()
        else:
            env.rollout_steps(steps, chains=chains, ring=ring)    # the same K batched steps, enqueued by one C-ABI call (sgx_step_n / sgx_rollout / sgx_step_ring)
            _tally(env, steps)

    def counters():
        if DRY_RUN:
            return env.counters()
        return [int(env.env_info()[:, 1].to(torch.int64).sum()), 0]

    elapsed, dev_ms, own, (games, _) = timed_steps(rk, run_warmup, run_timed, counters,
                                                   settle=(lambda: gpu_settle(env, SETTLE_SECONDS)) if (settle and SETTLE_SECONDS > 0 and not DRY_RUN) else None)
    env.bench_steps_played += warmup + steps
    _, (invalid,) = rk.reduce([], [int(env.invalid_action.sum())])
    return elapsed, dev_ms, own, games, invalid


def verify_against_oracle(env, version, n_check, both=False):
    """Ties the number to verified outputs: `n_check` sampled envs of the env object that has just been timed -- the first and the
    last of the rank plus an even spread -- must hold, after env.bench_steps_played steps, exactly what the CPU oracle holds
    after replaying that many steps of the same global env ids: the last step's mask / observation(s) / rewards / flags (one
    FNV digest, so_rollout_ex's `last_digests`) and the turn / game / game-over / player counters.  Raises on any difference.
    Returns the number of envs checked."""
    import numpy as np
    import torch
    n_check = min(int(n_check), env.num_envs)
    if n_check <= 0 or env.bench_steps_played <= 0:
        return 0
    orc, cv = oracle_variant(version)
    ids = np.unique(np.concatenate([[0, env.num_envs - 1], np.linspace(0, env.num_envs - 1, n_check).astype(np.int64)]))[:max(n_check, 2)]
    idx = torch.from_numpy(ids).to(env.device)
    if getattr(env, 'compact', False):             # compact outputs: what the decode ops make of them is what must equal the oracle
        mk, ob = env.decode_mask()[idx].cpu().numpy(), env.decode_obs()[idx].cpu().numpy()
    else:
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
    fo = env.fobs[idx].cpu().numpy() if both else None
    rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
    ei, info = env.ending_invalid[idx].cpu().numpy(), env.env_info()[idx].cpu().numpy()
    cores = usable_cores()
    for i, e in enumerate(ids):
        r = orc.rollout_ex(cv, BASE_SEED, env.env_id_offset + int(e), 1, env.bench_steps_played, both=both, threads=1)
        got = orc.step_digest(mk[i], ob[i], rw[i], dn[i], pl[i], ei[i], fobs=None if fo is None else fo[i])
        if int(r['last_digests'][0]) != got or not np.array_equal(r['info'][0], info[i]):
            raise SystemExit("bench.py: env %d (global id %d) differs from the CPU oracle after %d steps (oracle turn/game/over/player %s, "
                             "GPU %s)" % (int(e), env.env_id_offset + int(e), env.bench_steps_played, r['info'][0].tolist(), info[i].tolist()))
    del cores
    return len(ids)


def outputs_checksum(env):
    """Checksum of checksums over ALL envs of this rank (size-independent property: the sum over ranks does not depend on the
    sharding): per env a 40-bit mix of its observation words, mask bytes, rewards and flags, summed."""
    import torch
    n = env.num_envs
    tot = torch.zeros((), dtype=torch.int64, device=env.device)
    chunk = 8192
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        o = env.obs[a:b].reshape(b - a, -1).view(torch.int32).to(torch.int64)
        w = (torch.arange(o.shape[1], device=env.device, dtype=torch.int64) * 2654435761 + 12345) & 0xFFFF
        c = (o * (w + 1)).sum(dim=1)
        m = env.mask[a:b].reshape(b - a, -1).to(torch.int64)
        wm = (torch.arange(m.shape[1], device=env.device, dtype=torch.int64) * 40503 + 7) & 0xFFFF
        c = c + (m * (wm + 1)).sum(dim=1) * 3
        c = c + env.reward[a:b].view(torch.int32).to(torch.int64).sum(dim=1) * 5 + env.done[a:b].to(torch.int64) * 7
        c = c + env.player[a:b].to(torch.int64) * 11 + env.next_actions[a:b].to(torch.int64) * 13
        tot = tot + (c & ((1 << 40) - 1)).sum()
    return int(tot)


def ranks_sharing_device(args):
    """How many local ranks allocate on this rank's GPU: 1 unless --devices maps several ranks onto one device."""
    if not args.devices:
        return 1
    dmap = [int(x) for x in args.devices.split(',')]
    me = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    return max(1, dmap.count(dmap[me])) if me < len(dmap) else 1


def placement_budgets(args, budget):
    """(first-pass budget, wide-pass budget) of the placement search in bytes, both capped by what is FREE on this rank's device:
    every rank searches on its own and, while it does, holds up to its budget beyond the buffers it keeps (the wide pass' pads
    reached 54 GB for a moment in round 3) -- at most --placement-free-fraction (default 0.4) of the free device memory divided by the
    ranks that share the device, so that eight ranks' searches cannot exhaust a GPU between them whatever else lives on it."""
    import torch
    free, _total = torch.cuda.mem_get_info()
    cap = int(free * args.placement_free_fraction / ranks_sharing_device(args))
    wide = int(getattr(args, 'placement_wide_gb', 0.0) * (1 << 30))
    return min(budget, cap), min(wide, cap)


def place_outputs(env, args):
    """Library-owned output buffers from sgx_alloc_outputs' bounded placement trial, unless --placement plain.
    -> {'candidates', 'plain_us' (the allocation a caller would have got first), 'kept_us', 'median_us', 'max_us', 'peak_extra_gb'}."""
    if args.placement != 'trial' or getattr(env, 'compact', False):
        return None
    # No candidate of the fast class (>= 14 % below the slowest) in the first budget: tune_placement's second pass
    # samples a much wider range with the same number of candidates and is kept only if it found something faster.  (One box: 32
    # candidates at 338-340 us within 8 GiB, 275.9 us in the 64 GB pass.)
    # (a 7 GB observation buffer -- 262,144 games -- would have 1 GB of the default budget left for its candidates: four buffer sizes then)
    budget = max(int(args.placement_gb * (1 << 30)), 4 * env.obs.numel() * 4) if args.placement_gb > 0 else 0
    if env.obs.numel() * 4 <= 300e6 and env.fobs is None:
        budget = 0     # a launch whose observations fit the Infinity Cache has no placement classes (toy boards: 64 candidates within 1 %): no search
    budget, wide = placement_budgets(args, budget)
    rep = env.tune_placement(args.placement_trials, max_extra_bytes=budget, wide_extra_bytes=wide)
    t = rep.get('obs') or []
    out = {"candidates": len(t), "peak_extra_gb": round(getattr(env, 'placement_peak_extra_bytes', 0) / 2.0 ** 30, 2)}
    if t:
        out.update({"plain_us": round(t[0], 1), "kept_us": round(min(t), 1), "median_us": round(sorted(t)[len(t) // 2], 1),
                    "max_us": round(max(t), 1)})
    if rep.get('fobs'):
        out["fobs_plain_us"], out["fobs_kept_us"] = round(rep['fobs'][0], 1), round(min(rep['fobs']), 1)
    w = rep.get('wide')
    if w:
        t2 = w['obs']
        out["wide_pass"] = {"budget_gb": args.placement_wide_gb, "candidates": len(t2), "kept_us": round(min(t2), 1) if t2 else None,
                            "max_us": round(max(t2), 1) if t2 else None, "peak_extra_gb": round(w['peak_extra_bytes'] / 2.0 ** 30, 2),
                            "used": w['used']}
        if w['used']:
            out["kept_us"] = round(min(t2), 1)
    return out


def roofline(version, v, n, launch_s, traffic_override=None, full_obs=False, first_us=None, rec_bytes=None, build_id=None,
             rotating=None, ring_sets=1, fused_steps=1):
    """The roofline object of one workload.  Every `frac*` is B_min x games / time / 8 TB/s -- bytes the kernel cannot avoid moving
    (b_min), so none of them overstates the traffic; `frac_dram` comes from the rotating-outputs leg (`rotating` = its launch
    seconds), where the Infinity Cache cannot hold anything back from DRAM."""
    key = version + ('+full_obs' if full_obs else '')
    if ring_sets > 1 and measured_traffic(key + '+rotating', n, build_id)[0]:
        key += '+rotating'
    per_step = b_min(v, full_obs, rec_bytes, fused_steps)
    min_bytes = per_step * n
    traffic, source = (traffic_override, "--traffic-bytes") if traffic_override is not None else measured_traffic(key, n, build_id)
    ach = min_bytes / launch_s / 1e9
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "frac_basis": "B_min = %d B per env step (packed record in + out, action in, next action out, float32 observation%s, uint8 mask, "
                         "results) x %d games per launch / launch time by HIP events; %s" % (
                             per_step, "s" if full_obs else "", n,
                             "outputs written round-robin into %d sets, so that no line a launch writes can still be cached when it is written "
                             "again: DRAM side" % ring_sets if ring_sets > 1 else
                             "outputs written in place (one set of tensors): memory side including the 256 MiB Infinity Cache where a set fits it"),
           "bytes_per_launch": min_bytes, "b_min_bytes_per_step": per_step,
           "kernel": ("%s<%d,%d%s> (multi-step launches of %g steps: launch_us and the bytes are per STEP; a kernel trace's duration of this kernel / its steps "
                      "is the same figure)" % ("lane_steps_kernel" if v.rows * v.columns <= 16 and (v.rows * v.columns) % 4 == 0 else "steps_kernel", v.rows, v.columns,
                                               "" if v.rows * v.columns <= 16 and (v.rows * v.columns) % 4 == 0 else (",1" if full_obs else ",0"), fused_steps)) if fused_steps > 1
                     else "step_kernel<%d,%d,%d,false>" % (v.rows, v.columns, 1 if full_obs else 0), "launch_us": launch_s * 1e6, "steps_per_launch": fused_steps,
           # (a multi-step launch on well-placed buffers reads 1.00-1.02 here: the guide's 8 TB/s is the data sheet's rounded figure; the pins' own
           #  rate is 8,192 GB/s, and the bytes are the run's own counter bytes -- `traffic`, rocprofv3 WRITE_SIZE + 2 x FETCH_SIZE)
           "peak_pin_rate": HBM_PIN_RATE_GBS, "frac_of_pin_rate": ach / HBM_PIN_RATE_GBS, "hbm_mclk_mhz": hbm_mclk_mhz(),
           "frac_dram": None, "traffic": traffic, "traffic_source": source,
           "traffic_over_b_min": (traffic / min_bytes) if traffic else None,
           # the same kernel writing the allocation the process got first (observe launch of the placement trial's first candidate):
           # what an integrator who passes plain torch.empty tensors may get
           # (a single launch of the per-step kernel: its own byte minimum, the record included)
           "frac_untuned": (b_min(v, full_obs, rec_bytes) * n / (first_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if first_us else None,
           # SURVEY 8d's formula (state as 32 dense int8 boards): a labelled comparison, not a fraction -- the kernel reads a 512-byte
           # record instead of 3,216 B of boards, so this rate counts bytes that are never moved
           "survey_8d": {"bytes_per_step": b_alg(v.rows, v.columns, full_obs), "gbps_if_those_bytes_moved": b_alg(v.rows, v.columns, full_obs) * n / launch_s / 1e9}}
    if rotating:
        out["frac_dram"] = min_bytes / rotating / 1e9 / HBM_PEAK_GBS
        out["frac_dram_basis"] = "B_min bytes over the launch time of a run that writes a ring of output sets (the headline itself, or the leg's rotating_outputs)"
    return out


WSTEPS_MAX_PER_LAUNCH = 256       # SGX_STEPS_MAX_PER_LAUNCH: a rollout call of the multi-step kernels goes out in launches of at most this many steps


def fused_launch(env):
    """True if the env's last rollout ran as multi-step launches (sgx_last_launch_kind): the games stay on the chip between the steps of a
    launch -- registers on boards of at most 16 cells (lane_steps_kernel), LDS elsewhere (steps_kernel) -- and the record travels once per
    LAUNCH."""
    from stratego_env_amd import _lib
    return (not DRY_RUN) and env.last_launch_kind in (_lib.LAUNCH_MULTI_STEP, _lib.LAUNCH_MULTI_STEP_WAVE)


def fused_steps_of(env, steps):
    """Steps per launch of the env's last rollout of `steps` steps (1 = one launch per step): what the per-step byte minimum divides the
    record traffic by.  Both multi-step kernels chunk a call into launches of at most WSTEPS_MAX_PER_LAUNCH steps."""
    if DRY_RUN or not fused_launch(env):
        return 1
    n_launches = -(-steps // WSTEPS_MAX_PER_LAUNCH)
    return steps / float(n_launches)


def solo_anchor(rk, env, steps, warmup, **kw):
    """The per-GPU workload of a multi-GPU leg timed on rank 0 ALONE -- same process, same env object, same output buffers, same K / W,
    the other ranks parked at the host-side gloo barrier with no GPU work -- directly before the ranks run it side by side.  The games
    are put back where they were afterwards (env.snapshot / restore: records, counters, game numbers), so the side-by-side run and its
    sharding-independent checksum are those of a run without the anchor.  Every rank calls this; -> {value, launch_us, ms_per_step}
    on rank 0, None elsewhere (and None on one GPU, where the headline is its own anchor)."""
    if rk.world == 1:
        return None
    out = None
    if rk.rank == 0:
        snap, played = env.snapshot(), env.bench_steps_played
        elapsed, dev_ms, _, games, invalid = time_workload(rk.solo(), env, steps, warmup, **kw)
        assert invalid == 0
        out = {"value": env.num_envs * steps / elapsed, "unit": "env steps/s", "games": env.num_envs, "ms_per_step": elapsed / steps * 1e3,
               "launch_us": (dev_ms or 0.0) / steps * 1e3,
               "how": "rank 0 alone (the other ranks parked at a host-side barrier), same env object and buffers, same --steps / --warmup, "
                      "directly before the side-by-side run; the games were put back afterwards (snapshot / restore)"}
        env.restore(snap)
        env.bench_steps_played = played
        snap.close()
    rk.barrier()
    return out


def scaling_fields(value_all, own, n, steps, solo, world):
    """{scaling_x, per_gpu_min_over_solo, ...} of a multi-GPU leg against its solo anchor (rank 0 only has one)."""
    if not solo:
        return {"solo": None, "scaling_x": None, "per_gpu_value_min_over_solo": None}
    return {"solo": solo, "scaling_x": value_all / solo["value"], "scaling_x_ideal": world,
            "per_gpu_value_min_over_solo": (n * steps / own[1]) / solo["value"]}


def ring_for(env, args, n_sets):
    """alloc_output_ring with the bench's placement budgets -> the per-extra-set (plain, kept) us report."""
    budget, wide = placement_budgets(args, max(int(args.placement_gb * (1 << 30)), 4 * env.obs.numel() * 4) if (args.placement == 'trial' and args.placement_gb > 0) else 0)
    reps = env.alloc_output_ring(n_sets, tune=budget >= env.obs.numel() * 4 and env.obs.numel() * 4 > 300e6, max_extra_bytes=budget,
                                 trials=args.placement_trials, wide_extra_bytes=wide)
    return reps


def scaling_leg(rk, args, per_gpu, total_envs, label, output_sets=1):
    """Another games-per-GPU size of the multi-GPU run, same K / W as the headline, anchored like it: rank 0 alone first (solo_anchor),
    then all ranks side by side: {value, ms_per_step, solo, scaling_x, ...}.  output_sets > 1: a ring of output sets like the 1-GPU
    headline's."""
    import torch
    first, n, total = shard_of(rk, per_gpu, total_envs)
    env = make_env(args.version, n, first, rk.device_index)
    try:
        ring = output_sets >= 2
        if not DRY_RUN:
            place_outputs(env, args)
            if ring:
                ring_for(env, args, output_sets)
        solo = solo_anchor(rk, env, args.steps, args.warmup, ring=ring)
        elapsed, dev_ms, own, games, invalid = time_workload(rk, env, args.steps, args.warmup, ring=ring)
        assert invalid == 0
        checked = verify_against_oracle(env, args.version, min(args.verify_envs, 8)) if (args.verify_envs and not DRY_RUN) else 0
        _, (checked,) = rk.reduce([], [checked])
        out = {"workload": label, "scaling": "strong" if total_envs else "weak", "total_games": total, "games_per_gpu": n,
               "output_sets": output_sets if ring else 1,
               "value": total * args.steps / elapsed, "unit": "env steps/s", "ms_per_step": elapsed / args.steps * 1e3,
               "per_gpu_value_min": n * args.steps / own[1], "per_gpu_value_max": n * args.steps / own[0],
               "launch_us": (dev_ms or 0.0) / args.steps * 1e3, "verified_envs": checked}
        out.update(scaling_fields(out["value"], own, n, args.steps, solo, rk.world))
        return out
    finally:
        env.close()
        del env
        if not DRY_RUN:
            torch.cuda.empty_cache()


def optional_leg(fn, *a, **kw):
    """One of the line's EXTRA legs (other workloads, compact outputs, trajectory buffer, facade).  A leg that cannot run on this box -- out of
    device memory next to another tenant, say -- reports {"failed": ...} and leaves the headline its line; a leg whose outputs differ from the
    oracle still ends the run (verify_against_oracle raises SystemExit, which is not caught here)."""
    try:
        return fn(*a, **kw)
    except Exception as e:      # noqa: BLE001
        import torch
        torch.cuda.empty_cache()
        return {"failed": "%s: %s" % (type(e).__name__, str(e)[:300]), "workload": "%s%r" % (fn.__name__, tuple(x for x in a[3:]))}


def line_summary(out):
    """The figures of the legs once more, as the LAST key of the line: whoever keeps only the tail of this (long) line still sees BASELINE
    configs 3 and 4, config 5's per-GPU size, the trajectory and facade legs and the store-only probe next to the headline."""
    def pick(d, *keys):
        return {k: (round(d[k], 4) if isinstance(d.get(k), float) else d.get(k)) for k in keys + ("failed",) if d and k in d} if d else None
    c, rf = out.get("config") or {}, out.get("roofline") or {}
    ow = c.get("other_workloads") or []
    names = ("config3_standard_262144", "config4_micro_65536", "both_observations_65536", "config5_per_gpu_size_barrage_262144")
    s = {"value": out.get("value"), "value_one_launch_per_step": out.get("value_one_launch_per_step"), "n_gpus": out.get("n_gpus"),
         "frac": rf.get("frac"), "frac_untuned": rf.get("frac_untuned"), "traffic_over_b_min": rf.get("traffic_over_b_min"),
         "store_peak_measured_gbps": rf.get("store_peak_measured"), "frac_of_store_peak": rf.get("frac_of_store_peak"),
         "verified_envs": out.get("verified_envs"), "scaling_x": c.get("scaling_x")}
    for name, w in zip(names, ow):
        s[name] = pick(w, "value", "launch_us", "frac", "verified_envs")
        if w and w.get("rotating_outputs"):
            s[name]["ring_of_3"] = pick(w["rotating_outputs"], "value", "launch_us", "frac_dram")
    s["in_place"] = pick(c.get("in_place"), "value", "launch_us", "rate_over_spec_peak", "one_launch_per_step")
    s["trajectory"] = pick(c.get("trajectory"), "slots", "value", "launch_us", "frac", "one_launch", "verified_envs")
    if s["trajectory"] and ((c.get("trajectory") or {}).get("ring_of_separately_placed_sets") or {}).get("value"):
        s["trajectory"]["ring_of_separately_placed_sets"] = pick(c["trajectory"]["ring_of_separately_placed_sets"], "sets", "value", "launch_us", "frac", "one_launch", "same_memory")
    s["compact_outputs"] = pick(c.get("compact_outputs"), "value", "launch_us", "frac", "trajectory_64_slots")
    s["as_it_comes"] = pick(c.get("as_it_comes"), "value", "launch_us", "frac", "in_place_one_launch_per_step", "verified_envs")
    s["consumer_in_loop"] = pick((c.get("consumer_in_loop") or {}).get("nt_stores"), "value", "step_kernel_us_in_loop")
    s["facade_n1"] = pick(c.get("facade_n1"), "steps_per_s", "env_step_calls_per_s")
    s["scaling_legs"] = [pick(l, "games_per_gpu", "value", "scaling_x", "per_gpu_value_min_over_solo") for l in (c.get("scaling_legs") or [])] or None
    cb = out.get("cpu_baseline") or {}
    s["cpu_baseline"] = pick(cb, "value", "cores", "kind")
    return s


def run_rank(args):      # noqa: C901
    global DRY_RUN, SETTLE_SECONDS
    dry = DRY_RUN = args.dry_run
    if not dry:
        import torch
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    backend = args.backend or ('gloo' if dry else 'nccl')
    rk = Rank(args.gpus, backend, use_cuda=not dry, devices=args.devices)
    backend = rk.backend if not rk.backend_note else "%s (nccl failed: %s)" % (rk.backend, rk.backend_note)
    first, n, total = shard_of(rk, args.envs, args.total_envs)
    if n <= 0:
        raise SystemExit("bench.py: rank %d got no games (%d games over %d ranks)" % (rk.rank, total, rk.world))
    SETTLE_SECONDS = args.settle_seconds
    legs_on = legs_for(rk.rank, rk.world, args, dry)
    if not dry:
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
    env = make_env(args.version, n, first, rk.device_index, full_obs=args.full_obs)
    build_id, rec_bytes = env.build_id, env.record_bytes
    placement = place_outputs(env, args) if not dry else None
    headline_ring = args.output_sets >= 2 and not args.unfused and args.chains == 1
    ring_report = None
    if headline_ring and not dry:      # the headline writes a ring of output sets: each extra set from its own placement trial, like the first
        reps = ring_for(env, args, args.output_sets)
        ring_report = [(round(r['obs'][0], 1), round(min(r['obs']), 1)) if (r and r.get('obs')) else None for r in reps[1:]]
        if reps[0]:            # the env's own set was slower than the extra ones and was searched again against them (alloc_output_ring)
            placement = dict(placement or {})
            placement["first_set_searched_again"] = {"before_us": round(reps[0]['before_us'], 1), "target_us": round(reps[0]['target_us'], 1),
                                                     "candidates": len(reps[0]['obs']), "kept_us": round(min(reps[0]['obs']), 1) if reps[0]['obs'] else None,
                                                     "used": reps[0]['used']}
            if reps[0]['used']:
                placement["kept_us"] = round(min(reps[0]['obs']), 1)
    # several GPUs: rank 0 times this per-GPU workload ALONE first (the others parked, no GPU work), so that the line carries its own anchor
    solo = solo_anchor(rk, env, args.steps, args.warmup, unfused=args.unfused, chains=args.chains, ring=headline_ring)
    elapsed, dev_ms, own, games, invalid = time_workload(rk, env, args.steps, args.warmup, args.unfused, args.chains, ring=headline_ring)
    assert invalid == 0, "rollout produced invalid actions"
    fused = fused_steps_of(env, args.steps) if not dry else 1           # multi-step launches: the record travels once per launch, not per step
    # outside the timed region: the envs that were just timed against the CPU oracle, and the sharding-independent checksum
    checked = verify_against_oracle(env, args.version, args.verify_envs, both=args.full_obs) if (args.verify_envs and not dry) else 0
    verified_steps = env.bench_steps_played
    _, (checked, checksum, covered) = rk.reduce([], [checked, outputs_checksum(env) if not dry else 0, n])
    in_place, two_chains, no_settle = None, None, None
    per_step = b_min(v, args.full_obs, rec_bytes, fused)
    if 'no_settle' in legs_on:
        # what gpu_settle is worth: the same K steps on the same env object and buffers once more, W warm-up steps straight into the bracket
        # (every rank takes part: same barriers)
        e0, d0, _, _, inv0 = time_workload(rk, env, args.steps, args.warmup, args.unfused, args.chains, ring=headline_ring, settle=False)
        assert inv0 == 0
        no_settle = {"workload": "the headline's K steps once more WITHOUT gpu_settle: %d warm-up steps, then the bracket" % args.warmup,
                     "value": total * args.steps / e0, "unit": "env steps/s", "launch_us": d0 / args.steps * 1e3,
                     "frac": per_step * n / (d0 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS}
    per_step_launches = None
    if fused > 1 and 'one_launch_per_step' in legs_on:
        # ... and with one launch per step (sgx_set_multi_step(0): what every round before this one measured), same env object and buffers
        env.set_multi_step(False)
        e4, d4, _, _, inv4 = time_workload(rk, env, args.steps, args.warmup, args.unfused, args.chains, ring=headline_ring)
        env.set_multi_step(True)
        assert inv4 == 0
        per_step_launches = {"workload": "the headline's K steps once more as K launches of the per-step kernel (step_kernel)", "value": total * args.steps / e4,
                             "unit": "env steps/s", "launch_us": d4 / args.steps * 1e3,
                             "frac": b_min(v, args.full_obs, rec_bytes) * n / (d4 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS}
    if 'in_place' in legs_on:
        # The same K / W on the same env object into ONE set of tensors (the set the ring wrote last), step after step: what rounds 1-3
        # reported as the headline.  Rewriting the same 1-2 GB back to back is 8-10 % faster than anything that cannot reuse its lines
        # (DESIGN.md section 4.1), so its rate is a memory-side figure that can touch the 8 TB/s spec peak: a ratio, not a DRAM fraction.
        e1, d1, _, g1, inv1 = time_workload(rk, env, args.steps, args.warmup)
        assert inv1 == 0
        in_place = {"workload": "the same rollout writing one set of output tensors in place", "value": total * args.steps / e1,
                    "unit": "env steps/s", "launch_us": d1 / args.steps * 1e3, "games_finished_in_timed_region": g1,
                    "rate_over_spec_peak": per_step * n / (d1 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                    "rate_is": "memory side including the 256 MiB Infinity Cache and whatever else favours rewriting the same lines; not a DRAM fraction",
                    "verified_envs": verify_against_oracle(env, args.version, min(args.verify_envs, 8), both=args.full_obs) if args.verify_envs else 0,
                    "verified_steps": env.bench_steps_played}
        if fused > 1:
            # ... and as K launches of the per-step kernel into the same tensors: what env.step() with a policy between the steps does by default
            env.set_multi_step(False)
            e5, d5, _, _, inv5 = time_workload(rk, env, args.steps, args.warmup)
            env.set_multi_step(True)
            assert inv5 == 0
            in_place["one_launch_per_step"] = {"value": total * args.steps / e5, "launch_us": d5 / args.steps * 1e3,
                                               "rate_over_spec_peak": b_min(v, args.full_obs, rec_bytes) * n / (d5 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS}
    if 'two_chains' in legs_on:
        # The same K steps (in place) with the batch split into two ranges of games whose launches overlap (sgx_rollout, chains = 2)
        e2, d2, _, _, inv2 = time_workload(rk, env, args.steps, args.warmup, False, 2)
        assert inv2 == 0
        two_chains = {"chains": 2, "outputs": "in place", "value": total * args.steps / e2, "us_per_step": d2 / args.steps * 1e3,
                      "rate_over_spec_peak": per_step * n / (d2 / 1e3 / args.steps) / 1e9 / HBM_PEAK_GBS,
                      "verified_envs": verify_against_oracle(env, args.version, args.verify_envs, both=args.full_obs) if args.verify_envs else 0,
                      "verified_steps": env.bench_steps_played}

    store_probe = None
    if 'store_probe' in legs_on:
        # the step kernel's store stream without the game, on the very buffers the headline wrote (they are re-rendered afterwards)
        store_probe = bench_legs.store_probe_leg(sys.modules[__name__], env, per_step * n / ((dev_ms or 0.0) / 1e3 / args.steps) / 1e9)
        env.observe()

    out = None
    if rk.rank == 0:
        total_steps = total * args.steps
        launch_s = (dev_ms or 0.0) / 1e3 / args.steps                 # average device time per batched step (HIP events)
        rf = None
        if not dry:
            rf = roofline(args.version, v, n, launch_s, args.traffic_bytes, full_obs=args.full_obs,
                          first_us=(placement or {}).get('fobs_plain_us' if args.full_obs else 'plain_us'), rec_bytes=rec_bytes, build_id=build_id,
                          rotating=launch_s if headline_ring else None, ring_sets=args.output_sets if headline_ring else 1, fused_steps=fused)
            rf["in_place_rate_over_spec_peak"] = in_place["rate_over_spec_peak"] if in_place else None
            # what a caller with a policy BETWEEN the steps gets from the same buffers: one launch per step
            rf["frac_one_launch_per_step"] = per_step_launches["frac"] if per_step_launches else None
            if store_probe:
                rf.update(store_probe)
                sp = store_probe["store_peak_measured"]
                rf["frac_dram_basis"] = (
                    "B_min bytes over the HIP-event time of a run that writes a ring of output sets, over the spec peak: a rate at the device's memory "
                    "boundary, not a count of DRAM pin transfers.  The store-only kernel of the same store shape (store_probe: payloads, streams at once, "
                    "wave lifetime, rewriting, plain / non-temporal mix) takes at most %.0f GB/s from the same buffers in this process, so this launch runs "
                    "at %.3f of it: a store-only stream is NOT an upper bound for the game kernel, and where the rate exceeds the pins' 8,192 GB/s the "
                    "excess is unexplained (payload, cache residue and the reported clocks are excluded: DESIGN.md section 4.2)" % (sp, store_probe["frac_of_store_peak"] or 0.0))
        value = total_steps / elapsed
        out = {
            "metric": "env steps/sec", "value": None if dry else value, "unit": "env steps/s",
            "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong" if args.total_envs else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "none (launcher self-test)" if dry else "synthetic", "build_id": build_id,
            # `value` is a rollout call (all K steps known in advance: multi-step launches); with a consumer between the steps -- env.step(), a
            # policy, env.step() -- every step is a launch of its own: the same K steps on the same buffers that way
            "value_one_launch_per_step": per_step_launches["value"] if per_step_launches else None,
            "verified_envs": checked, "verified_steps": verified_steps,
            "config": {"workload": "%d concurrent %s games per GPU (%dx%d)%s, random-valid-action rollout with auto-reset, "
                                   "%s step+sample%s" % (n, args.version, v.rows, v.columns,
                                                         ", BOTH_OBSERVATIONS (67 + 79 channels)" if args.full_obs else "",
                                                         "separate" if args.unfused else "fused",
                                                         ", outputs written round-robin into %d sets (a trajectory buffer of the last %d steps)" % (args.output_sets, args.output_sets) if headline_ring else ", outputs written in place"),
                       "games_per_gpu": n, "total_games": total, "games_covered_by_ranks": covered, "version": args.version, "seed": BASE_SEED,
                       "arithmetic": "game logic on int8 / uint8 boards (dtype u8); outputs: float32 observation (85 % of the bytes), uint8 mask",
                       "games_finished_in_timed_region": games, "b_min_bytes_per_step": per_step, "steps_per_launch": fused,
                       "record_bytes": rec_bytes,
                       "untimed_before_bracket": "%d warm-up steps, then %.2f s of state-preserving sgx_observe launches (gpu_settle)" % (args.warmup, SETTLE_SECONDS),
                       "no_settle": no_settle, "one_launch_per_step": per_step_launches,
                       # multi-step launches of the WHOLE process so far and the steps they played (a kernel trace of this command: total
                       # duration of steps_kernel / lane_steps_kernel over these steps = the per-step time of the legs that used them)
                       "multi_step_tally": dict(MULTI_STEP_TALLY),
                       "output_sets": args.output_sets if headline_ring else 1, "ring_placement_plain_and_kept_us_per_extra_set": ring_report,
                       "concurrent_chains": args.chains, "in_place": in_place, "two_chains": two_chains,
                       "launched_by": os.environ.get('SGX_BENCH_LAUNCHER', 'external' if rk.world > 1 else 'direct'),
                       "reduction_backend": backend if rk.world > 1 else None, "reduction_bringup_s": round(rk.bringup_seconds, 2) if rk.world > 1 else None,
                       "legs_of_this_line": sorted(legs_on), "devices": args.devices,
                       "strong_leg_total_games": args.strong_total,
                       # the slowest / fastest rank's own K steps (no waiting for the others): weak scaling without a data-path
                       # collective loses nothing as long as these stay at the 1-GPU rate of the same games-per-GPU size
                       "per_gpu_value_min": n * args.steps / own[1], "per_gpu_value_max": n * args.steps / own[0],
                       "outputs_checksum": checksum,
                       # sgx_alloc_outputs' report: observe-launch time on the plain first allocation and on the candidate it kept
                       "placement": placement},
            "roofline": rf,
        }
        # several GPUs: the same per-GPU workload on rank 0 alone, and what the job makes of it (measured in this process, on this node)
        out["config"].update(scaling_fields(value, own, n, args.steps, solo, rk.world))
        if dry:
            out["dry_run"] = True
            out["launched_by"] = out["config"]["launched_by"]
            out["config"]["stub_steps_x_games"] = games
    env.close()
    del env
    if not dry:
        import torch
        torch.cuda.empty_cache()
    legs = None
    if not args.no_scaling_legs and rk.world > 1 and not args.total_envs and args.leg_envs:
        # the other per-GPU sizes of the scaling study (every rank takes part), each with its own solo anchor
        legs = [scaling_leg(rk, args, args.leg_envs, 0,
                            "%s%d games per GPU (%d on %d GPUs), outputs in place%s"
                            % ("BASELINE config 5: " if args.leg_envs == GAMES_PER_GPU_MULTI else "", args.leg_envs, args.leg_envs * rk.world, rk.world,
                               " (8 GB per set streams past every cache)" if args.leg_envs == GAMES_PER_GPU_MULTI else ""))]
        if args.strong_total and args.strong_total != args.leg_envs * rk.world:
            legs.append(scaling_leg(rk, args, 0, args.strong_total, "%d games in total split over the GPUs (BASELINE config 5's total, strong scaling)" % args.strong_total))
    elif args.strong_total and not args.no_scaling_legs:
        legs = [scaling_leg(rk, args, 0, args.strong_total, "%d games in total split over the GPUs (BASELINE config 5's total, strong scaling)" % args.strong_total)]
    if rk.rank == 0:
        out["config"]["scaling_legs"] = legs
        out["config"]["other_workloads"] = None
        out["config"]["consumer_in_loop"] = None
        if 'other_workloads' in legs_on:
            if 'consumer_in_loop' in legs_on:
                out["config"]["consumer_in_loop"] = bench_legs.consumer_leg(sys.modules[__name__], rk, args)
            B = sys.modules[__name__]
            # (first, while this process's device memory has seen only the headline's buffers: a learner allocates its 128 GB buffer early too;
            #  after the other legs' allocations the same buffer ran 10 % slower in one line -- profiles/r06_default_bench_line_run4.json)
            traj_leg = optional_leg(bench_legs.trajectory_leg, B, rk, args, slots=args.trajectory_slots) if 'trajectory' in legs_on else None
            out["config"]["other_workloads"] = [optional_leg(bench_legs.other_workload, B, rk, args, 'standard', 262144, chains=2),
                                                optional_leg(bench_legs.other_workload, B, rk, args, 'micro', 65536, chains=2, rotate_sets=args.rotate_sets),
                                                optional_leg(bench_legs.other_workload, B, rk, args, 'barrage', 65536, full_obs=True),
                                                # BASELINE config 5's per-GPU size on ONE GPU: the G = 1 anchor of the 1 / 2 / 4 / 8 curve
                                                optional_leg(bench_legs.other_workload, B, rk, args, 'barrage', GAMES_PER_GPU_MULTI)]
            out["config"]["compact_outputs"] = optional_leg(bench_legs.compact_leg, B, rk, args)
            out["config"]["trajectory"] = traj_leg
            out["config"]["as_it_comes"] = optional_leg(bench_legs.as_it_comes_leg, B, rk, args)
        if 'facade_n1' in legs_on:
            out["config"]["facade_n1"] = optional_leg(bench_legs.facade_leg, sys.modules[__name__])
        if 'live_traffic' in legs_on:
            # the counter bytes of the headline's kernel, measured now (after every timed region): two children under rocprofv3 --pmc
            rf = out["roofline"]
            rf["traffic_static"], rf["traffic_static_source"] = rf["traffic"], rf["traffic_source"]
            live, src = live_traffic(args.version, n, args.output_sets if headline_ring else 1, full_obs=args.full_obs)
            if live:
                rf["traffic"], rf["traffic_source"] = live, src
                rf["traffic_over_b_min"] = live / rf["bytes_per_launch"]
            else:
                rf["traffic_live_failed"] = src
        if 'cpu_baseline' in legs_on:         # the CPU leg is timed on rank 0 of the 1-GPU run only
            out["cpu_baseline"] = cpu_baseline(args.version, BASE_SEED, args.cpu_seconds)
        else:
            out["cpu_baseline"] = None
        out["summary_at_the_end_of_the_line"] = line_summary(out)
        print(json.dumps(out), flush=True)
    rk.close()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.traffic_probe:
        return traffic_probe(args)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))     # nothing in this process has touched the GPU (torch is not even imported)
    run_rank(args)


if __name__ == '__main__':
    main()
