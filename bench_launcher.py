"""bench.py's process plumbing (kept out of bench.py, which holds the contract, the headline and the roofline):

  launch_ranks     `python bench.py --gpus N` outside any launcher starts the N ranks itself as fresh child processes BEFORE anything touches the GPU
                   (the parent never imports torch; GPUs are counted from sysfs: visible_gpus)
  Rank             one rank of the job: device, the host-side gloo group (barriers of the timed bracket, agreement), the two reporting reductions
                   over RCCL when EVERY rank brought it up, else gloo for all
tests/test_bench_launcher_cpu.py exercises all of it over gloo with 2 / 3 / 4 / 8 ranks and simulated RCCL failures."""
import glob
import os
import socket
import subprocess
import sys
import time

BENCH_PY = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench.py')


# ---------------------------------------------------------------------------------------------------------------
# Launcher: `python bench.py --gpus N` outside any launcher starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Number of AMD GPUs this process could open, WITHOUT loading torch or the HIP runtime (a launcher parent must never
    initialise the GPU before it starts its ranks): KFD topology nodes with SIMDs, narrowed by *_VISIBLE_DEVICES."""
    n = 0
    for props in glob.glob('/sys/class/kfd/kfd/topology/nodes/*/properties'):
        try:
            for line in open(props):
                if line.startswith('simd_count') and int(line.split()[1]) > 0:
                    n += 1
        except Exception:
            pass
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(',') if x.strip() != '']))
    return n


def launch_ranks(args, argv):
    """Start one fresh `python bench.py` process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relay rank 0's
    stdout, return the first non-zero exit code (the other ranks are then terminated).  This process never initialises HIP
    and never imports torch: GPUs are counted from the KFD topology in sysfs."""
    n = args.gpus
    if not args.dry_run:
        have = visible_gpus()
        need = n if not args.devices else len(set(args.devices.split(',')))
        if have < need:
            print("bench.py: --gpus %d but only %d GPU(s) are visible" % (n, have), file=sys.stderr)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), SGX_BENCH_LAUNCHER='bench.py')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        procs.append(subprocess.Popen([sys.executable, BENCH_PY] + list(argv), env=env,
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
    silently started fewer ranks is an error, not a smaller run.

    Process groups.  The DEFAULT group is always gloo on the host: it carries the barriers of the timed bracket (ranks parked at a
    gloo barrier do no GPU work: the solo anchors of the scaling legs need that) and the agreement below.  The reporting reductions
    (one MAX, one SUM) run over RCCL (backend nccl, a second group on the same store: no second port) when -- and only when -- EVERY
    rank brought it up: (1) each rank checks what it can check alone (a GPU of its own; SGX_BENCH_FAIL_NCCL_RANKS simulates a failure)
    and the ranks all-gather the verdicts over gloo; (2) only if all passed do they create the nccl group and probe it with one
    all-reduce, and all-gather the outcome again.  One rank failing at either stage moves ALL ranks to gloo for the reductions, and
    the line says which rank and why (config.reduction_backend).  Nothing on the data path depends on any of this: the games never
    interact."""

    def __init__(self, gpus, backend, use_cuda, devices=None):
        import torch
        self.rank = int(os.environ.get('RANK', '0'))
        self.world = int(os.environ.get('WORLD_SIZE', '1'))
        self.local_rank = int(os.environ.get('LOCAL_RANK', str(self.rank)))
        if self.world != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (gpus, self.world))
        if not 0 <= self.rank < self.world:
            raise SystemExit("bench.py: RANK=%d outside WORLD_SIZE=%d" % (self.rank, self.world))
        self.device_index = self.local_rank
        dmap = None
        if devices:
            dmap = [int(x) for x in devices.split(',')]
            if self.local_rank >= len(dmap):
                raise SystemExit("bench.py: --devices lists %d devices, LOCAL_RANK=%d" % (len(dmap), self.local_rank))
            self.device_index = dmap[self.local_rank]
        self.use_cuda = use_cuda
        self.dist = None
        self.red_group = None                  # None = the default (gloo) group
        self.bringup_seconds = 0.0
        self.backend, self.backend_note = ('gloo' if self.world > 1 else backend), None
        self.reduce_device = 'cpu'
        if use_cuda:
            torch.cuda.set_device(self.device_index)
        if self.world > 1:
            import datetime
            import torch.distributed as dist
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29500')
            # (gloo announces its connections on stdout, which has to stay ONE JSON line: send that to stderr)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group('gloo', rank=self.rank, world_size=self.world, timeout=datetime.timedelta(seconds=600))
                dist.barrier()
                if dist.get_world_size() != gpus:
                    raise SystemExit("bench.py: process group has %d ranks, --gpus %d" % (dist.get_world_size(), gpus))
                self.dist = dist
                if backend == 'nccl':
                    t_up = time.perf_counter()
                    self._bring_up_rccl(torch, dist, dmap, datetime.timedelta(seconds=int(os.environ.get('SGX_BENCH_NCCL_TIMEOUT', '120'))))
                    self.bringup_seconds = time.perf_counter() - t_up         # (config.reduction_bringup_s: a first-time RCCL bring-up must not stall the job)
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    def _agree(self, dist, ok, note):
        """All ranks learn every rank's verdict (over gloo) -> (everybody ok, [(rank, note) of the ranks that failed])."""
        verdicts = [None] * self.world
        dist.all_gather_object(verdicts, (bool(ok), note))
        failed = [(r, v[1]) for r, v in enumerate(verdicts) if not v[0]]
        return not failed, failed

    def _bring_up_rccl(self, torch, dist, dmap, timeout):
        fake = os.environ.get('SGX_BENCH_FAKE_NCCL') == '1'      # CPU tests: a second gloo group stands in for RCCL
        fail_ranks = [int(x) for x in os.environ.get('SGX_BENCH_FAIL_NCCL_RANKS', '').split(',') if x.strip() != '']
        # ---- stage 1: what a rank can check on its own
        ok, note = True, None
        try:
            if self.rank in fail_ranks:
                raise RuntimeError("simulated failure (SGX_BENCH_FAIL_NCCL_RANKS)")
            if not fake:
                if not self.use_cuda:
                    raise RuntimeError("nccl needs a GPU per rank (dry run)")
                if dmap is not None and dmap.count(self.device_index) > 1 and os.environ.get('SGX_BENCH_SKIP_DEVICE_CHECK') != '1':
                    raise RuntimeError("device %d is shared by %d ranks (RCCL refuses that: invalid usage)" % (self.device_index, dmap.count(self.device_index)))
        except Exception as e:              # noqa: BLE001
            ok, note = False, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else '')
        all_ok, failed = self._agree(dist, ok, note)
        group = None
        if all_ok:
            # ---- stage 2: the collective bring-up, probed with one all-reduce
            try:
                if self.rank in [int(x) for x in os.environ.get('SGX_BENCH_FAIL_NCCL_STAGE2_RANKS', '').split(',') if x.strip() != '']:
                    raise RuntimeError("simulated failure inside the collective bring-up (SGX_BENCH_FAIL_NCCL_STAGE2_RANKS)")
                if fake:
                    group = dist.new_group(backend='gloo', timeout=timeout)
                    probe = torch.ones(1)
                else:
                    # (a collective that cannot complete must RAISE after the timeout, so that this rank joins the agreement below, instead of
                    #  having the watchdog abort the process: blocking wait; the group only ever carries two tiny reductions)
                    os.environ.setdefault('TORCH_NCCL_BLOCKING_WAIT', '1')
                    group = dist.new_group(backend='nccl', timeout=timeout)
                    probe = torch.ones(1, device='cuda')
                dist.all_reduce(probe, group=group)
                if not fake:
                    torch.cuda.synchronize()
                if int(probe.item()) != self.world:
                    raise RuntimeError("all_reduce of ones over %d ranks gave %r" % (self.world, probe.item()))
            except Exception as e:          # noqa: BLE001 -- whatever RCCL / the rendezvous raises
                ok, note = False, "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:160] if str(e) else '')
            all_ok, failed = self._agree(dist, ok, note)
        if all_ok:
            self.red_group = group
            self.backend = 'nccl (simulated by a second gloo group)' if fake else 'nccl'
            self.reduce_device = 'cpu' if fake else 'cuda'
            return
        self.backend_note = "; ".join("rank %d: %s" % (r, n) for r, n in failed[:4]) + (" (+%d more)" % (len(failed) - 4) if len(failed) > 4 else "")
        print("bench.py rank %d: nccl (RCCL) group not usable by every rank (%s); ALL ranks use gloo for the reporting reductions"
              % (self.rank, self.backend_note), file=sys.stderr, flush=True)
        if group is not None:
            try:
                dist.destroy_process_group(group)
            except Exception:               # noqa: BLE001
                pass
        self.backend, self.reduce_device, self.red_group = 'gloo', 'cpu', None

    def sync(self):
        if self.use_cuda:
            import torch
            torch.cuda.synchronize()

    def barrier(self):
        """barrier + device synchronize on both sides (the bench contract's bracket of the timed region).  The barrier is the host-side
        gloo one: a rank waiting in it puts no work on its GPU."""
        self.sync()
        if self.dist:
            self.dist.barrier()
        self.sync()

    def reduce(self, maxes, sums):
        """MAX over ranks of the float list `maxes`, SUM over ranks of the int list `sums`; the only collectives of the run."""
        if not self.dist:
            return list(maxes), list(sums)
        import torch
        t = torch.tensor(list(maxes), dtype=torch.float64, device=self.reduce_device)
        if len(maxes):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.red_group)
        c = torch.tensor(list(sums), dtype=torch.int64, device=self.reduce_device)
        if len(sums):
            self.dist.all_reduce(c, op=self.dist.ReduceOp.SUM, group=self.red_group)
        return [float(x) for x in t], [int(x) for x in c]

    def solo(self):
        """This rank on its own: same device, no process group -- what times a leg's per-GPU workload ALONE (the other ranks parked at
        the gloo barrier) before the ranks run it side by side."""
        return _SoloRank(self)

    def close(self):
        if self.dist:
            self.dist.destroy_process_group()


class _SoloRank:
    def __init__(self, rk):
        self.rank, self.world, self.local_rank, self.device_index, self.use_cuda, self.dist = rk.rank, 1, rk.local_rank, rk.device_index, rk.use_cuda, None
        self.sync = rk.sync

    def barrier(self):
        self.sync()

    def reduce(self, maxes, sums):
        return list(maxes), list(sums)
