"""Instruction counts of the step kernel by phase: run under `rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS`;
dispatches alternate between the full step and steps with the observation / mask / both switched off (NULL output pointers).
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d OUT -- python3 tools/phase_insts.py medium
    python3 tools/phase_insts.py --parse OUT medium"""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ORDER = ['full', 'no obs', 'no mask', 'no obs, no mask', 'no obs, no mask, no sampler']
OBSERVE = ['observe: stage + mask generation only']


def run(version, n):
    import torch
    from stratego_env_amd.vec_env import VecStrategoEnv
    env = VecStrategoEnv(version, n, seed=3, auto_reset=True)
    env.reset()
    env.sample_valid_actions()
    for _ in range(30):
        env.rollout_step()
    for rep in range(3):
        for obs, mask, fused in ((True, True, True), (False, True, True), (True, False, True), (False, False, True), (False, False, False)):
            env.step(env.next_actions, want_next_actions=fused, emit_obs=obs, emit_mask=mask)
            if not fused:
                env.sample_valid_actions()
        env.observe(emit_obs=False, emit_mask=False)
    torch.cuda.synchronize()
    env.close()


def parse(out, n):
    rows = []
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        rows += list(csv.DictReader(open(f)))
    steps = {}
    for r in rows:
        if 'step_kernel' in r['Kernel_Name']:
            steps.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    obs_rows = {}
    for r in rows:
        if 'observe_kernel' in r['Kernel_Name']:
            obs_rows.setdefault(int(r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    last = [obs_rows[i] for i in sorted(obs_rows)[-3:]]
    print("%-30s per game: VALU %7.1f  SALU %7.1f  LDS %6.1f" % (OBSERVE[0], *(sum(s[c] for s in last) / len(last) / n for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS'))))
    ids = sorted(steps)[-15:]                      # the 15 measured dispatches come last
    for k, name in enumerate(ORDER):
        sel = [steps[i] for j, i in enumerate(ids) if j % 5 == k]
        avg = {c: sum(s[c] for s in sel) / len(sel) for c in sel[0]}
        print("%-30s per game: VALU %7.1f  SALU %7.1f  LDS %6.1f" % (name, avg['SQ_INSTS_VALU'] / n, avg['SQ_INSTS_SALU'] / n, avg['SQ_INSTS_LDS'] / n))


if __name__ == '__main__':
    if sys.argv[1] == '--parse':
        parse(sys.argv[2], int(sys.argv[4]) if len(sys.argv) > 4 else 65536)
    else:
        run(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 65536)
