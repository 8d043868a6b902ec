#!/bin/bash
# round 6, fourth GPU call: the store probe with persistent waves, inside the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_trajectory.py -x -q -k "barrage-48 or micro-130" 2>&1 | tail -2
timeout 1200 python bench.py --no-other-workloads --no-cpu-baseline --no-live-traffic > $O/bench_probe.json 2> $O/bench_probe.err
echo "bench rc $?"
tail -c 600 $O/bench_probe.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_probe.json') if l.startswith('{')][0])
rf = d['roofline']
print('value %.1fM one-launch-per-step %s frac %.3f achieved %.0f' % (d['value'] / 1e6, d.get('value_one_launch_per_step'), rf['frac'], rf['achieved']))
print('store_peak', rf.get('store_peak_measured'), 'frac_of_store_peak', rf.get('frac_of_store_peak'))
sp = rf.get('store_probe') or {}
print('best', sp.get('best_waves_per_cu_pace_persistent'))
for r in sp.get('streams_at_once_sweep_set0_observation_like', []):
    print('   ', r)
print(json.dumps(sp.get('gbps_by_set_and_payload'), indent=1))
print(json.dumps(sp.get('long_launch')))
PY
