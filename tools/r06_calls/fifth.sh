#!/bin/bash
# round 6, fifth GPU call: the store probe with rewriting waves (dwell / ring), on a 10x10-only build of the current sources
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
export SGX_LIB_PATH=tools/_dev/barrage_cur.so SGX_ALLOW_FOREIGN_BUILD=1
timeout 1200 python -W ignore bench.py --no-other-workloads --no-cpu-baseline --no-live-traffic --no-facade-leg > $O/bench_probe2.json 2> $O/bench_probe2.err
echo "bench rc $?"
tail -c 600 $O/bench_probe2.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_probe2.json') if l.startswith('{')][0])
rf = d['roofline']
print('value %.1fM one-launch-per-step %s frac %.3f achieved %.0f' % (d['value'] / 1e6, d.get('value_one_launch_per_step'), rf['frac'], rf['achieved']))
print('in place rate', rf.get('in_place_rate_over_spec_peak'))
print('store_peak', rf.get('store_peak_measured'), 'frac_of_store_peak', rf.get('frac_of_store_peak'))
sp = rf.get('store_probe') or {}
print('best', sp.get('best_waves_per_cu_pace_persistent'))
for r in sp.get("rewriting_waves_set0_observation_like", []) + sp.get("plain_and_non_temporal_mix_set0_observation_like", []):
    print('   ', r)
print(json.dumps(sp.get('long_launch')))
PY
