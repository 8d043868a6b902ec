#!/bin/bash
# round 6, third GPU call: the whole GPU suite on the new ABI, the mutant check, bench lines with the store-probe sweep
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
bash tools/mutant_check.sh > $O/mutant_check.log 2>&1
cat $O/mutant_check.log
timeout 900 python tools/facade_breakdown.py > $O/facade_breakdown.log 2>&1; grep -v "^/opt" $O/facade_breakdown.log
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc $?"
tail -c 600 $O/bench_default.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_default.json') if l.startswith('{')][0])
rf = d['roofline']
print('value %.1fM one-launch-per-step %s frac %.3f' % (d['value'] / 1e6, d.get('value_one_launch_per_step'), rf['frac']))
print('store_peak', rf.get('store_peak_measured'), 'frac_of_store_peak', rf.get('frac_of_store_peak'))
sp = rf.get('store_probe') or {}
print('best', sp.get('best_waves_per_cu_pace_persistent'))
for r in sp.get('streams_at_once_sweep_set0_observation_like', []):
    print('   ', r)
print(json.dumps(sp.get('gbps_by_set_and_payload'), indent=1))
print(json.dumps(sp.get('long_launch')))
print('trajectory', json.dumps(d['config'].get('trajectory'))[:1200])
print('facade', json.dumps(d['config'].get('facade_n1')))
print('basis', rf.get('frac_dram_basis'))
PY
