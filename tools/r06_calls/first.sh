#!/bin/bash
# round 6, first GPU call: the new per-step pinning tests of the multi-step kernels, then a driver-style bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_multi_step.py tests/test_gpu_lane_kernel.py -x -q > $O/pytest_new.log 2>&1
echo "pytest rc $?" >> $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 300 python tools/noobs_small_ab.py > $O/noobs_small_ab.log 2>&1
cat $O/noobs_small_ab.log
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "bench rc $?"
tail -c 1500 $O/bench_driver_style.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_driver_style.json') if l.startswith('{')][0])
rf = d['roofline']
print('value %.1fM one-launch-per-step %s frac %.3f' % (d['value'] / 1e6, d.get('value_one_launch_per_step'), rf['frac']))
print('store_peak', rf.get('store_peak_measured'), 'frac_of_store_peak', rf.get('frac_of_store_peak'))
print(json.dumps(rf.get('store_probe'), indent=1)[:3000])
print('trajectory', json.dumps(d['config'].get('trajectory'))[:800])
print('facade', json.dumps(d['config'].get('facade_n1')))
PY
