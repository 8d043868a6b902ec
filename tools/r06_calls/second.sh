#!/bin/bash
# round 6, second GPU call: new tests again (after the fix), placement-lead probe, Micro ring experiments, driver-style bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_trajectory.py tests/test_gpu_multi_step.py tests/test_gpu_lane_kernel.py -x -q > $O/pytest_new.log 2>&1
echo "pytest rc $?" >> $O/pytest_new.log
tail -8 $O/pytest_new.log
timeout 600 python tools/r06_spread_probe.py > $O/spread_probe.log 2>&1
cat $O/spread_probe.log
SGX_ALLOW_FOREIGN_BUILD=1 timeout 600 python tools/lib_ab.py micro 65536 tools/_dev/micro_base.so tools/_dev/micro_e6w4.so tools/_dev/micro_e5w6.so tools/_dev/micro_u16.so tools/_dev/micro_u4.so tools/_dev/micro_sub16.so tools/_dev/micro_sub4.so --steps 256 --rounds 3 > $O/micro_variants_ab.log 2>&1
cat $O/micro_variants_ab.log | tail -30
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err
echo "bench rc $?"
tail -c 1000 $O/bench_driver_style.err
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
timeout 1200 bash tools/kstep_profile.sh r06_micro micro 65536 256 > $O/kstep_profile.log 2>&1
cp gpurun_out/prof_r06_micro/summary.txt $O/micro_ring_counters.txt
tail -60 $O/micro_ring_counters.txt
