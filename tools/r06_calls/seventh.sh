#!/bin/bash
# round 6, seventh GPU call: the tests added after the campaign (mask-only trajectories, the C example's trajectory mode, the facade floor), the
# multi-rank tests after the bench.py split, one default line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_trajectory.py tests/test_c_example.py tests/test_gpu_facade.py tests/test_gpu_two_ranks.py -x -q > $O/pytest_late.log 2>&1
echo "pytest rc $?" >> $O/pytest_late.log
tail -6 $O/pytest_late.log
timeout 1200 python bench.py > $O/bench_default_late.json 2> $O/bench_default_late.err
echo "bench rc $?"; tail -c 300 $O/bench_default_late.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_default_late.json') if l.startswith('{')][0])
rf = d['roofline']
print('value %.1fM frac %.3f store_peak %.0f long %s' % (d['value'] / 1e6, rf['frac'], rf['store_peak_measured'], rf['store_probe']['long_launch']))
PY
