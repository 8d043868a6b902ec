#!/bin/bash
# round 6, eighth GPU call: rings of more than 8 separate sets in one launch (pointer table) -- the whole GPU suite, then a default line whose
# trajectory leg also times a ring of 64 separately placed sets
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu2.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu2.log
tail -5 $O/pytest_gpu2.log
timeout 1500 python bench.py > $O/bench_default_ring64.json 2> $O/bench_default_ring64.err
echo "bench rc $?"; tail -c 400 $O/bench_default_ring64.err
python - <<'PY'
import json
d = json.loads([l for l in open('gpurun_out/r06/bench_default_ring64.json') if l.startswith('{')][0])
print(json.dumps(d['config']['trajectory'], indent=1)[:2500])
print(json.dumps(d['summary_at_the_end_of_the_line'])[:1800])
PY
