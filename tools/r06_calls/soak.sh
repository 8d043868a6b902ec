#!/bin/bash
# round 6: long soaks on the final binary (every output of every step / every functional-API result against the oracle, fresh seeds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
python tools/soak_parity.py ${1:-1200} > $O/soak_parity_long.log 2>&1; tail -1 $O/soak_parity_long.log
python tools/soak_trajectory.py ${4:-600} > $O/soak_trajectory_long.log 2>&1; tail -1 $O/soak_trajectory_long.log
python tools/soak_procedural.py ${2:-480} > $O/soak_procedural_long.log 2>&1; tail -1 $O/soak_procedural_long.log
python tools/soak_general_states.py ${3:-240} > $O/soak_general_states_long.log 2>&1; tail -1 $O/soak_general_states_long.log
timeout 1200 python bench.py > $O/bench_default_summary.json 2> $O/bench_default_summary.err; echo "bench rc $?"
tail -c 1500 $O/bench_default_summary.json
