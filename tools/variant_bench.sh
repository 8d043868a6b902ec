#!/bin/bash
# steps/s and roofline fraction of every board size on this box (one bench.py line each; plain first allocation unless $1 = tuned).
# Each line: one launch per step (sgx_step_n), and the same steps with the batch split over two concurrent chains (sgx_rollout).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TRIALS="--placement plain"; [ "$1" = "tuned" ] && TRIALS=""
for spec in "barrage 65536" "standard 262144" "micro 65536" "tiny 65536" "fives 65536" "medium 65536" "octa_barrage 65536" "standard2 32768"; do
  set -- $spec
  python3 $R/bench.py --version $1 --envs $2 --steps 256 --warmup 32 --no-cpu-baseline --no-other-workloads $TRIALS 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; t=d['config']['two_chains']; print('%-14s %8d games  %8.1f M steps/s  launch %8.1f us  frac %.3f   | two chains: %8.1f M steps/s  %8.1f us per step  frac %.3f' % ('$1', $2, d['value']/1e6, r['launch_us'], r['frac_algorithmic'], t['value']/1e6, t['us_per_step'], t['frac_algorithmic']))"
done
