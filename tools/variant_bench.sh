#!/bin/bash
# steps/s and roofline fractions of every board size on this box (one bench.py line each; plain first allocation unless $1 = tuned).
# Each line: the headline (one launch per step into a ring of three output sets: DRAM side), the in-place leg (one set of tensors
# rewritten every step) and the same in-place steps with the batch split over two concurrent chains (sgx_rollout).  Fractions are on
# B_min, the packed layout's own byte minimum per step (bench.py: b_min).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TRIALS="--placement plain"; [ "$1" = "tuned" ] && TRIALS=""
for spec in "barrage 65536" "standard 262144" "micro 65536" "tiny 65536" "fives 65536" "medium 65536" "octa_barrage 65536" "standard2 32768"; do
  set -- $spec
  python3 $R/bench.py --version $1 --envs $2 --steps 256 --warmup 32 --no-cpu-baseline --no-other-workloads --no-store-probe --no-facade-leg --no-live-traffic $TRIALS 2>/dev/null | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; i=d['config']['in_place']; t=d['config']['two_chains']; print('%-14s %8d games  ring of 3: %8.1f M steps/s %8.1f us frac %.3f | in place: %8.1f M %8.1f us rate/peak %.3f | two chains: %8.1f M %8.1f us rate/peak %.3f' % ('$1', $2, d['value']/1e6, r['launch_us'], r['frac'], i['value']/1e6, i['launch_us'], i['rate_over_spec_peak'], t['value']/1e6, t['us_per_step'], t['rate_over_spec_peak']))"
done
