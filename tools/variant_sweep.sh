#!/bin/bash
# Run bench.py over alternative builds of the HIP library (stratego_env_amd/_build/var_*.so) in ONE process-sequence
# on one box; prints launch_us per variant, interleaved over rounds (cdna guide rule 24).
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import torch,time
x=torch.empty(1<<28,device='cuda'); t=time.time()
while time.time()-t<3: x.fill_(1.0); torch.cuda.synchronize()
"
for round in 1 2 3; do
  for f in $R/stratego_env_amd/_build/var_*.so; do
    SGX_LIB_PATH=$f python3 $R/bench.py --steps 128 --warmup 16 --wake-seconds 0.5 --no-cpu-baseline "$@" | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$round', '$(basename $f)', round(d['roofline']['launch_us'],1), 'us', round(d['value']/1e6,1), 'M/s')"
  done
done
