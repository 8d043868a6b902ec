#!/bin/bash
# round 6, sixth GPU call: the whole GPU suite on the shipped build with two games per wave for the no-observation kinds, then the A/B on every board
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1
echo "pytest rc $?" >> $O/pytest_gpu.log
tail -6 $O/pytest_gpu.log
timeout 900 python -W ignore tools/half_wave_ab.py barrage standard octa_barrage medium 2>&1 | grep -v "^/opt" | tee $O/half_wave_ab_all.log
timeout 600 python tools/procedural_bench.py 2>&1 | grep -v "^/opt" | tee $O/procedural_bench.log | tail -25
