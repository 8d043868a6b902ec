#!/bin/bash
# round 6: the two-games-per-wave variant of the no-observation kernels, 10x10-only development build: parity tests, then the A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
export SGX_LIB_PATH=${1:-tools/_dev/barrage_half.so} SGX_ALLOW_FOREIGN_BUILD=1
timeout 1500 python -m pytest tests/test_gpu_no_obs_kind.py tests/test_gpu_parity.py tests/test_gpu_multi_step.py tests/test_gpu_procedural.py tests/test_gpu_trajectory.py tests/test_gpu_compact.py -x -q \
   -k "(barrage or standard) and not standard2 and not octa and not short_standard and not micro and not tiny" -W ignore > $O/pytest_half.log 2>&1
echo "pytest rc $?" >> $O/pytest_half.log
tail -12 $O/pytest_half.log
timeout 600 python -W ignore tools/half_wave_ab.py barrage standard > $O/half_wave_ab.log 2>&1
grep -v "^/opt" $O/half_wave_ab.log
