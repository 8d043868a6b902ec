#!/bin/bash
# Test the tests: a build of the library whose multi-step kernel LOSES the observation store of the fourth step of every launch
# (-DSGX_MUTANT_SKIP_STORE, sgx_step.h) must FAIL tests/test_gpu_trajectory.py -- the per-step oracle pinning of the kernels bench.py times --
# and the last-step comparisons of tests/test_gpu_multi_step.py (what pinned those kernels until round 5) must not notice it.
#   tools/_dev_build_variant.sh tools/_dev/barrage_mutant.so 10 10 -DSGX_MUTANT_SKIP_STORE     (build container)
#   bash tools/mutant_check.sh                                                                 (GPU box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export SGX_LIB_PATH=tools/_dev/barrage_mutant.so SGX_ALLOW_FOREIGN_BUILD=1
python -m pytest tests/test_gpu_trajectory.py -x -q -k "barrage-48-64" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
echo "trajectory test rc ${PIPESTATUS[0]} (must be non-zero: the mutant has to be caught)"
python -m pytest tests/test_gpu_multi_step.py -x -q -k "equals_one_launch_per_step and barrage-700" 2>&1 | grep -E "passed|failed" | tail -2
echo "(last-step self-comparison of round 5 on the same mutant, for contrast)"
