#!/bin/bash
# Test the tests: a build of the library whose multi-step kernel LOSES the observation store of the fourth step of every launch
# (-DSGX_MUTANT_SKIP_STORE, sgx_step.h) must FAIL tests/test_gpu_trajectory.py -- the per-step oracle pinning of the kernels bench.py times --
# at exactly that step: the slot keeps the poison the test wrote before the call.
#   tools/_dev_build_variant.sh tools/_dev/barrage_mutant.so 10 10 -DSGX_MUTANT_SKIP_STORE     (build container)
#   bash tools/mutant_check.sh                                                                 (GPU box)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export SGX_LIB_PATH=tools/_dev/barrage_mutant.so SGX_ALLOW_FOREIGN_BUILD=1
python -m pytest tests/test_gpu_trajectory.py -x -q -k "barrage-48-64" 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
echo "trajectory test rc ${PIPESTATUS[0]} (must be non-zero: the mutant has to be caught)"
