#!/bin/bash
# the long soaks on the round's final build (one gpurun call, ~23 GPU-minutes)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r06
mkdir -p $O
python -c "from stratego_env_amd import _lib; import ctypes; L = _lib.load(); print('build id', L.sgx_build_id().decode())" > $O/soak_long2.log 2>&1
python tools/soak_parity.py 600 2>&1 | tail -1 >> $O/soak_long2.log
python tools/soak_trajectory.py 420 2>&1 | tail -1 >> $O/soak_long2.log
python tools/soak_procedural.py 200 2>&1 | tail -1 >> $O/soak_long2.log
python tools/soak_general_states.py 120 2>&1 | tail -1 >> $O/soak_long2.log
cat $O/soak_long2.log
