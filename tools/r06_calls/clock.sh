#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r06
export SGX_LIB_PATH=tools/_dev/barrage_cur.so SGX_ALLOW_FOREIGN_BUILD=1
ls /sys/class/drm/card*/device/ | grep -i "pp_dpm\|clk" | sort -u | head -20
timeout 300 python -W ignore tools/clock_probe.py 2>&1 | grep -v "^/opt" | tee gpurun_out/r06/clock_probe.log
