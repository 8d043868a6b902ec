#!/bin/bash
# One-box diagnosis: partition modes, clocks, plain fill, store-pattern probe, step kernel.
for f in /sys/class/drm/card*/device/current_compute_partition /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/pp_dpm_mclk /sys/class/drm/card*/device/pp_dpm_fclk /sys/class/drm/card*/device/mem_info_vram_used /sys/class/drm/card*/device/power_dpm_force_performance_level; do [ -e $f ] && echo "$f: $(cat $f | tr '\n' ' ')"; done
rocm-smi --showmemuse --showpower --showmaxpower 2>/dev/null | grep -E "GPU\[" | head -8
python3 tools/alloc_probe.py 2>/dev/null | head -3
./tools/microbench/store_pattern | grep -E "WPB=8|WPB=4 xcd"
python3 bench.py --no-cpu-baseline --steps 128 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('step kernel', round(d['value']/1e6,1), 'M steps/s', round(d['roofline']['launch_us'],1), 'us', d['config']['placement_trial_us'])"
