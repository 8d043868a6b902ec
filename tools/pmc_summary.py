"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean over dispatches.

    python tools/pmc_summary.py <dir> ...                              text summary (what profiles/*_pmc_summary.txt hold)
    python tools/pmc_summary.py --traffic-entry KEY GAMES <dir> SRC     one profiles/traffic.json entry as JSON: the step kernel's counter
                                                                        bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, KiB -> bytes; the
                                                                        gfx950 correction of MI355X_MICROARCH.md), the L2 -> fabric write
                                                                        requests and how many of them were addressed to DRAM, and the
                                                                        build id of the library that ran
"""
import collections
import csv
import glob
import json
import os
import sys


def collect(pat):
    """{(kernel, counter): [values]} over every counter_collection.csv under the directory pattern."""
    agg = collections.defaultdict(list)
    files = sorted(glob.glob(pat + '/**/*_counter_collection.csv', recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            agg[(f, r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40], r['Counter_Name'])].append(float(r['Counter_Value']))
    return files, agg


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--traffic-entry':
        key, games, pat, src = sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
        total_steps = int(sys.argv[6]) if len(sys.argv) > 6 else 8        # steps a PMC pass of tools/gpu_profile.sh plays: --warmup 2 + --steps 6
        _, agg = collect(pat)
        mean = {}
        for (f, k, c), v in agg.items():
            if k.startswith('void step_kernel'):
                mean[c] = sum(v) / len(v)
        if not mean:      # boards of at most 16 cells: the steps of a call are ONE launch (lane_steps_kernel): counters per STEP = sum over launches / steps
            for (f, k, c), v in agg.items():
                if k.startswith('void lane_steps_kernel') or k.startswith('void steps_kernel'):
                    mean[c] = sum(v) / total_steps
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from stratego_env_amd import build as B
        e = {"games_per_launch": games, "fetch_size_kib": mean.get('FETCH_SIZE'), "write_size_kib": mean.get('WRITE_SIZE'),
             "hbm_bytes_per_launch": int(round((2 * mean['FETCH_SIZE'] + mean['WRITE_SIZE']) * 1024)),
             "tcc_ea0_wrreq": mean.get('TCC_EA0_WRREQ_sum'), "tcc_ea0_wrreq_dram": mean.get('TCC_EA0_WRREQ_DRAM_sum'),
             "tcc_ea0_wrreq_64b": mean.get('TCC_EA0_WRREQ_64B_sum'),
             "build_id": B.read_build_id(B.LIB_PATH), "source": src}
        print(json.dumps({key: e}, indent=1))
        return
    pats = sys.argv[1:] or ['gpurun_out/prof_*']
    for pat in pats:
        files, agg = collect(pat)
        for f in files:
            print('#', f)
            for (ff, k, c), v in sorted(agg.items()):
                if ff == f and any(x in k for x in ('step_kernel', 'steps_kernel', 'sample_kernel', 'lane_kernel', 'export_kernel', 'import_kernel', 'states_kernel', 'choose_kernel')):
                    print('%-42s %-26s n=%-3d mean=%.1f' % (k, c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
