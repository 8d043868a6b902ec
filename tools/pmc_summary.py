"""Summarise rocprofv3 --pmc counter_collection.csv files: per kernel, per counter, mean over dispatches."""
import collections
import csv
import glob
import sys


def main():
    pats = sys.argv[1:] or ['gpurun_out/prof_*']
    for pat in pats:
        for f in sorted(glob.glob(pat + '/**/*_counter_collection.csv', recursive=True)):
            agg = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                agg[(r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40], r['Counter_Name'])].append(float(r['Counter_Value']))
            print('#', f)
            for (k, c), v in sorted(agg.items()):
                if 'step_kernel' in k or 'sample_kernel' in k:
                    print('%-42s %-26s n=%-3d mean=%.1f' % (k, c, len(v), sum(v) / len(v)))


if __name__ == '__main__':
    main()
