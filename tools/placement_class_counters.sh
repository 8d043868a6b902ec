#!/bin/bash
# rocprofv3 --pmc passes over tools/placement_class_run.py: is the slow placement class of an output allocation a matter of address
# translation, like a ring beyond 16 GB (tools/ring_footprint_counters.sh), or of the DRAM side?
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_placement_class
mkdir -p $OUT
cd /tmp
i=0
for P in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" \
  "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/placement_class_run.py > $OUT/pmc$i.log 2>&1
done
cd $R
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if 'steps_kernel' in r['Kernel_Name']]
    by = collections.defaultdict(float)
    for r in rows:
        by[(r['Counter_Name'], int(r['Dispatch_Id']))] += float(r['Counter_Value'])
    ids = sorted({d for (_, d) in by})[-6:]
    for (c, d), v in by.items():
        if d in ids[:3]:
            agg[(c, 'fast')].append(v)
        elif d in ids[3:]:
            agg[(c, 'slow')].append(v)
print("# steps_kernel in place, 32 steps of 65,536 Barrage games per launch; counters per launch (mean of three launches)")
print("%-44s %16s %16s %8s" % ("counter", "fast-class set", "slow-class set", "ratio"))
for c in sorted({c for (c, _) in agg}):
    a, b = agg[(c, 'fast')], agg[(c, 'slow')]
    a, b = sum(a) / max(1, len(a)), sum(b) / max(1, len(b))
    print("%-44s %16.6g %16.6g %8.3f" % (c, a, b, b / a if a else float('nan')))
PY
for j in 1 2 3 4; do grep -h "us per step\|obs at" $OUT/pmc$j.log | sed "s/^/# pass $j: /" >> $OUT/summary.txt; done
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
