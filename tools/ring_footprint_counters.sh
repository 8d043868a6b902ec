#!/bin/bash
# rocprofv3 --pmc passes over tools/ring_footprint_run.py: translation (UTCL1 / UTCL2) and write-path counters of steps_kernel writing a ring that
# covers 16 GB against one that covers 128 GB (same kernel, same launch shape, every set of the fast class in place)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_ring_footprint
mkdir -p $OUT
cd /tmp
i=0
for P in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
  "TCP_CLIENT_UTCL1_INFLIGHT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_SERIALIZATION_STALL_sum" \
  "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_LFIFO_FULL_sum TCP_UTCL1_STALL_LFIFO_NO_RES_sum TCP_UTCL1_PERMISSION_MISS_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum" \
  "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
  "WRITE_SIZE" "FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/ring_footprint_run.py > $OUT/pmc$i.log 2>&1
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
    ids = sorted({d for (_, d) in by})
    if len(ids) != 6:
        print("# %s: %d steps_kernel dispatches (6 expected)" % (f, len(ids)))
        continue
    for (c, d), v in by.items():
        j = ids.index(d)
        if j in (1, 2):
            agg[(c, 16)].append(v)
        if j in (4, 5):
            agg[(c, 128)].append(v)
print("# steps_kernel, 64 steps of 65,536 Barrage games per launch; counters per launch (mean of two launches)")
print("%-48s %16s %16s %8s" % ("counter", "ring of 8 (16 GB)", "ring of 64 (128 GB)", "ratio"))
for c in sorted({c for (c, _) in agg}):
    a, b = agg[(c, 16)], agg[(c, 128)]
    a, b = sum(a) / max(1, len(a)), sum(b) / max(1, len(b))
    print("%-48s %16.6g %16.6g %8.3f" % (c, a, b, b / a if a else float('nan')))
PY
grep -h "us per step" $OUT/pmc1.log | sed 's/^/# under the first counter pass: /' >> $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
