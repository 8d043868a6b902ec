#!/bin/bash
# rocprofv3 --pmc passes over tools/probe_vs_game.py: write-path counters of steps_kernel against store_probe_kernel, per GB written
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_probe_vs_game
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/probe_vs_game.py > $OUT/trace.log 2>&1
i=0
for P in "WRITE_SIZE" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_STREAMING_REQ_sum" \
  "TCC_CYCLE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" \
  "SQ_WAIT_INST_ANY SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
  "TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_BUBBLE_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/probe_vs_game.py > $OUT/pmc$i.log 2>&1
done
cd $R
python3 - "$OUT" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
GB = {'steps_kernel': 256 * 65536 * 30516.0 / 1e9, 'store_probe_kernel': 64 * 65536 * 26800.0 / 1e9}
def kind(name):
    for k in GB:
        if k in name:
            return k
    return None
dur = collections.defaultdict(list)
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = kind(r['Kernel_Name'])
        if k:
            dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in dur.items():
    v = v[-3:] if k == 'steps_kernel' else v
    print("%-20s launches %s us -> %s GB/s" % (k, [round(x) for x in v], [round(GB[k] / (x * 1e-6)) for x in v]))
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if kind(r['Kernel_Name'])]
    by = collections.defaultdict(list)
    for r in rows:
        by[(kind(r['Kernel_Name']), r['Counter_Name'], int(r['Dispatch_Id']))].append(float(r['Counter_Value']))
    disp = collections.defaultdict(set)
    for (k, c, d) in by:
        disp[k].add(d)
    for (k, c, d), v in by.items():
        ids = sorted(disp[k])
        # steps_kernel: the last three dispatches are the 256-step ring launches; store_probe_kernel: dispatches 1,2 = nt 1 (first is the untimed touch), 4,5 = every 8th sweep plain
        if k == 'steps_kernel' and d in ids[-3:]:
            agg[(c, 'game (steps_kernel, ring of 3)')].append(sum(v))
        if k == 'store_probe_kernel':
            j = ids.index(d)
            if j in (1, 2):
                agg[(c, 'probe, all non-temporal')].append(sum(v))
            if j in (4, 5):
                agg[(c, 'probe, every 8th sweep plain')].append(sum(v))
print("# counters per GB written (mean over the timed launches)")
for (c, what), v in sorted(agg.items()):
    g = GB['steps_kernel'] if what.startswith('game') else GB['store_probe_kernel']
    print("%-36s %-34s n=%d  per GB %.5g" % (c, what, len(v), sum(v) / len(v) / g))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
