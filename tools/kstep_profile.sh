#!/bin/bash
# rocprofv3 passes over tools/kstep_probe.py (lane_steps_kernel: 3 in-place launches, then 3 ring launches): per-dispatch durations and counters
TAG=$1; VERSION=${2:-micro}; GAMES=${3:-65536}; STEPS=${4:-256}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/kstep_probe.py $VERSION $GAMES $STEPS > $OUT/trace.log 2>&1
i=0
for P in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT" \
  "WRITE_SIZE" "FETCH_SIZE" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_STALL_sum" \
  "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_BUSY_sum" \
  "TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_SRC_FIFO_FULL_sum TCC_LATENCY_FIFO_FULL_sum" \
  "TCP_PENDING_STALL_CYCLES_sum TCC_WRITE_sum TCC_WRITEBACK_sum TCC_STREAMING_REQ_sum" \
  "TCC_CYCLE_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/kstep_probe.py $VERSION $GAMES $STEPS > $OUT/pmc$i.log 2>&1
done
cd $R
python3 - "$OUT" "$STEPS" "$GAMES" > $OUT/summary.txt <<'PY'
import csv, glob, sys, collections
out, steps, games = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
# tools/kstep_probe.py launches lane_steps_kernel 7 times: 8 warm-up steps, then 3 x `steps` in place, then 3 x `steps` into a ring of three
def mode_of(k, n):
    return 'warm-up' if k == 0 else ('in place' if k < n - 3 else 'ring of 3')
for f in glob.glob(out + '/trace/**/*kernel_trace.csv', recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if 'lane_steps_kernel' in r['Kernel_Name']]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    for k, r in enumerate(rows):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        n = 8 if k == 0 else steps
        print("launch %d (%s, %d steps): %.1f us = %.2f us per step" % (k, mode_of(k, len(rows)), n, d, d / n))
agg = collections.defaultdict(list)
for f in sorted(glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if 'lane_steps_kernel' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    for r in rows:
        k = ids.index(int(r['Dispatch_Id']))
        agg[(r['Counter_Name'], mode_of(k, len(ids)))].append(float(r['Counter_Value']))
print("# counters of %d games: mean per launch of %d steps, per step, per game and step" % (games, steps))
for (c, mode), v in sorted(agg.items()):
    if mode == 'warm-up':
        continue
    m = sum(v) / len(v)
    print("%-24s %-10s n=%d per launch %.4g   per step %.4g   per game-step %.4g" % (c, mode, len(v), m, m / steps, m / steps / games))
PY
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt
