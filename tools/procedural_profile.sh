#!/bin/bash
# Runs on the GPU box (via gpurun): tools/procedural_bench.py (the int64 operator paths: export / import / sgx_step_states) under
# rocprofv3 -- one --kernel-trace --stats pass and separate --pmc passes -- results under gpurun_out/prof_<tag>/.
# usage: tools/procedural_profile.sh <tag> [version] [games]
TAG=$1; VERSION=${2:-barrage}; GAMES=${3:-65536}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/tools/procedural_bench.py $VERSION $GAMES > $OUT/stats.log 2>&1
# one path per trace where several paths share a kernel symbol (states_kernel<..., false, false> plays get_next_state AND is_move_valid_*)
for ONLY in get_next_state is_move_valid_by_1d_index; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$ONLY -- python3 $R/tools/procedural_bench.py $VERSION $GAMES $ONLY > $OUT/stats_$ONLY.log 2>&1
done
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/tools/procedural_bench.py $VERSION $GAMES > $OUT/pmc$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py "gpurun_out/prof_$TAG" > $OUT/summary.txt 2>&1
python3 - "$OUT" >> $OUT/summary.txt <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-90s calls %5s avg %10.1f us" % (r['Name'].replace('(anonymous namespace)::', '')[:90], r['Calls'], float(r['AverageNs']) / 1e3))
for only in ('get_next_state', 'is_move_valid_by_1d_index'):
    for f in glob.glob(sys.argv[1] + '/stats_%s/**/*kernel_stats.csv' % only, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'states_kernel' in r['Name']:
                print("ONLY %-26s %-60s calls %5s avg %10.1f us" % (only, r['Name'].replace('(anonymous namespace)::', '')[:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
cat $OUT/stats.log | grep "us per batch" >> $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
cat $OUT/summary.txt | grep -v "^#"
