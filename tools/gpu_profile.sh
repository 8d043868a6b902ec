#!/bin/bash
# Runs on the GPU box (via gpurun): kernel stats + PMC passes of bench.py, results under gpurun_out/prof_<tag>/.
# (bench.py runs with --placement plain here to keep the profiled runs short; sgx_observe launches -- placement trials, reset() --
# have their own kernel symbol, observe_kernel, and do not mix into step_kernel's statistics.  --no-in-place-leg / --no-two-chains keep every
# step_kernel launch of a run to ONE launch shape: the headline's -- a ring of three output sets by default, `--output-sets 1` = in place.)
# usage: tools/gpu_profile.sh <tag> <traffic key or -> <games per launch> [extra bench args]
#   a traffic key (e.g. barrage+rotating for the default ring headline, barrage with --output-sets 1) merges the pass' counter bytes into gpurun_out/prof_<tag>/traffic_entry.json
TAG=$1; KEY=$2; GAMES=$3; shift 3
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
python3 -c "import torch,time; x=torch.empty(1<<28,device='cuda'); t=time.time()
while time.time()-t<3: x.fill_(1.0); torch.cuda.synchronize()"
COMMON="--no-cpu-baseline --no-other-workloads --no-two-chains --no-in-place-leg --no-settle-leg --no-live-traffic --no-store-probe --no-facade-leg --placement plain"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 64 --warmup 8 $COMMON "$@" > $OUT/stats.log 2>&1
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum" \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
  "GRBM_GUI_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $P --output-format csv -d $OUT/pmc$i -- python3 $R/bench.py --steps 6 --warmup 2 $COMMON "$@" > $OUT/pmc$i.log 2>&1
done
cd $R
python3 tools/pmc_summary.py "gpurun_out/prof_$TAG" > $OUT/summary.txt 2>&1
if [ "$KEY" != "-" ]; then python3 tools/pmc_summary.py --traffic-entry "$KEY" "$GAMES" "gpurun_out/prof_$TAG" "profiles/${TAG}_pmc_summary.txt" > $OUT/traffic_entry.json; fi
cat $OUT/stats/*/*kernel_stats.csv | head -4 >> $OUT/summary.txt
grep '^{' $OUT/stats.log > $OUT/stats_line.json
# raw traces are large (gpurun merges back at most 64 MiB): keep the summaries only
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*.db" -delete
du -sh $OUT | sed 's/^/# kept: /'
cat $OUT/summary.txt | grep -v "^#"
