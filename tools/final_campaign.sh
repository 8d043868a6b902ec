#!/bin/bash
# The round's measurement campaign, ONE gpurun call on the GPU box:  gpurun --timeout 4000 -- 'bash tools/final_campaign.sh <outdir name>'
# Everything lands under gpurun_out/; tools/collect_profiles.py (in the build container, afterwards) copies what is quoted into profiles/.
OUT=${1:-final}
RND=${2:-r06}                                  # prefix of everything this round commits under profiles/
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out/$OUT
python -m pytest tests -m gpu -q > gpurun_out/$OUT/pytest.log 2>&1; echo PYTEST_RC $?; tail -3 gpurun_out/$OUT/pytest.log
python bench.py > gpurun_out/$OUT/bench_default.json 2> gpurun_out/$OUT/bench_default.err; echo BENCH_RC $?
python bench.py --steps 20 --warmup 5 > gpurun_out/$OUT/bench_driver_style.json 2> gpurun_out/$OUT/bench_driver_style.err; echo BENCH2_RC $?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/$OUT/smoke.log 2>&1; tail -1 gpurun_out/$OUT/smoke.log
bash tools/gpu_profile.sh ${RND}_untuned_headline barrage+rotating 65536 > gpurun_out/prof_${RND}_untuned_headline.log 2>&1
bash tools/gpu_profile.sh ${RND}_inplace barrage 65536 --output-sets 1 > gpurun_out/prof_${RND}_inplace.log 2>&1
bash tools/gpu_profile.sh ${RND}_micro micro 65536 --version micro --output-sets 1 > gpurun_out/prof_${RND}_micro.log 2>&1
bash tools/gpu_profile.sh ${RND}_standard standard 262144 --version standard --envs 262144 --warmup 300 --output-sets 1 > gpurun_out/prof_${RND}_standard.log 2>&1
bash tools/gpu_profile.sh ${RND}_both barrage+full_obs 65536 --full-obs --output-sets 1 > gpurun_out/prof_${RND}_both.log 2>&1
cd /tmp; mkdir -p $R/gpurun_out/${RND}_tuned
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_tuned/headline -- python3 $R/bench.py --no-other-workloads --no-cpu-baseline --no-two-chains --no-in-place-leg --no-live-traffic --no-store-probe --no-facade-leg > $R/gpurun_out/${RND}_tuned/headline_line.json 2> $R/gpurun_out/${RND}_tuned/headline.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${RND}_tuned/inplace -- python3 $R/bench.py --no-other-workloads --no-cpu-baseline --no-two-chains --no-live-traffic --no-store-probe --no-facade-leg --output-sets 1 > $R/gpurun_out/${RND}_tuned/inplace_line.json 2> $R/gpurun_out/${RND}_tuned/inplace.err
find $R/gpurun_out/${RND}_tuned -name "*kernel_trace.csv" -delete; find $R/gpurun_out/${RND}_tuned -name "*.db" -delete
cd $R
bash tools/variant_bench.sh tuned > gpurun_out/$OUT/variant_bench.log 2>&1
python tools/soak_parity.py ${SOAK_SECONDS:-150} > gpurun_out/$OUT/soak_parity.log 2>&1; tail -1 gpurun_out/$OUT/soak_parity.log
python tools/soak_trajectory.py 120 > gpurun_out/$OUT/soak_trajectory.log 2>&1; tail -1 gpurun_out/$OUT/soak_trajectory.log
python tools/soak_procedural.py 60 > gpurun_out/$OUT/soak_procedural.log 2>&1; tail -1 gpurun_out/$OUT/soak_procedural.log
python tools/lane_ab.py --specs micro:65536,tiny:65536,micro:262144 --rounds 2 > gpurun_out/$OUT/lane_ab.log 2>&1
bash tools/procedural_profile.sh ${RND}_procedural barrage 65536 > gpurun_out/$OUT/procedural_profile.log 2>&1
bash tools/kstep_profile.sh ${RND}_kstep_micro micro 65536 256 > gpurun_out/$OUT/kstep_micro.log 2>&1
python tools/procedural_bench.py > gpurun_out/$OUT/procedural_bench.log 2>&1
for v in barrage standard micro tiny fives; do python tools/phase_cost.py $v 65536 2>&1 | grep -v "^/opt"; done > gpurun_out/$OUT/phase_cost.log
python tools/multi_step_ab.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/multi_step_ab.log
python tools/ring_size_probe_tuned.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/ring_size_probe_tuned.log
for i in 2 3; do python bench.py > gpurun_out/$OUT/bench_default_run$i.json 2> gpurun_out/$OUT/bench_default_run$i.err; done
python -W ignore tools/half_wave_ab.py barrage standard octa_barrage medium 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/half_wave_ab.log
python tools/noobs_small_ab.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/noobs_small_ab.log
python tools/r06_spread_probe.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/spread_probe.log
python tools/facade_breakdown.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/facade_breakdown.log
python tools/clock_probe.py 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/clock_probe.log
python -W ignore tools/ring_footprint_probe.py 64 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/ring_footprint_probe.log
python -W ignore tools/ring_chunk_probe.py 64 2>&1 | grep -v "^/opt" > gpurun_out/$OUT/ring_chunk_probe.log
python tools/soak_general_states.py 40 > gpurun_out/$OUT/soak_general_states.log 2>&1; tail -1 gpurun_out/$OUT/soak_general_states.log
