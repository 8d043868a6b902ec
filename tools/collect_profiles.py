"""Copies what a tools/final_campaign.sh run left under gpurun_out/ into profiles/ (the tracked, quoted evidence), merges the counter bytes
into profiles/traffic.json and prints the figures DESIGN.md quotes.      python tools/collect_profiles.py <outdir name of the campaign>"""
import csv
import glob
import json
import os
import shutil
import sys


def newest(pat):
    fs = glob.glob(pat)
    fs.sort(key=os.path.getmtime)
    return fs[-1]


def step_row(f, multi_steps=None):
    """(calls, average us) of the per-step kernel in a kernel_stats.csv; with multi_steps = the steps the run's multi-step launches played:
    ('N launches of steps_kernel', total us / steps = us per STEP) for the multi-step kernel too."""
    out = []
    for r in csv.DictReader(open(f)):
        name = r['Name'].replace('(anonymous namespace)::', '')
        if name.startswith('void step_kernel'):
            out.append(('step_kernel', r['Calls'], round(float(r['AverageNs']) / 1e3, 2)))
        if multi_steps and (name.startswith('void steps_kernel') or name.startswith('void lane_steps_kernel')):
            out.append((name.split('<')[0].replace('void ', '') + ': %s launches, %d steps' % (r['Calls'], multi_steps), 'us per step',
                        round(float(r['TotalDurationNs']) / 1e3 / multi_steps, 2)))
    return out


def main():
    out = 'gpurun_out/' + sys.argv[1]
    rnd = sys.argv[2] if len(sys.argv) > 2 else 'r06'
    tags = tuple('%s_%s' % (rnd, t) for t in ('untuned_headline', 'inplace', 'micro', 'standard', 'both'))
    for tag in tags:
        d = 'gpurun_out/prof_%s' % tag
        shutil.copy(d + '/summary.txt', 'profiles/%s_pmc_summary.txt' % tag)
        shutil.copy(newest(d + '/stats/*/*kernel_stats.csv'), 'profiles/%s_kernel_stats.csv' % tag)
        shutil.copy(d + '/stats_line.json', 'profiles/%s_stats_line.json' % tag)
        shutil.copy(d + '/traffic_entry.json', 'profiles/%s_traffic_entry.json' % tag)
    for tag in ('headline', 'inplace'):
        shutil.copy(newest('gpurun_out/%s_tuned/%s/*/*kernel_stats.csv' % (rnd, tag)), 'profiles/%s_tuned_%s_kernel_stats.csv' % (rnd, tag))
        shutil.copy('gpurun_out/%s_tuned/%s_line.json' % (rnd, tag), 'profiles/%s_tuned_%s_line.json' % (rnd, tag))
    for src, dst in (('pytest.log', 'gputest.log'), ('smoke.log', 'smoke.log'), ('variant_bench.log', 'variant_bench.log'),
                     ('lane_ab.log', 'lane_ab.log'), ('procedural_bench.log', 'procedural_bench.log'),
                     ('soak_parity.log', 'soak_parity.log'), ('soak_procedural.log', 'soak_procedural.log'),
                     ('phase_cost.log', 'phase_cost_final.log'), ('bench_default.json', 'default_bench_line.json'),
                     ('bench_driver_style.json', 'driver_style_line.json'), ('bench_default_run2.json', 'default_bench_line_run2.json'),
                     ('bench_default_run3.json', 'default_bench_line_run3.json'),
                     ('multi_step_ab.log', 'multi_step_ab_plain_buffers.log'), ('ring_size_probe_tuned.log', 'ring_size_probe_tuned.log'),
                     ('half_wave_ab.log', 'half_wave_ab.log'), ('noobs_small_ab.log', 'noobs_small_ab.log'), ('spread_probe.log', 'spread_probe_final.log'),
                     ('facade_breakdown.log', 'facade_breakdown.log'), ('clock_probe.log', 'clock_probe.log'), ('soak_general_states.log', 'soak_general_states.log'), ('soak_trajectory.log', 'soak_trajectory.log'),
                     ('ring_footprint_probe.log', 'ring_footprint_probe.log'), ('ring_chunk_probe.log', 'ring_chunk_probe.log')):
        if os.path.exists(os.path.join(out, src)):
            shutil.copy(os.path.join(out, src), 'profiles/%s_%s' % (rnd, dst))
    for tag in ('procedural', 'kstep_micro'):
        f = 'gpurun_out/prof_%s_%s/summary.txt' % (rnd, tag)
        if os.path.exists(f):
            shutil.copy(f, 'profiles/%s_%s_summary.txt' % (rnd, tag))
    t = json.load(open('profiles/traffic.json'))
    for tag in tags:
        t.update(json.load(open('profiles/%s_traffic_entry.json' % tag)))
    json.dump(t, open('profiles/traffic.json', 'w'), indent=1)
    for k, v in t.items():
        if k != '_comment':
            print(k, v['hbm_bytes_per_launch'], v['build_id'], v['tcc_ea0_wrreq'], v['tcc_ea0_wrreq_dram'], v['fetch_size_kib'], v['write_size_kib'])
    for tag in tags:
        d = json.load(open('profiles/%s_stats_line.json' % tag))
        print(tag, step_row('profiles/%s_kernel_stats.csv' % tag, (d['config'].get('multi_step_tally') or {}).get('steps') or (8 + 64 if d['config']['output_sets'] > 1 else 64)),
              'line launch_us %.2f' % d['roofline']['launch_us'], 'sets', d['config']['output_sets'], d['build_id'])
    for tag in ('headline', 'inplace'):
        d = json.load(open('profiles/%s_tuned_%s_line.json' % (rnd, tag)))
        print('tuned', tag, step_row('profiles/%s_tuned_%s_kernel_stats.csv' % (rnd, tag), (d['config'].get('multi_step_tally') or {}).get('steps') or (2 * (64 + 512) if d['config']['output_sets'] > 1 else 2 * 512)),
              'line launch_us %.2f value %.1fM frac %.3f' % (d['roofline']['launch_us'], d['value'] / 1e6, d['roofline']['frac']), d['config']['output_sets'])
    for f in ('bench_default', 'bench_default_run2', 'bench_default_run3', 'bench_driver_style'):
        if not os.path.exists('%s/%s.json' % (out, f)):
            continue
        d = json.loads(open('%s/%s.json' % (out, f)).read().strip().splitlines()[-1])
        r, c = d['roofline'], d['config']
        print(f, 'value %.1fM' % (d['value'] / 1e6), 'launch %.1f' % r['launch_us'],
              'frac %.3f untuned %.3f inplace %.3f' % (r['frac'], r['frac_untuned'] or 0, r['in_place_rate_over_spec_peak'] or 0), d['build_id'], r['traffic_source'][:40])
        print('  in_place %.1fM %.1fus' % (c['in_place']['value'] / 1e6, c['in_place']['launch_us']),
              'two %.1fM %.1fus %.3f' % (c['two_chains']['value'] / 1e6, c['two_chains']['us_per_step'], c['two_chains']['rate_over_spec_peak']),
              c['placement'], c['ring_placement_plain_and_kept_us_per_extra_set'])
        for w in c['other_workloads']:
            print('  ', w['workload'][:50], '%.1fM %.1fus frac %.3f dram %s two %s' % (w['value'] / 1e6, w['launch_us'], w['frac'], w.get('frac_dram'),
                  (w['concurrent_chains'] or {}).get('frac')), (w.get('rotating_outputs') or {}).get('launch_us'))
        ci = c['consumer_in_loop']
        print('  consumer nt %.1f plain %.1f loop %.2fM ratio %.3f' % (ci['nt_stores']['step_kernel_us_in_loop'], ci['plain_stores']['step_kernel_us_in_loop'],
              ci['nt_stores']['value'] / 1e6, ci['nt_over_plain_step_kernel']))
        co = c['compact_outputs']
        print('  compact %.1fM %.1fus decode %.1f/%.1f frac %.3f' % (co['value'] / 1e6, co['launch_us'], co['decode_obs_us_per_batch'], co['decode_mask_us_per_batch'], co['frac']))
        print('  cpu %.0f (1 thread %.0f) cores %d' % (d['cpu_baseline']['value'], d['cpu_baseline']['value_1_thread'], d['cpu_baseline']['cores']))


if __name__ == '__main__':
    main()
