"""The legs of bench.py's 1-GPU line other than the headline (kept out of bench.py, which holds the contract, the headline and the roofline):

  rotating_leg       a workload's steps into a ring of output sets (the DRAM-side figure of the legs)
  consumer_leg       config.consumer_in_loop: sgx_step alternating with a device policy that READS the observation and the mask
  compact_leg        config.compact_outputs: the opt-in 4-bit-code / mask-bit outputs
  other_workload     config.other_workloads: BASELINE configs 3 and 4, BOTH_OBSERVATIONS, config 5's per-GPU size
  as_it_comes_leg    config.as_it_comes: VecStrategoEnv with every default on the headline's workload

  store_probe_leg    roofline.store_peak_measured: the step kernel's store stream WITHOUT the game (sgx_store_probe) on the very ring buffers
                     the headline wrote, in the same process -- zeros, observation-like floats and incompressible bits as payload; one launch
                     long enough that less than 0.1 % of its bytes can still sit in L2 + Infinity Cache when it ends; the memory clock
                     sampled WHILE it runs.
  trajectory_leg     config.trajectory: the headline's rollout into a trajectory buffer [T][N]... of T = 64 slots (sgx_step_traj: one launch,
                     any number of slots; the reference hands back a fresh observation array per step, impl:905, maenv:447-497).
  facade_leg         config.facade_n1: BASELINE config 1 -- ONE Barrage game behind the reference's dict API (StrategoMultiAgentEnv.step,
                     loop:34-63), random valid actions, steps per second of the whole loop and of env.step() alone.

Every function takes `B`, the bench module object that calls it (bench.py runs as __main__: importing it again from here would make a
second copy of its globals).  The oracle is used as the checker only (B.verify_against_oracle)."""
import ctypes as C
import threading
import time


def _mclk_sampler(B, stop, seen, period=0.002):
    while not stop.is_set():
        m = B.hbm_mclk_mhz()
        if m:
            seen.append(m)
        time.sleep(period)


def store_probe_leg(B, env, achieved_gbs, passes_short=8, launches_short=4, long_bytes=320e9):
    """The store-only kernel on the observation buffers of the env's ring (or its one set).  OVERWRITES them: the caller re-renders the
    current outputs afterwards (env.observe()).  Two axes: the payload (zeros / observation-like floats / incompressible bits) and how many
    store streams the memory sees at once (resident waves per CU, sleeps between a wave's 1 KiB sweeps): the step kernel's waves do not store
    back to back.  -> dict for roofline['store_probe'], plus store_peak_measured (the best rate of any configuration) / frac_of_store_peak."""
    import torch
    from stratego_env_amd import _lib
    L = env._L
    sets = [s[0] for s in env._ring] if env._ring else [env.obs]
    seg = int(sets[0][0].numel() * 4)                      # bytes of one game's observation: the step kernel's segment
    stream = env._stream()
    names = ("zeros", "observation_like", "random_bits")
    us, gbs = C.c_float(), C.c_float()

    def probe(t, passes, payload, nt, waves, pace, launches, persistent=0, dwell=1, ring=1):
        _lib.check(L.sgx_store_probe(env.device.index, C.c_void_p(t.data_ptr()), int(t.numel() * 4), seg, passes, payload, nt, waves, pace, persistent,
                                     dwell, ring, launches, stream, C.byref(us), C.byref(gbs)), L)
        return float(gbs.value)

    with torch.cuda.device(env.device):
        # (1) how many streams at once: observation-like payload on the first set
        sweep, best, best_cfg = [], 0.0, (24, 0, 0)
        for persistent in (0, 1):
            for waves in (24, 16, 8):
                for pace in (0, 1, 2, 4, 8):
                    g = probe(sets[0], passes_short, 1, 1, waves, pace, launches_short, persistent)
                    sweep.append({"persistent_waves": persistent, "waves_per_cu": waves, "pace": pace, "gbps": round(g, 1)})
                    if g > best:
                        best, best_cfg = g, (waves, pace, persistent)
        # (1b) what the multi-step kernel's waves do and a walking store stream does not: REWRITE the same segment step after step (dwell), in place
        # or cycling through the sets of a ring -- the live window of the resident waves is 6,144 x 26.8 KB = 165 MB per set
        rewrite = []
        for ring in (1, 3, 8):
            for dwell in (1, 4, 32):
                g = probe(sets[0], 1 if dwell > 1 else 8, 1, 1, 24, 0, launches_short, 1, dwell, ring)
                rewrite.append({"persistent_waves": 1, "ring_sub_ranges": ring, "dwell": dwell, "gbps": round(g, 1)})
        # (1c) the MIX of the step kernel: most of its bytes are non-temporal observation stores, 12 % (mask, edge lines, results) plain stores through
        # L2 -- every N-th sweep of the probe plain
        mix = []
        for every in (2, 4, 8, 16):
            for waves, pace, persistent in ((24, 0, 0), tuple(best_cfg)):
                g = probe(sets[0], passes_short, 1, every, waves, pace, launches_short, persistent)
                mix.append({"plain_every_nth_sweep": every, "waves_per_cu": waves, "pace": pace, "persistent_waves": persistent, "gbps": round(g, 1)})
                best = max(best, g)
        # (2) the payload, on every set: back to back (24 waves per CU, no pacing: every resident wave stores all the time) and at the best point of (1)
        per_set = []
        for t in sets:
            row = {"bytes": int(t.numel() * 4)}
            for payload in (0, 1, 2):
                row[names[payload]] = round(probe(t, passes_short, payload, 1, 24, 0, launches_short), 1)
                g = probe(t, passes_short, payload, 1, best_cfg[0], best_cfg[1], launches_short, best_cfg[2])
                row[names[payload] + "_at_best"] = round(g, 1)
                best = max(best, g)
            row["observation_like_plain_stores"] = round(probe(t, passes_short, 1, 0, 24, 0, launches_short), 1)
            per_set.append(row)
        # (3) ONE long launch at the best point, incompressible bits: passes x the first set's bytes >= long_bytes, so that what L2 (32 MiB) + the
        # Infinity Cache (256 MiB) can still hold at its end is < 0.1 % of what it wrote; the memory clock sampled while it runs
        t = sets[0]
        passes = max(1, int(-(-long_bytes // (t.numel() * 4))))
        stop, seen = threading.Event(), []
        th = threading.Thread(target=_mclk_sampler, args=(B, stop, seen))
        th.start()
        try:
            long_cfg = (24, 0, 0)                        # the plain back-to-back configuration (the sweep's maximum is within its own noise of it)
            g_long = probe(t, passes, 2, 1, long_cfg[0], long_cfg[1], 2, long_cfg[2])
            us_long = float(us.value)
        finally:
            stop.set()
            th.join()
        long_launch = {"payload": "random_bits", "waves_per_cu": long_cfg[0], "pace": long_cfg[1], "persistent_waves": long_cfg[2], "passes": passes,
                       "bytes_per_launch": int(t.numel() * 4) * passes, "us_per_launch": round(us_long, 1),
                       "gbps": round(g_long, 1), "cache_residue_frac_at_most": (288 << 20) / float(int(t.numel() * 4) * passes),
                       "mclk_mhz_seen_during": sorted(set(seen)), "mclk_samples": len(seen)}
        best = max(best, g_long)
    return {"store_probe": {"kernel": "store_probe_kernel: one wave per %d-byte segment, 16 B per lane, 1 KiB sweeps on 1 KiB address boundaries, whole lines "
                                      "non-temporal, edge lines through L2, 512-thread workgroups, eight XCD ranges per pass (the step kernel's observation "
                                      "stream without the game)" % seg,
                            "buffers": "the observation tensors of the ring the headline wrote (same process, same allocations)",
                            "passes_per_launch": passes_short, "streams_at_once_sweep_set0_observation_like": sweep,
                            "rewriting_waves_set0_observation_like": rewrite, "plain_and_non_temporal_mix_set0_observation_like": mix,
                            "best_waves_per_cu_pace_persistent": list(best_cfg), "gbps_by_set_and_payload": per_set, "long_launch": long_launch},
            "store_peak_measured": best, "store_peak_unit": "GB/s",
            "frac_of_store_peak": achieved_gbs / best if best > 0 else None}


def trajectory_leg(B, rk, args, version='barrage', n=65536, slots=64, passes=3, verify=8):
    """The headline's rollout written into a trajectory buffer of `slots` slots (obs [T,N,R,C,67], mask, per-slot results, drawn actions):
    n_steps = n_slots = T per call, ONE multi-step launch; plain torch.empty tensors (a 112 GB buffer has no placement search)."""
    import torch
    from stratego_env_amd import _lib
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    env = B.make_env(version, n, 0, rk.device_index)
    try:
        per_slot = env.obs.numel() * 4 + env.mask.numel() + n * 16
        free, _ = torch.cuda.mem_get_info()
        while slots > 4 and slots * per_slot > 0.7 * free:
            slots //= 2
        traj = env.alloc_trajectory(slots)
        env.sample_valid_actions()
        env.rollout_trajectory(slots, traj)                       # first touch of every slot, untimed
        env.bench_steps_played += slots
        kind = env.last_launch_kind
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for _ in range(passes):
            env.rollout_trajectory(slots, traj)
        e1.record()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        steps = passes * slots
        env.bench_steps_played += steps
        B.MULTI_STEP_TALLY["launches"] += passes + 1
        B.MULTI_STEP_TALLY["steps"] += steps + slots
        launch_s = e0.elapsed_time(e1) / 1e3 / steps
        assert int(traj['invalid_action'].sum()) == 0
        checked = B.verify_against_oracle(env, version, verify) if verify else 0
        # the same launch with the workgroups' waves left to drift (sgx_set_steps_barrier 0; the default keeps them in step beyond 8 slots)
        env.set_steps_barrier(0)
        env.rollout_trajectory(slots, traj)
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record()
        for _ in range(passes):
            env.rollout_trajectory(slots, traj)
        d1.record()
        torch.cuda.synchronize()
        env.set_steps_barrier(-1)
        env.bench_steps_played += (passes + 1) * slots
        B.MULTI_STEP_TALLY["launches"] += passes + 1
        B.MULTI_STEP_TALLY["steps"] += (passes + 1) * slots
        drift_us = d0.elapsed_time(d1) * 1e3 / steps
        # The strided code path against the pointer-per-set path on THE SAME memory: a one-slot "trajectory" that is the env's own output set
        # (in place) against rollout_steps into that set -- what the slot arithmetic costs, with the allocation lottery taken out
        del traj
        torch.cuda.empty_cache()
        own = env.alloc_trajectory(1)
        env.obs, env.mask = own['obs'][0], own['mask'][0]
        env.reward, env.done, env.player = own['reward'][0], own['done'][0], own['player'][0]
        env.invalid_action, env.ending_invalid = own['invalid_action'][0], own['ending_invalid'][0]
        env.observe()

        def timed(fn, k, reps=3):
            best = 1e9
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record(); fn(); b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e3 / k)
            return best
        k = 96
        us_ptr = timed(lambda: env.rollout_steps(k), k)
        us_strided = timed(lambda: env.rollout_trajectory(k, own), k)
        env.bench_steps_played += 6 * k
        B.MULTI_STEP_TALLY["launches"] += 6
        B.MULTI_STEP_TALLY["steps"] += 6 * k
        # The same T steps into a RING of T SEPARATE sets, each from its own placement search (sgx_step_ring: beyond 8 sets the pointers travel in a
        # device table; still one launch per call): what a trajectory buffer costs when every slot's memory is of the fast class
        ring = None
        torch.cuda.empty_cache()                  # (the 64-slot buffer was still referenced by the env's views when the cache was emptied above)
        free, _ = torch.cuda.mem_get_info()
        if args.placement != 'trial' or slots * per_slot >= 0.6 * free:
            ring = {"skipped": "--placement plain" if args.placement != 'trial' else "%.0f GB free, %.0f GB needed" % (free / 1e9, slots * per_slot / 0.6 / 1e9)}
        else:
            del own
            B.place_outputs(env, args)
            budget, wide = B.placement_budgets(args, max(int(args.placement_gb * (1 << 30)), 4 * env.obs.numel() * 4))
            # (1) the table path against the kernel-argument path on THE SAME memory: a ring of three placed sets, then `slots` ring entries
            # over those three buffers (entry i = set i mod 3)
            env.alloc_output_ring(3, tune=True, max_extra_bytes=budget, wide_extra_bytes=wide)
            env.rollout_steps(slots, ring=True)
            us_ring3 = timed(lambda: env.rollout_steps(slots, ring=True), slots, reps=5)
            env.repeat_output_ring(slots)
            env.rollout_steps(slots, ring=True)
            tab_kind = env.last_launch_kind
            us_tab = timed(lambda: env.rollout_steps(slots, ring=True), slots, reps=5)
            env.bench_steps_played += 12 * slots
            B.MULTI_STEP_TALLY["launches"] += 12
            B.MULTI_STEP_TALLY["steps"] += 12 * slots
            checked_tab = B.verify_against_oracle(env, version, verify) if verify else 0
            # (2) `slots` SEPARATE sets, each from a search of its own
            t_search = time.perf_counter()
            reps = env.alloc_output_ring(slots, tune=True, max_extra_bytes=budget, trials=24, wide_extra_bytes=wide)
            t_search = time.perf_counter() - t_search
            env.rollout_steps(slots, ring=True)                       # first touch, untimed
            ring_kind = env.last_launch_kind
            us_ring = timed(lambda: env.rollout_steps(slots, ring=True), slots)
            env.bench_steps_played += 4 * slots
            B.MULTI_STEP_TALLY["launches"] += 4
            B.MULTI_STEP_TALLY["steps"] += 4 * slots
            kept = sorted(min(r['obs']) for r in reps[1:] if r and r.get('obs'))
            multi = (_lib.LAUNCH_MULTI_STEP_WAVE, _lib.LAUNCH_MULTI_STEP)
            frac_of = lambda us: B.b_min(v, False, env.record_bytes, float(slots)) * n / (us * 1e-6) / 1e9 / B.HBM_PEAK_GBS
            ring = {"sets": slots, "one_launch": ring_kind in multi, "launch_us": round(us_ring, 2), "value": n / (us_ring * 1e-6), "frac": frac_of(us_ring),
                    "placement_search_seconds": round(t_search, 1),
                    "kept_in_place_us_min_median_max": [round(kept[0], 1), round(kept[len(kept) // 2], 1), round(kept[-1], 1)] if kept else None,
                    "sets_of_the_slow_class (in place > 1.1 x the fastest)": sum(1 for k in kept if k > 1.1 * kept[0]) if kept else None,
                    "verified_envs": B.verify_against_oracle(env, version, verify) if verify else 0,
                    "same_memory": {"what": "a ring of 3 placed sets (pointers in the kernel arguments) against %d ring entries over the same three buffers "
                                            "(pointers in the device table), %d steps per launch" % (slots, slots),
                                    "ring_of_3_us": round(us_ring3, 2), "table_of_%d_us" % slots: round(us_tab, 2), "table_over_ring_of_3": round(us_tab / us_ring3, 4),
                                    "one_launch": tab_kind in multi, "frac_table": frac_of(us_tab), "verified_envs": checked_tab}}
        fused = float(slots)
        per_step = B.b_min(v, False, env.record_bytes, fused) + 4            # + the drawn action of every step (actions log)
        return {"workload": "%d concurrent %s games, rollout into a trajectory buffer of %d slots (sgx_step_traj: obs / mask / rewards / flags / drawn "
                            "action of EVERY step kept; n_steps = n_slots per call)" % (n, version, slots),
                "slots": slots, "buffer_gb": round(slots * per_slot / 1e9, 1), "buffers": "plain torch.empty",
                "one_launch": kind == _lib.LAUNCH_MULTI_STEP_WAVE or kind == _lib.LAUNCH_MULTI_STEP, "launch_kind": kind,
                "value": n * steps / elapsed, "unit": "env steps/s", "steps": steps, "launch_us": launch_s * 1e6,
                "b_min_bytes_per_step": per_step, "frac": per_step * n / launch_s / 1e9 / B.HBM_PEAK_GBS,
                "launch_us_with_drifting_waves (sgx_set_steps_barrier 0)": round(drift_us, 2),
                "same_memory_in_place_us_per_step": {"pointer_per_set_path (sgx_step_n)": round(us_ptr, 2), "strided_slot_path (sgx_step_traj, 1 slot)": round(us_strided, 2)},
                "ring_of_separately_placed_sets": ring,
                "bound": "address translation, not DRAM: the same launch runs at 8.2 TB/s while the sets it writes cover <= 16 GB and falls to 6.9 TB/s "
                         "from 64 GB on, whatever the placement class of each set (tools/ring_footprint_probe.py); under it GRBM_UTCL2_BUSY is 41 % of the "
                         "cycles against 0.3 %, TCP_UTCL1_TRANSLATION_MISS 196 x (profiles/r06_ring_footprint_counters.txt); DRAM-side counters equal.  Beyond 8 sets "
                         "the waves of a workgroup are kept in step (sgx_set_steps_barrier), which recovers 3-5 % of it",
                "verified_envs": checked, "verified_steps": env.bench_steps_played}
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()


def as_it_comes_leg(B, rk, args, version='barrage', n=65536, steps=128, verify=8):
    """What a caller who changes nothing gets: VecStrategoEnv(...) with its default placement (the bounded search of its first reset()),
    alloc_output_ring(3) with its defaults, rollout_steps(..., ring=True) -- the headline's workload without bench.py's own, larger search --
    and the same env stepping in place with one launch per step (env.step()'s shape)."""
    import torch
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.vec_env import VecStrategoEnv
    v = VARIANTS[version]
    env = VecStrategoEnv(version, n, device=rk.device_index, seed=B.BASE_SEED, env_id_offset=0, auto_reset=True)
    try:
        t0 = time.perf_counter()
        env.reset()
        torch.cuda.synchronize()
        t_reset = time.perf_counter() - t0
        env.bench_steps_played = 0
        env.sample_valid_actions()
        ring_reps = env.alloc_output_ring(3)

        def timed(fn, k, reps=3):
            best = 1e9
            for _ in range(reps):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record(); fn(); b.record()
                torch.cuda.synchronize()
                best = min(best, a.elapsed_time(b) * 1e3 / k)
            return best
        env.rollout_steps(steps, ring=True)
        us_ring = timed(lambda: env.rollout_steps(steps, ring=True), steps)
        env.set_multi_step(False)
        env.rollout_steps(16)
        us_step = timed(lambda: env.rollout_steps(steps // 2), steps // 2)
        env.set_multi_step(True)
        env.bench_steps_played += 4 * steps + 16 + 3 * (steps // 2)
        B.MULTI_STEP_TALLY["launches"] += 4
        B.MULTI_STEP_TALLY["steps"] += 4 * steps
        rep = env.placement_report or {}
        return {"workload": "%d concurrent %s games, VecStrategoEnv with every default (placement='search' at the first reset()), alloc_output_ring(3), "
                            "rollout_steps(%d, ring=True)" % (n, version, steps),
                "value": n / (us_ring * 1e-6), "unit": "env steps/s", "launch_us": us_ring,
                "frac": B.b_min(v, False, env.record_bytes, float(steps)) * n / (us_ring * 1e-6) / 1e9 / B.HBM_PEAK_GBS,
                "in_place_one_launch_per_step": {"value": n / (us_step * 1e-6), "launch_us": us_step,
                                                 "rate_over_spec_peak": B.b_min(v, False, env.record_bytes) * n / (us_step * 1e-6) / 1e9 / B.HBM_PEAK_GBS},
                "first_reset_seconds": round(t_reset, 3), "search_candidates_us": [round(x, 1) for x in rep.get('obs', [])],
                "ring_sets_candidates_us": [[round(x, 1) for x in (r or {}).get('obs', [])] for r in ring_reps[1:]],
                "held_for_a_moment_gb": round(env.placement_peak_extra_bytes / 1e9, 2),
                "verified_envs": B.verify_against_oracle(env, version, verify) if verify else 0, "verified_steps": env.bench_steps_played}
    finally:
        env._ring = None
        env.close()
        del env
        torch.cuda.empty_cache()


def facade_leg(B, n_steps=3000, version='barrage', runs=3, good_enough=None):
    """BASELINE config 1 on the GPU: one game behind StrategoMultiAgentEnv (dict in / dict out), a random valid action per step
    (loop:34-63).  -> steps per second of the whole loop, and of the env.step() calls alone (the caller's action choice and reset()
    excluded).  Best of three runs of n_steps."""
    import numpy as np
    from stratego_env_amd import GameVersions, ObservationModes
    from stratego_env_amd.multiagent_env import StrategoMultiAgentEnv
    env = StrategoMultiAgentEnv({'version': GameVersions(version), 'human_inits': True, 'observation_mode': ObservationModes.PARTIALLY_OBSERVABLE})
    MASK = 'valid_actions_mask'
    best = None
    try:
        for run in range(1 + runs):
            np.random.seed(run)
            obs = env.reset()
            steps = games = 0
            in_step = 0.0
            limit = 300 if run == 0 else n_steps              # (run 0: warm-up)
            t0 = time.perf_counter()
            while steps < limit:
                (p,) = obs.keys()
                valid = np.flatnonzero(obs[p][MASK])
                a = int(valid[np.random.randint(valid.size)])
                s0 = time.perf_counter()
                obs, rew, done, info = env.step({p: a})
                in_step += time.perf_counter() - s0
                steps += 1
                if done['__all__']:
                    games += 1
                    obs = env.reset()
            dt = time.perf_counter() - t0
            if run and (best is None or steps / dt > best["steps_per_s"]):
                best = {"steps_per_s": steps / dt, "us_per_step": dt / steps * 1e6, "env_step_calls_per_s": steps / in_step,
                        "us_per_env_step_call": in_step / steps * 1e6, "steps": steps, "games_finished": games}
            if run and good_enough and best["steps_per_s"] >= good_enough:
                break                                      # (the GPU suite's floor test: no need for more runs once one is above the floor)
    finally:
        env.close()
    best["runs"] = run
    best["workload"] = ("ONE %s game behind StrategoMultiAgentEnv.step (dict in, dict out; outputs in host memory the kernel writes directly), "
                        "a random valid action per step chosen on the host, reset() between games; best of %d runs" % (version, best["runs"]))
    return best


def rotating_leg(B, rk, env, args, version, v, steps, warmup, n_sets, full_obs=False, verify=8):
    """The rotating-outputs leg on the env object that was just timed: n_sets output sets (the env's own + n_sets - 1 more, each from
    its own placement trial) written round-robin, sgx_step_ring.  With 3 x 2 GB of outputs nothing a launch writes can still be in
    the 256 MiB Infinity Cache when the same addresses are written again, three launches later: this leg's launch time is DRAM's."""
    import torch
    budget = max(int(args.placement_gb * (1 << 30)), 4 * env.obs.numel() * 4) if (args.placement == 'trial' and args.placement_gb > 0) else 0
    budget, wide = B.placement_budgets(args, budget)
    tune = budget >= env.obs.numel() * 4 and env.obs.numel() * 4 > 300e6
    reports = env.alloc_output_ring(n_sets, tune=tune, max_extra_bytes=budget, trials=args.placement_trials, wide_extra_bytes=wide)
    elapsed, dev_ms, _, games, invalid = B.time_workload(rk, env, steps, warmup, ring=True)
    assert invalid == 0
    fused = B.fused_steps_of(env, steps)
    checked = B.verify_against_oracle(env, version, verify, both=full_obs) if verify else 0
    launch_s = dev_ms / 1e3 / steps
    per_set = [(round(r['obs'][0], 1), round(min(r['obs']), 1)) if (r and r.get('obs')) else None for r in reports]
    set_bytes = env.obs.numel() * 4 + env.mask.numel() + (env.fobs.numel() * 4 if env.fobs is not None else 0)
    return {"workload": "the same rollout writing %d output sets round-robin (sgx_step_ring: a trajectory buffer of the last %d steps)" % (n_sets, n_sets),
            "output_sets": n_sets, "bytes_per_set": set_bytes, "exceeds_infinity_cache": bool((n_sets - 1) * set_bytes > (256 << 20)),
            "value": env.num_envs * steps / elapsed, "unit": "env steps/s", "steps": steps, "warmup": warmup, "launch_us": launch_s * 1e6,
            "frac_dram": B.b_min(v, full_obs, env.record_bytes, fused) * env.num_envs / launch_s / 1e9 / B.HBM_PEAK_GBS,
            "steps_per_launch": fused,
            "placement_plain_and_kept_us_per_extra_set": per_set[1:], "games_finished_in_timed_region": games,
            "verified_envs": checked, "verified_steps": env.bench_steps_played}, launch_s


def consumer_leg(B, rk, args, version='barrage', n=65536, rounds=3, steps_per_round=16, n_check=8):
    """A consumer in the loop (examples/basic_game_loop.py:6-31, 48-63 for a batch): every step is the policy of
    stratego_env_amd/examples/batched_policy_loop.py -- logits from the observation (mean over the board, a fixed linear read-out: the
    stand-in for a network) -- then the library's chooser sgx_choose_actions (invalid actions masked out, softmax, one sample per game
    with the env's counter RNG: the logits and the mask the step wrote are READ on the device), then sgx_step with the chosen actions.
    In-process A/B of the observation store policy under that reader: rounds of `steps_per_round` steps alternate between
    sgx_set_nt_stores(1) (non-temporal interior lines: the default at this size) and (0) (plain stores) on the same env object and
    buffers; reported per policy: whole-loop env steps/s, the step kernel's and the chooser's own time inside the loop (HIP events).
    One more round runs the round-4 chooser composed from torch ops (masked_fill, softmax, multinomial) for comparison.  The actions of
    `n_check` sampled envs are logged on the device and replayed on the CPU oracle afterwards (same setups by the counter RNG,
    auto-reset included): the last step's mask / observation / rewards / flags must match bit for bit."""
    import numpy as np
    import torch
    from stratego_env_amd.config import VARIANTS
    from stratego_env_amd.examples.batched_policy_loop import choose_actions
    v = VARIANTS[version]
    env = B.make_env(version, n, 0, rk.device_index)
    try:
        trial = B.place_outputs(env, args)
        dev = env.device
        g = torch.Generator(device=dev)
        g.manual_seed(1234)
        readout = torch.randn(env.obs.shape[-1], env.mask[0].numel(), device=dev, generator=g) * 0.5
        logits_buf = torch.empty((n, env.mask[0].numel()), dtype=torch.float32, device=dev)
        chosen = torch.empty((n,), dtype=torch.int32, device=dev)
        ids = np.unique(np.linspace(0, n - 1, n_check).astype(np.int64))
        idx = torch.from_numpy(ids).to(dev)
        total_steps = 2 * rounds * steps_per_round + steps_per_round + 8
        act_log = torch.zeros((total_steps, len(ids)), dtype=torch.int32, device=dev)
        done_log = torch.zeros((total_steps, len(ids)), dtype=torch.uint8, device=dev)
        obs, mask = env.obs, env.mask
        played = 0

        def loop(k, events=None, fused=True):
            nonlocal played, obs, mask
            for i in range(k):
                if fused:
                    logits = torch.matmul(obs.mean(dim=(1, 2)), readout, out=logits_buf)
                    if events is not None:
                        events[i][2].record()
                    a = env.choose_actions(logits, 1.0, out=chosen)
                else:
                    a = choose_actions(obs, mask, readout, g)
                act_log[played] = a[idx]
                if events is not None:
                    events[i][0].record()
                obs, mask, _, done, _ = env.step(a)
                if events is not None:
                    events[i][1].record()
                done_log[played] = done[idx]
                played += 1

        loop(4)                                             # untimed: allocator warm-up of the policy's temporaries
        loop(4, fused=False)                                # ... and of the torch-composed chooser's (its first calls load kernels and grow the cache)
        res = {m: {"s": 0.0, "kernel_ms": 0.0, "chooser_ms": 0.0, "steps": 0} for m in (1, 0, 'torch')}
        for rnd in range(rounds + 1):
            for mode in ((1, 0) if rnd < rounds else ('torch',)):
                env.set_nt_stores('auto' if mode == 'torch' else bool(mode))
                ev = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(3)) for _ in range(steps_per_round)]
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                loop(steps_per_round, ev, fused=mode != 'torch')
                torch.cuda.synchronize()
                res[mode]["s"] += time.perf_counter() - t0
                res[mode]["kernel_ms"] += sum(e[0].elapsed_time(e[1]) for e in ev)
                if mode != 'torch':
                    res[mode]["chooser_ms"] += sum(e[2].elapsed_time(e[0]) for e in ev)      # (includes the copy of 8 logged actions)
                res[mode]["steps"] += steps_per_round
        env.set_nt_stores('auto')
        assert int(env.invalid_action.sum()) == 0
        # the chooser on its own: back-to-back calls on the last logits and the current mask (the in-loop figure brackets the copy of the
        # logged actions too and starts from the caches the matmul left behind)
        ce = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(3):
            env.choose_actions(logits_buf, 1.0, out=chosen)
        ce[0].record()
        for _ in range(20):
            env.choose_actions(logits_buf, 1.0, out=chosen)
        ce[1].record()
        torch.cuda.synchronize()
        chooser_us = ce[0].elapsed_time(ce[1]) / 20 * 1e3
        # ---- replay the logged actions of the sampled envs on the CPU oracle
        orc, cv = B.oracle_variant(version)
        acts, dones = act_log.cpu().numpy(), done_log.cpu().numpy()
        mk, ob = env.mask[idx].cpu().numpy(), env.obs[idx].cpu().numpy()
        rw, dn, pl = env.reward[idx].cpu().numpy(), env.done[idx].cpu().numpy(), env.player[idx].cpu().numpy()
        ei = env.ending_invalid[idx].cpu().numpy()
        for c, e in enumerate(ids):
            oe = orc.OracleEnv(v.rows, v.columns, v.max_turns, v.obstacle_locations, v.piece_counts)
            game = 0
            oe.reset(initial_state_override=orc.reset_state(cv, B.BASE_SEED, int(e), game))
            for t in range(played):
                o, r, d, info = oe.step({oe.player: int(acts[t, c])})
                if bool(d["__all__"]) != bool(dones[t, c]):
                    raise SystemExit("bench.py consumer leg: env %d done flag differs from the oracle at step %d" % (int(e), t))
                last = (o, r, d, info)
                if d["__all__"]:
                    game += 1
                    first = oe.reset(initial_state_override=orc.reset_state(cv, B.BASE_SEED, int(e), game))
                    last_obs, last_player = first[1], 1
                else:
                    last_player = oe.player
                    last_obs = o[last_player]
            o, r, d, info = last
            want_mask = last_obs[oe.MASK].astype(np.uint8)
            want_obs = last_obs[oe.POBS]
            want_rw = np.asarray([r.get(1, 0), r.get(-1, 0)], dtype=np.float32) if d["__all__"] else np.zeros(2, np.float32)
            want_ei = int(bool(d["__all__"]) and info[1]['game_result_was_invalid'])
            ok = (np.array_equal(want_mask, mk[c]) and want_obs.tobytes() == ob[c].tobytes() and np.array_equal(want_rw, rw[c])
                  and int(dn[c]) == int(d["__all__"]) and int(pl[c]) == last_player and int(ei[c]) == want_ei)
            if not ok:
                raise SystemExit("bench.py consumer leg: env %d differs from the CPU oracle replaying its %d logged actions" % (int(e), played))

        def rep(m):
            r = res[m]
            out = {"value": n * r["steps"] / r["s"], "unit": "env steps/s", "ms_per_loop_step": r["s"] / r["steps"] * 1e3,
                   "step_kernel_us_in_loop": r["kernel_ms"] / r["steps"] * 1e3, "steps": r["steps"]}
            if m != 'torch':
                out["chooser_us_in_loop"] = r["chooser_ms"] / r["steps"] * 1e3
            return out
        nt, plain = rep(1), rep(0)
        na = env.mask[0].numel()
        return {"workload": "%d concurrent %s games: policy logits from the observation (mean over the board + a fixed linear read-out, torch), "
                            "sgx_choose_actions (reads the logits + the mask: masked softmax, one sample per game with the counter RNG), then sgx_step; "
                            "%d rounds x %d steps per store policy, interleaved" % (n, version, rounds, steps_per_round),
                "nt_stores": nt, "plain_stores": plain,
                "chooser": {"kernel": "choose_kernel<%d,%d,4,false>" % (v.rows, v.columns), "bytes_per_game": 4 * na + na + 4 + 32,
                            "us_per_call_back_to_back": chooser_us, "bound": "hbm (reads)",
                            "frac": (4 * na + na + 36) * n / (chooser_us * 1e-6) / 1e9 / B.HBM_PEAK_GBS},
                "torch_composed_chooser": dict(rep('torch'), note="the round-4 consumer: masked_fill + softmax + multinomial as torch ops"),
                "default_policy_at_this_size": "nt_stores (observation bytes per launch > 300 MB)",
                "nt_over_plain_step_kernel": nt["step_kernel_us_in_loop"] / plain["step_kernel_us_in_loop"],
                "verified_envs": int(len(ids)), "verified_steps": played, "placement": trial}
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()


def compact_leg(B, rk, args, version='barrage', n=65536, seconds=0.5, verify=8):
    """Opt-in compact outputs (SGX_STEP_COMPACT_OBS / _MASK; never the headline): the same rollout writing 4-bit codes + mask bits -- 1/8 of
    the bytes per step -- and, separately, the decode ops that expand a batch to the contract's float32 observation / uint8 mask.
    Verified like every leg (the DECODED last step against the oracle)."""
    import torch
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    env = B.make_env(version, n, 0, rk.device_index, compact=True)
    try:
        _, probe_ms, _, _, _ = B.time_workload(rk, env, 8, 8)
        steps = int(max(16, min(4096, seconds * 1e3 / max(probe_ms / 8, 1e-3))))
        elapsed, dev_ms, _, games, invalid = B.time_workload(rk, env, steps, 4)
        assert invalid == 0
        checked = B.verify_against_oracle(env, version, verify) if verify else 0
        launch_s = dev_ms / 1e3 / steps
        per_step = 2 * env.record_bytes + 8 + env.compact_obs_stride + 4 * env.compact_mask_words + 12
        obs_out, mask_out = env.decode_obs(), env.decode_mask()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(20):
            env.decode_obs(obs_out)
        ev[1].record()
        for _ in range(20):
            env.decode_mask(mask_out)
        ev[2].record()
        torch.cuda.synchronize()
        dec_obs_us, dec_mask_us = ev[0].elapsed_time(ev[1]) / 20 * 1e3, ev[1].elapsed_time(ev[2]) / 20 * 1e3
        # a 64-slot trajectory buffer of compact outputs: 1/8 of the float32 buffer's bytes -- inside the reach of the address translation caches,
        # which a 128 GB float32 buffer is not (trajectory leg)
        slots = 64
        traj = env.alloc_trajectory(slots)
        env.rollout_trajectory(slots, traj)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            env.rollout_trajectory(slots, traj)
        e1.record()
        torch.cuda.synchronize()
        env.bench_steps_played += 4 * slots
        traj_us = e0.elapsed_time(e1) * 1e3 / (3 * slots)
        traj_gb = sum(t.numel() * t.element_size() for t in traj.values()) / 1e9
        traj_checked = B.verify_against_oracle(env, version, verify) if verify else 0
        del traj
        return {"workload": "%d concurrent %s games, same rollout with COMPACT outputs (opt-in: uint8 codes [N,%d] + int32 mask bits [N,%d])"
                            % (n, version, env.compact_obs_stride, env.compact_mask_words),
                "value": n * steps / elapsed, "unit": "env steps/s", "steps": steps, "launch_us": launch_s * 1e6,
                "bytes_per_step": per_step, "frac": per_step * n / launch_s / 1e9 / B.HBM_PEAK_GBS,
                "decode_obs_us_per_batch": dec_obs_us, "decode_mask_us_per_batch": dec_mask_us,
                "decode_obs_frac": (env.compact_obs_stride + 4 * 67 * v.rows * v.columns) * n / (dec_obs_us * 1e-6) / 1e9 / B.HBM_PEAK_GBS,
                "trajectory_64_slots": {"buffer_gb": round(traj_gb, 1), "launch_us": round(traj_us, 2), "value": n / (traj_us * 1e-6),
                                        "frac": (per_step + 4) * n / (traj_us * 1e-6) / 1e9 / B.HBM_PEAK_GBS, "verified_envs": traj_checked},
                "games_finished_in_timed_region": games, "verified_envs": checked}
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()


def other_workload(B, rk, args, version, n, seconds=1.0, chains=1, full_obs=False, verify=8, rotate_sets=0):
    """One of the other BASELINE configs on this GPU, about `seconds` of timed steps; output buffers built like the headline's."""
    import torch
    from stratego_env_amd.config import VARIANTS
    v = VARIANTS[version]
    env = B.make_env(version, n, 0, rk.device_index, full_obs=full_obs)
    try:
        trial = B.place_outputs(env, args)
        _, probe_ms, _, _, _ = B.time_workload(rk, env, 8, 8)
        steps = int(max(16, min(4096, seconds * 1e3 / max(probe_ms / 8, 1e-3))))
        elapsed, dev_ms, _, games, invalid = B.time_workload(rk, env, steps, 4)
        fused = B.fused_steps_of(env, steps)                 # multi-step launches: the games stay on the chip between the steps, the record travels once per launch
        two, per_step_launches = None, None
        if chains > 1:                     # the same steps with the batch split over concurrent chains of launches (sgx_rollout)
            e2, d2, _, _, inv2 = B.time_workload(rk, env, steps, 4, chains=chains)
            assert inv2 == 0
            two = {"chains": chains, "value": n * steps / e2, "us_per_step": d2 / steps * 1e3,
                   "frac": B.b_min(v, full_obs, env.record_bytes) * n / (d2 / 1e3 / steps) / 1e9 / B.HBM_PEAK_GBS}
        if fused > 1 and chains > 1:       # ... and one launch per step (what every round before this one measured), same env object
            env.set_multi_step(False)
            e3, d3, _, _, inv3 = B.time_workload(rk, env, steps, 4)
            env.set_multi_step(True)
            assert inv3 == 0
            per_step_launches = {"value": n * steps / e3, "us_per_step": d3 / steps * 1e3,
                                 "frac": B.b_min(v, full_obs, env.record_bytes) * n / (d3 / 1e3 / steps) / 1e9 / B.HBM_PEAK_GBS}
        assert invalid == 0
        checked = B.verify_against_oracle(env, version, verify, both=full_obs) if verify else 0
        launch_s = dev_ms / 1e3 / steps
        rot, rot_s = None, None
        if rotate_sets >= 2:               # (toy boards: 3 x 248 MB of outputs rotate past the Infinity Cache too)
            rot, rot_s = rotating_leg(B, rk, env, args, version, v, steps, 4, rotate_sets, full_obs=full_obs, verify=verify)
        rf = B.roofline(version, v, n, launch_s, full_obs=full_obs, rec_bytes=env.record_bytes, build_id=env.build_id, rotating=rot_s, fused_steps=fused)
        # on the plain first allocation: this leg's step time scaled by the trial's observe launches, first candidate / kept one (the
        # observe launch itself is not this leg's step: cheaper on the toy boards, and in BOTH mode the candidates were timed per buffer)
        tr = trial or {}
        first, kept = tr.get('fobs_plain_us' if full_obs else 'plain_us'), tr.get('fobs_kept_us' if full_obs else 'kept_us')
        rf["frac_untuned"] = rf["frac"] * kept / first if (first and kept) else None
        return {"workload": "%d concurrent %s games (%dx%d)%s, same rollout" % (n, version, v.rows, v.columns,
                                                                                 ", BOTH_OBSERVATIONS (67 + 79 channels)" if full_obs else ""),
                "value": n * steps / elapsed, "unit": "env steps/s", "steps": steps, "launch_us": launch_s * 1e6,
                "frac": rf["frac"], "frac_dram": rf["frac_dram"], "frac_untuned": rf["frac_untuned"],
                "b_min_bytes_per_step": rf["b_min_bytes_per_step"], "traffic": rf["traffic"], "traffic_source": rf["traffic_source"],
                "survey_8d": rf["survey_8d"], "kernel": rf["kernel"],
                "games_finished_in_timed_region": games, "concurrent_chains": two, "rotating_outputs": rot, "verified_envs": checked,
                "steps_per_launch": fused, "one_launch_per_step": per_step_launches,
                "placement": trial}
    finally:
        env.close()
        del env
        torch.cuda.empty_cache()
