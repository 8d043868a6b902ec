"""Legs of bench.py's 1-GPU line that were added in round 6 (kept out of bench.py, which holds the contract and the headline):

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
            g_long = probe(t, passes, 2, 1, best_cfg[0], best_cfg[1], 2, best_cfg[2])
            us_long = float(us.value)
        finally:
            stop.set()
            th.join()
        long_launch = {"payload": "random_bits", "waves_per_cu": best_cfg[0], "pace": best_cfg[1], "persistent_waves": best_cfg[2], "passes": passes,
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
        fused = float(slots)
        per_step = B.b_min(v, False, env.record_bytes, fused) + 4            # + the drawn action of every step (actions log)
        return {"workload": "%d concurrent %s games, rollout into a trajectory buffer of %d slots (sgx_step_traj: obs / mask / rewards / flags / drawn "
                            "action of EVERY step kept; n_steps = n_slots per call)" % (n, version, slots),
                "slots": slots, "buffer_gb": round(slots * per_slot / 1e9, 1), "buffers": "plain torch.empty",
                "one_launch": kind == _lib.LAUNCH_MULTI_STEP_WAVE or kind == _lib.LAUNCH_MULTI_STEP, "launch_kind": kind,
                "value": n * steps / elapsed, "unit": "env steps/s", "steps": steps, "launch_us": launch_s * 1e6,
                "b_min_bytes_per_step": per_step, "frac": per_step * n / launch_s / 1e9 / B.HBM_PEAK_GBS,
                "same_memory_in_place_us_per_step": {"pointer_per_set_path (sgx_step_n)": round(us_ptr, 2), "strided_slot_path (sgx_step_traj, 1 slot)": round(us_strided, 2)},
                "verified_envs": checked, "verified_steps": env.bench_steps_played}
    finally:
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
