#!/usr/bin/env python3
"""Headline benchmark: env-steps/sec of one full PPO iteration (lockstep batched act over N
synthetic envs x T steps -> Monte-Carlo return scan + normalisation -> K-epoch clipped update)
on N GPUs of one node, weak scaling (fixed envs per GPU).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
         --master-port 29511 bench.py --gpus 8 --steps 5 --warmup 2
  python bench.py --config c1        # reference-shaped: one env, T single-state select_action calls + update()

A "step" is one full iteration over one batch of synthetic input.  The timed loop runs the default
product path (the calls a user of the PPO class makes); per-kernel times come from one extra,
untimed, instrumented iteration afterwards.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # the headline configuration: synthetic vectorised CartPole, 4096 envs/GPU x 128 steps
    "c4": dict(name="synthetic CartPole-shaped (S=4, A=2 discrete), {n_envs} envs/GPU x {horizon} steps, K={K}",
               S=4, A=2, continuous=False, std=None, n_envs=4096, horizon=128, K=80,
               reward_kind=0, p_terminal=1.0 / 200.0),
    # the large configuration: HalfCheetah-shaped continuous (obs 26, act 6), 65536 envs/GPU
    "c5": dict(name="synthetic HalfCheetah-shaped (S=26, A=6 continuous), {n_envs} envs/GPU x {horizon} steps, K={K}",
               S=26, A=6, continuous=True, std=0.6, n_envs=65536, horizon=128, K=80,
               reward_kind=1, p_terminal=-1.0),
    # reference-shaped configurations: ONE env, the reference's per-step protocol (train.py:173-188):
    # `horizon` single-state select_action calls with scalar reward / terminal appends, then update()
    "c1": dict(name="CartPole-v1-shaped (S=4, A=2 discrete), 1 env x {horizon} steps per update, K={K}, "
                    "single-state select_action protocol",
               S=4, A=2, continuous=False, std=None, n_envs=1, horizon=1600, K=80,
               reward_kind=0, p_terminal=1.0 / 200.0),
    "c2": dict(name="LunarLander-v2-shaped (S=8, A=4 discrete), 1 env x {horizon} steps per update, K={K}, "
                    "single-state select_action protocol",
               S=8, A=4, continuous=False, std=None, n_envs=1, horizon=900, K=30,
               reward_kind=1, p_terminal=1.0 / 300.0),
    "c3": dict(name="BipedalWalker-v2-shaped (S=24, A=4 continuous, std 0.6), 1 env x {horizon} steps per "
                    "update, K={K}, single-state select_action protocol",
               S=24, A=4, continuous=True, std=0.6, n_envs=1, horizon=6000, K=80,
               reward_kind=1, p_terminal=1.0 / 1500.0),
}
PEAK_FP32_TFLOPS = 157.3   # MI355X dense fp32 (vector = matrix) peak
PEAK_HBM_GBS = 8000.0
SCAN_BYTES_PER_STEP = 9        # fp32 reward + u8 terminal in, fp32 return out
NORMALIZE_BYTES_PER_STEP = 16  # return + old value in, normalised return + advantage out


def flops_per_sample_epoch(S, A):
    F = 2 * S * 64 + 2 * 64 * 64 + 64 * (A + 1)          # MACs of one actor+critic forward
    return 2 * (3 * F - 2 * S * 64)                      # fwd + wgrad + dgrad (no input gradient)


def host_description(threads):
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    import torch
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(),
            "torch_num_threads": torch.get_num_threads(), "blas_threads_used": threads}


def cpu_baseline(cfg, threads):
    """The CPU restatement (oracle/, numpy) timed on a bounded sample of the same workload."""
    import numpy as np
    from threadpoolctl import threadpool_limits
    from oracle import ppo_oracle as O
    S, A, cont, std, K = cfg["S"], cfg["A"], cfg["continuous"], cfg["std"], cfg["K"]
    rng = np.random.default_rng(0)
    shapes = O.param_shapes(S, A)
    P = {k: (rng.standard_normal(s) * 0.1).astype(np.float32) for k, s in shapes.items()}
    flat = cfg["n_envs"] == 1
    if flat:     # the whole workload: T single-state act calls, scan + normalise, K epochs over T samples
        Ts, n_act, n_ep = cfg["horizon"], cfg["horizon"], K
    else:        # 10-15 s of CPU work on the GPU box's host cores, at the full buffer size
        Ts, n_act, n_ep = 524288, 150000, 16
    x = rng.standard_normal((Ts, S)).astype(np.float32)
    acts = (rng.standard_normal((Ts, A)).astype(np.float32) if cont else rng.integers(0, A, Ts))
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        for i in range(n_act):           # the reference acts one state per call (PPO.py:175-198)
            j = i % Ts
            O.act_logprob_value(P, x[j:j + 1], acts[j:j + 1], cont, std)
        t_act = (time.perf_counter() - t0) / n_act
        r = rng.standard_normal(Ts).astype(np.float32)
        term = rng.random(Ts) < 1 / 200
        t0 = time.perf_counter()
        ret = O.returns_scan(r, term, 0.99)
        _, _, _, rn = O.normalize_returns(ret)
        t_scan = (time.perf_counter() - t0) / Ts
        lp, v = O.act_logprob_value(P, x, acts, cont, std)
        adv = rn - v
        st = O.adam_init(P)
        t0 = time.perf_counter()
        for _ in range(n_ep):
            _, g = O.ppo_loss_and_grads(P, x, acts, lp, adv, rn, 0.2, cont, std)
            O.adam_step(P, g, st, 3e-4, 1e-3)
        t_ep = (time.perf_counter() - t0) / (n_ep * Ts)
    per_step = t_act + t_scan + K * t_ep
    sample = (f"numpy restatement, the whole workload once: {n_act} single-state act calls, scan+normalise "
              f"of {Ts} steps, {n_ep} update epochs over {Ts} samples") if flat else \
             (f"numpy restatement: {n_act} single-state act calls, scan+normalise of {Ts} "
              f"steps, {n_ep} update epochs over {Ts} samples; extrapolated to K={K}")
    return {"value": 1.0 / per_step, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "host": host_description(threads), "sample": sample,
            "us_per_step": {"act": t_act * 1e6, "scan": t_scan * 1e6, "epoch": t_ep * 1e6}}


_watchdog_file = None

# the other BASELINE.json configurations, measured after the headline one by the default single-GPU run
# (a few iterations each) and reported under "other_configs" of the same JSON line
OTHER_CONFIGS = (("c5", 2, 1), ("c1", 8, 2), ("c2", 8, 2), ("c3", 5, 2))   # (config, steps, warmup)


def measure(config_key, cfg, steps, warmup, world, rank, custom_shape, with_cpu, cpu_threads):
    """One configuration: `warmup` untimed + `steps` timed iterations of the default product path between
    barriers (max over ranks), then one iteration with a host sync between the phases and one
    instrumented iteration for the kernel times.  Returns the fields of the JSON line (rank 0: all of
    them; other ranks: None)."""
    import contextlib
    import ctypes
    import io
    import numpy as np
    import torch
    import torch.distributed as td
    with contextlib.redirect_stdout(io.StringIO()):
        from ppo_amd import _lib, dist as _pdist, ppo
    flat = cfg["n_envs"] == 1
    S, A, N, T, K = cfg["S"], cfg["A"], cfg["n_envs"], cfg["horizon"], cfg["K"]
    torch.manual_seed(0)                       # identical initial weights on every rank
    agent = ppo.PPO(S, A, 3e-4, 1e-3, 0.99, K, 0.2, cfg["continuous"], cfg["std"])
    agent.buffer.reserve(N * T)
    env_seed = 1234 + rank
    it = [0]
    host_rng = np.random.default_rng(env_seed)
    if flat:    # the host-side "env" of the reference-shaped configurations: precomputed draws
        flat_states = host_rng.standard_normal((T, S))                       # float64, as gym hands over
        flat_rewards = [1.0] * T if cfg["reward_kind"] == 0 else host_rng.standard_normal(T).tolist()
        flat_terms = (host_rng.random(T) < cfg["p_terminal"]).tolist()

    def rollout():
        if flat:
            # the loop of train.py:173-181: one state in, one action out, scalar appends
            b = agent.buffer
            for t in range(T):
                agent.select_action(flat_states[t])
                b.rewards.append(flat_rewards[t])
                b.is_terminals.append(flat_terms[t])
        else:
            # T lockstep steps: synthetic env step (obs, reward, terminal) then batched act on the N
            # states, written straight into the device buffer
            agent.synthetic_rollout(N, T, cfg["reward_kind"], cfg["p_terminal"], step0=it[0] * T,
                                    env_seed=env_seed)

    def iteration(timed_phases=None):
        rollout()
        if timed_phases is not None:
            torch.cuda.synchronize()
            timed_phases["rollout"] = time.perf_counter()
        agent.update()
        it[0] += 1

    def barrier():
        # drain this rank's device work, meet the other ranks, drain again (an RCCL barrier is itself
        # device work on the current stream)
        torch.cuda.synchronize()
        if world > 1:
            td.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        iteration()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        iteration()
    barrier()
    dt = time.perf_counter() - t0
    in_group = td.is_available() and td.is_initialized()      # (also the one-rank rehearsal group)
    on_device = in_group and td.get_backend() == "nccl"
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_device else "cpu")
        td.all_reduce(tmax, op=td.ReduceOp.MAX)
        dt = float(tmax.item())

    # one more iteration with a host sync between phases, for the per-phase split (untimed)
    ph = {}
    torch.cuda.synchronize()
    p0 = time.perf_counter()
    iteration(ph)
    torch.cuda.synchronize()
    p2 = time.perf_counter()
    phases = {"rollout_ms": (ph["rollout"] - p0) * 1e3, "update_ms": (p2 - ph["rollout"]) * 1e3}
    if agent.last_dp_enqueue_s is not None:
        # data-parallel path: host time spent enqueueing the K-epoch loop (launches + collectives)
        phases["dp_host_enqueue_us_per_epoch"] = agent.last_dp_enqueue_s / K * 1e6

    # one instrumented iteration (untimed): the library brackets every launch of the K-epoch loop with
    # events on the launch stream (ppo_kernel_timings; the default product path, as in the timed loop);
    # the data-parallel path, which has no such loop in C, collects the events itself (and events around
    # every gradient all-reduce).  Events around the scan and normalise launches either way.
    c_timed = not _pdist.is_distributed()
    if c_timed:
        _lib.check(_lib.lib().ppo_kernel_timings_enable(1))
    else:
        agent.kernel_events = []
        agent.comm_events = []
    agent.phase_events = []
    iteration()
    torch.cuda.synchronize()
    events, pev, cev = agent.kernel_events, agent.phase_events, agent.comm_events
    agent.kernel_events = agent.phase_events = agent.comm_events = None
    reduce_adam_ms = None
    if c_timed:
        tbuf = (ctypes.c_float * 3)()
        _lib.check(_lib.lib().ppo_kernel_timings(tbuf, 3))
        _lib.check(_lib.lib().ppo_kernel_timings_enable(0))
        kern_ms, reduce_adam_ms, n_timed = float(tbuf[0]), float(tbuf[1]), int(tbuf[2])
    else:
        kern_ms, n_timed = sum(a.elapsed_time(b) for a, b in events) / max(len(events), 1), len(events)
    phase_ms = {name: a.elapsed_time(b) for name, a, b in pev}
    # the per-epoch cost next to the dominant kernel (reduction + Adam launch, kernel boundaries)
    phases["update_minus_fwdbwd_ms"] = phases["update_ms"] - K * kern_ms
    if reduce_adam_ms is not None:
        phases["reduce_adam_us_per_epoch"] = reduce_adam_ms * 1e3

    # what the first multi-rank run saw: backend, ranks, and whether the replicas agree bit for bit
    dist_info = None
    if in_group:
        flatp = agent.policy.flat.double()
        mine = torch.stack([flatp.sum(), flatp.abs().sum(), agent.last_loss[-1].double(),
                            agent.policy_old.flat.double().sum()])
        mine = mine if on_device else mine.cpu()
        hi, lo = mine.clone(), mine.clone()
        td.all_reduce(hi, op=td.ReduceOp.MAX)
        td.all_reduce(lo, op=td.ReduceOp.MIN)
        spread = (hi - lo).cpu().tolist()
        dist_info = {"backend": td.get_backend(), "world_size": td.get_world_size(),
                     "replica_check": {"param_sum_spread": spread[0], "param_abs_sum_spread": spread[1],
                                       "last_loss_spread": spread[2], "policy_old_sum_spread": spread[3],
                                       "replicas_identical": all(v == 0.0 for v in spread)},
                     "allreduce_us_per_epoch": (sum(a.elapsed_time(b) for a, b in cev) / max(len(cev), 1) * 1e3
                                                if cev else None),
                     "allreduce_bytes": int(agent.optimizer.grad.numel() * 4),
                     "dp_host_enqueue_us_per_epoch": phases.get("dp_host_enqueue_us_per_epoch")}

    traffic, traffic_source, phase_traffic = None, None, {}
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            pmc = json.load(f)
        rec = pmc.get(config_key, {})
        if not custom_shape:
            phase_traffic = pmc.get("hbm_phases_" + config_key, {})
        if rec and not custom_shape:   # measured for the default shape only
            traffic = rec.get("bytes_per_launch")
            traffic_source = f"{rec.get('source')} (rocprofv3 --pmc passes of this workload, not measured by this run)"
    except OSError:
        pass
    final_loss = float(agent.last_loss[-1].item())
    agent.close()
    del agent
    torch.cuda.empty_cache()
    if rank != 0:
        return None
    n = N * T
    steps_total = n * world * steps
    flops_launch = flops_per_sample_epoch(S, A) * n
    achieved = flops_launch / (kern_ms * 1e-3) / 1e12
    hbm = []
    for name, kernel, pmc_key, bps in (("scan", "scan_flat_reduce/carries/apply + moments_finalize" if flat else
                                        "scan_rows_kernel + moments_finalize", "scan_rows_kernel",
                                        SCAN_BYTES_PER_STEP),
                                       ("normalize", "normalize_adv_kernel", "normalize_adv_kernel",
                                        NORMALIZE_BYTES_PER_STEP)):
        if name in phase_ms:
            gbs = bps * n / (phase_ms[name] * 1e-3) / 1e9
            hbm.append({"bound": "hbm", "kernel": kernel, "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": gbs / PEAK_HBM_GBS,
                        "traffic": phase_traffic.get(pmc_key, {}).get("bytes_per_launch"),
                        "avg_launch_ms": phase_ms[name], "bytes_per_step": bps})
    out = {
        "metric": "env-steps/sec (act+return scan+K-epoch update)",
        "value": steps_total / dt, "unit": "env-steps/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": cfg["name"], "n_envs_per_gpu": N, "horizon": T, "K_epochs": K,
                   "state_dim": S, "action_dim": A, "parallelism": f"dp{world}"},
        "roofline": {"bound": "mfma", "kernel": "fwdbwd_kernel", "achieved": achieved,
                     "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
                     "traffic_source": traffic_source,
                     "avg_launch_ms": kern_ms, "launches_timed": n_timed,
                     "flop_per_sample_epoch": flops_per_sample_epoch(S, A),
                     "timed_in": "a separate instrumented iteration after the timed loop, events on the launch "
                                 "stream around every launch" + (" (ppo_kernel_timings)" if c_timed else "")},
        "roofline_hbm_phases": hbm,
        "phases": phases,
        "final_loss": final_loss,
    }
    if dist_info is not None:
        out["distributed"] = dist_info
    if with_cpu and world == 1:
        out["cpu_baseline"] = cpu_baseline(cfg, cpu_threads)
    return out


def main():
    global _watchdog_file
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", default="c4", choices=sorted(CONFIGS))
    ap.add_argument("--k-epochs", type=int, default=None)
    ap.add_argument("--n-envs", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the short runs of the other BASELINE.json configurations after the headline one")
    ap.add_argument("--cpu-threads", type=int, default=16)
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    if args.k_epochs:
        cfg["K"] = args.k_epochs
    if args.n_envs:
        cfg["n_envs"] = args.n_envs
    cfg["name"] = cfg["name"].format(**cfg)
    custom_shape = bool(args.n_envs or args.k_epochs)

    import torch
    import torch.distributed as td
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("PPO_BENCH_WATCHDOG"):     # rehearsal aid: dump the Python stacks of a stuck rank
        import faulthandler
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        # the file object has to stay alive: faulthandler only holds its descriptor
        _watchdog_file = open(os.path.join(ROOT, "gpurun_out", f"watchdog_rank{rank}.txt"), "w")
        faulthandler.dump_traceback_later(int(os.environ["PPO_BENCH_WATCHDOG"]), exit=True, file=_watchdog_file)
    grouped = world > 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # one rank per GPU over RCCL; PPO_BENCH_BACKEND=gloo lets several ranks share one GPU to
        # rehearse the distributed code path on a single-GPU box (not a measurement)
        backend = os.environ.get("PPO_BENCH_BACKEND", "nccl")
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        os.environ["LOCAL_RANK"] = str(local)
        torch.cuda.set_device(local)
        if backend == "nccl":
            td.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            td.init_process_group(backend)
    elif os.environ.get("PPO_BENCH_FORCE_GROUP") == "1":
        # rehearsal: a one-rank RCCL group; with PPO_DIST_FORCE=1 the update takes the data-parallel path
        torch.cuda.set_device(0)
        td.init_process_group("nccl", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1,
                              device_id=torch.device("cuda", 0))
        grouped = True

    # the CPU leg of a multi-rank run is timed by rank 0 after the group is gone, so that no rank waits
    # for it inside a collective
    out = measure(args.config, cfg, args.steps, args.warmup, world, rank, custom_shape,
                  with_cpu=not args.no_cpu_baseline and world == 1, cpu_threads=args.cpu_threads)
    if world == 1 and not grouped and args.config == "c4" and not custom_shape and not args.no_other_configs:
        others = {}
        for key, steps, warmup in OTHER_CONFIGS:
            c = dict(CONFIGS[key])
            c["name"] = c["name"].format(**c)
            try:        # an extra must never cost the headline line
                o = measure(key, c, steps, warmup, 1, 0, False, with_cpu=not args.no_cpu_baseline,
                            cpu_threads=args.cpu_threads)
            except Exception as e:   # noqa: BLE001
                others[key] = {"workload": c["name"], "error": f"{type(e).__name__}: {e}"}
                continue
            others[key] = {"workload": c["name"], "value": o["value"], "unit": o["unit"], "steps": steps,
                           "warmup": warmup, "ms_per_step": o["ms_per_step"],
                           "roofline_frac": o["roofline"]["frac"],
                           "fwdbwd_avg_launch_ms": o["roofline"]["avg_launch_ms"], "phases": o["phases"],
                           "cpu_baseline_value": o.get("cpu_baseline", {}).get("value"),
                           "cpu_baseline_sample": o.get("cpu_baseline", {}).get("sample")}
        out["other_configs"] = others
    if grouped:
        if world > 1:
            td.barrier()
        td.destroy_process_group()
    if rank == 0:
        if world > 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, args.cpu_threads)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
