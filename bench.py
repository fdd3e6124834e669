#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric on MI355X.

Workload (BASELINE.json configs[1]): the batched edge_computing environment, 4096 envs x 10 agents per
GPU, task_size_max=6000, uniform-random policy over the availability mask.  One bench "step" = ONE WHOLE EPISODE of
the rollout hot path over the whole batch (warm up on whole episodes, time whole episodes): reset,
first observation + mask, first actions, then episode_limit (100) env steps -- offload, 10 mini-slots of
execute/receive, statistics, generate_task, the next observation + availability mask and the random policy's draw
for the next step -- as ParallelRunner.run does.  `value` stays in agent-steps/s (envs x agents x episode_limit per
bench step); `ms_per_step` is per episode.  Inputs and all state are resident in HBM; nothing crosses PCIe inside
the timed region.

--launch rollout (default): ec_rollout_random_policy, one launch per episode (the looping instantiation of k_step).
--launch graph / eager:     ec_step_random_policy, one launch per step (replayed from a hipGraph per episode / eager);
                            the N=1 line reports this path too, under "per_step_launch".
Every step writes its reward, counters, termination flag, observation, mask and next actions in both.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...            (no launcher: starts the torch.distributed.run line above itself, as a
                                             child process, before anything here touches the GPU)

Rank 0 prints ONE JSON line.  Multi-GPU: BASELINE.json's metric is quoted for 4096 envs on the WHOLE node, so
--gpus N shards those 4096 envs (rank r owns env ids [r*4096/N, (r+1)*4096/N): same seeds, hence same trajectories,
as the single-GPU batch; no data-path collective) and the line says "scaling": "strong".  The weak-scaling figure
(4096 envs per GPU) rides beside it under "weak_scaling", and "ppo_iteration" times rollout + update on all ranks
with the learner's flat-gradient all-reduce over RCCL.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20, help="timed bench steps; one step = one whole episode")
    p.add_argument("--warmup", type=int, default=5, help="untimed episodes first")
    p.add_argument("--envs", type=int, default=4096, help="environments of the whole job (sharded over --gpus)")
    p.add_argument("--nodes", type=int, default=10, help="edge_node_num")
    p.add_argument("--launch", choices=("rollout", "graph", "eager"), default="rollout")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU baseline budget (0 = skip)")
    p.add_argument("--roofline-steps", type=int, default=400)
    p.add_argument("--geometry", default="", help="override the step kernel's launch geometry: epw,mw[,producers]")
    p.add_argument("--rehearse", action="store_true",
                   help="multi-rank logic on a box with ONE GPU: every rank uses cuda:0 and the process group is gloo "
                        "(not a measurement: the line is marked)")
    p.add_argument("--ppo-iters", type=int, default=3,
                   help="also time this many PPO iterations (rollout + update) at N=1 and report them beside the metric")
    return p.parse_args()


class Rollout:
    """The hot path of ParallelRunner.run (parallel_episode_runner.py:32-113) on device tensors."""

    def __init__(self, env, launch):
        import torch as th
        self.th = th
        self.env = env
        self.launch = launch
        self.T = env.episode_limit
        self.actions = th.empty(env.n_envs, env.n_agents, dtype=th.int64, device=env.device)
        self.counter_dev = th.zeros(2, dtype=th.int64, device=env.device)
        self.graph = None
        self.steps_done = 0
        if launch == "graph":
            # warm the kernels on a side stream, then capture one whole episode
            s = th.cuda.Stream(device=env.device)
            s.wait_stream(th.cuda.current_stream(env.device))
            with th.cuda.stream(s):
                self._episode(3)
            th.cuda.current_stream(env.device).wait_stream(s)
            th.cuda.synchronize(env.device)
            self.graph = th.cuda.CUDAGraph()
            with th.cuda.graph(self.graph):
                self._episode(self.T)

    def _episode(self, n):
        # the first actions of an episode come from the policy kernel; every step then draws the next step's actions
        # from the mask it has just produced (in place): n launches of ec_step_random_policy, or one of
        # ec_rollout_random_policy
        self.env.reset()
        self.env.observe()
        self.env.random_actions(seed=0, counter=0, out=self.actions, counter_dev=self.counter_dev)
        if self.launch == "rollout":
            self.env.rollout_random_policy(self.actions, n, seed=0)
        else:
            for _ in range(n):
                self.env.step_random_policy(self.actions, seed=0)

    def run(self, n_episodes):
        """Run exactly n_episodes whole episodes (one bench step each)."""
        for _ in range(n_episodes):
            if self.graph is not None:
                self.graph.replay()
            else:
                self._episode(self.T)
        self.steps_done += n_episodes * self.T


def algorithmic_bytes_per_env_step(N, live, draws):
    """I/O (actions, obs, avail, reward, counters/time, terminated) + 16 B/node header
    read+write + 40 B canonical task record read+write per live task + 16 B per consumed MT19937 double."""
    io = 4 * N + 36 * N + 4 * N * (N + 1) + 4 + 20 + 1
    return io + 2 * 16 * N + 2 * 40 * live + 16 * draws


def recorded_traffic(name="pmc_traffic_k_step.json"):
    """HBM bytes per k_step launch from the PMC passes committed under profiles/ (separate rocprofv3 --pmc runs of
    tools/prof_step.py for FETCH_SIZE and WRITE_SIZE, corrected as MI355X_MICROARCH.md prescribes; bench.py cannot
    run under the profiler itself).  None when no such record exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)))
    if not files:
        return None, None
    with open(files[-1]) as f:
        rec = json.load(f)
    return rec.get("traffic_bytes_per_launch"), os.path.relpath(files[-1], ROOT)


def measure_roofline(env, roll, n_steps):
    """Average kernel durations from HIP events recorded on the launch stream around each launch (passes with no host
    synchronisation inside): the one-step k_step (ec_step_random_policy) and the looping k_step of
    ec_rollout_random_policy (episode_limit steps per launch); then the measured L-bar (live tasks/env) and D-bar
    (MT19937 doubles/env-step) of a further pass over the same number of steps."""
    import torch as th

    def run_rollout(events):
        for ep in range(max(1, n_steps // env.episode_limit)):
            env.reset()
            env.observe()
            env.random_actions(seed=0, counter=0, out=roll.actions, counter_dev=roll.counter_dev)
            a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            a.record()
            env.rollout_random_policy(roll.actions, env.episode_limit, seed=0)
            b.record()
            events.append((a, b))
        th.cuda.synchronize(env.device)

    def run(sample_live, events):
        env.reset()
        env.observe()
        live = []
        for t in range(n_steps):
            if t and t % env.episode_limit == 0:
                env.reset()
                env.observe()
            if t % env.episode_limit == 0:
                env.random_actions(seed=0, counter=0, out=roll.actions, counter_dev=roll.counter_dev)
            if events is not None:
                a, b = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                a.record()
                env.step_random_policy(roll.actions, seed=0)
                b.record()
                events.append((a, b))
            else:
                env.step_random_policy(roll.actions, seed=0)
            if sample_live and t % 10 == 5:
                live.append(env.live_tasks() / env.n_envs)
        th.cuda.synchronize(env.device)
        return live

    th.cuda.synchronize(env.device)
    pairs = []
    run(False, pairs)
    ms = sorted(a.elapsed_time(b) for a, b in pairs)
    avg_ms = sum(ms) / len(ms)
    cur0 = env.debug_rng_cursors().astype("float64").sum()
    live = run(True, None)
    cur1 = env.debug_rng_cursors().astype("float64").sum()
    draws = (cur1 - cur0) / 2.0 / (n_steps * env.n_envs)
    lbar = sum(live) / max(1, len(live))
    pairs = []
    run_rollout(pairs)
    rms = sorted(a.elapsed_time(b) for a, b in pairs)
    return avg_ms, ms[len(ms) // 2], sum(rms) / len(rms), rms[len(rms) // 2], lbar, draws


def cpu_baseline(args_ns, tables, n_envs, budget_s):
    """The CPU oracle (a C port of the reference env) doing the same rollout work on the host cores."""
    from marl_amd._abi import config_from_args
    from oracle import ec_oracle
    threads = max(1, min(16, os.cpu_count() or 1))
    ora = ec_oracle.OracleEnv(config_from_args(args_ns), tables, n_envs, n_threads=threads)
    T = args_ns.episode_limit
    ora.rollout_random(T)  # warm-up episode
    steps = 0
    t0 = time.perf_counter()
    while True:
        ora.rollout_random(T, seed=steps)
        steps += T
        dt = time.perf_counter() - t0
        if dt >= budget_s or steps >= 50 * T:
            break
    ora.close()
    return {"value": n_envs * args_ns.edge_node_num * steps / dt, "unit": "agent-steps/s", "cores": threads,
            "kind": "port",
            "sample": "%d envs x %d steps (observe + random policy + step), OpenMP C oracle, %.1f s" % (n_envs, steps, dt)}


def ppo_iterations(ns, B, N, iters, rank=0, world=1, trace=lambda m: None):
    """BASELINE.json's second figure: wall time of one PPO iteration = ParallelRunner.run(test_mode=False):
    the episode as one ec_rollout_mlp_policy launch (Actor + env step, episode_limit times, for all B envs) and one
    ec_store_episode launch into the device buffer, then PPOAgent.train() (reward normalisation, GAE kernel, `epochs` full-batch actor/critic updates)."""
    import torch as th
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    import torch.distributed as dist
    ns.n_threads, ns.algo_type, ns.obs_shape, ns.n_agents, ns.n_actions = B, "rl", 9, N, N + 1
    ns.ec_graph_rollout = True  # only used where ec_rollout_mlp_policy is not compiled for the env shape
    ns.ec_env_id_base = rank * B  # this rank's shard of the job's envs
    th.manual_seed(100 + rank)
    agent = agent_register["ppo"](ns)
    trace("ppo: agent built")
    runner = runner_register["parallel"](ns, agent)
    trace("ppo: runner built")
    train = agent.train
    spans = []

    def timed_train():
        th.cuda.synchronize()
        trace("ppo: rollout done, train")
        t0 = time.perf_counter()
        train()
        th.cuda.synchronize()
        spans.append(time.perf_counter() - t0)
        trace("ppo: train done")

    agent.train = timed_train
    total = []
    for it in range(iters + 1):  # first iteration is warm-up (allocator, kernel load, RCCL communicator)
        th.cuda.synchronize()
        if world > 1:
            dist.barrier()
            th.cuda.synchronize()
        t0 = time.perf_counter()
        runner.run(test_mode=False)
        th.cuda.synchronize()
        total.append(time.perf_counter() - t0)
    mode = "ec_rollout_mlp_policy" if runner._kernel_rollout_ok() else "per-step launches (hipGraph)"
    runner.close_env()
    total, upd = sorted(total[1:]), sorted(spans[1:])
    med = lambda v: v[len(v) // 2]
    t_tot, t_upd = med(total), med(upd)
    if world > 1:  # the iteration is over when the slowest rank is
        from marl_amd.utils import dist_utils
        t = th.tensor([t_tot, t_upd], dtype=th.float64, device="cuda")
        dist_utils.all_reduce_(t, op=dist.ReduceOp.MAX)
        t_tot, t_upd = t.tolist()
    return {"total_ms": 1e3 * t_tot, "update_ms": 1e3 * t_upd, "rollout_ms": 1e3 * (t_tot - t_upd),
            "iters": iters, "envs": B * world, "envs_per_gpu": B, "agents": N, "episode_limit": ns.episode_limit,
            "epochs": ns.epochs, "rows_per_update": B * world * N * ns.episode_limit,
            "rollout": mode,
            "gradient_all_reduce": "one flat buffer per optimiser step over RCCL (nccl backend)" if world > 1 else None,
            "rollout_agent_steps_per_s": B * world * N * ns.episode_limit / (t_tot - t_upd)}


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it: start the contract's torch.distributed.run line as a
    CHILD process (nothing in this process has touched the GPU yet, and it never will), pass its output through and
    exit with its code.  Rank 0 of the child prints the JSON line."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # stdout carries the JSON line only: whatever else the ranks print there (gloo's connection notes) goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        dst = sys.stdout if line.lstrip().startswith("{") else sys.stderr
        dst.write(line)
        dst.flush()
    return proc.wait()


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    import torch as th
    import torch.distributed as dist
    from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv
    from marl_amd.utils import dist_utils
    from tests.util import make_args

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    if a.rehearse:
        local = 0
    th.cuda.set_device(local)
    dev = th.device("cuda:%d" % local)
    if world > 1 and a.rehearse:
        dist.init_process_group("gloo")
    elif world > 1:
        dist.init_process_group("nccl", device_id=dev)
    if a.geometry:  # "envs per wave, main waves[, 0 = no RNG producer waves]" (default: chosen by ec_create)
        g = [int(x) for x in a.geometry.split(",")]
        os.environ["MARL_AMD_EC_EPW"], os.environ["MARL_AMD_EC_MW"] = str(g[0]), str(g[1])
        if len(g) > 2:
            os.environ["MARL_AMD_EC_PRODUCERS"] = str(g[2])

    def trace(msg):  # stage markers on stderr (MARL_AMD_BENCH_TRACE=1): where a multi-rank run is, per rank
        if os.environ.get("MARL_AMD_BENCH_TRACE"):
            print("[bench rank %d %.1fs] %s" % (rank, time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    t_start = time.perf_counter()
    if os.environ.get("MARL_AMD_BENCH_TRACE", "").isdigit() and int(os.environ["MARL_AMD_BENCH_TRACE"]) > 1:
        import faulthandler  # where every thread is if the run is still going after that many seconds
        faulthandler.dump_traceback_later(int(os.environ["MARL_AMD_BENCH_TRACE"]), exit=True)
    ns = make_args(["--task_size_max=6000", "--seed=100", "--edge_node_num=%d" % a.nodes])
    N = a.nodes
    if a.envs % world:
        raise SystemExit("--envs %d is not a multiple of --gpus %d" % (a.envs, world))
    B = a.envs // world  # this rank's shard: strong scaling of the job BASELINE.json names

    def fence():
        th.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            th.cuda.synchronize(dev)

    def timed(envs_per_rank):
        """W untimed + K timed episodes of envs_per_rank envs on every rank; max over ranks"""
        env_ = BatchedEdgeComputingEnv(ns, n_envs=envs_per_rank, env_id_base=rank * envs_per_rank, device=dev)
        roll_ = Rollout(env_, a.launch)
        roll_.run(a.warmup)
        fence()
        t0 = time.perf_counter()
        roll_.run(a.steps)
        fence()
        dt_ = time.perf_counter() - t0
        if world > 1:
            t = th.tensor([dt_], dtype=th.float64, device=dev)
            dist_utils.all_reduce_(t, op=dist.ReduceOp.MAX)
            dt_ = float(t.item())
        st = env_.status()
        if st & ~32:
            raise SystemExit("device status 0x%x after the timed region" % st)
        return env_, roll_, dt_

    trace("process group up")
    env, roll, dt = timed(B)
    trace("timed region done: %.3f s" % dt)

    out = None
    if rank == 0:
        T_ep = env.episode_limit
        value = world * B * N * T_ep * a.steps / dt
        out = {
            "metric": "env agent-steps/sec", "value": value, "unit": "agent-steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64/u32", "data": "synthetic",
            "config": {"workload": "edge_computing env rollout, %d envs in all (%d per GPU on %d GPU%s) x %d agents, "
                                   "task_size_max=6000, uniform-random policy over the avail mask; one bench step = "
                                   "one whole episode for all envs (reset + first observation/actions + %d env steps); "
                                   "value = envs x agents x %d x steps / time (BASELINE.json configs[%d])"
                                   % (B * world, B, world, "" if world == 1 else "s", N, T_ep, T_ep,
                                      1 if world == 1 else 2),
                       "envs_total": B * world, "envs_per_gpu": B, "agents": N,
                       "env_steps_per_bench_step": T_ep, "us_per_env_step": 1e6 * dt / (a.steps * T_ep),
                       "launch": {"rollout": "ec_rollout_random_policy, one launch per episode",
                                  "graph": "ec_step_random_policy per step, hipGraph per episode",
                                  "eager": "ec_step_random_policy per step, eager"}[a.launch],
                       "geometry": env.info_geometry()},
        }
    if rank == 0:
        # the dominant kernel on rank 0's shard (no collectives inside; the other ranks wait at the next barrier)
        avg_ms, med_ms, ravg_ms, rmed_ms, lbar, draws = measure_roofline(env, roll, a.roofline_steps)
        bytes_step = algorithmic_bytes_per_env_step(N, lbar, draws) * B
        T = env.episode_limit
        std = B == 4096 and N == 10 and world == 1
        one = {"bound": "hbm", "achieved": bytes_step / (avg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "kernel": "k_step<.., one step>", "env_steps_per_launch": 1,
               "kernel_avg_us": avg_ms * 1e3, "kernel_median_us": med_ms * 1e3,
               "algorithmic_bytes_per_launch": bytes_step}
        one["traffic"], one["traffic_source"] = recorded_traffic() if std else (None, None)
        one["note"] = ("one launch per env step from a Python loop: the host issues slower than the kernel runs, so an event "
                       "pair around a launch also spans its launch latency; the median is the figure free of it (rocprofv3 "
                       "kernel time of the same launches: profiles/r02/bench_kernel_stats.csv, k_step<.., false>)")
        many = {"bound": "hbm", "achieved": bytes_step * T / (ravg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "kernel": "k_step<.., looping>", "env_steps_per_launch": T,
                "kernel_avg_us": ravg_ms * 1e3, "kernel_median_us": rmed_ms * 1e3,
                "algorithmic_bytes_per_launch": bytes_step * T}
        many["traffic"], many["traffic_source"] = recorded_traffic("pmc_traffic_k_step_rollout.json") if std else (None, None)
        for r in (one, many):
            r["frac"] = r["achieved"] / HBM_PEAK_GBS
            r["live_tasks_per_env"], r["rng_doubles_per_env_step"] = lbar, draws
        # the dominant kernel of the timed region above; the other launch mode beside it
        out["roofline"] = many if a.launch == "rollout" else one
        other = one if a.launch == "rollout" else many
        other["agent_steps_per_s_kernel_only"] = B * N * other["env_steps_per_launch"] / (other["kernel_avg_us"] * 1e-6)
        out["per_step_launch" if a.launch == "rollout" else "rollout_launch"] = other
        trace("roofline done")
        if a.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(ns, env.tables, B, a.cpu_seconds)
            # the reference itself cannot travel to the GPU box: its own figures, measured where it is importable
            out["cpu_baseline"]["reference_python"] = {
                "rollout_agent_steps_per_s": 32000.0, "env_only_agent_steps_per_s_1core": 24000.0, "cores": 8,
                "what": "the reference's ParallelRunner + PPO policy, n_threads=8 (3 iterations: 0.75 s rollout + "
                        "0.32 s update); single env, random-valid policy on one core for the second figure",
                "where": "measured in the build container by importing /root/reference (SURVEY.md section 6), "
                         "not on this box: a recorded constant, not a measurement of this run"}
    env.close()
    del roll
    if world > 1 and a.envs * world <= 65536:
        # weak scaling beside it: --envs environments on EVERY rank (what round 1 reported as the N > 1 value)
        trace("weak-scaling run")
        env_w, roll_w, dt_w = timed(a.envs)
        trace("weak-scaling run done: %.3f s" % dt_w)
        env_w.close()
        del roll_w
        if rank == 0:
            out["weak_scaling"] = {"envs_per_gpu": a.envs, "envs_total": a.envs * world,
                                   "value": world * a.envs * N * ns.episode_limit * a.steps / dt_w,
                                   "unit": "agent-steps/s",
                                   "ms_per_step": 1e3 * dt_w / a.steps}
    if a.ppo_iters > 0:
        trace("ppo iterations")
        ppo = ppo_iterations(ns, B, N, a.ppo_iters, rank, world, trace)
        trace("ppo iterations done")
        if rank == 0:
            out["ppo_iteration"] = ppo
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if a.rehearse:
            out["rehearsal"] = "all ranks on one GPU over gloo: checks the multi-rank code path, measures nothing"
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
