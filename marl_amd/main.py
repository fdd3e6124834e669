"""Entry point with the reference's command line (main.py:17-82):

    python -m marl_amd.main --name=ppo --task_size_max=6000 --seed=100 --runner=parallel --n_threads=4096
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 -m marl_amd.main --name=ppo ...

`--name=<algo>` must come first, as in the reference (main.py:19).  With torch.distributed (one process per GPU,
backend nccl = RCCL) every rank owns `n_threads` environments, env ids [rank * n_threads, (rank+1) * n_threads)."""
import argparse
import os
import sys
import time

import numpy as np
import torch as th


def build_args(argv):
    from marl_amd.agent_register import agent_config_register
    from marl_amd.default_config import add_default_args
    from marl_amd.envs.env_register import env_config_register
    env_name = "edge_computing"
    if not argv or not argv[0].startswith("--name="):
        raise SystemExit("usage: python -m marl_amd.main --name=<%s> [flags]" % "|".join(sorted(agent_config_register)))
    algo = argv[0].split("=")[1]
    parser = argparse.ArgumentParser(conflict_handler="resolve")
    parser = add_default_args(parser)
    parser = env_config_register[env_name](parser)
    parser = agent_config_register[algo](parser)
    return parser.parse_args(argv)


def main(argv=None):
    from marl_amd.agent_register import agent_register
    from marl_amd.runners.runner_register import runner_register
    from marl_amd.utils import run_utils
    args = build_args(list(sys.argv[1:] if argv is None else argv))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if th.cuda.is_available():
        th.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl" if th.cuda.is_available() else "gloo",
                                **({"device_id": th.device("cuda:%d" % local)} if th.cuda.is_available() else {}))
        args.ec_env_id_base = getattr(args, "ec_env_id_base", 0) + rank * args.n_threads
    run_utils.logger.info(args)
    np.random.seed(args.seed)
    th.manual_seed(args.seed + rank)

    # env_info as main.py:37-43, without constructing a throw-away env: the batched env's geometry is static
    args.n_agents = args.edge_node_num
    args.n_actions = args.edge_node_num + 1
    args.obs_shape = 9
    args.state_shape = 9 * args.edge_node_num
    runner_name = args.runner if args.runner in runner_register else "parallel"
    agent = agent_register[args.name](args)
    runner = runner_register[runner_name](args, agent)
    if args.load_models:
        run_utils.load_models(args, agent)
    out = None
    if args.test_models:
        t0 = time.time()
        out = run_utils.only_test_models(args, runner)
        run_utils.logger.info("The average running time is: %s", (time.time() - t0) / args.test_nepisode)
        runner.close_env()
    elif args.train_models:
        if getattr(args, "expert", False) and args.name == "gail":
            run_utils.train_expert_models(args, agent, runner)
        else:
            out = run_utils.train_models(args, agent, runner)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()
