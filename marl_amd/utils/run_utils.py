"""Training / evaluation driver loops with the reference's names and metrics (utils/run_utils.py:10-355):
`train_models`, `train_expert_models`, `test_models`, `only_test_models`, `load_models`.

Same control flow and scalar tags (`Training/Episode_Reward`, `Training/Cumulative_Avg_Reward`,
`Training/Moving_Avg_Reward_100`, `success_rate_avg`, `success_rate_std`, `Test/Episode_Return_Avg|Std`) and the same
results/ directory layout for logs and checkpoints.  Scalars go to tensorboard when it is importable and always to a
JSON-lines file beside it; the matplotlib figures of the reference are not reproduced.  Under torch.distributed only
rank 0 writes files (metrics are already whole-job values: the runners all-reduce them).
"""
import json
import logging
import os
from datetime import datetime

import numpy as np

from marl_amd.utils.dist_utils import dist_on

logger = logging.getLogger("marl_amd")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s %(levelname)s %(message)s"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)


def _rank0():
    if not dist_on():
        return True
    import torch.distributed as dist
    return dist.get_rank() == 0


class ScalarWriter:
    """add_scalar(tag, value, step) -> results/tensorboard/<folder>/<run>/scalars.jsonl (+ tensorboard if present)"""

    def __init__(self, log_dir):
        self.enabled = _rank0()
        self._tb = None
        self._f = None
        if not self.enabled:
            return
        os.makedirs(log_dir, exist_ok=True)
        self._f = open(os.path.join(log_dir, "scalars.jsonl"), "a")
        try:
            from torch.utils.tensorboard import SummaryWriter
            self._tb = SummaryWriter(log_dir=log_dir)
        except Exception:
            self._tb = None

    def add_scalar(self, tag, value, step):
        if not self.enabled:
            return
        self._f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")
        if self._tb is not None:
            self._tb.add_scalar(tag, value, step)

    def flush(self):
        if self._f:
            self._f.flush()
        if self._tb is not None:
            self._tb.flush()

    def close(self):
        self.flush()
        if self._f:
            self._f.close()
        if self._tb is not None:
            self._tb.close()


def load_models(args, agent):
    """run_utils.py:10-12"""
    agent.load_models(os.path.join("results", "models", args.folder, args.checkpoint_path))


def _merge_hops(total, part):
    for hop, count in part.items():
        total[hop] = total.get(hop, 0) + count


def test_models(args, runner):
    """run_utils.py:295-326"""
    success, returns, hops = [], [], {}
    for _ in range(max(1, args.test_nepisode // args.n_threads)):
        info = runner.run(test_mode=True)
        success.append(info["success_rate"])
        if "episode_return" in info:
            returns.append(info["episode_return"])
        _merge_hops(hops, info.get("max_hop_dict", {}))
    res = {"success_rate_avg": np.mean(success, axis=0), "success_rate_std": np.std(success, axis=0), "max_hop_dict": hops}
    if returns:
        res.update(episode_return_avg=np.mean(returns), episode_return_std=np.std(returns),
                   episode_return_max=np.max(returns), episode_return_min=np.min(returns))
    return res


def _run_name(args, with_nodes=True):
    time = datetime.today().strftime("%Y.%m.%d-%H-%M-%S")
    parts = [args.name] + ([str(args.edge_node_num)] if with_nodes else []) + [args.log_tag, time]
    return "-".join(parts)


def _save(args, agent, file_name):
    if args.save_models and _rank0():
        path = os.path.join(os.getcwd(), "results", "models", args.folder, file_name)
        os.makedirs(path, exist_ok=True)
        agent.save_models(path)
        logger.info("models saved to %s", path)


def train_models(args, agent, runner):
    """run_utils.py:15-140"""
    file_name = _run_name(args)
    if args.offline_models_path != "":
        agent.load_offline_models(os.path.join(os.getcwd(), "results", "models", args.offline_models_path))
    writer = ScalarWriter(os.path.join(os.getcwd(), "results", "tensorboard", args.folder, file_name))
    test_steps = 0
    hops_all = {}
    rewards = []
    total = 0.0
    while runner.t_env <= args.t_max:
        info = runner.run(test_mode=False)
        if "episode_return" in info:
            r = info["episode_return"]
            rewards.append(r)
            total += r
            n = len(rewards)
            writer.add_scalar("Training/Episode_Reward", r, n)
            writer.add_scalar("Training/Cumulative_Avg_Reward", total / n, n)
            recent = rewards[-min(100, n):]
            writer.add_scalar("Training/Moving_Avg_Reward_100", sum(recent) / len(recent), n)
            if n % 50 == 0 and _rank0():
                logger.info("Episode %d: reward %.4f, cumulative average %.4f, moving average %.4f", n, r, total / n,
                            sum(recent) / len(recent))
        if runner.t_env // args.test_interval > test_steps:
            test_steps += 1
            res = test_models(args, runner)
            if _rank0():
                logger.info("t_env %d success_rate_avg %s", runner.t_env, res["success_rate_avg"])
            writer.add_scalar("success_rate_avg", res["success_rate_avg"], args.t_start + runner.t_env)
            writer.add_scalar("success_rate_std", res["success_rate_std"], args.t_start + runner.t_env)
            if "episode_return_avg" in res:
                writer.add_scalar("Test/Episode_Return_Avg", res["episode_return_avg"], args.t_start + runner.t_env)
                writer.add_scalar("Test/Episode_Return_Std", res["episode_return_std"], args.t_start + runner.t_env)
            _merge_hops(hops_all, res["max_hop_dict"])
    if rewards and _rank0():
        logger.info("Training completed - episodes %d, final_avg_reward %.4f, max %.4f, min %.4f", len(rewards),
                    total / len(rewards), max(rewards), min(rewards))
    runner.close_env()
    writer.close()
    _save(args, agent, file_name)
    return {"episodes": len(rewards), "episode_rewards": rewards, "max_hop_dict": hops_all, "run": file_name}


def train_expert_models(args, agent, runner):
    """run_utils.py:211-250 (runners whose run() takes t_expert: GAILRunner)"""
    file_name = _run_name(args, with_nodes=False)
    writer = ScalarWriter(os.path.join(os.getcwd(), "results", "tensorboard", args.folder, file_name))
    test_steps = 0
    while runner.t_env <= args.t_max:
        t_expert = runner.t_env
        runner.run(test_mode=False, t_expert=t_expert)
        if runner.t_env // args.test_interval > test_steps:
            test_steps += 1
            success = [runner.run(test_mode=True, t_expert=t_expert)["success_rate"]
                       for _ in range(max(1, args.test_nepisode // args.n_threads))]
            if _rank0():
                logger.info("t_env %d success_rate_avg %s", runner.t_env, np.mean(success))
            writer.add_scalar("success_rate_avg", np.mean(success, axis=0), args.t_start + runner.t_env)
            writer.add_scalar("success_rate_std", np.std(success, axis=0), args.t_start + runner.t_env)
    runner.close_env()
    writer.close()
    _save(args, agent, file_name)


def only_test_models(args, runner):
    """run_utils.py:329-355"""
    n = max(1, args.test_nepisode // args.n_threads)
    acc = {"success_rate": 0.0, "drop_rate": 0.0, "failure_rate": 0.0, "task_completion_time": 0.0}
    success = []
    for _ in range(n):
        info = runner.run(test_mode=True)
        for k in acc:
            acc[k] += info[k]
        success.append(info["success_rate"])
    res = {k: v / n for k, v in acc.items()}
    res["success_rate_std"] = float(np.std(success, axis=0))
    if _rank0():
        logger.info("success rate %s std %s drop rate %s failure rate %s task completion time %s", res["success_rate"],
                    res["success_rate_std"], res["drop_rate"], res["failure_rate"], res["task_completion_time"])
    return res
