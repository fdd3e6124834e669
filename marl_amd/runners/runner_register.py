"""name -> runner class, as the reference's runners/runner_register.py:9-17 (only the runners of the batched
on-policy path exist here; `episode` is the parallel runner with n_threads envs, which for n_threads=1 is what
the reference's EpisodeRunner does)."""
from marl_amd.runners.dgail_runner import DGAILRunner
from marl_amd.runners.gail_runner import GAILRunner
from marl_amd.runners.parallel_episode_runner import ParallelRunner

runner_register = {
    "parallel": ParallelRunner,
    "episode": ParallelRunner,
    "gail_runner": GAILRunner,
    "dgail_runner": DGAILRunner,
}
