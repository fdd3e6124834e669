"""name -> agent class / flag adder, as the reference's agent_register.py:31-51 (on-policy learners of the
batched path)."""
from marl_amd.rl.policy_gradient_rl.dgail.dgail_agent import DGAILAgent
from marl_amd.rl.policy_gradient_rl.dgail.dgail_config import add_dgail_args
from marl_amd.rl.policy_gradient_rl.gail.gail_agent import GAILAgent
from marl_amd.rl.policy_gradient_rl.gail.gail_config import add_gail_args
from marl_amd.rl.policy_gradient_rl.ppo.ppo_agent import PPOAgent
from marl_amd.rl.policy_gradient_rl.ppo.ppo_config import add_ppo_args

agent_register = {"ppo": PPOAgent, "gail": GAILAgent, "dgail": DGAILAgent}
agent_config_register = {"ppo": add_ppo_args, "gail": add_gail_args, "dgail": add_dgail_args}
