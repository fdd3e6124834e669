"""name -> agent class / flag adder, as the reference's agent_register.py:31-51 (on-policy learners of the
batched path)."""
from marl_amd.rl.policy_gradient_rl.ppo.ppo_agent import PPOAgent
from marl_amd.rl.policy_gradient_rl.ppo.ppo_config import add_ppo_args

agent_register = {"ppo": PPOAgent}
agent_config_register = {"ppo": add_ppo_args}
