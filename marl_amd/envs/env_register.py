"""name -> class registries, as envs/env_register.py:4-10 in the reference.

`edge_computing`          one environment per instance, Python scalars / numpy out (the reference's contract)
`edge_computing_batched`  args.n_threads environments per instance, device tensors out (what DeviceRunner uses)
Both run on the HIP library; there is no CPU implementation in this package.
"""
from marl_amd.envs.edge_computing.multihop.edge_computing_config import add_edge_computing_env_args
from marl_amd.envs.edge_computing.multihop.edge_computing_env import BatchedEdgeComputingEnv, EdgeComputingEnv

env_register = {
    "edge_computing": EdgeComputingEnv,
    "edge_computing_batched": BatchedEdgeComputingEnv,
}

env_config_register = {
    "edge_computing": add_edge_computing_env_args,
    "edge_computing_batched": add_edge_computing_env_args,
}
