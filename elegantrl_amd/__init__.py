"""elegantrl_amd -- MI355X-native (gfx950) hot path for an ElegantRL-compatible vectorised actor-learner.

Same Python surface as `elegantrl` for the path named in BASELINE.json (train_agent / Config / AgentBase /
AgentPPO / ReplayBuffer); the arithmetic runs in hand-written HIP kernels behind the C ABI in
include/erl_hip.h (elegantrl_amd/lib/liberl_hip.so).  No CPU fallback, no CUDA shims.
"""
__version__ = "0.1.0"

from .train.config import Config, get_gym_env_args  # noqa: E402
from .train.run import (train_agent, train_agent_multiprocessing,  # noqa: E402
                        train_agent_multiprocessing_multi_gpu, train_agent_single_process)
