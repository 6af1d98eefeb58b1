from .config import Config, build_env, get_gym_env_args, kwargs_filter
from .evaluator import Evaluator
from .replay_buffer import ReplayBuffer
from .run import (train_agent, train_agent_multiprocessing, train_agent_multiprocessing_multi_gpu,
                  train_agent_single_process)
