"""`Config` and `build_env`: the argument surface of `elegantrl.train.config` kept attribute-for-attribute.

Reference: elegantrl/train/config.py:10-136.  Only the attribute names, defaults and the on/off-policy
decision rule are shared with the reference (they are the API); the subprocess `VecEnv` fan-out
(config.py:212-311, capped at 64 CPU envs) is out of scope -- environments here are GPU resident.
"""
from __future__ import annotations

import inspect
import os
import shutil
from typing import Any, Dict, Optional

import numpy as np
import torch as th

_ON_POLICY_TAGS = ("SARSA", "VPG", "A2C", "A3C", "TRPO", "PPO", "MPO")  # config.py:110

_COMMON_DEFAULTS: Dict[str, Any] = dict(
    gamma=0.99, reward_scale=2 ** 0,
    net_dims=[128, 128], learning_rate=6e-5, clip_grad_norm=3.0, state_value_tau=0, soft_update_tau=5e-3,
    continue_train=False,
    gpu_id=0, num_workers=2, num_threads=8, random_seed=None, learner_gpu_ids=(),
    cwd=None, if_remove=True, break_step=np.inf, break_score=np.inf, if_keep_save=True, if_over_write=False,
    if_save_buffer=False, save_gap=8, eval_times=3, eval_per_step=int(2e4), eval_env_class=None, eval_env_args=None,
    eval_record_step=0,
)
_OFF_POLICY_DEFAULTS: Dict[str, Any] = dict(
    batch_size=64, horizon_len=512, buffer_size=int(1e6), repeat_times=1.0, if_use_per=False, lambda_fit_cum_r=0.0,
    buffer_init_size=64 * 8,
)
_ON_POLICY_DEFAULTS: Dict[str, Any] = dict(
    batch_size=128, horizon_len=2048, buffer_size=None, repeat_times=8.0, if_use_vtrace=True, buffer_init_size=None,
)
_ENV_KEYS = ("env_name", "num_envs", "max_step", "state_dim", "action_dim", "if_discrete")


class Config:
    """Plain attribute bag (same names and defaults as the reference's `Config`)."""

    def __init__(self, agent_class=None, env_class=None, env_args: Optional[dict] = None):
        self.agent_class = agent_class
        self.if_off_policy = self.get_if_off_policy()
        self.env_class = env_class
        self.env_args = env_args
        ea = dict(env_name=None, num_envs=1, max_step=12345, state_dim=None, action_dim=None, if_discrete=None)
        if env_args is not None:
            env_args.setdefault("num_envs", 1)
            env_args.setdefault("max_step", 12345)
            ea.update({k: env_args[k] for k in _ENV_KEYS})
        for k in _ENV_KEYS:
            setattr(self, k, ea[k])
        for k, v in _COMMON_DEFAULTS.items():
            setattr(self, k, list(v) if isinstance(v, list) else v)
        for k, v in (_OFF_POLICY_DEFAULTS if self.if_off_policy else _ON_POLICY_DEFAULTS).items():
            setattr(self, k, v)

    def get_if_off_policy(self) -> bool:
        name = self.agent_class.__name__ if self.agent_class else ""
        return not any(tag in name for tag in _ON_POLICY_TAGS)

    def init_before_training(self):
        """seeds, thread count, working directory (config.py:85-106)."""
        if self.random_seed is None:
            self.random_seed = max(0, self.gpu_id)
        np.random.seed(self.random_seed)
        th.manual_seed(self.random_seed)
        th.set_num_threads(self.num_threads)
        th.set_default_dtype(th.float32)
        if self.cwd is None:
            self.cwd = f"./{self.env_name}_{self.agent_class.__name__[5:]}_{self.random_seed}"
        if self.if_remove is None:
            self.if_remove = bool(input(f"| Arguments PRESS 'y' to REMOVE: {self.cwd}? ") == "y")
        if self.if_remove:
            shutil.rmtree(self.cwd, ignore_errors=True)
            print(f"| Arguments Remove cwd: {self.cwd}", flush=True)
        else:
            print(f"| Arguments Keep cwd: {self.cwd}", flush=True)
        os.makedirs(self.cwd, exist_ok=True)

    def print_config(self):
        from pprint import pprint
        pprint(vars(self))


def kwargs_filter(function, kwargs: dict) -> dict:
    """keep only the keyword arguments `function` accepts (config.py:139-146)."""
    accepted = set(inspect.signature(function).parameters)
    return {k: v for k, v in kwargs.items() if k in accepted}


def build_env(env_class=None, env_args: Optional[dict] = None, gpu_id: int = -1):
    """env = env_class(**accepted env_args); the six protocol attributes are (re)set on the instance
    (config.py:118-136).  `if_build_vec_env` (CPU subprocess fan-out) is not supported here."""
    env_args = dict(env_args)
    env_args["gpu_id"] = gpu_id
    if env_args.get("if_build_vec_env"):
        raise NotImplementedError("subprocess VecEnv (CPU gym fan-out, <= 64 envs) is out of scope; use a GPU-resident "
                                  "vectorised env such as elegantrl_amd.envs.SynVecEnv / PendulumVecEnv")
    env = env_class(**kwargs_filter(env_class.__init__, env_args.copy()))
    env_args.setdefault("num_envs", 1)
    env_args.setdefault("max_step", 12345)
    for k in _ENV_KEYS:
        setattr(env, k, env_args[k])
    return env


def get_gym_env_args(env, if_print: bool = True) -> dict:
    """env_args of an already constructed env exposing the protocol attributes."""
    env_args = {k: getattr(env, k) for k in _ENV_KEYS if hasattr(env, k)}
    env_args.setdefault("num_envs", 1)
    if if_print:
        print("env_args =", env_args)
    return env_args
