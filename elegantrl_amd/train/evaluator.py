"""Thin evaluator: same constructor / `evaluate_and_save` contract, console table, checkpoint file names and
`recorder.npy` as elegantrl/train/evaluator.py:12-155, evaluated fully on the device for vectorised envs.
Off the hot path (runs every `eval_per_step` training steps); "next" row f3 of SURVEY.md section 8.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch as th

from .config import Config

TEN = th.Tensor


class Evaluator:
    def __init__(self, cwd: str, env, args: Config, if_tensorboard: bool = False):
        self.cwd, self.env = cwd, env
        self.agent_id = args.gpu_id
        self.total_step = 0
        self.start_time = time.time()
        self.eval_times = args.eval_times
        self.eval_per_step = args.eval_per_step
        self.eval_step_counter = -self.eval_per_step
        self.save_gap, self.save_counter = args.save_gap, 0
        self.if_keep_save, self.if_over_write = args.if_keep_save, args.if_over_write
        self.recorder_path = f"{cwd}/recorder.npy"
        self.recorder = []
        self.recorder_step = args.eval_record_step
        self.max_r = -np.inf
        self.tensorboard = None
        print(f"{'#' * 80}\n{'ID':<3}{'Step':>8}{'Time':>8} |{'avgR':>8}{'stdR':>7}{'avgS':>7}{'stdS':>6} |"
              f"{'expR':>8}{'objC':>7}{'objA':>7}{'etc.':>7}", flush=True)

    def evaluate_and_save(self, actor, steps: int, exp_r: float, logging_tuple: tuple):
        self.total_step += steps
        if self.total_step < self.recorder_step or self.total_step < self.eval_step_counter + self.eval_per_step:
            return
        self.eval_step_counter = self.total_step
        rs = self.get_cumulative_rewards_and_step(actor)
        returns, ep_steps = rs[:, 0], rs[:, 1]
        avg_r, avg_s = returns.mean().item(), ep_steps.mean().item()
        std_r = returns.std().item() if returns.numel() > 1 else 0.0
        std_s = ep_steps.std().item() if ep_steps.numel() > 1 else 0.0
        train_time = int(time.time() - self.start_time)
        value_tuple = [v for v in logging_tuple if isinstance(v, (int, float))]
        logging_str = logging_tuple[-1] if len(logging_tuple) and isinstance(logging_tuple[-1], str) else ""
        self.recorder.append((self.total_step, avg_r, std_r, exp_r, *value_tuple))
        prev_max_r = self.max_r
        self.max_r = max(self.max_r, avg_r)
        print(f"{self.agent_id:<3}{self.total_step:8.2e}{train_time:8.0f} |{avg_r:8.2f}{std_r:7.1f}{avg_s:7.0f}{std_s:6.0f} |"
              f"{exp_r:8.2f}{''.join(f'{n:7.2f}' for n in value_tuple)} {logging_str}", flush=True)
        if_best = avg_r > prev_max_r
        if if_best:
            self.save_training_curve_jpg()
        if not self.if_keep_save:
            return
        self.save_counter += 1
        actor_path = None
        if if_best:
            actor_path = f"{self.cwd}/actor.pt" if self.if_over_write else \
                f"{self.cwd}/actor__{self.total_step:012}_{self.max_r:09.3f}.pt"
        elif self.save_counter == self.save_gap:
            self.save_counter = 0
            actor_path = f"{self.cwd}/actor.pt" if self.if_over_write else f"{self.cwd}/actor__{self.total_step:012}.pt"
        if actor_path:
            th.save(actor, actor_path)
            self.save_training_curve_jpg()

    def get_cumulative_rewards_and_step(self, actor) -> TEN:
        if getattr(self.env, "num_envs", 1) == 1:                                  # evaluator.py:145-148
            out = [get_rewards_and_steps(self.env, actor) for _ in range(self.eval_times)]
            return th.tensor(out, dtype=th.float32)
        rounds = max(1, self.eval_times // self.env.num_envs)                       # evaluator.py:150-155
        return th.cat([get_cumulative_rewards_and_step_from_vec_env(self.env, actor) for _ in range(rounds)], dim=0)

    def save_or_load_recoder(self, if_save: bool):
        if if_save:
            np.save(self.recorder_path, np.array(self.recorder))
        elif os.path.exists(self.recorder_path):
            self.recorder = [tuple(i) for i in np.load(self.recorder_path)]
            self.total_step = self.recorder[-1][0]

    def save_training_curve_jpg(self):
        if not self.recorder:
            return
        recorder = np.array(self.recorder)
        np.save(self.recorder_path, recorder)
        try:  # the plot is optional; the recorder file is the durable artefact
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            fig, ax = plt.subplots(2, 1, figsize=(8, 8))
            ax[0].plot(recorder[:, 0], recorder[:, 1], label="avgR")
            ax[0].fill_between(recorder[:, 0], recorder[:, 1] - recorder[:, 2], recorder[:, 1] + recorder[:, 2], alpha=0.3)
            ax[0].legend()
            if recorder.shape[1] > 5:
                ax[1].plot(recorder[:, 0], recorder[:, 4], label="objC")
                ax[1].plot(recorder[:, 0], recorder[:, 5], label="objA")
                ax[1].legend()
            fig.savefig(f"{self.cwd}/LearningCurve.jpg")
            plt.close(fig)
        except Exception:
            pass


def get_rewards_and_steps(env, actor, if_render: bool = False) -> tuple:
    """one episode of a single (numpy) env with the deterministic policy `actor(state)` (evaluator.py:161-198)."""
    device = next(actor.parameters()).device
    max_step = env.max_step
    state, _ = env.reset()
    episode_steps, cumulative = 0, 0.0
    for episode_steps in range(max_step):
        ten = th.as_tensor(state, dtype=th.float32, device=device).unsqueeze(0)
        action = actor(ten).detach().cpu().numpy()[0]
        state, reward, terminated, truncated, _ = env.step(action)
        cumulative += reward
        if if_render:
            env.render()
        if terminated or truncated:
            break
    cumulative = getattr(getattr(env, "unwrapped", env), "cumulative_returns", cumulative)
    return cumulative, episode_steps + 1


def get_cumulative_rewards_and_step_from_vec_env(env, actor) -> TEN:
    """(return, length) of EVERY episode the sub-envs complete within `max_step` steps of the deterministic policy, as the
    reference counts them (evaluator.py:201-238): the rollout runs on the env's device without host round trips, the reward
    and done planes come to the host once, episodes still open at the end are dropped, and an env that exposes
    `cumulative_returns` reports that (with length max_step) instead.  -> (n_episodes, 2) float32 on the CPU."""
    device = next(actor.parameters()).device
    n, max_step = env.num_envs, env.max_step
    state, _ = env.reset()
    rewards = th.empty((max_step, n), dtype=th.float32, device=device)
    dones = th.empty((max_step, n), dtype=th.bool, device=device)
    with th.no_grad():
        for t in range(max_step):
            state, reward, terminal, truncate, _ = env.step(actor(state.to(device)))
            rewards[t] = reward
            dones[t] = th.logical_or(terminal, truncate)
    if hasattr(env, "cumulative_returns"):
        ret = th.as_tensor(env.cumulative_returns, dtype=th.float32).reshape(-1).cpu()
        return th.stack((ret, th.full_like(ret, float(max_step))), dim=1)
    r, d = rewards.cpu().numpy().astype(np.float64), dones.cpu().numpy()
    cs = np.vstack((np.zeros((1, n)), np.cumsum(r, axis=0)))      # cs[t] = sum of the first t rewards (exact enough in fp64)
    out = []
    for i in range(n):
        ends = np.flatnonzero(d[:, i]) + 1
        starts = np.concatenate(([0], ends[:-1]))
        out.extend(zip(cs[ends, i] - cs[starts, i], ends - starts))
    return th.tensor(out, dtype=th.float32).reshape(-1, 2)
