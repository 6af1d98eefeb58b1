"""`train_agent` and friends: the entry points of elegantrl/train/run.py kept callable.

Process topology is deliberately different from the reference's (Learner / Worker / Evaluator processes
talking through pickled Pipes, run.py:141-475): here every GPU runs ONE in-process actor-learner (rollout,
GAE and update never leave the device), and multi-GPU is data parallel -- env shards one-per-GPU, a flat
gradient all-reduce per minibatch over RCCL (elegantrl_amd/parallel.py).  The training loop itself follows
run.py:39-138 (including the `{cwd}/stop` file convention and what is printed / saved).
"""
from __future__ import annotations

import os
import time

import torch as th

from .. import parallel
from .config import Config, build_env
from .evaluator import Evaluator
from .replay_buffer import ReplayBuffer


def train_agent(args: Config, if_single_process: bool = False):
    if len(getattr(args, "learner_gpu_ids", ())) > 1:
        print(f"| train_agent_multiprocessing_multi_gpu() with GPU_ID {args.learner_gpu_ids}", flush=True)
        train_agent_multiprocessing_multi_gpu(args)
    else:
        print(f"| train_agent_single_process() with GPU_ID {args.gpu_id}", flush=True)
        train_agent_single_process(args)


def train_agent_multiprocessing(args: Config):
    """elegantrl/train/run.py:141-190 spawns one Learner, `num_workers` Worker and one Evaluator process per GPU and moves
    rollouts through pipes; the Learner trains on the `num_workers` x `num_envs` env columns it gathers per horizon
    (`num_seqs = num_envs * num_workers * num_learners`, run.py:245).  Here the env shard lives on the learner's device and the
    rollout is one kernel launch, so one in-process actor-learner per GPU replaces that process tree -- and `num_workers` keeps its
    MEANING: the shard is widened to `num_workers * num_envs` envs (same data per iteration as the reference's topology), for
    vectorised env classes that take `num_envs` from `env_args`.  A single non-vectorised env cannot be widened: that is said,
    not ignored.  Several `learner_gpu_ids` -> one data-parallel rank per GPU, as `train_agent`."""
    n_workers = max(1, int(getattr(args, "num_workers", 1)))
    if n_workers > 1 and int(args.num_envs) > 1 and isinstance(getattr(args, "env_args", None), dict) and "num_envs" in args.env_args:
        wide = int(args.num_envs) * n_workers
        print(f"| train_agent_multiprocessing(): {n_workers} workers x {args.num_envs} envs -> one in-process actor-learner per GPU "
              f"with a shard of {wide} envs (no Worker processes, no pipes)", flush=True)
        args.env_args = dict(args.env_args, num_envs=wide)
        args.num_envs = wide
        args.num_workers = 1
    elif n_workers > 1:
        print(f"| train_agent_multiprocessing(): num_workers = {n_workers} asks for {n_workers} copies of a non-vectorised env; this "
              f"package runs ONE in-process actor-learner per GPU -- use a vectorised env (env_args['num_envs']) to widen the shard",
              flush=True)
    else:
        print("| train_agent_multiprocessing(): one in-process actor-learner per GPU (no Learner / Worker / Evaluator processes)", flush=True)
    train_agent(args)


def train_agent_single_process(args: Config):
    args.init_before_training()
    th.set_grad_enabled(False)
    rank, world = getattr(args, "rank", 0), getattr(args, "world_size", 1)

    env = build_env(args.env_class, args.env_args, args.gpu_id)
    agent = args.agent_class(args.net_dims, args.state_dim, args.action_dim, gpu_id=args.gpu_id, args=args)
    if args.continue_train:
        agent.save_or_load_agent(args.cwd, if_save=False)
    if world > 1 and hasattr(agent, "_flat"):        # identical initial weights on every rank
        parallel.broadcast_(agent._flat)

    state, _ = env.reset()
    if args.num_envs == 1:
        assert state.shape == (args.state_dim,)
        state = th.tensor(state, dtype=th.float32, device=agent.device).unsqueeze(0)
    else:
        state = state.to(agent.device)
    assert state.shape == (args.num_envs, args.state_dim)
    agent.last_state = state.detach()

    if args.if_off_policy:
        buffer = ReplayBuffer(gpu_id=args.gpu_id, num_seqs=args.num_envs, max_size=args.buffer_size, state_dim=args.state_dim,
                              action_dim=1 if args.if_discrete else args.action_dim, if_use_per=args.if_use_per,
                              if_discrete=args.if_discrete, args=args)
    else:
        buffer = []

    evaluator = None
    if rank == 0:
        eval_env_class = args.eval_env_class if args.eval_env_class else args.env_class
        eval_env_args = args.eval_env_args if args.eval_env_args else args.env_args
        evaluator = Evaluator(cwd=args.cwd, env=build_env(eval_env_class, eval_env_args, args.gpu_id), args=args)

    cwd, break_step, horizon_len = args.cwd, args.break_step, args.horizon_len
    if_off_policy, if_save_buffer = args.if_off_policy, args.if_save_buffer
    total_step, start = 0, time.time()
    if_train = True
    n_iter = 0
    lazy_ok = bool(getattr(args, "lazy_logs", True)) and bool(getattr(agent, "supports_lazy_logs", False)) and not if_off_policy
    pending = None

    def report(exp_r, logs, steps):
        logging_tuple = logs.result() if hasattr(logs, "result") else logs
        logging_tuple = (*logging_tuple, agent.explore_rate, "")
        if evaluator is not None:
            evaluator.evaluate_and_save(actor=agent.act, steps=steps, exp_r=float(exp_r), logging_tuple=logging_tuple)

    while if_train:
        n_iter += 1
        if n_iter == 3 and getattr(args, "gc_freeze", True):
            # the interpreter's full collections walk every object torch created at import (~40-65 ms each with the GPU idle, measured:
            # one PPO iteration is ~2.4 ms); the long-lived ones go to the permanent generation once the loop has warmed up
            import gc
            gc.collect()
            gc.freeze()
        buffer_items = agent.explore_env(env, horizon_len)
        if if_off_policy:
            buffer.update(buffer_items)
        else:
            buffer[:] = buffer_items
        exp_r = buffer_items[2].mean()               # (for on-policy this is mean(logprobs), as in run.py:122); read below

        # Agents whose update_net can hand its logged objectives over later (AgentPPO's fused paths: lazy=True -> PendingLogs) are read
        # ONE ROLLOUT LATE: iteration k's numbers reach the evaluator after iteration k + 1's rollout has been enqueued, so the GPU runs
        # that rollout while the interpreter blocks on the (already finished) logs instead of idling behind every update_net
        # (args.lazy_logs = False: read them at once, as the reference does)
        if pending is not None:
            report(*pending)
            pending = None
        th.set_grad_enabled(True)
        if lazy_ok:
            logs = agent.update_net(buffer, lazy=True)
        else:
            logs = agent.update_net(buffer)
        th.set_grad_enabled(False)

        total_step += horizon_len
        if lazy_ok and hasattr(logs, "result"):
            pending = (exp_r, logs, horizon_len)
        else:
            report(exp_r, logs, horizon_len)
        stop = (total_step > break_step) or os.path.exists(f"{cwd}/stop")
        if world > 1:
            stop = parallel.all_reduce_max_float(float(stop), device=agent.device) > 0
        if_train = not stop

    if pending is not None:
        report(*pending)
    if rank == 0:
        env_steps = total_step * args.num_envs * world
        print(f"| UsedTime: {time.time() - start:>7.0f} | SavedDir: {cwd} | env-steps: {env_steps:.3e}", flush=True)
        evaluator.save_training_curve_jpg()
        agent.save_or_load_agent(cwd, if_save=True)
        if if_save_buffer and hasattr(buffer, "save_or_load_history"):
            buffer.save_or_load_history(cwd, if_save=True)
    env.close() if hasattr(env, "close") else None


def _dp_worker(local_rank: int, args: Config, gpu_ids, port: int):
    world = len(gpu_ids)
    os.environ.update(RANK=str(local_rank), WORLD_SIZE=str(world), LOCAL_RANK=str(gpu_ids[local_rank]),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    parallel.init_from_env()
    args.gpu_id = int(gpu_ids[local_rank])
    args.rank, args.world_size = local_rank, world
    args.env_args = dict(args.env_args, seed=int(args.env_args.get("seed", 0)) + 7919 * local_rank)   # rank-distinct env streams
    if local_rank != 0:
        args.if_remove = False
    train_agent_single_process(args)
    parallel.shutdown()


def train_agent_multiprocessing_multi_gpu(args: Config):
    """one process per GPU in `args.learner_gpu_ids`; gradients averaged with RCCL (see parallel.py)."""
    import torch.multiprocessing as mp
    gpu_ids = tuple(args.learner_gpu_ids)
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_dp_worker, args=(args, gpu_ids, port), nprocs=len(gpu_ids), join=True)
