"""Interpreter time on the launch path of one PPO iteration (the GPU idles behind update_net's host sync while explore_env gets
to its launch): median host time of explore_env, a cProfile of it, and update_net's wall time.  Run on the GPU box."""
import sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import torch as th
from elegantrl_amd.agents import AgentPPO
from elegantrl_amd.envs import SynVecEnv
from elegantrl_amd.train import Config
N, S, A, H, B = 4096, 64, 8, 32, 16384
args = Config(AgentPPO, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
args.net_dims, args.horizon_len, args.batch_size = [128, 128], H, B
args.repeat_times = 40 * B / H
agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
for _ in range(5):
    agent.update_net(list(agent.explore_env(env, H)))
th.cuda.synchronize()
te = []
for _ in range(50):
    th.cuda.synchronize()
    t0 = time.perf_counter(); items = agent.explore_env(env, H); te.append(time.perf_counter() - t0)
print(f"explore_env host time: {1e6*sorted(te)[len(te)//2]:.0f} us (median)")
pr = cProfile.Profile()
for _ in range(200):
    th.cuda.synchronize()
    pr.enable(); items = agent.explore_env(env, H); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
# update_net tail: time from GPU done to return
tu = []
for _ in range(20):
    items = agent.explore_env(env, H)
    t0 = time.perf_counter(); out = agent.update_net(list(items)); tu.append(time.perf_counter() - t0)
print(f"update_net wall (incl. GPU loop): {1e6*sorted(tu)[len(tu)//2]:.0f} us")
