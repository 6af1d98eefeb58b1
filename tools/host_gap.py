import sys, time
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
te, tu, tt = [], [], []
for _ in range(20):
    t0 = time.perf_counter()
    items = agent.explore_env(env, H)
    t1 = time.perf_counter()
    out = agent.update_net(list(items))
    t2 = time.perf_counter()
    te.append(t1 - t0); tu.append(t2 - t1)
print(f"host time in explore_env (no sync inside): {1e6*sum(te)/len(te):.0f} us; update_net incl. its final sync: {1e6*sum(tu)/len(tu):.0f} us; iteration {1e6*(sum(te)+sum(tu))/len(te):.0f} us")
