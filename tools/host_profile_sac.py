"""Interpreter time of one SAC iteration (explore_env + buffer.update + 64 x [buffer.sample + the fused update step]) at config 3's
shape (64 envs, 1e6 ring, batch 256, net [256,256], 4 critics) against its wall time, and a cProfile of it.  Run on the GPU box."""
import cProfile
import io
import pstats
import sys
import time

sys.path.insert(0, "/root/repo")
import torch as th  # noqa: E402
from elegantrl_amd.agents import AgentSAC  # noqa: E402
from elegantrl_amd.envs import SynVecEnv  # noqa: E402
from elegantrl_amd.train import Config, ReplayBuffer  # noqa: E402

dev = th.device("cuda:0")
N, S, A, H, B, UPD, NET = 64, 11, 3, 64, 256, 64, [256, 256]
max_size = 1_000_000 // N
args = Config(AgentSAC, SynVecEnv, {"env_name": "SynVecEnv", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
args.net_dims, args.horizon_len, args.batch_size = NET, H, B
args.repeat_times = UPD * B / max_size
args.gpu_id, args.random_seed = 0, 0
th.manual_seed(0)
agent = AgentSAC(args.net_dims, S, A, gpu_id=0, args=args)
env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
buf = ReplayBuffer(max_size=max_size, state_dim=S, action_dim=A, gpu_id=0, num_seqs=N)
g = th.Generator(device=dev).manual_seed(1)
for _ in range(2):
    buf.update((th.randn((max_size // 2 + 7, N, S), device=dev, generator=g), th.randn((max_size // 2 + 7, N, A), device=dev, generator=g).tanh(),
                th.randn((max_size // 2 + 7, N), device=dev, generator=g), th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.99,
                th.rand((max_size // 2 + 7, N), device=dev, generator=g) < 0.995))


def step():
    buf.update(agent.explore_env(env, H))
    return agent.update_net(buf)


for _ in range(3):
    step()
th.cuda.synchronize()
n = 8
t0 = time.perf_counter()
for _ in range(n):
    step()
th.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"iteration (64 updates + rollout): {dt * 1e3:.2f} ms = {dt / 64 * 1e6:.0f} us per update")
pr = cProfile.Profile()
pr.enable()
for _ in range(4):
    step()
th.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(24)
print(s.getvalue()[:5200])
