#!/usr/bin/env python3
"""One config-4-shaped rollout (4096 envs, S = 64, A = 8, net [128,128], H = 32, injected noise) under the library named by ERL_HIP_LIB:
the first time row's actions and values against an fp64 evaluation of the same networks on the same states, and the rollout's buffers
saved to gpurun_out/rollout_ab_<tag>.npz so that two libraries (two arithmetics of the hidden layers) can be compared element by element:
    ERL_HIP_LIB=.../liberl_hip_old.so python tools/rollout_arith_ab.py old
    python tools/rollout_arith_ab.py new
    python tools/rollout_arith_ab.py --compare old new"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
OUT = "gpurun_out"


def compare(a, b):
    x, y = np.load(f"{OUT}/rollout_ab_{a}.npz"), np.load(f"{OUT}/rollout_ab_{b}.npz")
    for k in ("actions", "logprobs", "values", "rewards", "states"):
        d = np.abs(x[k].astype(np.float64) - y[k])
        per_t = d.reshape(d.shape[0], -1).max(axis=1)
        print(f"{k:9s} {a} vs {b}: max |diff| {d.max():.3e} (scale {np.abs(x[k]).max():.3g}); rows bit-identical {int((per_t == 0).sum())}/{len(per_t)}; "
              f"per time row t=0 {per_t[0]:.2e}, t=1 {per_t[1]:.2e}, t=8 {per_t[8]:.2e}, t=31 {per_t[-1]:.2e}")
    for k in ("undones", "unmasks"):
        print(f"{k:9s} differing elements: {int((x[k] != y[k]).sum())}")


def main():
    if sys.argv[1] == "--compare":
        return compare(sys.argv[2], sys.argv[3])
    tag = sys.argv[1]
    import torch as th
    from elegantrl_amd.agents import AgentPPO
    from elegantrl_amd.envs import SynVecEnv
    from elegantrl_amd.train import Config
    N, S, A, H, net = 4096, 64, 8, 32, (128, 128)
    dev = "cuda:0"
    args = Config(AgentPPO, None, {"env_name": "syn", "num_envs": N, "max_step": 1000, "state_dim": S, "action_dim": A, "if_discrete": False})
    args.net_dims, args.random_seed = list(net), 7
    args.fused_rollout = True
    th.manual_seed(3)
    agent = AgentPPO(args.net_dims, S, A, gpu_id=0, args=args)
    g = th.Generator(device=dev).manual_seed(4)
    with th.no_grad():
        for m in (agent.act, agent.cri):
            m.state_avg[:] = 0.1 * th.randn(S, device=dev, generator=g)
            m.state_std[:] = 1.0 + 0.2 * th.rand(S, device=dev, generator=g)
        agent.act.action_std_log[:] = -0.3 + 0.1 * th.randn(A, device=dev, generator=g)
    env = SynVecEnv(N, S, A, max_step=1000, gpu_id=0, seed=5)
    agent.last_state = env.reset()[0]
    noise = th.randn((H, N, A), device=dev, generator=th.Generator(device=dev).manual_seed(11))
    items = agent._explore_vec_env(env, H, noise=noise)
    values = agent._rollout_cache["values"]
    th.cuda.synchronize()
    states, actions, logprobs, rewards, undones, unmasks = items

    def f64(module, x):                      # the torch module's layers in fp64 on the same states (normalisation included, no tanh)
        import copy
        m = copy.deepcopy(module).double()
        with th.no_grad():
            return m.net(m.state_norm(x.double()))
    with th.no_grad():
        mean64 = f64(agent.act, states[0])
        act64 = mean64 + agent.act.action_std_log.double().exp() * noise[0].double()
        val64 = f64(agent.cri, states[0]).reshape(-1)
    ea = (actions[0].double() - act64).abs().max().item()
    ev = (values[0].double().reshape(-1) - val64).abs().max().item()
    print(f"[{tag}] {os.environ.get('ERL_HIP_LIB', 'liberl_hip.so')}: first time row against fp64: actions max |err| {ea:.3e} "
          f"(scale {act64.abs().max().item():.3g}), values max |err| {ev:.3e} (scale {val64.abs().max().item():.3g})")
    os.makedirs(OUT, exist_ok=True)
    np.savez(f"{OUT}/rollout_ab_{tag}.npz", states=states.cpu().numpy(), actions=actions.cpu().numpy(), logprobs=logprobs.cpu().numpy(),
             rewards=rewards.cpu().numpy(), undones=undones.cpu().numpy(), unmasks=unmasks.cpu().numpy(), values=values.cpu().numpy())


if __name__ == "__main__":
    main()
