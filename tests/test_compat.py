"""The `elegantrl` name ("drops in under elegantrl/agents and elegantrl/train"): scripts written against the
reference's import paths run unmodified from a checkout of this repository.  Every case runs in its own interpreter so that
the import hook never meets a mounted reference inside the pytest process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code: str):
    env = dict(os.environ, PYTHONPATH=ROOT, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], cwd="/", env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    return out.stdout


def test_reference_import_paths_resolve_to_this_package():
    out = _run("""
import elegantrl
from elegantrl import train_agent, train_agent_multiprocessing
from elegantrl.agents import AgentPPO, AgentSAC, AgentA2C, AgentDiscretePPO
from elegantrl.agents.AgentPPO import AgentPPO as A2, ActorPPO, CriticPPO
from elegantrl.agents.AgentBase import AgentBase, build_mlp
from elegantrl.train.config import Config, build_env, get_gym_env_args
from elegantrl.train.run import train_agent as t2, train_agent_single_process
from elegantrl.train.replay_buffer import ReplayBuffer
from elegantrl.train.evaluator import Evaluator
from elegantrl.train import ReplayBuffer as R2
import elegantrl_amd, elegantrl_amd.agents.AgentPPO as real
assert elegantrl is elegantrl_amd and A2 is AgentPPO is real.AgentPPO and t2 is train_agent and R2 is ReplayBuffer
assert AgentPPO.__module__ == "elegantrl_amd.agents.AgentPPO"           # one module object, one class identity
args = Config(AgentPPO, None, {"env_name": "x", "num_envs": 4, "max_step": 10, "state_dim": 3, "action_dim": 1, "if_discrete": False})
assert not args.if_off_policy and Config(AgentSAC, None, {"env_name": "x", "num_envs": 4, "max_step": 10, "state_dim": 3,
                                                          "action_dim": 1, "if_discrete": False}).if_off_policy
try:
    from elegantrl.agents import AgentDQN                                 # outside SURVEY 8: the ordinary ImportError
    raise SystemExit("AgentDQN should not exist")
except ImportError:
    pass
try:
    import elegantrl.agents.AgentTD3
    raise SystemExit("AgentTD3 should not exist")
except ModuleNotFoundError:
    pass
print("ok")
""")
    assert out.strip().endswith("ok")


def test_install_is_explicit_and_reversible():
    out = _run("""
import sys
sys.path = [p for p in sys.path if p != %r]          # no repository root on the path: only the explicit hook
sys.path.append(%r)                                    # (appended: found after site-packages, as an installed package would be)
import elegantrl_amd.compat as compat
compat.install(); compat.install()
from elegantrl.train.config import Config
import elegantrl, elegantrl_amd
assert elegantrl is elegantrl_amd
compat.uninstall()
assert "elegantrl" not in sys.modules and "elegantrl.train.config" not in sys.modules
print("ok")
""" % (ROOT, ROOT))
    assert out.strip().endswith("ok")


@pytest.mark.gpu
def test_survey_appendix_b_snippet_with_reference_imports():
    """SURVEY.md Appendix B ("How the oracle was exercised"): one PPO iteration at the config-4 shape written against the
    REFERENCE's import paths and protocol -- Config, AgentPPO(net_dims, state_dim, action_dim, gpu_id, args), last_state,
    explore_env, update_net -- on the GPU (the snippet's gpu_id = -1 selects the reference's CPU path, which this package
    does not have)."""
    out = _run("""
import torch as th
from elegantrl.agents import AgentPPO
from elegantrl.train.config import Config
from elegantrl.envs import SynVecEnv
args = Config(AgentPPO, None, {'env_name':'syn','num_envs':4096,'max_step':100,'state_dim':64,'action_dim':8,'if_discrete':False})
args.horizon_len, args.batch_size = 32, 16384; args.repeat_times = 8*16384/32
agent = AgentPPO(args.net_dims, 64, 8, gpu_id=0, args=args)
env = SynVecEnv(4096, 64, 8, max_step=100, gpu_id=0, seed=0)
agent.last_state = env.reset()[0]
th.set_grad_enabled(False)
items = agent.explore_env(env, 32)
assert [tuple(x.shape) for x in items] == [(32, 4096, 64), (32, 4096, 8), (32, 4096), (32, 4096), (32, 4096), (32, 4096)]
th.set_grad_enabled(True)
objs = agent.update_net(list(items))
th.set_grad_enabled(False)
assert len(objs) == 3 and all(o == o for o in objs)
import elegantrl_amd._hip as h
assert h._lib is not None                                                # the HIP library did the work
print("ok", objs)
""")
    assert "ok" in out
