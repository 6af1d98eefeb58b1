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


def test_every_kernel_launching_agent_method_pins_its_device():
    """`AgentSAC.explore_action` had lost its `@_hip.on_device` (and `_explore_vec_env` carried it twice): with
    `gpu_id != current device` the binding then refuses the agent's tensors.  Every public method that enqueues kernels is wrapped
    exactly once (the wrapper makes the agent's device current for the call)."""
    from elegantrl_amd.agents import AgentModSAC, AgentPPO, AgentSAC
    from elegantrl_amd.agents.AgentPPO import AgentDiscretePPO

    def depth(f):
        n = 0
        while hasattr(f, "__wrapped__"):
            f, n = f.__wrapped__, n + 1
        return n

    for cls, names in ((AgentPPO, ("explore_action", "_explore_vec_env", "_explore_one_env", "get_values", "get_advantages", "update_net")),
                       (AgentDiscretePPO, ("explore_action", "_explore_vec_env", "_explore_one_env")),
                       (AgentSAC, ("explore_action", "_explore_vec_env", "update_objectives", "update_net")),
                       (AgentModSAC, ("explore_action", "_explore_vec_env", "update_objectives", "update_net"))):
        for n in names:
            assert depth(getattr(cls, n)) == 1, f"{cls.__name__}.{n}: wrapped {depth(getattr(cls, n))} times by _hip.on_device"


@pytest.mark.gpu
def test_config0_helloworld_shaped_pendulum_run_with_four_envs(tmp_path):
    """BASELINE configs[0] (helloworld/helloworld_PPO_single_file.py on Pendulum-v1, num_envs = 4: plumbing).  A script written the way
    the helloworld's `train_ppo_for_pendulum` + `train_agent` are (:490-517, :535-553) -- Config(agent_class, env_class, env_args), its
    net_dims / gamma / repeat_times, the explicit explore_env -> buffer[:] -> update_net loop, then `train_agent(args)` -- against the
    REFERENCE's import paths (`elegantrl.*`), on the GPU-resident Pendulum with num_envs = 4 (gymnasium is not in the image; the env
    class follows CustomGymEnv.py:42-44's action / reward scaling)."""
    out = _run(f"""
import os
import numpy as np
import torch as th
from elegantrl.train.config import Config, build_env
from elegantrl.train.run import train_agent
from elegantrl.train.evaluator import Evaluator
from elegantrl.agents import AgentPPO
from elegantrl.envs import PendulumVecEnv

env_args = {{'env_name': 'Pendulum', 'num_envs': 4, 'max_step': 200, 'state_dim': 3, 'action_dim': 1, 'if_discrete': False}}
args = Config(AgentPPO, PendulumVecEnv, env_args)
args.break_step = int(3 * 512)
args.net_dims = [64, 32]
args.gamma = 0.97
args.repeat_times = 16
args.horizon_len = 512
args.gpu_id = 0
args.cwd = {str(tmp_path / 'hello')!r}
args.eval_times, args.eval_per_step = 4, 512
assert args.num_envs == 4 and not args.if_off_policy and args.batch_size == 128

# the helloworld's own loop (:500-517)
args.init_before_training()
th.set_grad_enabled(False)
env = build_env(args.env_class, args.env_args, args.gpu_id)
agent = args.agent_class(args.net_dims, args.state_dim, args.action_dim, gpu_id=args.gpu_id, args=args)
agent.last_state, info_dict = env.reset()
assert agent.last_state.shape == (4, 3)
buffer = []
for it in range(3):
    buffer_items = agent.explore_env(env, args.horizon_len)
    buffer[:] = buffer_items
    assert [tuple(x.shape) for x in buffer] == [(512, 4, 3), (512, 4, 1), (512, 4), (512, 4), (512, 4), (512, 4)]
    assert buffer[4].dtype == th.bool and buffer[5].dtype == th.bool
    th.set_grad_enabled(True)
    logging_tuple = agent.update_net(buffer)
    th.set_grad_enabled(False)
    assert len(logging_tuple) == 3 and all(np.isfinite(x) for x in logging_tuple), logging_tuple
print('loop ok', logging_tuple)

# ... and through train_agent (run.py:39-138)
train_agent(args, if_single_process=True)
files = os.listdir(args.cwd)
assert 'act.pth' in files and 'recorder.npy' in files, files
rec = np.load(os.path.join(args.cwd, 'recorder.npy'))
assert np.isfinite(rec).all() and rec.shape[0] >= 2
print('ok')
""")
    assert out.strip().endswith("ok")
