"""`import elegantrl` -> this package.  The drop-in boundary of the path (SURVEY.md 8b / BASELINE.json north_star: "drops in
under elegantrl/agents and elegantrl/train") is the reference's class protocol, so scripts written against the reference --
`from elegantrl.agents import AgentPPO`, `from elegantrl.train.config import Config`, `from elegantrl.train.run import
train_agent`, `from elegantrl.train.replay_buffer import ReplayBuffer`, `from elegantrl import train_agent` ... -- should run
unmodified.  `install()` puts an import hook at the front of `sys.meta_path` that resolves `elegantrl` and every
`elegantrl.<x>` to the very module object of `elegantrl_amd` / `elegantrl_amd.<x>` (no second copy of any module, so
`elegantrl.agents.AgentPPO is elegantrl_amd.agents.AgentPPO`).  Names the reference has and this package does not (the DQN /
TD3 families, gym wrappers: outside SURVEY.md 8) fail with the ordinary ImportError / AttributeError.

Two ways in: `import elegantrl_amd.compat; elegantrl_amd.compat.install()` before the script's imports, or simply keep the
repository root on `sys.path`: the `elegantrl/` directory there is a three-line package that calls `install()` and hands
its name over.  A reference checkout EARLIER on `sys.path` (as oracle/make_golden.py arranges) still wins as long as the
hook is not installed in that process.
"""
from __future__ import annotations

import importlib
import importlib.abc
import importlib.machinery
import sys

_ALIAS, _REAL = "elegantrl", "elegantrl_amd"


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname != _ALIAS and not fullname.startswith(_ALIAS + "."):
            return None
        real = _REAL + fullname[len(_ALIAS):]
        try:
            mod = importlib.import_module(real)
        except ImportError:
            return None                       # not part of this package: the ordinary ModuleNotFoundError follows
        spec = importlib.machinery.ModuleSpec(fullname, self, is_package=hasattr(mod, "__path__"))
        spec._erl_real = mod
        return spec

    def create_module(self, spec):
        return spec._erl_real                 # the SAME module object: classes keep one identity under both names

    def exec_module(self, module):
        pass


_finder = None


def install() -> None:
    """idempotent; after it `import elegantrl[.x]` yields `elegantrl_amd[.x]`."""
    global _finder
    if _finder is None:
        _finder = _AliasFinder()
        sys.meta_path.insert(0, _finder)
    sys.modules[_ALIAS] = importlib.import_module(_REAL)


def uninstall() -> None:
    global _finder
    if _finder is not None:
        sys.meta_path.remove(_finder)
        _finder = None
    for name in [n for n in sys.modules if n == _ALIAS or n.startswith(_ALIAS + ".")]:
        if getattr(sys.modules[name], "__name__", "").startswith(_REAL):
            del sys.modules[name]
