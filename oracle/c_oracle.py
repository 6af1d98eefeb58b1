"""ctypes binding of oracle/gae_scan.c (oracle; test infrastructure only)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "liberl_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gae_scan.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def gae(rewards, undones, unmasks, values, next_value, gamma, lam, use_v_trace=True):
    """Returns (adv, ret, rewards_after, undones_after); inputs untouched."""
    r = np.ascontiguousarray(rewards, np.float32).copy()
    u = np.ascontiguousarray(undones).astype(np.uint8)
    m = np.ascontiguousarray(unmasks).astype(np.uint8)
    v = np.ascontiguousarray(values, np.float32)
    nv = np.ascontiguousarray(next_value, np.float32)
    H, N = r.shape
    adv = np.empty_like(r)
    ret = np.empty_like(r)
    rc = lib().erl_oracle_gae_f32(_p(r), _p(u), _p(m), _p(v), _p(nv), _p(adv), _p(ret), ctypes.c_int64(H),
                                  ctypes.c_int64(N), ctypes.c_float(gamma), ctypes.c_float(lam),
                                  ctypes.c_int(int(use_v_trace)), ctypes.c_int(1))
    assert rc == 0
    return adv, ret, r, u.astype(bool)


def gae_cols_inplace(r, u, m, v, nv, adv, ret, n0, n1, gamma, lam):
    """Timing entry (mutates r/u like the reference); arrays must be C-contiguous of the right dtype."""
    H, N = r.shape
    return lib().erl_oracle_gae_f32_cols(_p(r), _p(u), _p(m), _p(v), _p(nv), _p(adv), _p(ret), ctypes.c_int64(H),
                                         ctypes.c_int64(N), ctypes.c_int64(n0), ctypes.c_int64(n1),
                                         ctypes.c_float(gamma), ctypes.c_float(lam))
