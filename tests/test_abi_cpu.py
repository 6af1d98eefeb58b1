"""CPU-side checks of the C-ABI boundary (no GPU, no compute calls): liberl_hip.so loads, exports every symbol that
include/erl_hip.h declares, the ctypes table in elegantrl_amd/_hip.py covers exactly those symbols, and the pure host
entry points (parameter counts, workspace sizes, error reporting) behave."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "erl_hip.h")


def header_symbols():
    txt = open(HEADER).read()
    return sorted(set(re.findall(r"ERL_API\s+[\w\s\*]+?\b(erl_\w+)\s*\(", txt)))


def test_header_declares_what_the_binding_binds():
    from elegantrl_amd import _hip
    hdr = header_symbols()
    assert len(hdr) >= 30
    assert sorted(_hip.EXPORTED_SYMBOLS) == hdr, (set(hdr) ^ set(_hip.EXPORTED_SYMBOLS))
    txt = open(HEADER).read()
    assert int(re.search(r"#define ERL_ABI_VERSION (\d+)", txt).group(1)) == _hip.ABI_VERSION


def test_library_exports_every_declared_symbol():
    from elegantrl_amd import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), f"{name} is declared in erl_hip.h but not exported by liberl_hip.so"
    assert _hip.lib().erl_abi_version() == _hip.ABI_VERSION


def test_host_only_entry_points():
    from elegantrl_amd import _hip, ops
    L = _hip.lib()
    # flat parameter counts = what nn.Linear layers hold (AgentBase.py:345-360 layout)
    assert L.erl_mlp_param_count(64, 128, 128, 8, 1) == 64 * 128 + 128 + 128 * 128 + 128 + 128 * 8 + 8 + 8
    assert L.erl_mlp_param_count(64, 128, 128, 1, 0) == 64 * 128 + 128 + 128 * 128 + 128 + 128 + 1
    assert L.erl_mlp_param_count(64, 100, 128, 8, 1) == -1           # fused kernels: widths are multiples of 32
    assert L.erl_ppo_slab_stride(64, 128, 128, 8) == (25872 + 24961 + 4 + 31) // 32 * 32       # whole 128-byte lines
    assert L.erl_ppo_num_slabs(16384) == 128 and L.erl_ppo_num_slabs(1) == 1 and L.erl_ppo_num_slabs(0) == -1
    assert L.erl_gae_workspace_bytes(32, 4096) >= 32 * 4096 * 2
    spec = ops.MlpSpecN([17, 256, 128, 64, 5], True)
    assert spec.count == 17 * 256 + 256 + 256 * 128 + 128 + 128 * 64 + 64 + 64 * 5 + 5 + 5
    assert spec.workspace_bytes(1000, True) > spec.workspace_bytes(1000, False) > 0
    sac = ops.SacSpec(11, 3, [64, 32], 4)
    assert (sac.actor_count, sac.critic_count) == (3046, 9412)
    with pytest.raises(_hip.HipExtensionError):
        ops.MlpSpecN([8] + [16] * 9 + [2], True).count               # more than ERL_MAX_LAYERS hidden layers


def test_errors_are_reported_not_swallowed():
    """bad arguments come back as ERL_EINVAL with a message (no device needed: validation happens before any launch)."""
    from elegantrl_amd import _hip
    L = _hip.lib()
    rc = L.erl_split_ids_i64(None, 4, 0, None, None, None)
    assert rc == -1 and b"erl_split_ids_i64" in L.erl_last_error_string()
    rc = L.erl_ppo_step_f32(*([None] * 6), 64, 128, 128, 8, *([None] * 6), 32, 4096, None, 16384, 0.25, 0.001, 1.0, 0, None, 128, None)
    assert rc == -1 and b"NULL" in L.erl_last_error_string()


def test_product_package_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under elegantrl_amd/ may import it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "elegantrl_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports oracle"
    cpu_tensor_guard = open(os.path.join(ROOT, "elegantrl_amd", "_hip.py")).read()
    assert "no CPU path" in cpu_tensor_guard
