"""CPU: the C-ABI library builds, loads, and exports exactly what include/a2c_mi355x.h declares
(no compute calls here: there is no GPU in the build container)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes(header="a2c_mi355x.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(a2c_[a-z0-9_]+)\s*\(", src)))


def test_header_lists_the_path_functions():
    names = header_prototypes()
    for n in ("a2c_discount_scan", "a2c_gae_returns_fused", "a2c_frame_stack_push", "a2c_softmax_sample",
              "a2c_rollout_record", "a2c_loss_fwd_bwd", "a2c_gemm_f32", "a2c_conv2d_fwd", "a2c_conv2d_bwd_data",
              "a2c_conv2d_bwd_weight", "a2c_gru_gates", "a2c_layernorm_fwd", "a2c_clip_rmsprop", "a2c_clip_adam"):
        assert n in names


def test_library_exports_every_declared_symbol():
    from a2c_amd import _lib
    lib = _lib.load()                      # raises if the .so is missing or a symbol is absent
    names = header_prototypes()
    assert sorted(_lib.SIGNATURES) == names, set(_lib.SIGNATURES) ^ set(names)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (a2c_[a-z0-9_]+)", out))
    assert set(names) <= exported, set(names) - exported
    assert lib.a2c_version() == 2
    assert lib.a2c_error_string(-1) == b"invalid argument"


def test_hostpool_library_exports_every_declared_symbol():
    """include/a2c_hostpool.h <-> liba2c_hostpool.so (plain C, loaded by the env worker processes) and it
    must not depend on the HIP runtime"""
    from a2c_amd import hostpool
    hostpool.pool_lib()
    names = header_prototypes("a2c_hostpool.h")
    assert sorted(hostpool.POOL_SIGNATURES) == names, set(hostpool.POOL_SIGNATURES) ^ set(names)
    out = subprocess.run(["nm", "-D", "--defined-only", hostpool.POOL_LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (a2c_[a-z0-9_]+)", out))
    assert set(names) <= exported, set(names) - exported
    ldd = subprocess.run(["ldd", hostpool.POOL_LIB_PATH], capture_output=True, text=True).stdout
    assert "amdhip" not in ldd and "hsa" not in ldd


def test_argument_validation_without_gpu():
    """launchers reject bad arguments before touching the device"""
    from a2c_amd import _lib
    lib = _lib.load()
    assert lib.a2c_discount_scan(None, None, None, -1, 4, 0.9, None, None) == -1
    assert lib.a2c_discount_scan(None, None, None, 0, 4, 0.9, None, None) == 0          # empty input is a no-op
    assert lib.a2c_gemm_f32(0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, 0, None, 0, 0, 1, None, 0, None) == -1
    assert lib.a2c_loss_fwd_bwd(None, 3, None, 1, None, None, None, None, 4, 4, 64, 1.0, .5, .005, None, 3, None, 1,
                                None, None, None) == -1
    assert lib.a2c_moments(None, 4, None, None, None) == -1 and lib.a2c_gradnorm_sq(None, 4, None, None, None) == -1
    d = _lib.ConvDesc(3, 8, 8, 16, 3, 1, 1, 8, 8)       # Cin % 4 != 0
    import ctypes
    assert lib.a2c_conv2d_prep_floats(ctypes.byref(d), 0) == 0
    # one-launch rollout step: shape gate is host logic, bad argument blocks are refused
    assert lib.a2c_a3c_step_supported(4, 84, 84, 6) == 1
    assert lib.a2c_a3c_step_supported(3, 84, 84, 6) == 0 and lib.a2c_a3c_step_supported(4, 84, 82, 6) == 0
    assert lib.a2c_a3c_step_supported(4, 200, 200, 6) == 0          # state does not fit one CU's LDS
    assert lib.a2c_a3c_step(None, None) == -1
    args = _lib.A3CStepArgs(B=2, C=4, H=84, W=84, n_actions=6)       # all pointers NULL
    assert lib.a2c_a3c_step(ctypes.byref(args), None) == -1
    assert lib.a2c_a3c_rollout(None, None) == -1
    rargs = _lib.A3CRolloutArgs(B=2, C=4, H=84, W=84, n_actions=3, T=4)
    assert lib.a2c_a3c_rollout(ctypes.byref(rargs), None) == -1
    assert lib.a2c_frame_stack_push_u8(None, 7056, None, None, 0, None, 0, 2, 4, 7056, None) == -1
    # round 6: the mask-bit entry points
    assert lib.a2c_lanemask_from_act(None, None, 0, None) == 0 and lib.a2c_lanemask_from_act(None, None, 100, None) == -1
    assert lib.a2c_lanemask_from_act(None, None, 256, None) == -1                      # NULL pointers
    assert lib.a2c_small_n_bwd_data_bits(None, 4, None, None, 2592, None, 324, 0, 3, 2592, None) == 0      # no rows
    assert lib.a2c_small_n_bwd_data_bits(None, 4, None, None, 2592, None, 324, 8, 3, 2592, None) == -1     # NULL pointers
    assert lib.a2c_small_n_bwd_data_bits(None, 4, None, None, 2592, None, 324, 8, 9, 2592, None) == -1     # N > 8
    assert lib.a2c_small_n_bwd_data_bits(None, 4, None, None, 2590, None, 324, 8, 3, 2590, None) == -1     # K % 8
    d2 = _lib.ConvDesc(16, 20, 20, 32, 4, 2, 0, 9, 9)
    assert lib.a2c_conv2d_bwd_data_lanemask_supported(ctypes.byref(d2), 0) == 0
    assert lib.a2c_conv2d_bwd_data_lanemask(ctypes.byref(d2), None, None, None, None, 4096, None) == -1
    # the rank-head forms of conv2's two backward passes and the prebuilt-image form of the bf16 x 6 GEMM: argument checks
    assert lib.a2c_conv2d_bwd_rank_supported(ctypes.byref(d2), 5, 4096) == 0 and lib.a2c_conv2d_bwd_rank_supported(ctypes.byref(d2), 3, 0) == 0
    assert lib.a2c_conv2d_bwd_data_lanemask_rank(ctypes.byref(d2), None, 4, 3, None, None, 324, None, None, None, 4096, None) == -1
    assert lib.a2c_conv2d_bwd_data_lanemask_rank(ctypes.byref(d2), None, 4, 3, None, None, 324, None, None, None, 0, None) == 0
    assert lib.a2c_conv2d_bwd_weight_rank(ctypes.byref(d2), None, 6400, None, 4, 3, None, None, 324, None, None, 4096, None, 0, None) == -1
    assert lib.a2c_gemm_x6_image_bytes(256, 28224) == 3 * 2 * 256 * 28224 and lib.a2c_gemm_x6_image_bytes(0, 5) == 0
    assert lib.a2c_gemm_x6_image_bytes(257, 17) == 3 * 2 * 512 * 32
    assert lib.a2c_gemm_x6_split(None, 8, 4, 8, 1, None, None) == -1
    assert lib.a2c_gemm_x6_images(4, 4, 8, None, None, None, 4, None, 0, None, 0, 0, 1, None, 0, None) == -1
    rargs2 = _lib.A3CRolloutArgs(B=2, C=4, H=84, W=84, n_actions=3, T=4)
    assert "a1_lanemask_rows" in dict(rargs2._fields_) and "a2_maskbit_rows" in dict(rargs2._fields_)
    assert lib.a2c_memcpy_async(None, None, 0, 1, None) == 0 and lib.a2c_memcpy_async(None, None, 8, 1, None) == -1


def test_product_refuses_cpu_tensors():
    import torch
    from a2c_amd import ops, utils
    with pytest.raises(RuntimeError):
        ops.discount_rows(torch.zeros(4), torch.zeros(4), 0.9, 1, 4)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            utils.discount(torch.zeros(4), torch.zeros(4), 0.9)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pytorch-a2c_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                for line in open(os.path.join(dp, f)).read().splitlines():
                    low = line.lower()
                    assert not ("oracle" in low and ("import" in low or "include" in low)), (f, line)


def test_torch_library_ops_are_registered_without_cpu_kernels():
    """torch.ops.a2c_mi355x.* (TORCH_LIBRARY shim over the C ABI): registered, HIP dispatch key only"""
    import torch
    from a2c_amd import ops
    o = ops.load_torch_ops()
    for name in ("discount", "gae_returns", "softmax_sample", "frame_stack_push", "loss_fwd_bwd", "linear", "clip_rmsprop_"):
        assert hasattr(o, name), name
    with pytest.raises(NotImplementedError):
        o.discount(torch.zeros(4), torch.zeros(4), 0.9, 1)


def test_generated_abi_ops_are_current_and_registered():
    """csrc/torch_ops_abi.inc is generated from the header (tools/gen_torch_abi_ops.py): the committed file is what the
    generator writes today, and every kernel-launching entry point it covers is a registered torch op (HIP key only)"""
    import importlib.util
    import sys
    import torch
    from a2c_amd import _lib, ops
    gen_path = os.path.join(ROOT, "tools", "gen_torch_abi_ops.py")
    assert subprocess.run([sys.executable, gen_path, "--check"]).returncode == 0
    spec = importlib.util.spec_from_file_location("_gen_abi", gen_path)
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    planned, skipped = gen.plan()
    o = ops.load_torch_ops()
    names = [n for n, _, _ in planned]
    assert len(names) >= 55 and set(names) | set(skipped) <= set(_lib.SIGNATURES)
    for n in ("a2c_gae_returns_fused", "a2c_loss_fwd_bwd", "a2c_gemm_f32", "a2c_conv2d_bwd_data", "a2c_conv2d_bwd_weight",
              "a2c_conv2d_bwd_data_lanemask", "a2c_clip_rmsprop", "a2c_clip_adam", "a2c_gru_cell_bwd", "a2c_layernorm_bwd"):
        assert n in names, n
    for n, params, kinds in planned:
        assert hasattr(o, "abi_" + n[4:]), n
        # the ctypes table and the header agree on how many parameters there are (+ the stream)
        assert len(_lib.SIGNATURES[n][1]) == len(params) + 1, n
    with pytest.raises(NotImplementedError):
        o.abi_add([torch.zeros(4), torch.zeros(4), torch.zeros(4)], [0, 0, 0], [4], [])
