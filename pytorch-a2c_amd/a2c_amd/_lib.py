"""ctypes binding of liba2c_mi355x.so (the C ABI declared in include/a2c_mi355x.h).

The library is built in-tree by ``pytorch-a2c_amd/csrc/Makefile`` (``__graft_entry__.build()``).
There is no CPU fallback: if the library is missing, or a kernel is asked to run on a
non-CUDA tensor, this module raises.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint32, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liba2c_mi355x.so")


class ConvDesc(Structure):
    """a2c_conv_desc"""
    _fields_ = [("Cin", c_int), ("H", c_int), ("W", c_int), ("Cout", c_int), ("ks", c_int),
                ("stride", c_int), ("pad", c_int), ("OH", c_int), ("OW", c_int)]


P = c_void_p
PD = POINTER(ConvDesc)


class A3CStepArgs(Structure):
    """a2c_a3c_step_args"""
    _fields_ = [("B", c_int), ("C", c_int), ("H", c_int), ("W", c_int), ("n_actions", c_int),
                ("prev", P), ("prev_stride", c_int64), ("frame_new", P), ("reset_mask", P),
                ("out", P), ("out_stride", c_int64),
                ("wfrag1", P), ("bias1", P), ("wfrag2", P), ("bias2", P), ("Wc", P), ("bc", P),
                ("heads", P), ("ldh", c_int64), ("u", P), ("actions", P), ("act_stride", c_int64),
                ("rew", P), ("done", P), ("val_prev", P), ("rewards", P), ("dones", P), ("deltas", P),
                ("T", c_int64), ("t_rec", c_int64), ("slot0", c_int64), ("gamma", c_float),
                ("pong", c_int), ("bootstrap", c_int), ("frame_u8", P), ("frame_stride", c_int64),
                ("a1_out", P), ("a1_stride", c_int64), ("a2_out", P), ("a2_stride", c_int64),
                ("heads_out", P), ("heads_out_stride", c_int64)]


class A3CRolloutArgs(Structure):
    """a2c_a3c_rollout_args"""
    _fields_ = [("B", c_int), ("C", c_int), ("H", c_int), ("W", c_int), ("n_actions", c_int),
                ("states", P), ("bookmark", P),
                ("wfrag1", P), ("bias1", P), ("wfrag2", P), ("bias2", P), ("Wc", P), ("bc", P),
                ("heads", P), ("ldh", c_int64), ("u", P), ("u_stride", c_int64), ("actions", P),
                ("val_prev", P), ("rewards", P), ("dones", P), ("deltas", P),
                ("T", c_int64), ("slot0", c_int64), ("gamma", c_float), ("pong", c_int),
                ("cmd", P), ("rec", P), ("frames", P), ("frame_stride", c_int64),
                ("seq0", c_uint32), ("env0", c_int), ("err", P), ("timeout_ticks", c_int64), ("a1_rows", P), ("a2_rows", P), ("heads_rows", P), ("heads_rows_ld", c_int64),
                ("frame_store", P), ("frame_store_slot_stride", c_int64), ("nvalid_rows", P), ("nvalid_carry", P),
                ("frame_bits", c_int), ("conv1_weight", P), ("states_lazy", c_int),
                ("tagged", P), ("tagged_stride", c_int64), ("tagged_chunks", c_int), ("a1_lanemask_rows", P),
                ("a2_maskbit_rows", P)]


PS = POINTER(A3CStepArgs)
PR = POINTER(A3CRolloutArgs)

# name -> (restype, argtypes); one entry per prototype in include/a2c_mi355x.h
SIGNATURES = {
    "a2c_version": (c_int, []),
    "a2c_error_string": (c_char_p, [c_int]),
    "a2c_discount_scan": (c_int, [P, P, P, c_int64, c_int64, c_float, P, P]),
    "a2c_gae_returns_fused": (c_int, [P, P, P, P, P, c_int64, c_int64, c_float, c_float, P, P]),
    "a2c_moments": (c_int, [P, c_int64, P, P, P]),
    "a2c_normalize": (c_int, [P, P, c_int64, P, c_int64, c_float, P]),
    "a2c_add": (c_int, [P, P, P, c_int64, P]),
    "a2c_frame_stack_push": (c_int, [P, P, P, c_int64, P, c_int64, c_int, c_int, c_int, P]),
    "a2c_frame_stack_push_u8": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, c_int, c_int, c_int, P]),
    "a2c_pool_publish_actions": (c_int, [P, P, c_int64, c_int, P, ctypes.c_uint32, P]),
    "a2c_pool_ingest": (c_int, [P, P, c_int64, c_int, c_int, P, ctypes.c_uint32, c_int64, P, P, P, P, c_int64, P]),
    "a2c_pool_ingest_bits": (c_int, [P, P, c_int64, c_int, c_int, P, ctypes.c_uint32, c_int64, P, P, P, P, c_int64, P]),
    "a2c_unpack_bits": (c_int, [P, c_int64, P, c_int64, c_int, c_int, P]),
    "a2c_pool_ingest_post": (c_int, [c_int, P, P, c_int64, c_int, c_int, P, ctypes.c_uint32, c_int64, P, P, P, P, c_int64,
                                     P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_int, P, P, c_int, P, c_int64, P,
                                     P, P, P]),
    "a2c_store_u32_system": (c_int, [P, ctypes.c_uint32, P]),
    "a2c_softmax_sample": (c_int, [P, c_int64, P, P, c_int64, P, c_int, c_int, P]),
    "a2c_sample_probs": (c_int, [P, P, P, c_int64, c_int, P]),
    "a2c_rollout_record": (c_int, [P, P, P, c_int64, P, P, P, P, P, P, c_int, c_int, c_int64, c_int64, c_int64,
                                    c_float, c_int, P]),
    "a2c_rollout_post": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_int, P, P, P, c_int64,
                                  P, c_int64, c_int, c_int, c_int, P]),
    "a2c_rollout_post_rec": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_int, P, P, c_int64, P, P,
                             c_int64, P, c_int64, c_int, c_int, c_int, P, P, c_int, P, c_int64, P, P]),
    "a2c_rollout_post_u8": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_int, P, c_int64, P, P,
                                     c_int64, P, c_int64, c_int, c_int, c_int, P]),
    "a2c_rollout_bootstrap": (c_int, [P, c_int64, P, P, P, P, c_int, c_int64, c_int64, c_float, P]),
    "a2c_copy_rows": (c_int, [P, c_int64, P, c_int64, c_int, c_int64, P]),
    "a2c_mask_rows": (c_int, [P, c_int64, P, c_int64, c_int, c_int, P]),
    "a2c_permute_rows": (c_int, [P, P, c_int64, c_int64, c_int64, P]),
    "a2c_loss_fwd_bwd": (c_int, [P, c_int64, P, c_int64, P, P, P, P, c_int64, c_int64, c_int, c_float, c_float,
                                  c_float, P, c_int64, P, c_int64, P, P, P]),
    "a2c_gemm_ws_bytes": (c_size_t, [c_int64, c_int64, c_int]),
    "a2c_gemm_x9_ws_bytes": (c_size_t, [c_int64, c_int64, c_int64]),
    "a2c_gemm_x6_image_bytes": (c_size_t, [c_int64, c_int64]),
    "a2c_gemm_x6_split": (c_int, [P, c_int64, c_int64, c_int64, c_int, P, P]),
    "a2c_gemm_x6_images": (c_int, [c_int64, c_int64, c_int64, P, P, P, c_int64, P, c_int, P, c_int64, c_int, c_int, P, c_size_t, P]),
    "a2c_gemm_f32": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64,
                              P, c_int, P, c_int64, c_int, c_int, P, c_size_t, P]),
    "a2c_gemm_splits": (c_int, [c_int64, c_int]),
    "a2c_small_n_bwd_data_bits": (c_int, [P, c_int64, P, P, c_int64, P, c_int64, c_int64, c_int, c_int64, P]),
    "a2c_gemm_f32_partial": (c_int, [c_int, c_int, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, c_int, P, c_size_t, P]),
    "a2c_heads_fused": (c_int, [P, c_int, c_int64, c_int64, P, c_int, P, c_int64, P, P, P, c_int64, c_int64, c_int, c_int,
                                 P, c_int, P, c_int64, P]),
    "a2c_heads_fused_publish": (c_int, [P, c_int, c_int64, c_int64, P, c_int, P, c_int64, P, P, P, c_int64, c_int64, c_int, c_int,
                                         P, c_int, P, c_int64, P, P, ctypes.c_uint32, P]),
    "a2c_gemm_f32_nt": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int, P]),
    "a2c_gemm_f32_nn": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, P, c_int64, P]),
    "a2c_gemm_f32_tn": (c_int, [c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, c_int64, c_int, P,
                                 c_size_t, P]),
    "a2c_colsum_ws_bytes": (c_size_t, [c_int64]),
    "a2c_colsum": (c_int, [P, c_int64, c_int64, c_int64, P, P, c_size_t, P]),
    "a2c_conv2d_prep_floats": (c_size_t, [PD, c_int]),
    "a2c_compose_heads": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "a2c_a3c_step_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "a2c_a3c_ring_supported": (c_int, [c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "a2c_a3c_step": (c_int, [PS, P]),
    "a2c_a3c_rollout": (c_int, [PR, P]),
    "a2c_rollout_buffer_create": (c_int, [c_char_p, c_size_t, POINTER(c_void_p), POINTER(c_void_p)]),
    "a2c_rollout_buffer_destroy": (c_int, [c_char_p, P, c_size_t]),
    "a2c_pinned_register": (c_int, [P, c_size_t, POINTER(c_void_p)]),
    "a2c_pinned_unregister": (c_int, [P]),
    "a2c_push_buffer_alloc": (c_int, [c_size_t, POINTER(c_void_p)]),
    "a2c_push_buffer_free": (c_int, [P]),
    "a2c_memcpy_async": (c_int, [P, P, c_size_t, c_int, P]),
    "a2c_device_pci_bus_id": (c_int, [c_char_p, c_int]),
    "a2c_set_blocking_sync": (c_int, [c_int]),
    "a2c_pack_update_scalars": (c_int, [P, P, P, P, P]),
    "a2c_conv2d_prep_weights": (c_int, [PD, c_int, P, P, P]),
    "a2c_conv2d_fwd": (c_int, [PD, P, c_int64, P, P, c_int, P, c_int64, c_int, P]),
    "a2c_conv2d_bwd_data": (c_int, [PD, P, P, P, P, c_int, P]),
    "a2c_conv2d_sign_words": (c_int64, [PD]),
    "a2c_conv2d_fwd_signs": (c_int, [PD, P, c_int64, P, P, c_int, P, c_int64, P, c_int64, c_int, P]),
    "a2c_conv2d_bwd_data_signs_supported": (c_int, [PD]),
    "a2c_conv2d_bwd_data_w1_frames_ws_bytes": (c_size_t, [PD, PD, c_int]),
    "a2c_conv2d_bwd_data_w1_frames": (c_int, [PD, P, P, P, c_int64, PD, P, c_int64, c_int64, P, P, P, c_int, P, c_size_t, P]),
    "a2c_lanemask_from_act": (c_int, [P, P, c_int64, P]),
    "a2c_conv2d_bwd_data_lanemask_supported": (c_int, [PD, c_int]),
    "a2c_conv2d_bwd_rank_supported": (c_int, [PD, c_int, c_int]),
    "a2c_conv2d_bwd_data_lanemask_rank": (c_int, [PD, P, c_int64, c_int, P, P, c_int64, P, P, P, c_int, P]),
    "a2c_conv2d_bwd_weight_rank": (c_int, [PD, P, c_int64, P, c_int64, c_int, P, P, c_int64, P, P, c_int, P, c_size_t, P]),
    "a2c_conv2d_bwd_data_lanemask": (c_int, [PD, P, P, P, P, c_int, P]),
    "a2c_conv2d_bwd_data_signs": (c_int, [PD, P, P, P, c_int64, P, c_int, P]),
    "a2c_conv2d_bwd_weight_ws_bytes": (c_size_t, [PD, c_int]),
    "a2c_conv2d_bwd_data_w1_ws_bytes": (c_size_t, [PD, PD, c_int]),
    "a2c_conv2d_bwd_data_w1": (c_int, [PD, P, P, P, PD, P, c_int64, P, P, c_int, P, c_size_t, P]),
    "a2c_conv2d_bwd_weight": (c_int, [PD, P, c_int64, P, P, P, c_int, P, c_size_t, P]),
    "a2c_conv2d_bwd_weight_frames": (c_int, [PD, P, c_int64, c_int64, P, P, P, P, c_int, P, c_size_t, P]),
    "a2c_conv2d_fwd_frames_supported": (c_int, [PD]),
    "a2c_conv2d_fwd_chain_supported": (c_int, [PD, c_int]),
    "a2c_conv2d_fwd_chain": (c_int, [PD, c_int, P, c_int64, P, P, c_int, P, P, P, P, c_int, P]),
    "a2c_conv2d_fwd_frames": (c_int, [PD, P, c_int64, c_int64, P, c_int64, P, P, c_int, P, c_int64, P, c_int64, c_int, P]),
    "a2c_frame_prep_u8": (c_int, [P, c_int64, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, c_int64, c_int, P]),
    "a2c_frame_store_begin": (c_int, [P, c_int64, c_int64, c_int, c_int, P, P, c_int, P]),
    "a2c_rollout_post_frames": (c_int, [P, P, P, c_int64, P, P, P, P, c_int64, c_int64, c_int64, c_float, c_int, c_int, P, P,
                                        c_int, P, c_int64, P, P, P, P]),
    "a2c_frames_to_states": (c_int, [P, c_int64, P, c_int64, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "a2c_gru_gates": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_gru_out": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_gru_cell_bwd": (c_int, [P, P, P, c_int64, P, P, P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_gru_cell_fwd": (c_int, [P, c_int64, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P]),
    "a2c_gru_out_bwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_gru_out_bwd_carry": (c_int, [P, P, P, c_int64, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_gru_gates_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, c_int, P]),
    "a2c_layernorm_fwd": (c_int, [P, P, P, P, P, P, c_int64, c_int, P]),
    "a2c_layernorm_bwd": (c_int, [P, P, P, P, P, P, P, c_int64, c_int, c_int, P]),
    "a2c_gradnorm_sq": (c_int, [P, c_int64, P, P, P]),
    "a2c_clip_rmsprop": (c_int, [P, P, P, c_int64, P, c_double, c_double, c_double, c_double, P, P]),
    "a2c_clip_adam": (c_int, [P, P, P, P, c_int64, P, c_double, c_double, c_double, c_double, c_double,
                               c_int64, P, P]),
}

_lib = None


class A2CKernelError(RuntimeError):
    pass


def load():
    """Load the library (once) and declare every prototype.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build the HIP kernels first (python -c 'import __graft_entry__ as g; "
            "g.build()' or make -C pytorch-a2c_amd/csrc). a2c_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().a2c_error_string(rc).decode()
        raise A2CKernelError(f"{what}: {msg} (code {rc})")
