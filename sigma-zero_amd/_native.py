"""ctypes binding of libsigmazero_hip.so (include/sigmazero.h).  Fails loudly: there is no CPU fallback
for the search path — if the library is missing, build it (`python -c "import __graft_entry__ as g; g.build()"`)."""
import ctypes as C
import os

from . import build as _build

SZ_RING, SZ_PLANES, SZ_ACTIONS, SZ_MAX_MOVES, SZ_POS_BYTES, SZ_MASK_WORDS = 256, 119, 4672, 218, 80, 73
SZ_OK, SZ_ERR_INVALID, SZ_ERR_HIP, SZ_ERR_CAPACITY, SZ_ERR_NO_DEVICE, SZ_ERR_STATE, SZ_ERR_ZERO_VISITS = 0, -1, -2, -3, -4, -5, -6
SZ_PLANES_F32, SZ_PLANES_BF16, SZ_PLANES_NHWC128_BF16, SZ_PLANES_NHWC128_BITS = 0, 1, 2, 3
SZ_NN_W16 = 0x40000
SZ_NN_IN_BITS = 0x1000000
SZ_NN_SPLIT_WGB1, SZ_NN_SPLIT_WGB2 = 0x2000000, 0x4000000
SZ_NN_F16 = 0x8000000
SZ_NN_TOWER_WGB1, SZ_NN_TOWER_WGB2 = 0x10000000, 0x20000000


class sz_config(C.Structure):
    _fields_ = [("n_boards", C.c_int32), ("num_searches", C.c_int32), ("c_puct", C.c_float), ("learning", C.c_int32),
                ("noise_value", C.c_float), ("chess960", C.c_int32), ("edges_per_board", C.c_int32),
                ("planes_dtype", C.c_int32), ("device", C.c_int32), ("reuse_subtree", C.c_int32)]


class sz_stats(C.Structure):
    _fields_ = [("simulations", C.c_uint64), ("expansions", C.c_uint64), ("terminal_hits", C.c_uint64), ("sum_depth", C.c_uint64),
                ("sum_children", C.c_uint64), ("max_edges_used", C.c_uint64), ("boards_pending", C.c_int32),
                ("boards_done", C.c_int32), ("boards_error", C.c_int32), ("first_error", C.c_int32)]


class NativeError(RuntimeError):
    def __init__(self, code, what=""):
        self.code = code
        msg = lib().sz_error_string(code).decode() if _lib is not None else str(code)
        super().__init__("sigmazero native call failed%s: %s (%d)" % (" in " + what if what else "", msg, code))


EXPORTS = {
    # engine
    "sz_create": (C.c_int, [C.POINTER(sz_config), C.POINTER(C.c_void_p)]),
    "sz_destroy": (C.c_int, [C.c_void_p]),
    "sz_new_games": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_upload_game": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]),
    "sz_set_active": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_compact": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.c_void_p]),
    "sz_search_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_search_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_get_stats": (C.c_int, [C.c_void_p, C.POINTER(sz_stats), C.c_void_p]),
    "sz_root_children": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    "sz_play": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_fetch_ply": (C.c_int, [C.c_void_p] + [C.c_void_p] * 9 + [C.c_void_p]),
    "sz_debug_pending": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_void_p]),
    "sz_debug_position": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]),
    "sz_debug_tree": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [C.POINTER(C.c_int32), C.c_void_p]),
    "sz_debug_select": (C.c_int, [C.c_void_p] * 8 + [C.c_int32, C.c_void_p]),
    "sz_nn_conv_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_int32] * 4 + [C.c_void_p]),
    "sz_nn_block_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_int32] * 2 + [C.c_void_p]),
    "sz_nn_tower_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_tower_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_debug_split_stamps": (C.c_int, [C.c_void_p, C.c_int32]),
    "sz_nn_split_stream_elems": (C.c_int64, [C.c_int32]),
    "sz_nn_pack_split_stream": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_forward_split": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_pack_split_head": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_nn_pack_split_head_f16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_nn_pack_split_stream_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_conv3x3_split_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "sz_nn_wgrad3x3_split_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "sz_nn_pack_conv_split_dev": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_nn_pack_conv_split_both": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "sz_bn_act_train_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_bn_act_train_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_conv3x3_train_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "sz_nn_conv3x3_train_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_value_mlp": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "sz_debug_stream_read": (C.c_int, [C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p, C.c_void_p]),
    "sz_nn_policy_head_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] * 2 + [C.c_void_p]),
    "sz_debug_step_stamps": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_set_root_noise": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_nn_debug_tower_stamps": (C.c_int, [C.c_void_p, C.c_int32]),
    "sz_nn_heads_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_float] + [C.c_void_p] * 3 + [C.c_float] + [C.c_void_p] * 3 + [C.c_int32] * 2 + [C.c_void_p]),
    "sz_nn_value_head_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    "sz_nn_pack_head16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_nn_pack_head16_f16": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sz_nn_pack_weights16_f16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_pack_weights16": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_nn_pack_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "sz_error_string": (C.c_char_p, [C.c_int]),
    "sz_device_count": (C.c_int, []),
    # host mirror
    "szh_game_new": (C.c_void_p, [C.c_int, C.c_int]),
    "szh_game_from_fen": (C.c_void_p, [C.c_char_p, C.c_int]),
    "szh_game_copy": (C.c_void_p, [C.c_void_p]),
    "szh_game_free": (None, [C.c_void_p]),
    "szh_legal_actions": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "szh_action_to_move": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "szh_move_to_action": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "szh_push_action": (C.c_int, [C.c_void_p, C.c_int]),
    "szh_push_move": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "szh_status": (None, [C.c_void_p, C.POINTER(C.c_int32)]),
    "szh_planes": (None, [C.c_void_p, C.c_void_p]),
    "szh_perft": (C.c_uint64, [C.c_void_p, C.c_int]),
    "szh_bitboards": (None, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "szh_export": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "szh_is_chess960": (C.c_int, [C.c_void_p]),
    "szh_plane_bits_mismatches": (C.c_int, [C.c_void_p]),
}

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        raise RuntimeError("libsigmazero_hip.so is not built (%s). Run: python -c \"import __graft_entry__ as g; g.build()\". "
                           "The search path has no CPU fallback." % path)
    L = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(L, name)            # AttributeError here = the library does not export what include/sigmazero.h declares
        fn.restype, fn.argtypes = res, args
    _lib = L
    return L


def check(code, what=""):
    if code != SZ_OK:
        raise NativeError(code, what)
