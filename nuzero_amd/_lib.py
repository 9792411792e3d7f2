"""ctypes binding of include/nuzero_amd.h.

The shared library is built in-tree by ``python -m nuzero_amd.build``.  There is
no fallback: if it is missing, importing this module raises.
"""
import ctypes
import os

# PyTorch-ROCm ships its own libamdhip64.so.7; the engine library is linked against
# the same SONAME.  Importing torch first makes the process use ONE HIP runtime (torch's)
# whatever the import order of the caller -- two runtimes in one process do not see
# each other's devices or streams.
import torch  # noqa: F401

from ctypes import POINTER, Structure, c_char_p, c_double, c_int32, c_int64, c_uint32, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NZ_LIB_PATH") or os.path.join(_HERE, "csrc", "libnuzero_amd.so")   # env: timing-only ablation builds

NZ_OK, NZ_ERR_ARG, NZ_ERR_HIP, NZ_ERR_STATE, NZ_ERR_OVERFLOW = range(5)
NZ_GAME_TIC_TAC_TOE = 0
NZ_ACT_TANH, NZ_ACT_RELU = 0, 1
NZ_ARCH_RECURRENT, NZ_ARCH_RESNET, NZ_ARCH_CONVNET = 0, 1, 2
NZ_LOSS_CE, NZ_LOSS_KLD, NZ_LOSS_MSE = 0, 1, 2
NZ_LOSS_SE, NZ_LOSS_AE = 0, 1


class SearchCfg(Structure):
    _fields_ = [("mcts_simulations", c_int32), ("keep_subtree", c_int32),
                ("pb_c_base", c_double), ("pb_c_init", c_double),
                ("number_of_softmax_moves", c_int32), ("training", c_int32),
                ("epsilon_softmax_exploration", c_double), ("epsilon_random_exploration", c_double),
                ("value_factor", c_double), ("root_exploration_fraction", c_double),
                ("root_dist_alpha", c_double), ("root_dist_beta", c_double)]


class GameDesc(Structure):
    _fields_ = [("game", c_int32), ("negate_player", c_int32)]


class NetDesc(Structure):
    _fields_ = [("in_channels", c_int32), ("policy_channels", c_int32), ("width", c_int32),
                ("num_blocks", c_int32), ("recall", c_int32), ("value_activation", c_int32),
                ("arch", c_int32), ("kernel_size", c_int32), ("hex", c_int32)]


class Dims(Structure):
    _fields_ = [("n_games", c_int32), ("n_slots", c_int32), ("num_actions", c_int32), ("max_moves", c_int32),
                ("state_channels", c_int32), ("rows", c_int32), ("cols", c_int32),
                ("node_capacity", c_int32)]


class ScsDesc(Structure):
    _fields_ = [("rows", c_int32), ("cols", c_int32), ("turns", c_int32), ("stacking", c_int32),
                ("terrain", c_void_p), ("n_vp", c_int32 * 2), ("vp", c_void_p), ("n_units", c_int32),
                ("units", c_void_p), ("arrival", c_void_p)]


# name -> (restype, argtypes); every symbol include/nuzero_amd.h declares
SIGNATURES = {
    "nz_version": (c_char_p, []),
    "nz_last_error": (c_char_p, [c_void_p]),
    "nz_engine_create": (c_int32, [POINTER(c_void_p), POINTER(SearchCfg), POINTER(GameDesc), c_int32, c_int32]),
    "nz_engine_create_ex": (c_int32, [POINTER(c_void_p), POINTER(SearchCfg), POINTER(GameDesc), c_int32, c_int32,
                                      c_int32]),
    "nz_engine_destroy": (None, [c_void_p]),
    "nz_engine_dims": (c_int32, [c_void_p, POINTER(Dims)]),
    "nz_engine_set_weights": (c_int32, [c_void_p, POINTER(NetDesc), POINTER(c_void_p), c_int32, c_int32]),
    "nz_engine_set_table": (c_int32, [c_void_p, c_void_p, c_int32]),
    "nz_engine_reset": (c_int32, [c_void_p, c_void_p]),
    "nz_engine_root_children": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_engine_search": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_engine_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_engine_last_actions": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_engine_alive": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_engine_move": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_engine_live_games": (c_int32, [c_void_p, POINTER(c_int32), c_void_p]),
    "nz_engine_play": (c_int32, [c_void_p, c_uint64, c_void_p]),
    "nz_engine_play_next": (c_int32, [c_void_p, c_uint64, c_int32, c_uint64, c_void_p]),
    "nz_engine_play_lockstep": (c_int32, [c_void_p, c_uint64, c_void_p]),
    "nz_engine_desync_count": (c_int32, [c_void_p, POINTER(c_int64)]),
    "nz_engine_export": (c_int32, [c_void_p] + [c_void_p] * 8 + [c_void_p]),
    "nz_engine_export_trace": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_engine_counters": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int64), c_void_p]),
    "nz_engine_counters_ex": (c_int32, [c_void_p, POINTER(c_int64), c_void_p]),
    "nz_engine_counters_n": (c_int32, [c_void_p, POINTER(c_int64), c_int32, c_void_p]),
    "nz_engine_net_flops": (c_int32, [c_void_p, POINTER(c_double)]),
    "nz_engine_net_matrix_flops": (c_int32, [c_void_p, POINTER(c_double), POINTER(c_double)]),
    "nz_net_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_net_forward_stamps": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, POINTER(c_double)]),
    "nz_engine_profile": (c_int32, [c_void_p, c_int32]),
    "nz_engine_profile_read": (c_int32, [c_void_p, POINTER(c_double), POINTER(c_int64), POINTER(c_int64)]),
    "nz_engine_phase_stamps": (c_int32, [c_void_p, c_int32, POINTER(c_double)]),
    "nz_scs_create": (c_int32, [POINTER(c_void_p), POINTER(ScsDesc), c_int32, c_int32]),
    "nz_scs_destroy": (None, [c_void_p]),
    "nz_scs_last_error": (c_char_p, [c_void_p]),
    "nz_scs_dims": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32)]),
    "nz_scs_reset": (c_int32, [c_void_p, c_void_p]),
    "nz_scs_step": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_legal_mask": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_state_image": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_status": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_create": (c_int32, [POINTER(c_void_p), POINTER(ScsDesc), POINTER(SearchCfg), c_int32, c_int32,
                                       c_int32]),
    "nz_scs_search_destroy": (None, [c_void_p]),
    "nz_scs_search_last_error": (c_char_p, [c_void_p]),
    "nz_scs_search_reset": (c_int32, [c_void_p, c_void_p]),
    "nz_scs_search_root_children": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_begin_move": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_select": (c_int32, [c_void_p, c_void_p, c_void_p, POINTER(c_int32), c_void_p]),
    "nz_scs_search_expand": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_end_move": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_status": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_export": (c_int32, [c_void_p] + [c_void_p] * 9 + [POINTER(c_int64), c_void_p]),
    "nz_boardnet_create": (c_int32, [POINTER(c_void_p), POINTER(NetDesc), c_int32, c_int32, c_int32, c_int32]),
    "nz_boardnet_destroy": (None, [c_void_p]),
    "nz_boardnet_last_error": (c_char_p, [c_void_p]),
    "nz_boardnet_set_weights": (c_int32, [c_void_p, POINTER(c_void_p), c_int32, c_int32]),
    "nz_boardnet_forward": (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_boardnet_input_rows": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_int32)]),
    "nz_boardnet_forward_rows": (c_int32, [c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_boardnet_flops": (c_int64, [c_void_p]),
    "nz_boardnet_dims": (c_int32, [c_void_p] + [POINTER(c_int32)] * 5),
    "nz_boardnet_fused": (c_int32, [c_void_p, c_int32, POINTER(c_int32)]),
    "nz_scs_search_play": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_play_moves": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "nz_scs_search_play_round": (c_int32, [c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "nz_scs_search_export_round": (c_int32, [c_void_p] + [c_void_p] * 10 + [c_void_p, c_void_p]),
    "nz_scs_search_waves": (c_int32, [c_void_p, POINTER(c_int64)]),
    "nz_scs_search_apply": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_last_actions": (c_int32, [c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_cache": (c_int32, [c_void_p, c_int64]),
    "nz_scs_search_cache_stats": (c_int32, [c_void_p, POINTER(c_int64)]),
    "nz_scs_search_limits": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "nz_scs_search_set_games": (c_int32, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_scs_set_maps": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_persistent": (c_int32, [c_void_p, c_int32, POINTER(c_int32)]),
    "nz_scs_search_persist_ticks": (c_int32, [c_void_p, POINTER(c_int64)]),
    "nz_scs_search_persist_profile": (c_int32, [c_void_p, c_int32, POINTER(ctypes.c_double)]),
    "nz_scs_netbench": (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    "nz_boardnet_wide_stamps": (c_int32, [c_void_p]),
    "nz_scs_search_record": (c_int32, [c_void_p, c_void_p, c_int32, c_int32]),
    "nz_scs_search_record_read": (c_int32, [c_void_p, c_int32, POINTER(c_int32), c_void_p, c_void_p, c_void_p]),
    "nz_scs_search_phase_ticks": (c_int32, [c_void_p, POINTER(c_int64)]),
    "nz_replay_create": (c_int32, [POINTER(c_void_p), c_int64, c_int32, c_int32, c_int32]),
    "nz_replay_destroy": (None, [c_void_p]),
    "nz_replay_last_error": (c_char_p, [c_void_p]),
    "nz_replay_dims": (c_int32, [c_void_p, POINTER(c_int64), POINTER(c_int32), POINTER(c_int32)]),
    "nz_replay_append": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                   c_void_p, c_int32, c_void_p, c_int64, c_int32, c_void_p]),
    "nz_replay_gather": (c_int32, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_replay_check": (c_int32, [c_void_p, c_void_p]),
    "nz_loss_forward_backward": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                           c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "nz_loss_last_error": (c_char_p, []),
    "nz_rng_create": (c_void_p, [c_uint32]),
    "nz_rng_create_state": (c_void_p, [c_void_p, c_int32, c_int32, c_double]),
    "nz_rng_clone": (c_void_p, [c_void_p]),
    "nz_rng_destroy": (None, [c_void_p]),
    "nz_rng_seed": (None, [c_void_p, c_uint32]),
    "nz_rng_u32": (c_uint32, [c_void_p]),
    "nz_rng_double": (c_double, [c_void_p]),
    "nz_rng_gamma": (None, [c_void_p, c_double, c_double, c_int32, POINTER(c_double)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m nuzero_amd.build` "
            "(the engine has no CPU or PyTorch fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class NzError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"nuzero_amd error {code}: {message}")
        self.code = code


def check(status, handle=None):
    if status != NZ_OK:
        msg = lib.nz_last_error(handle)
        raise NzError(status, msg.decode() if msg else "")
