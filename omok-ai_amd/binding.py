"""ctypes binding of libomok_mi355x.so (include/omok_mi355x.h).  No fallbacks: if the HIP library
is missing or no GPU is present every entry point raises."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# OMOK_MI355X_LIB: another build of the same library (A/B timing of kernel variants inside one GPU session)
LIB_PATH = os.environ.get("OMOK_MI355X_LIB") or os.path.join(_HERE, "libomok_mi355x.so")

OK = 0
NET_F16X3, NET_F32, NET_F16X3_ROWS, NET_F16X3_FP6, NET_F16X3_F16, NET_F16X3_MIXED = 0, 1, 2, 3, 4, 5
FC0_FORMATS = {-1: "f32", 0: "fp6", 1: "f16", 2: "mixed"}
MODE_PLAYER, MODE_OPPONENT = 0, 1
STAT_NAMES = ["sims", "evals", "ply_games", "finished", "ms_tree", "ms_trunk", "ms_fc0", "ms_tail", "ms_ply",
              "fc0_launches", "fc0_rows", "tree_bytes", "round_launches", "ms_round", "peak_nodes", "peak_tables",
              "fc0_format", "probe_rows", "probe_dp_fp6", "probe_dv_fp6", "probe_dp_f16", "probe_dv_f16", "probe_limit", "probe_logit_max",
              "children2_launches", "children1_launches", "probe_dlogit_fp6", "probe_dlogit_f16", "probe_round_rows",
              "probe_round_dp_fp6", "probe_round_dv_fp6", "probe_round_dlogit_fp6", "probe_round_dp_mixed", "probe_round_dv_mixed", "probe_round_dlogit_mixed",
              "probe_round_dp_f16", "probe_round_dv_f16", "probe_round_dlogit_f16", "probe_logit_limit", "probe_outside",
              "work_diff_runs", "work_diff_singles", "work_diff_children", "work_copy_runs", "work_copy_singles", "work_copy_children", "work_diff_full_runs",
              "work_win_pixels", "work_win_tiles", "work_full_tiles"]

# every symbol include/omok_mi355x.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "omok_create", "omok_destroy", "omok_last_error", "omok_net_num_tensors", "omok_net_tensor_size", "omok_net_load",
    "omok_net_commit", "omok_net_load_file", "omok_net_save_file", "omok_evaluate_pv", "omok_evaluate_logits", "omok_env_play", "omok_env_place_stone", "omok_encode_nn_input", "omok_selfplay_reset", "omok_set_episode", "omok_execute", "omok_execute_shared", "omok_execute_shared_recorded",
    "omok_compute_policy", "omok_play_actions", "omok_set_actions", "omok_root_children",
    "omok_sample_actions", "omok_advance", "omok_selfplay_run", "omok_selfplay_run_slots", "omok_round_generate", "omok_round_inputs",
    "omok_round_eval", "omok_round_outputs", "omok_round_logits", "omok_round_inject", "omok_round_scatter", "omok_mirror_generate",
    "omok_mirror_inputs", "omok_mirror_eval", "omok_mirror_outputs", "omok_mirror_inject", "omok_mirror_apply",
    "omok_alive_count", "omok_current_ply", "omok_game_info", "omok_tree_dump", "omok_tree_root", "omok_replay_game",
    "omok_operand_row_bytes", "omok_debug_operand_rows", "omok_debug_set_base_cache", "omok_debug_set_children_kernel", "omok_debug_set_window_rects",
    "omok_replay_pack_dev", "omok_replay_record_bytes", "omok_replay_augment_dev", "omok_replay_augmented_game", "omok_get_stats", "omok_reset_stats", "omok_set_profiling",
]


class Config(C.Structure):
    _fields_ = [("board_size", C.c_int32), ("games", C.c_int32), ("max_nodes", C.c_int32), ("max_tables", C.c_int32),
                ("max_batch_k", C.c_int32), ("device", C.c_int32), ("net_mode", C.c_int32), ("max_tree_waves", C.c_int32),
                ("seed", C.c_uint64), ("game_offset", C.c_int64)]


class OmokError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"omok error {code}: {msg}")
        self.code = code


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(_HERE, "csrc")])


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OmokError(-2, f"{LIB_PATH} is missing: build it with __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                            "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    u8p = C.POINTER(C.c_uint8)
    H = C.c_void_p
    L.omok_create.argtypes = [C.POINTER(Config), C.POINTER(H)]
    L.omok_destroy.argtypes = [H]
    L.omok_destroy.restype = None
    L.omok_last_error.argtypes = [H]
    L.omok_last_error.restype = C.c_char_p
    L.omok_net_tensor_size.argtypes = [H, C.c_int]
    L.omok_net_tensor_size.restype = C.c_int64
    L.omok_net_load.argtypes = [H, C.c_int, fp, C.c_int64]
    L.omok_net_commit.argtypes = [H]
    L.omok_net_load_file.argtypes = [H, C.c_char_p]
    L.omok_net_save_file.argtypes = [H, C.c_char_p]
    L.omok_evaluate_pv.argtypes = [H, fp, C.c_int32, fp, fp]
    L.omok_evaluate_logits.argtypes = [H, fp, C.c_int32, fp, fp]
    L.omok_env_play.argtypes = [H, ip, C.c_int32, C.c_int32, ip, u8p, u8p, C.POINTER(C.c_uint16)]
    L.omok_encode_nn_input.argtypes = [H, u8p, u8p, C.c_int32, C.c_int32, fp]
    L.omok_selfplay_reset.argtypes = [H]
    L.omok_execute_shared_recorded.argtypes = [H, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, u8p, u8p, ip, fp, fp, C.c_int32, ip, ip]
    L.omok_operand_row_bytes.argtypes = [H]
    L.omok_operand_row_bytes.restype = C.c_int64
    L.omok_debug_operand_rows.argtypes = [H, C.c_int32, C.c_int32, C.c_void_p]
    L.omok_debug_set_base_cache.argtypes = [H, C.c_int32]
    L.omok_debug_set_children_kernel.argtypes = [H, C.c_int32]
    L.omok_debug_set_window_rects.argtypes = [H, C.c_int32]
    L.omok_set_episode.argtypes = [H, C.c_uint64]
    L.omok_env_place_stone.argtypes = [H, u8p, u8p, C.POINTER(C.c_uint16), ip, C.c_int32, ip]
    L.omok_compute_policy.argtypes = [H, fp, u8p]
    L.omok_play_actions.argtypes = [H, ip]
    L.omok_set_actions.argtypes = [H, ip]
    L.omok_root_children.argtypes = [H, C.c_int32, C.c_int32, ip, C.POINTER(C.c_uint32), fp, fp, C.c_int32]
    L.omok_execute.argtypes = [H, C.c_int32, C.c_int32, C.c_float, C.c_float]
    L.omok_execute_shared.argtypes = [H, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32]
    L.omok_sample_actions.argtypes = [H, C.c_float, C.c_int32, ip]
    L.omok_advance.argtypes = [H]
    L.omok_selfplay_run.argtypes = [H, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32,
                                    C.POINTER(C.c_double)]
    L.omok_selfplay_run_slots.argtypes = [H, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_int64,
                                          C.POINTER(C.c_int64), ip, ip, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    L.omok_round_generate.argtypes = [H, C.c_int32, C.c_int32, C.c_float, C.c_float, ip]
    L.omok_round_inputs.argtypes = [H, fp]
    L.omok_round_eval.argtypes = [H]
    L.omok_round_outputs.argtypes = [H, fp, fp]
    L.omok_round_logits.argtypes = [H, fp, fp]
    L.omok_round_inject.argtypes = [H, fp, fp]
    L.omok_round_scatter.argtypes = [H]
    L.omok_mirror_generate.argtypes = [H, ip]
    L.omok_mirror_inputs.argtypes = [H, fp]
    L.omok_mirror_eval.argtypes = [H]
    L.omok_mirror_outputs.argtypes = [H, fp]
    L.omok_mirror_inject.argtypes = [H, fp]
    L.omok_mirror_apply.argtypes = [H]
    L.omok_alive_count.argtypes = [H]
    L.omok_current_ply.argtypes = [H]
    L.omok_game_info.argtypes = [H, u8p, u8p, ip]
    L.omok_tree_dump.argtypes = [H, C.c_int32, C.c_int32, ip, fp, C.c_int32]
    L.omok_tree_root.argtypes = [H, C.c_int32, C.c_int32, C.POINTER(C.c_uint32), fp, ip, ip]
    L.omok_replay_game.argtypes = [H, C.c_int32, u8p, u8p, fp, fp, C.c_int32]
    L.omok_replay_pack_dev.argtypes = [H, C.c_void_p, C.c_int64]
    L.omok_replay_pack_dev.restype = C.c_int64
    L.omok_replay_augment_dev.argtypes = [H, C.c_void_p, C.c_int64]
    L.omok_replay_augment_dev.restype = C.c_int64
    L.omok_replay_augmented_game.argtypes = [H, C.c_int32, u8p, u8p, fp, fp, C.c_int32]
    L.omok_replay_record_bytes.argtypes = [H]
    L.omok_get_stats.argtypes = [H, C.POINTER(C.c_double)]
    L.omok_reset_stats.argtypes = [H]
    L.omok_set_profiling.argtypes = [H, C.c_int32]
    _lib = L
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))
