"""ctypes binding of libqttt_hip.so (include/qttt.h).  There is no CPU fallback: if the HIP
library is missing or a call fails, this raises."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# QTTT_LIB_PATH: load another build of the same ABI (A/B diagnostics); default = the in-tree build
LIB_PATH = os.environ.get("QTTT_LIB_PATH") or os.path.join(_HERE, "libqttt_hip.so")

ABI_VERSION = 5
FLAG_AUTO_RESET = 1
FLAG_FUSED = 2
BOARD_RECORD_BYTES = 64
SIM_STRIDE = 16
EXPAND_ROLLOUT_MAX_SIMS = 128
OP_MAKE_MOVE, OP_UPDATE_QSTRUCTS, OP_CHECK_WIN = 0, 1, 2

class EnvRecord(ctypes.Structure):
    """include/qttt.h: struct qttt_env."""
    _fields_ = [("state", ctypes.c_void_p), ("n", ctypes.c_int64), ("board_offset", ctypes.c_int64),
                ("seed", ctypes.c_uint64), ("flags", ctypes.c_uint32), ("reserved", ctypes.c_uint32),
                ("reward", ctypes.c_void_p), ("terminated", ctypes.c_void_p), ("classical", ctypes.c_void_p),
                ("q_p1", ctypes.c_void_p), ("q_p1_len", ctypes.c_void_p), ("q_p2", ctypes.c_void_p),
                ("q_p2_len", ctypes.c_void_p), ("turn", ctypes.c_void_p), ("step_counter", ctypes.c_void_p)]


ENV_STEP, ENV_STEP_OBSERVE, ENV_STEP_RANDOM, ENV_SAMPLE = 0, 1, 2, 3

# every symbol include/qttt.h declares: name -> (restype, argtypes)
_vp, _i32, _i64, _u64, _u32 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint64,
                               ctypes.c_uint32)
SIGNATURES = {
    "qttt_abi_version": (_i32, []),
    "qttt_state_bytes": (_i64, [_i64]),
    "qttt_reset": (_i32, [_vp, _i64, _vp]),
    "qttt_reset_observe": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_step": (_i32, [_vp, _vp, _vp, _u64, _u32, _i64, _u32, _vp, _vp, _i64, _vp]),
    "qttt_step_observe": (_i32, [_vp, _vp, _vp, _u64, _u32, _i64, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                 _i64, _vp]),
    "qttt_step_many": (_i32, [_vp, _vp, _vp, _u64, _u32, _i64, _u32, _vp, _vp, _i64, _i64, _i32, _vp]),
    "qttt_step_random_many": (_i32, [_vp, _u64, _u32, _i64, _u32, _vp, _vp, _vp, _i64, _vp, _i64, _i32, _vp]),
    "qttt_observe": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_check_win": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "qttt_export": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_import": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_board_op": (_i32, [_vp, _vp, _i64, _vp]),
    "qttt_board_op_sync": (_i32, [_vp, _vp, _i64, _vp]),
    "qttt_board_op_host": (_i32, [_vp, _vp, _i64, _vp]),
    "qttt_board_mailbox_retire": (_i32, [_i32]),
    "qttt_sample_actions": (_i32, [_vp, _u64, _u32, _i64, _u32, _vp, _i64, _vp]),
    "qttt_node_info": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_state_key": (_u64, [_u64, _u64]),
    "qttt_expand": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "qttt_expand_rollout": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u64, _u32, _i64, _i32, _vp, _vp,
                                   _i64, _vp]),
    "qttt_rollout": (_i32, [_vp, _u64, _u32, _i64, _vp, _vp, _vp, _i64, _vp]),
    "qttt_rollout_many": (_i32, [_vp, _u64, _u32, _i64, _i32, _vp, _vp, _i64, _vp]),
    "qttt_encode": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "qttt_set_tuning": (_i32, [_i32, _i32]),
    "qttt_step_launch_shape": (_i32, [_i64, _u32, _i32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "qttt_env_step": (_i32, [ctypes.POINTER(EnvRecord), _vp, _vp, _u32, _i32, _vp]),
    "qttt_counter_add": (_i32, [_vp, _u32, _vp]),
    "qttt_step_random": (_i32, [_vp, _u64, _u32, _i64, _u32, _vp, _vp, _vp, _i64, _vp]),
    "qttt_hash": (_u64, [_u64, _u64, _u32]),
}

_lib = None

# Process settings the measurements of this repo were made with.  Importing the package does NOT touch os.environ
# (an embedding host owns its environment): a launcher calls recommended_env(apply=True) before anything initialises
# HIP, or exports the variables itself.  INTEGRATION.md §3.
RECOMMENDED_ENV = {
    # kernel arguments placed in device memory: ROCm 7.2's default on MI355X, read when the HIP runtime starts.  Worth
    # 1.2 us per launch (tools/stepbench, profiles/r04/kernarg_placement.txt: 7.13 us with it, 8.32 without at 1 M
    # boards; 3.80 / 4.77 at 262 144)
    "HIP_FORCE_DEV_KERNARG": "1",
    # multi-process RCCL / tensor sharing on this pool needs dmabuf IPC
    "HSA_ENABLE_IPC_MODE_LEGACY": "0",
}


def recommended_env(apply=False):
    """The environment variables bench.py / the examples run with, as a dict.  apply=True sets those that the process
    has not set itself (os.environ.setdefault) — call it before the first HIP call (torch.cuda.*, lib()): the HIP
    runtime reads them once, when it starts."""
    if apply:
        for k, v in RECOMMENDED_ENV.items():
            os.environ.setdefault(k, v)
    return dict(RECOMMENDED_ENV)


def flag_shape(boards_per_lane=0, workgroup_size=0):
    """include/qttt.h QTTT_FLAG_SHAPE: the launch shape carried by one call's flags (0 = the library's choice)."""
    if boards_per_lane not in (0, 1, 2, 4) or workgroup_size not in (0, 256, 512, 1024):
        raise ValueError("boards per lane 0|1|2|4 and workgroup size 0|256|512|1024")
    return (boards_per_lane << 8) | ({0: 0, 256: 1, 512: 2, 1024: 3}[workgroup_size] << 12)


def step_launch_shape(n, flags=0, observe=False):
    """(boards per lane, workgroup size) a step call with these flags uses for a batch of n boards."""
    bpl, blk = ctypes.c_int(0), ctypes.c_int(0)
    check(lib().qttt_step_launch_shape(int(n), int(flags), int(bool(observe)), ctypes.byref(bpl), ctypes.byref(blk)),
          "qttt_step_launch_shape")
    return bpl.value, blk.value


class QtttNativeError(RuntimeError):
    pass


def retire_mailbox(wait=True):
    """include/qttt.h qttt_board_mailbox_retire: asks the resident wave behind the single-board façades (Board.make_move,
    Env.step) to leave now — before a device-wide synchronise, or before handing the GPU to something else.  A no-op when
    none is resident; the next Board / Env call launches it again."""
    check(lib().qttt_board_mailbox_retire(1 if wait else 0), "qttt_board_mailbox_retire")


def lib():
    """Loads the HIP library (once).  Raises QtttNativeError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise QtttNativeError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        if L.qttt_abi_version() != ABI_VERSION:
            raise QtttNativeError("libqttt_hip.so ABI %d != expected %d (stale build?)"
                                  % (L.qttt_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


_ARG_ERRORS = {-1: "null pointer", -2: "bad size / offset", -3: "misaligned actions"}


def check(rc, what):
    if rc != 0:
        if rc < 0:
            raise QtttNativeError("%s: argument error %d (%s)" % (what, rc, _ARG_ERRORS.get(rc, "?")))
        raise QtttNativeError("%s: hipError_t %d" % (what, rc))
