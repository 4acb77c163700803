"""ctypes binding of libttsk.so (the C ABI declared in include/ttsk.h).

The product path has no CPU fallback: if the HIP library is missing or no
device is present every compute entry point raises.
"""
import ctypes
import os
import re
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_int16, c_int64,
                    c_size_t, c_uint64, c_void_p)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTSK_LIB") or os.path.join(_HERE, "libttsk.so")      # TTSK_LIB: another build of the library (A/B runs)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ttsk.h")

TTSK_ERR_ARG = -2
TTSK_ERR_UNSUPPORTED = -3
NUM_STREAMS = 8


class TtskError(RuntimeError):
    """A HIP / RCCL level failure inside libttsk."""


class TtskUnsupported(TtskError):
    """The fused kernel does not cover this shape (caller composes from ttsk_gemm)."""


class GemmDesc(Structure):
    _fields_ = [(n, c_int64) for n in
                ("batch", "M", "N", "Ko", "Ki", "a_b", "a_m", "a_ko", "a_ki",
                 "b_b", "b_ko", "b_ki", "b_n", "c_b", "c_m", "c_n")] + [
        ("alpha", c_double), ("accumulate", c_int), ("split_k", c_int)]


_lib = None


def declared_symbols():
    """Every function name include/ttsk.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(ttsk_[a-z0-9_]+)\s*\(", text)))


def _bind(lib):
    P, I, S = c_void_p, c_int, c_size_t
    sig = {
        "ttsk_init": [I], "ttsk_shutdown": [], "ttsk_device_info": [c_char_p, S, POINTER(I), POINTER(S)],
        "ttsk_malloc": [POINTER(P), S], "ttsk_free": [P], "ttsk_memset": [P, I, S, I],
        "ttsk_h2d": [P, P, S, I], "ttsk_d2h": [P, P, S, I], "ttsk_d2d": [P, P, S, I],
        "ttsk_sync": [I], "ttsk_stream_wait": [I, I],
        "ttsk_timer_start": [I], "ttsk_timer_stop": [I, POINTER(c_float)],
        "ttsk_graph_begin": [I], "ttsk_graph_end": [I, POINTER(P)], "ttsk_graph_launch": [P, I],
        "ttsk_graph_free": [P],
        "ttsk_gemm": [POINTER(GemmDesc), P, P, P, P, I],
        "ttsk_copy_strided": [P, P, I, POINTER(c_int64), POINTER(c_int64), POINTER(c_int64), I],
        "ttsk_axpby": [P, P, c_double, c_double, S, I],
        "ttsk_sum_slices": [P, P, I, S, S, I, I],
        "ttsk_tt_sketch": [I] + [POINTER(c_int64)] * 8 + [POINTER(P)] * 3 + [P, I, I],
        "ttsk_tt_sketch_batch": [I, I] + [POINTER(c_int64)] * 8 + [POINTER(P)] * 3 + [P, c_int64, I, I],
        "ttsk_tt_sketch_sum": [I, I] + [POINTER(c_int64)] * 8 + [POINTER(P)] * 3 + [P, I, I],
        "ttsk_chain_step": [I, I, I, I, I, I, POINTER(P), c_int64, POINTER(P), c_int64, c_int64, c_int64, c_int64, P,
                            POINTER(P), POINTER(P), I],
        "ttsk_chain_step_wide": [I, I, I, I, I, I, POINTER(P), c_int64, POINTER(P), c_int64, c_int64, c_int64, c_int64, P,
                                 POINTER(P), POINTER(P), I],
        "ttsk_chain_step_sum": [I, I, I, I, I, I, POINTER(P), c_int64, POINTER(P), c_int64, c_int64, c_int64, c_int64, P,
                                P, c_int64, c_int64, c_int64, POINTER(P), I],
        "ttsk_prof_enable": [I],
        "ttsk_mfma_f64_peak_probe": [POINTER(c_double)],
        "ttsk_ndtri_rate_probe": [POINTER(c_double)],
        "ttsk_prof_read": [I, POINTER(c_int64), POINTER(c_double), POINTER(c_double)],
        "ttsk_prof_kernel_name": [I, c_char_p, S],
        "ttsk_hash_u64": [P, S],
        "ttsk_inds_to_rand_double": [P, P, I, S, I, I, c_uint64, P],
        "ttsk_inds_to_normal": [P, P, I, S, I, I, c_uint64, P],
        "ttsk_inds_to_sparse_sign": [P, P, I, S, I, I, I, I, c_uint64, P],
        "ttsk_sparse_normal_dev": [P, c_int64, POINTER(I), POINTER(c_uint64), I, S, I, I, c_uint64, P, I],
        "ttsk_sparse_sign_dev": [P, c_int64, POINTER(I), POINTER(c_uint64), I, S, I, I, I, I, c_uint64, P, I],
        "ttsk_fill_normal": [P, S, c_uint64, c_double, I],
        "ttsk_fill_normal_many": [I, POINTER(P), POINTER(S), POINTER(c_uint64), POINTER(c_double), I],
        "ttsk_sparse_ttdrm_step": [P, c_int64, P, c_int64, c_int64, P, S, P, I],
        "ttsk_sparse_densedrm_gather": [P, c_int64, c_int64, P, c_int64, POINTER(I), POINTER(c_int64), I, S, P, I],
        "ttsk_sparse_psi": [P, P, P, S, P, c_int64, P, c_int64, c_int64, P, I],
        "ttsk_sparse_normal_table": [POINTER(c_uint64), I, I, I, c_uint64, P, I],
        "ttsk_sparse_sign_table": [POINTER(c_uint64), I, I, I, I, I, c_uint64, P, I],
        "ttsk_sparse_flat_mult": [POINTER(c_uint64), I, POINTER(c_uint64)],
        "ttsk_sparse_mode_order": [P, c_int64, S, POINTER(I), POINTER(c_uint64), I, I, c_int64, P, I],
        "ttsk_sparse_mode_stream": [P, c_int64, P, S, POINTER(I), POINTER(c_uint64), I, POINTER(I), POINTER(c_uint64), I, I,
                                    P, P, P, P, P, I],
        "ttsk_sparse_mode_stream_u32": [P, c_int64, P, S, POINTER(I), POINTER(c_uint64), I, POINTER(I), POINTER(c_uint64), I, I,
                                    P, P, P, P, P, I],
        "ttsk_sparse_gauss_pass": [P, P, P, P, S, c_int64, P, P, P, I, P, P, I],
        "ttsk_sparse_gauss_pass_u32": [P, P, P, P, S, c_int64, P, P, P, I, P, P, I],
        "ttsk_sparse_sort_mode": [P, S, c_int64, P, I],
        "ttsk_pinv": [P, c_int64, c_int64, c_double, P, POINTER(I), I],
        "ttsk_pinv_begin": [P, c_int64, c_int64, c_double, P, I],
        "ttsk_pinv_end": [P, c_int64, c_int64, c_double, P, POINTER(I), I],
        "ttsk_triu": [P, c_int64, c_int64, I],
        "ttsk_svd_small": [P, c_int64, c_int64, P, P, P, I],
        "ttsk_qr_thin": [P, c_int64, c_int64, I],
        "ttsk_orth_step": [P, c_int64, c_int64, P, c_int64, P, I],
        "ttsk_deferred_status": [I, POINTER(I)],
        "ttsk_pinv_batch_deferred": [I, POINTER(P), c_int64, c_int64, POINTER(P), I],
        "ttsk_pinv_batch": [I, POINTER(P), c_int64, c_int64, POINTER(P), I],
        "ttsk_dense_first_pass": [P, c_int64, c_int64, c_int64, P, c_int64, P, c_int64, P, P, I],
        "ttsk_dense_left_pass": [P, c_int64, c_int64, c_int64, c_int64, c_int64, I, P, P, P, P, P, P, P, P, I],
        "ttsk_orth_step_pinv": [P, c_int64, c_int64, P, c_int64, P, I],
        "ttsk_tt_orth_sketch": [I] + [POINTER(c_int64)] * 4 + [POINTER(P)] * 5 + [I],
        "ttsk_tt_orth_sketch_batch": [I, I] + [POINTER(c_int64)] * 4 + [POINTER(P)] * 5 + [P, I],
        "ttsk_tt_assemble": [I] + [POINTER(c_int64)] * 3 + [POINTER(P)] * 4 + [I, I],
        "ttsk_comm_unique_id": [P], "ttsk_comm_init": [P, I, I],
        "ttsk_comm_allreduce_sum": [P, S, I], "ttsk_comm_reduce_sum": [P, S, I, I],
        "ttsk_comm_allgather": [P, P, S, I], "ttsk_comm_allreduce_max": [P, S, I],
        "ttsk_comm_destroy": [],
    }
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    if missing:
        raise TtskError(f"{LIB_PATH} does not export {missing}")
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_int
    lib.ttsk_last_error.restype = c_char_p
    lib.ttsk_last_error.argtypes = []
    lib.ttsk_tt_sketch_size.restype = c_int64
    lib.ttsk_tt_sketch_size.argtypes = [I] + [POINTER(c_int64)] * 5


def lib():
    """Loads libttsk.so (no device needed) and checks the exported symbol set."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TtskError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                "tt_sketch_amd has no CPU fallback")
        handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        _bind(handle)
        _lib = handle
    return _lib


def check(rc):
    if rc == 0:
        return
    msg = lib().ttsk_last_error().decode(errors="replace")
    if rc == TTSK_ERR_ARG:
        raise ValueError(msg)
    if rc == TTSK_ERR_UNSUPPORTED:
        raise TtskUnsupported(msg)
    raise TtskError(msg)


_sync_epoch = 0
# Which library streams have had work queued since they were last drained, and how often each has
# been drained: device.py tags a released buffer with this so that it is handed out again only to a
# user on the same stream, or after the streams that may still be touching it have been synchronised.
_stream_gen = [0] * NUM_STREAMS
_dirty = set()
# entry points whose LAST argument is the library stream their work is queued on
_STREAM_LAST = frozenset((
    "ttsk_memset", "ttsk_d2d", "ttsk_gemm", "ttsk_copy_strided", "ttsk_axpby", "ttsk_sum_slices",
    "ttsk_tt_sketch", "ttsk_tt_sketch_batch", "ttsk_tt_sketch_sum", "ttsk_chain_step", "ttsk_chain_step_wide", "ttsk_sparse_normal_dev", "ttsk_sparse_sign_dev",
    "ttsk_fill_normal", "ttsk_fill_normal_many", "ttsk_sparse_ttdrm_step", "ttsk_sparse_densedrm_gather", "ttsk_sparse_psi",
    "ttsk_sparse_sort_mode", "ttsk_sparse_normal_table", "ttsk_sparse_sign_table", "ttsk_sparse_mode_order", "ttsk_sparse_mode_stream", "ttsk_sparse_mode_stream_u32", "ttsk_sparse_gauss_pass", "ttsk_sparse_gauss_pass_u32", "ttsk_pinv", "ttsk_pinv_begin", "ttsk_pinv_end", "ttsk_triu", "ttsk_svd_small",
    "ttsk_qr_thin", "ttsk_orth_step", "ttsk_orth_step_pinv", "ttsk_pinv_batch_deferred", "ttsk_pinv_batch", "ttsk_dense_first_pass", "ttsk_tt_orth_sketch", "ttsk_tt_orth_sketch_batch", "ttsk_tt_assemble", "ttsk_comm_allreduce_sum", "ttsk_comm_reduce_sum", "ttsk_comm_allgather", "ttsk_comm_allreduce_max", "ttsk_graph_launch", "ttsk_timer_start"))
_BLOCKING = frozenset(("ttsk_h2d", "ttsk_d2h"))          # return only after their stream has drained
_TWO_STREAMS = frozenset(("ttsk_tt_sketch", "ttsk_tt_sketch_batch", "ttsk_tt_sketch_sum", "ttsk_tt_orth_sketch"))   # fork a helper on stream + 1, joined back
_ALL_STREAMS = frozenset(("ttsk_tt_assemble", "ttsk_tt_orth_sketch_batch"))           # fork every other library stream, all joined back


def sync_epoch() -> int:
    """Number of device-wide synchronisations so far."""
    return _sync_epoch


def dirty_snapshot() -> dict:
    """{stream: drain generation} of every stream with work queued since its last drain.  A helper stream whose work
    was joined back into its caller's stream (the one-call TT sketches) counts as that stream: whatever is queued there
    later is ordered behind the helper's work too, so a buffer released now may go straight to a user on the caller's
    stream (every temporary of a loop of one-call sketches used to miss the pool and cost a hipMalloc)."""
    out = {}
    for s in _dirty:
        t = _joined_into.get(s, s)
        out[t] = _stream_gen[t]
    return out


def drained_since(stream: int, gen: int) -> bool:
    return _stream_gen[stream] > gen


# helper stream -> the stream its work was joined back into (the one-call TT sketches fork stream + 1 and join it
# before they return): draining the caller's stream then covers the helper too, unless something else was queued on
# the helper directly in the meantime
_joined_into = {}


def _mark(name, args):
    if name in _STREAM_LAST:
        s = int(args[-1])
        if 0 <= s < NUM_STREAMS:
            _dirty.add(s)
            _joined_into.pop(s, None)          # direct work on s: it is no longer merely a joined helper
            helpers = ((s + 1) % NUM_STREAMS,) if name in _TWO_STREAMS else \
                (tuple(h for h in range(NUM_STREAMS) if h != s) if name in _ALL_STREAMS else ())
            for h in helpers:
                if h not in _dirty or _joined_into.get(h) == s:
                    _joined_into[h] = s
                _dirty.add(h)


def _drained(s):
    _stream_gen[s] += 1
    _dirty.discard(s)
    _joined_into.pop(s, None)
    for h, into in list(_joined_into.items()):
        if into == s:                          # everything the helper had in flight was ordered before this drain
            _stream_gen[h] += 1
            _dirty.discard(h)
            del _joined_into[h]


def call(name, *args):
    global _sync_epoch
    _mark(name, args)                      # before the call: a failing call may have queued part of its work
    check(getattr(lib(), name)(*args))
    if name == "ttsk_sync" and args:
        s = int(args[0])
        if s < 0:
            _sync_epoch += 1
            for i in range(NUM_STREAMS):
                _stream_gen[i] += 1
            _dirty.clear()
            _joined_into.clear()
        elif s < NUM_STREAMS:
            _drained(s)
    elif name in _BLOCKING:
        s = int(args[-1])
        if 0 <= s < NUM_STREAMS:
            _drained(s)
    elif name == "ttsk_deferred_status":   # waits for its stream
        s = int(args[0])
        if 0 <= s < NUM_STREAMS:
            _drained(s)
