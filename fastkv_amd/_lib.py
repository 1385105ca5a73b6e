"""ctypes binding of fastkv_amd/lib/libfastkv_hip.so (C ABI: include/fastkv_hip.h).

The library is the product: there is NO CPU or torch fallback.  If it cannot be loaded the
import of the ops fails loudly (FastKVNativeError)."""
from __future__ import annotations

import ctypes
import os

from . import _build


# return codes of the C ABI (include/fastkv_hip.h)
FASTKV_OK, FASTKV_EINVAL, FASTKV_EWORKSPACE, FASTKV_ELAUNCH, FASTKV_EUNSUPPORTED, FASTKV_EABORTED, FASTKV_EOVERFLOW, FASTKV_EBOUNDS, FASTKV_EPLACEMENT = 0, -1, -2, -3, -4, -5, -6, -7, -8


class FastKVNativeError(RuntimeError):
    """`code` = the C ABI's return code (None when the library itself could not be loaded).  Callers may fall back to another
    path on FASTKV_EUNSUPPORTED only; FASTKV_EABORTED / FASTKV_ELAUNCH mean a launch of this process went wrong and must
    reach the user (fastkv_amd/cluster.py: DeferredCompression)."""

    def __init__(self, msg: str, code=None):
        super().__init__(msg)
        self.code = code


class Problem(ctypes.Structure):
    """struct fastkv_problem (include/fastkv_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("B", "H", "Hkv", "S", "D", "window", "kernel", "pooling", "capacity",
                                              "tsp_len", "order", "reserved")]


class SPWindow(ctypes.Structure):
    """struct fastkv_sp_window (include/fastkv_hip.h)."""
    _fields_ = [(n, ctypes.c_int32) for n in ("ncols", "pos0", "own_lo", "own_hi", "S_glob", "Sp")]


EXPORTS = ["fastkv_workspace_bytes", "fastkv_workspace_init", "fastkv_update_kv_f16", "fastkv_update_kv_strided_f16", "fastkv_score_f16", "fastkv_select_f16",
           "fastkv_select_workspace_bytes", "fastkv_compact_f16", "fastkv_compact_ranked_f16", "fastkv_gather_rows", "fastkv_tsp_propagate", "fastkv_head_sum_f16", "fastkv_sp_workspace_bytes", "fastkv_sp_logits_f16", "fastkv_sp_rowmax_f16", "fastkv_sp_rowsum_f16",
           "fastkv_sp_scores_f16", "fastkv_sp_pack_f16", "fastkv_sp_unpack_f16", "fastkv_sp_pick",
           "fastkv_sp_compact_f16", "fastkv_decode_workspace_bytes", "fastkv_decode_append_f16", "fastkv_decode_attention_f16",
           "fastkv_decode_rmsnorm_f16", "fastkv_decode_rope_f16", "fastkv_decode_silu_mul_f16", "fastkv_decode_greedy_f16", "fastkv_decode_rotary_f16", "fastkv_decode_gemv_f16", "fastkv_update_kv_ptrs_f16", "fastkv_fused_entries_f16", "fastkv_pool_f16", "fastkv_decode_step_attention_f16", "fastkv_debug_contract", "fastkv_debug_mfma16", "fastkv_debug_granule_areas", "fastkv_debug_occupy", "fastkv_debug_fused_placement", "fastkv_placement_violations", "fastkv_set_placement_policy", "fastkv_set_no_wait_mode", "fastkv_set_fused_rolling", "fastkv_no_wait_mode", "fastkv_last_status", "fastkv_profile_enable",
           "fastkv_profile_kernels", "fastkv_profile_kernel_name", "fastkv_profile_read", "fastkv_strerror", "fastkv_version"]

_lib = None


def load(build_if_missing: bool = True) -> ctypes.CDLL:
    """Load (building first if the sources are newer) the HIP library and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    try:
        if build_if_missing and _build.needs_build():
            path = _build.build()
        L = ctypes.CDLL(path)
    except Exception as e:   # noqa: BLE001 - any failure here means the native path is unusable
        raise FastKVNativeError(f"fastkv_amd: cannot load the HIP extension {path}: {e}") from e
    vp, i64, ci, sz = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32, ctypes.c_size_t
    i64p, pp = ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(Problem)
    L.fastkv_workspace_bytes.argtypes = [pp]
    L.fastkv_workspace_bytes.restype = sz
    L.fastkv_workspace_init.argtypes = [vp, sz, vp]
    L.fastkv_workspace_init.restype = ctypes.c_int
    L.fastkv_update_kv_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, vp, vp, vp, vp, vp, sz, vp]
    L.fastkv_update_kv_f16.restype = ci
    L.fastkv_update_kv_strided_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, i64p, vp, vp, i64p, vp, vp, vp, vp, sz, vp]
    L.fastkv_update_kv_strided_f16.restype = ci
    L.fastkv_score_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, vp, vp, sz, vp]
    L.fastkv_score_f16.restype = ci
    L.fastkv_select_f16.argtypes = [vp, i64, i64, i64, i64, ci, ci, vp, vp, sz, vp]
    L.fastkv_select_f16.restype = ci
    L.fastkv_head_sum_f16.argtypes = [vp, i64, i64, i64, vp, vp]
    L.fastkv_head_sum_f16.restype = ci
    L.fastkv_select_workspace_bytes.argtypes = [i64, i64, i64]
    L.fastkv_select_workspace_bytes.restype = sz
    L.fastkv_compact_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, vp, vp, vp]
    L.fastkv_compact_f16.restype = ci
    L.fastkv_compact_ranked_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, vp, i64, vp, vp, vp, vp, sz, vp]
    L.fastkv_compact_ranked_f16.restype = ci
    L.fastkv_gather_rows.argtypes = [vp, i64, i64, vp, i64, i64, i64, i64, i64, vp, vp]
    L.fastkv_gather_rows.restype = ci
    L.fastkv_tsp_propagate.argtypes = [vp, i64, i64, vp, i64, vp, i64, i64, i64, i64, i64, vp, vp, vp]
    L.fastkv_tsp_propagate.restype = ci
    wp = ctypes.POINTER(SPWindow)
    L.fastkv_sp_workspace_bytes.argtypes = [pp]
    L.fastkv_sp_workspace_bytes.restype = sz
    L.fastkv_sp_logits_f16.argtypes = [pp, vp, i64p, vp, i64p, vp, i64, i64, vp, sz, vp]
    L.fastkv_sp_logits_f16.restype = ci
    L.fastkv_sp_rowmax_f16.argtypes = [pp, vp, wp, vp, vp]
    L.fastkv_sp_rowmax_f16.restype = ci
    L.fastkv_sp_rowsum_f16.argtypes = [pp, vp, wp, vp, vp, vp]
    L.fastkv_sp_rowsum_f16.restype = ci
    L.fastkv_sp_scores_f16.argtypes = [pp, vp, wp, vp, vp, vp, vp, vp, sz, vp]
    L.fastkv_sp_scores_f16.restype = ci
    L.fastkv_sp_pack_f16.argtypes = [vp, i64, i64, vp, i64, i64, i64, vp, vp]
    L.fastkv_sp_pack_f16.restype = ci
    L.fastkv_sp_unpack_f16.argtypes = [vp, i64, i64, ci, i64, i64, vp, i64, vp]
    L.fastkv_sp_unpack_f16.restype = ci
    L.fastkv_sp_pick.argtypes = [vp, i64, i64, i64, i64, vp, i64, i64, i64, vp, vp]
    L.fastkv_sp_pick.restype = ci
    L.fastkv_sp_compact_f16.argtypes = [ci, ci, ci, ci, ci, ci, vp, i64p, vp, i64p, vp, i64, ci, vp, vp, vp]
    L.fastkv_sp_compact_f16.restype = ci
    L.fastkv_decode_workspace_bytes.argtypes = [ci, ci, ci, ci]
    L.fastkv_decode_workspace_bytes.restype = sz
    L.fastkv_decode_append_f16.argtypes = [ci, ci, ci, vp, i64p, vp, i64p, vp, vp, i64p, ci, vp, vp]
    L.fastkv_decode_append_f16.restype = ci
    L.fastkv_decode_attention_f16.argtypes = [ci, ci, ci, ci, vp, i64p, vp, vp, i64p, ci, vp, ctypes.c_float, ci, vp, vp, sz, vp]
    L.fastkv_decode_attention_f16.restype = ci
    L.fastkv_decode_rmsnorm_f16.argtypes = [vp, i64, i64, ci, vp, ctypes.c_float, vp, vp]
    L.fastkv_decode_rmsnorm_f16.restype = ci
    L.fastkv_decode_rope_f16.argtypes = [ci, ci, ci, ci, vp, i64p, vp, i64p, vp, vp, i64, vp]
    L.fastkv_decode_rope_f16.restype = ci
    L.fastkv_decode_silu_mul_f16.argtypes = [vp, vp, i64, vp, vp]
    L.fastkv_decode_silu_mul_f16.restype = ci
    L.fastkv_decode_greedy_f16.argtypes = [ci, ci, vp, i64, vp, vp, vp, vp, vp, ci, vp]
    L.fastkv_decode_greedy_f16.restype = ci
    L.fastkv_decode_rotary_f16.argtypes = [ci, ci, vp, vp, ctypes.c_float, vp, vp, vp]
    L.fastkv_decode_rotary_f16.restype = ci
    L.fastkv_decode_step_attention_f16.argtypes = [ci, ci, ci, ci, vp, i64p, vp, i64p, vp, i64p, vp, vp, i64, vp, vp, i64p, ci, vp,
                                                   ctypes.c_float, ci, vp, vp, sz, vp, vp]
    L.fastkv_decode_step_attention_f16.restype = ci
    L.fastkv_update_kv_ptrs_f16.argtypes = [ctypes.POINTER(Problem), vp, i64p, vp, i64p, vp, i64p, vp, vp, i64p, vp, vp, vp, sz, vp]
    L.fastkv_update_kv_ptrs_f16.restype = ci
    L.fastkv_fused_entries_f16.argtypes = [ctypes.POINTER(Problem)]
    L.fastkv_fused_entries_f16.restype = ci
    L.fastkv_pool_f16.argtypes = [vp, i64, i64, i64, ci, ci, vp, i64, vp]
    L.fastkv_pool_f16.restype = ci
    L.fastkv_decode_gemv_f16.argtypes = [ci, ci, vp, i64, vp, ctypes.c_float, ci, ctypes.POINTER(vp), ctypes.POINTER(ci), ci, vp, i64, vp, i64, vp]
    L.fastkv_decode_gemv_f16.restype = ci
    L.fastkv_debug_occupy.argtypes = [ci, ci, i64, vp]
    L.fastkv_debug_occupy.restype = ci
    L.fastkv_debug_fused_placement.argtypes = [ci, vp, ctypes.c_size_t]
    L.fastkv_debug_fused_placement.restype = ci
    L.fastkv_placement_violations.argtypes = [ci]
    L.fastkv_placement_violations.restype = ci
    L.fastkv_set_placement_policy.argtypes = [ci]
    L.fastkv_set_placement_policy.restype = ci
    L.fastkv_set_no_wait_mode.argtypes = [ci]
    L.fastkv_set_no_wait_mode.restype = ci
    L.fastkv_set_fused_rolling.argtypes = [ci]
    L.fastkv_set_fused_rolling.restype = ci
    L.fastkv_no_wait_mode.argtypes = []
    L.fastkv_no_wait_mode.restype = ci
    L.fastkv_last_status.argtypes = []
    L.fastkv_last_status.restype = ci
    L.fastkv_debug_mfma16.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    L.fastkv_debug_mfma16.restype = ci
    L.fastkv_debug_contract.argtypes = [ci, vp, vp, vp, vp, ci, vp]
    L.fastkv_debug_contract.restype = ci
    L.fastkv_profile_enable.argtypes = [ci]
    L.fastkv_profile_enable.restype = None
    L.fastkv_profile_kernels.restype = ci
    L.fastkv_profile_kernel_name.argtypes = [ci]
    L.fastkv_profile_kernel_name.restype = ctypes.c_char_p
    L.fastkv_profile_read.argtypes = [vp, vp]
    L.fastkv_profile_read.restype = ci
    L.fastkv_strerror.argtypes = [ci]
    L.fastkv_strerror.restype = ctypes.c_char_p
    L.fastkv_version.argtypes = []
    L.fastkv_version.restype = ctypes.c_char_p
    _lib = L
    return L


_violations_seen = 0


_no_wait_warned = False


def check(rc: int, what: str) -> None:
    if rc != 0:
        if rc == FASTKV_EPLACEMENT:
            # the violations behind this report have now been reported: they stay in the library's running total, but
            # raise_if_aborted does not warn about them a second time
            global _violations_seen
            _violations_seen = load().fastkv_placement_violations(0)
        if rc in (FASTKV_EABORTED, FASTKV_EPLACEMENT):
            # under the fail-safe policy the report that is being raised has switched the process to the no-wait kernels for good
            # (a given-up wait, a placement violation): say so once -- nothing else tells the caller why every later call is slower
            global _no_wait_warned
            if not _no_wait_warned and os.environ.get("FASTKV_FUSED", "1") != "0" and load().fastkv_no_wait_mode():
                _no_wait_warned = True
                import warnings
                warnings.warn(f"fastkv_amd.{what}: the library has switched this process to the no-wait kernels (staged scoring, "
                              "wait-free selection; as FASTKV_FUSED=0) with the report that follows -- ops.set_no_wait_mode(False) "
                              "switches back", RuntimeWarning, stacklevel=3)
        msg = load().fastkv_strerror(rc).decode()
        raise FastKVNativeError(f"fastkv_amd.{what}: {msg} (code {rc})", code=rc)



def raise_if_aborted(what: str = "last_status") -> None:
    """Host-only read of the library's asynchronous-error word (fastkv_last_status): raises FastKVNativeError(FASTKV_EABORTED /
    FASTKV_EOVERFLOW) if a launch of this process that has ALREADY RUN gave up a bounded in-kernel wait or overran a decode slab
    since the last report.  Costs a load of pinned memory: no synchronisation.  Call it behind a synchronisation point to learn
    about everything enqueued before it (benchmark/prefill.py, benchmark/e2e.py); the wiring also calls it un-synchronised at the
    end of every prefill, which reports what has completed by then."""
    L = load()
    check(L.fastkv_last_status(), what)
    global _violations_seen
    total = L.fastkv_placement_violations(0)                # (not reset: bench.py and tests read the running count)
    n, _violations_seen = total - _violations_seen if total >= _violations_seen else total, total
    if n:
        # only reached under the "count" policy (FASTKV_STRICT_PLACEMENT=0 / ops.set_placement_policy("count")): the default policy has
        # raised FASTKV_EPLACEMENT above and switched the process to the no-wait kernels
        import warnings
        warnings.warn(f"fastkv_amd.{what}: {n} workgroup(s) of fused scoring launches shared a compute unit with another head's workgroups "
                      "(foreign kernels on the GPU, or a launch that was not resident all at once); set FASTKV_FUSED=0 on a shared GPU "
                      "(include/fastkv_hip.h: fastkv_placement_violations)", RuntimeWarning, stacklevel=2)
