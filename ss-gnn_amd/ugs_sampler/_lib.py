"""ctypes binding of libugs_mi355.so (C ABI: include/ugs_mi355.h).  Loading fails loudly: there is no fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
LIB_PATH = os.environ.get("UGS_MI355_LIB", os.path.join(_CSRC, "libugs_mi355.so"))

i64p = C.POINTER(C.c_int64)
vp = C.c_void_p


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"ugs_sampler: {LIB_PATH} is missing -- build the HIP library first "
            f"(python {os.path.join(_CSRC, 'build.py')}  or  python -c 'import __graft_entry__ as g; g.build()'). "
            "This sampler has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.ugs_last_error.restype = C.c_char_p
    L.ugs_version.restype = C.c_char_p
    sig = {
        "ugs_device_count": [C.POINTER(C.c_int)],
        "ugs_set_device": [C.c_int],
        "ugs_set_stream": [vp, C.c_int],
        "ugs_create_preproc": [vp, C.c_int64, C.c_int64, C.c_int64, C.c_int, i64p],
        "ugs_destroy_preproc": [C.c_int64],
        "ugs_has_graphlets": [C.c_int64, C.POINTER(C.c_int)],
        "ugs_get_preproc_info": [C.c_int64, C.POINTER(C.c_int), i64p, i64p, C.POINTER(C.c_double), C.POINTER(C.c_int)],
        "ugs_preproc_dump": [C.c_int64] + [vp] * 9,
        "ugs_sample_begin": [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.POINTER(vp), i64p],
        "ugs_sample_finish": [vp, vp, vp, vp, vp, C.c_int],
        "ugs_sample_batch_begin": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.POINTER(vp), i64p],
        "ugs_sample_batch_finish": [vp, vp, vp, vp, vp, vp, C.c_int],
        "ugs_job_cancel": [vp],
        "ugs_sample_batch_stream": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64,
                                    vp, vp, vp, vp, vp, i64p],
        "ugs_stream_stats": [i64p, i64p],
        "ugs_sample_stream": [C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, vp, vp, vp, vp, i64p],
        "ugs_apx_sample_batch": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_double, vp, i64p],
        "ugs_apx_gpu_sample_batch": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_uint64, C.c_double, vp, i64p, vp, vp, C.c_int64],
        "ugs_eps_sample_batch_begin": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_double,
                                       C.POINTER(vp), i64p],
        "ugs_eps_sample_batch_finish": [vp, vp, vp, vp, vp, vp, C.c_int],
        "ugs_cache_clear": [],
        "ugs_cache_stats": [i64p, i64p, i64p],
        "ugs_batch_pass_stats": [i64p, i64p],
        "ugs_plan_create_batch": [vp, C.c_int64, C.c_int64, vp, C.c_int64, C.c_int, C.POINTER(vp)],
        "ugs_plan_create_handle": [C.c_int64, C.POINTER(vp)],
        "ugs_plan_release": [vp],
        "ugs_plan_graph_roots": [vp, C.c_int64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)] + [vp] * 6,
        "ugs_plan_twin": [vp, C.c_int, C.POINTER(vp)],
        "ugs_plan_info": [vp, C.c_int, i64p, i64p, i64p, i64p, C.POINTER(C.c_int)],
        "ugs_plan_walk": [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, vp, vp, vp, i64p],
        "ugs_plan_fill": [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, vp, C.c_int64, vp],
        "ugs_plan_step": [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int64, C.c_int64, vp, vp, vp, vp, C.c_int64, vp],
        "ugs_plan_graph_create": [vp, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, vp, vp, vp, C.c_int64, vp, C.POINTER(vp)],
        "ugs_plan_graph_launch": [vp, C.c_int, vp],
        "ugs_plan_graph_destroy": [vp],
        "ugs_collate_layout": [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, i64p, i64p],
        "ugs_collate_unpack": [vp, C.c_int, i64p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, vp, vp, C.c_int64, vp, vp, vp, vp],
        "ugs_plan_set_walk_share": [vp, C.c_int],
        "ugs_plan_set_timing": [vp, C.c_int],
        "ugs_plan_get_timing": [vp, C.POINTER(C.c_double), i64p],
        "ugs_plan_last_launch": [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), i64p],
    }
    for name, argtypes in sig.items():
        fn = getattr(L, name)          # AttributeError here = the library does not export what the header declares
        fn.argtypes = argtypes
        fn.restype = C.c_int
    return L, sorted(list(sig) + ["ugs_last_error", "ugs_version"])


lib, EXPORTS = _load()

UGS_E_BAD_ARG = -4
UGS_E_CAPACITY = -9


def check(rc):
    """Maps a C status to the exception the reference raises for the same condition: pybind turns its
    TORCH_CHECK / std::runtime_error into RuntimeError (reference src/sampler.cpp:105,149,
    src/ugs_sampler_batch_extension.cpp:85-90,213-230)."""
    if rc != 0:
        msg = (lib.ugs_last_error() or b"").decode() or f"ugs error {rc}"
        raise RuntimeError(msg)
