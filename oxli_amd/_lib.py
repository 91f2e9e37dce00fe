"""Loads libkct_hip.so (the C ABI of include/kct.h) with ctypes and declares every entry point.

There is no fallback: if the library is missing this raises, and if no gfx950 device is present
``kct_create`` fails with KCT_ERR_NO_DEVICE.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KCT_LIB_PATH") or os.path.join(HERE, "csrc", "libkct_hip.so")  # override: A/B builds of the same ABI

KCT_OK = 0
KCT_ERR_WRONG_KSIZE = 1
KCT_ERR_INVALID_DNA = 2
KCT_ERR_BAD_KMER = 3
KCT_ERR_KSIZE_MISMATCH = 4
KCT_ERR_NOMEM = 5
KCT_ERR_HIP = 6
KCT_ERR_ARG = 7
KCT_ERR_NO_DEVICE = 8
KCT_ERR_BUSY = 9

u8, u64, sz, vp, cp, ci = C.c_uint8, C.c_uint64, C.c_size_t, C.c_void_p, C.c_char_p, C.c_int
u64p = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); mirrors include/kct.h one to one (tests/test_abi.py checks both ways)
SIGNATURES = {
    "kct_last_error": (cp, []),
    "kct_device_count": (ci, []),
    "kct_create": (ci, [u8, u64, ci, C.POINTER(vp)]),
    "kct_destroy": (None, [vp]),
    "kct_clear": (ci, [vp]),
    "kct_reserve": (ci, [vp, u64]),
    "kct_resize": (ci, [vp, u64]),
    "kct_hash_kmer": (ci, [vp, vp, sz, u64p]),
    "kct_hash_windows": (ci, [vp, vp, sz, vp, sz, u64p, u64p]),
    "kct_count_hash": (ci, [vp, u64, u64p]),
    "kct_count": (ci, [vp, vp, sz, u64p]),
    "kct_get": (ci, [vp, vp, sz, u64p]),
    "kct_get_hash": (ci, [vp, u64, u64p]),
    "kct_get_hash_array": (ci, [vp, vp, sz, vp]),
    "kct_set_hash": (ci, [vp, u64, u64]),
    "kct_consume": (ci, [vp, vp, sz, ci, u64p]),
    "kct_consume_will_defer": (ci, [vp, sz, ci]),
    "kct_consume_batch": (ci, [vp, vp, vp, sz, ci, u64p, u64p, u64p]),
    "kct_consume_device": (ci, [vp, vp, sz, u64, u64p]),
    "kct_batch_timeline": (ci, [vp, C.POINTER(C.c_double)]),
    "kct_consume_device_packed": (ci, [vp, vp, vp, sz, u64, u64p]),
    "kct_pack_stream_device": (ci, [vp, sz, vp, vp, vp]),
    "kct_set_packed_upload": (ci, [vp, ci]),
    "kct_consume_device_routed": (ci, [vp, vp, sz, u64, C.c_uint32, C.c_uint32, vp, u64, u64p, u64p]),
    "kct_superkmer_split_device": (ci, [vp, vp, sz, C.c_uint32, C.POINTER(vp), u64p, u64p, u64p]),
    "kct_superkmer_streams": (C.c_uint32, [vp]),
    "kct_debug_inject_fault": (ci, [vp, ci, u64]),
    "kct_consume_file": (ci, [vp, cp, ci, u64p, u64p, u64p]),
    "kct_inflater_name": (cp, []),
    "kct_len": (ci, [vp, u64p]),
    "kct_sum_counts": (ci, [vp, u64p]),
    "kct_consumed": (ci, [vp, u64p]),
    "kct_add_consumed": (ci, [vp, u64]),
    "kct_ksize": (u8, [vp]),
    "kct_capacity": (ci, [vp, u64p]),
    "kct_dump": (ci, [vp, vp, vp, sz, ci, u64p]),
    "kct_add": (ci, [vp, vp, u64p, u64p]),
    "kct_export_device": (ci, [vp, vp, vp, sz, u64p]),
    "kct_merge_device": (ci, [vp, vp, vp, sz, u64p, u64p]),
    "kct_merge_host": (ci, [vp, vp, vp, sz, u64p, u64p]),
    "kct_export_by_owner_device": (ci, [vp, C.c_uint32, vp, sz, vp, u64p]),
    "kct_merge_pairs_device": (ci, [vp, vp, sz, u64p, u64p]),
    "kct_sync": (ci, [vp]),
    "kct_release_scratch": (ci, [vp]),
    "kct_save": (ci, [vp, cp, cp]),
    "kct_load": (ci, [cp, ci, C.POINTER(vp)]),
    "kct_load_rest_json": (cp, []),
    "kct_set_deferred": (ci, [vp, ci]),
    "kct_count_stats": (ci, [vp, u64p, u64p, C.POINTER(C.c_double)]),
    "kct_digest": (ci, [vp, u64p, u64p, u64p]),
    "kct_histogram": (ci, [vp, vp, vp, sz, u64p]),
    "kct_retain_counts": (ci, [vp, u64, u64, u64p]),
    "kct_remove_hash": (ci, [vp, u64, u64p]),
    "kct_compare": (ci, [vp, vp, u64p, u64p]),
    "kct_set_op": (ci, [vp, vp, ci, vp, sz, u64p]),
    "kct_set_path": (ci, [vp, ci]),
    "kct_set_stream": (ci, [vp, vp]),
    "kct_get_stream": (vp, [vp]),
    "kct_profile_enable": (ci, [vp, ci]),
    "kct_profile_reset": (ci, [vp]),
    "kct_profile_read": (ci, [vp, ci, vp, sz, u64p, C.POINTER(C.c_double)]),
    # include/kct_synth.h (measurement infrastructure)
    "kct_synth_genome_device": (ci, [vp, u64, u64, vp]),
    "kct_synth_reads_device": (ci, [vp, vp, u64, u64, u64, C.c_uint32, u64, vp]),
    "kct_synth_reads_device_ex": (ci, [vp, vp, u64, u64, u64, C.c_uint32, u64, C.c_uint32, C.c_uint32, u64, u64, vp]),
}



class ExchangeOps(C.Structure):
    """``kct_exchange_ops`` of include/kct.h: the collective the early multi-GPU route asks its caller for."""
    ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_uint64)
    RELEASE = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    SIZES = C.CFUNCTYPE(C.c_int, C.c_void_p, u64p, C.c_uint32, u64p)
    START = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, u64p, u64p, C.c_void_p, u64p, u64p)
    WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p)
    _fields_ = [("user", C.c_void_p), ("alloc", ALLOC), ("release", RELEASE), ("exchange_sizes", SIZES), ("start", START), ("wait", WAIT)]


# include/kct_rccl.h (libkct_rccl.so: the exchange and the late route's merge over RCCL)
RCCL_LIB_PATH = os.path.join(HERE, "csrc", "libkct_rccl.so")
RCCL_ID_BYTES = 128
RCCL_SIGNATURES = {
    "kct_rccl_unique_id": (ci, [vp]),
    "kct_rccl_create": (ci, [vp, ci, ci, ci, C.POINTER(vp)]),
    "kct_rccl_ops": (vp, [vp]),
    "kct_rccl_destroy": (None, [vp]),
    "kct_rccl_last_error": (cp, []),
    "kct_rccl_stats": (None, [vp, u64p, u64p, C.POINTER(C.c_double)]),
    "kct_rccl_merge_across_ranks": (ci, [vp, vp, u64p]),
    "kct_rccl_merge_when_alone": (None, [vp, ci]),
    "kct_rccl_release_buffers": (None, [vp]),
    "kct_rccl_release_above": (None, [vp, u64]),
}

_lib = None
_rccl = None


def load_rccl():
    """ctypes handle of libkct_rccl.so (optional: built when the RCCL headers are there).  libkct_hip.so is loaded first, so that the
    helper's dependency on it resolves to the SAME copy; in a process that carries PyTorch, RCCL resolves to PyTorch's librccl.so.1."""
    global _rccl
    if _rccl is None:
        load()
        if not os.path.exists(RCCL_LIB_PATH):
            raise ImportError(f"{RCCL_LIB_PATH} is missing (built by `make -C oxli_amd/csrc` when /opt/rocm/include/rccl/rccl.h exists)")
        lib = C.CDLL(RCCL_LIB_PATH)
        for name, (res, args) in RCCL_SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _rccl = lib
    return _rccl


def load():
    """Returns the ctypes handle of libkct_hip.so; raises ImportError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  oxli_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)   # (global: libkct_rccl.so binds to this copy's kct_* symbols)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the library does not export what kct.h declares
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error():
    msg = load().kct_last_error()
    return msg.decode("utf-8", "replace") if msg else ""
