"""ctypes binding of libbbmap_amd.so (the C ABI declared in include/bbmap_amd.h).

There is no fallback: if the shared library is missing, loading raises; if no gfx950 device is
present, bbmsa_create() returns BBMAP_E_NODEVICE and the wrappers raise.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libbbmap_amd.so")

# every symbol include/bbmap_amd.h declares
EXPORTS = [
    "bbmap_last_error", "bbmap_abi_version",
    "bbmsa_create", "bbmsa_destroy", "bbmsa_align_batch_device", "bbmsa_align_batch",
    "bbmsa_fill_packed", "bbmsa_fill_submit", "bbmsa_fill_collect", "bbmsa_legacy_stats", "bbmsa_last_kernel_ms", "bbmsa_last_kernel_ms3", "bbmsa_last_counts", "bbmsa_align_gapped_batch_device", "bbmsa_align_gapped_batch", "bbmsa_align_batch_device_indirect",
    "bbmsa_align_gapped_batch_device_indirect",
    "bbband_create", "bbband_destroy", "bbband_align_batch_device", "bbband_align_batch",
    "bbband_align_quadruple_batch", "bbband_align_quadruple_progressive_batch", "bbband_align_double_batch",
    "bbidx_create", "bbidx_destroy", "bbidx_find_batch_device", "bbidx_find_batch", "bbidx_find_batch_device_rc", "bbidx_last_stats", "bbidx_set_kernel", "bbidx_set_max_read_len", "bbidx_build", "bbidx_get_params", "bbidx_export_block",
    "bbmap_default_config", "bbmap_create", "bbmap_destroy", "bbmap_map_batch_device", "bbmap_map_batch", "bbmap_get_output", "bbmap_get_overflow_output", "bbmap_last_stats", "bbmap_pack_sites_device",
    "bbmap_copy_to_host", "bbidx_get_chrom_table",
    "bbpipe_revcomp_device", "bbpipe_quick_rescue_device",
    "bbidx_build_profile", "bbkeys_default_config", "bbkeys_make", "bbkeys_make_batch", "bbmap_default_config_profile",
    "bbmap_get_final", "bbmap_set_average_pair_dist", "bbmap_final_batch_device",
]


class bbmsa_job(C.Structure):
    _fields_ = [("read_off", C.c_int64), ("ref_off", C.c_int64),
                ("read_len", C.c_int32), ("ref_len", C.c_int32),
                ("refStartLoc", C.c_int32), ("refEndLoc", C.c_int32),
                ("minScore", C.c_int32), ("flags", C.c_int32)]


class bbmsa_result(C.Structure):
    _fields_ = [("result", C.c_int32 * 5), ("status", C.c_int32), ("iterations", C.c_int64),
                ("score", C.c_int32 * 8), ("score_len", C.c_int32), ("match_len", C.c_int32),
                ("fill_kind", C.c_int32), ("columns", C.c_int32)]


class bbmsa_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("maxRows", C.c_int32), ("maxColumns", C.c_int32),
                ("bandwidth", C.c_int32), ("bandwidthRatio", C.c_float), ("reserved", C.c_int32 * 3)]


class bbband_job(C.Structure):
    _fields_ = [("query_off", C.c_int64), ("ref_off", C.c_int64), ("query_len", C.c_int32), ("ref_len", C.c_int32),
                ("qstart", C.c_int32), ("rstart", C.c_int32), ("maxEdits", C.c_int32), ("flags", C.c_int32)]


class bbband_result(C.Structure):
    _fields_ = [("edits", C.c_int32), ("lastQueryLoc", C.c_int32), ("lastRefLoc", C.c_int32), ("lastRow", C.c_int32),
                ("lastEdits", C.c_int32), ("lastOffset", C.c_int32), ("status", C.c_int32), ("reserved", C.c_int32)]


class bbband_config(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("semantics", C.c_int32), ("reserved", C.c_int32)]


assert C.sizeof(bbmsa_job) == 40 and C.sizeof(bbmsa_result) == 80
assert C.sizeof(bbband_job) == 40 and C.sizeof(bbband_result) == 32

_lib = None


class BBMapAmdError(RuntimeError):
    pass


def load():
    """Loads the HIP library; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    so_path = os.environ.get("BBMAP_AMD_SO") or SO_PATH          # (experiments: a variant build, scripts/build_variant.py)
    if not os.path.exists(so_path):
        raise BBMapAmdError(
            "libbbmap_amd.so is missing (%s). Build it with `python -m bbmap_amd.build`; "
            "there is no CPU fallback." % so_path)
    L = C.CDLL(so_path)
    L.bbmap_last_error.restype = C.c_char_p
    L.bbmap_abi_version.restype = C.c_int
    L.bbmsa_create.argtypes = [C.POINTER(bbmsa_config), C.POINTER(C.c_void_p)]
    L.bbmsa_create.restype = C.c_int
    L.bbmsa_destroy.argtypes = [C.c_void_p]
    L.bbmsa_destroy.restype = None
    L.bbmsa_align_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_batch_device.restype = C.c_int
    L.bbmsa_align_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64,
                                    C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_batch.restype = C.c_int
    L.bbmsa_align_gapped_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_gapped_batch_device.restype = C.c_int
    L.bbmsa_align_batch_device_indirect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_batch_device_indirect.restype = C.c_int
    L.bbmsa_align_gapped_batch_device_indirect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_gapped_batch_device_indirect.restype = C.c_int
    L.bbmsa_align_gapped_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                           C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
    L.bbmsa_align_gapped_batch.restype = C.c_int
    L.bbmsa_fill_packed.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    L.bbmsa_fill_packed.restype = C.c_int
    L.bbmsa_fill_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p]
    L.bbmsa_fill_submit.restype = C.c_int
    L.bbmsa_fill_collect.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bbmsa_fill_collect.restype = C.c_int
    L.bbmsa_legacy_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.bbmsa_legacy_stats.restype = C.c_int
    L.bbmsa_last_kernel_ms3.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.bbmsa_last_kernel_ms3.restype = C.c_int
    L.bbmsa_last_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    L.bbmsa_last_counts.restype = C.c_int
    L.bbmsa_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.bbmsa_last_kernel_ms.restype = C.c_int
    L.bbband_create.argtypes = [C.POINTER(bbband_config), C.POINTER(C.c_void_p)]
    L.bbband_create.restype = C.c_int
    L.bbband_destroy.argtypes = [C.c_void_p]
    L.bbband_destroy.restype = None
    L.bbband_align_batch_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bbband_align_batch_device.restype = C.c_int
    L.bbband_align_batch.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    L.bbband_align_batch.restype = C.c_int
    L.bbpipe_revcomp_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    L.bbpipe_revcomp_device.restype = C.c_int
    L.bbpipe_quick_rescue_device.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
    L.bbpipe_quick_rescue_device.restype = C.c_int
    _lib = L
    return L


def check(rc, what):
    if rc != 0:
        raise BBMapAmdError("%s failed (%d): %s" % (what, rc, load().bbmap_last_error().decode()))
