"""ctypes binding of libccsm (include/ccsm.h).  The HIP extension is mandatory: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCSM_LIB_PATH") or os.path.join(_HERE, "lib", "libccsm.so")   # override: kernel experiments

SEQ_LEN, HIDDEN, LAYERS, CLASSES = 21, 256, 3, 2
OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_CAPACITY = range(6)
H0_EXPLICIT, H0_ZERO, H0_DEVICE_RNG = 0, 1, 2
PRECISION_SPLIT3, PRECISION_SPLIT_MX, PRECISION_HYBRID, PRECISION_SPLIT_MXD = 3, 4, 5, 6      # ccsm_precision (include/ccsm.h); 0 = chosen by ccsm_create's probe

_FP = C.POINTER(C.c_float)


class Config(C.Structure):
    _fields_ = [("seq_len", C.c_int32), ("num_layers", C.c_int32), ("num_classes", C.c_int32), ("hidden_size", C.c_int32),
                ("is_npass", C.c_int32), ("is_sn", C.c_int32), ("is_map", C.c_int32), ("is_stds", C.c_int32),
                ("model_type", C.c_char_p), ("precision", C.c_int32)]


class Weights(C.Structure):
    _fields_ = [("embed_weight", C.c_void_p),
                ("weight_ih", (C.c_void_p * 2) * LAYERS), ("weight_hh", (C.c_void_p * 2) * LAYERS),
                ("bias_ih", (C.c_void_p * 2) * LAYERS), ("bias_hh", (C.c_void_p * 2) * LAYERS),
                ("att_wa", C.c_void_p), ("att_ua", C.c_void_p), ("att_va", C.c_void_p),
                ("fc1_weight", C.c_void_p), ("fc1_bias", C.c_void_p)]


class Strand(C.Structure):
    _fields_ = [("kmer", C.c_void_p), ("ipd", C.c_void_p), ("pw", C.c_void_p), ("npass", C.c_void_p),
                ("ipd_std", C.c_void_p), ("pw_std", C.c_void_p), ("sn", C.c_void_p), ("map", C.c_void_p)]   # is_stds / is_sn / is_map models only


class Batch(C.Structure):
    _fields_ = [("strand", Strand * 2), ("kmer_is_f32", C.c_int32), ("npass_per_base", C.c_int32)]


class H0(C.Structure):
    _fields_ = [("mode", C.c_int32), ("h0", C.c_void_p * 2), ("seed", C.c_uint64), ("offset", C.c_uint64),
                ("site_key", C.c_void_p), ("site_sub", C.c_void_p)]


class Reads(C.Structure):
    _fields_ = [("n_reads", C.c_int32), ("offset", C.c_void_p), ("length", C.c_void_p), ("seq", C.c_void_p),
                ("fi", C.c_void_p), ("ri", C.c_void_p), ("fp", C.c_void_p), ("rp", C.c_void_p), ("fn", C.c_void_p),
                ("rn", C.c_void_p), ("h0_key", C.c_void_p)]


class AggrWeights(C.Structure):
    _fields_ = [("weight_ih", C.c_void_p * 2), ("weight_hh", C.c_void_p * 2), ("bias_ih", C.c_void_p * 2),
                ("bias_hh", C.c_void_p * 2), ("att_wa", C.c_void_p), ("att_ua", C.c_void_p), ("att_va", C.c_void_p),
                ("fc1_weight", C.c_void_p), ("fc1_bias", C.c_void_p)]


class CcsmError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("libccsm status %d: %s" % (status, text))
        self.status = status


_lib = None

# every symbol include/ccsm.h declares (checked by tests/test_cabi_symbols.py without a GPU)
EXPORTS = ("ccsm_create", "ccsm_destroy", "ccsm_workspace_create", "ccsm_workspace_destroy", "ccsm_forward_host",
           "ccsm_submit_host", "ccsm_wait_host", "ccsm_forward_device", "ccsm_last_error", "ccsm_version",
           "ccsm_model_precision", "ccsm_model_probe_error", "ccsm_model_probe_error_hybrid", "ccsm_model_probe_error_of", "ccsm_model_probe_tail", "ccsm_model_probe_q999", "ccsm_model_probe_sites", "ccsm_model_quant_error",
           "ccsm_model_set_precision", "ccsm_workspace_force_split3", "ccsm_model_data_probe_add", "ccsm_model_data_probe_decide", "ccsm_model_data_probe_error", "ccsm_model_data_probe_q999",
           "ccsm_model_data_probe_sites", "ccsm_model_data_probe_verdict", "ccsm_workspace_bytes", "ccsm_workspace_set_timing", "ccsm_workspace_last_timing",
           "ccsm_selftest_mfma", "ccsm_debug_read", "ccsm_debug_rows_padded", "ccsm_debug_rows_capacity",
           "ccsm_group_add_device", "ccsm_group_run", "ccsm_group_pending", "ccsm_workspace_timing_mean",
           "ccsm_forward_reads_host", "ccsm_submit_reads_host", "ccsm_wait_reads_host", "ccsm_selftest_split_f8", "ccsm_selftest_split_mx",
           "ccsm_debug_fp8_e4m3", "ccsm_measure_mfma_ceiling",
           "ccsm_aggr_create", "ccsm_aggr_destroy", "ccsm_aggr_set_only_close", "ccsm_aggr_forward_host", "ccsm_aggr_forward_device")


def load():
    """Load libccsm.so (built in-tree by __graft_entry__.build()).  Raises if it is missing: the product has no
    other execution path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libccsm.so not built (%s): run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    try:  # share ONE HIP runtime with PyTorch when it is in the process (same SONAME libamdhip64.so.7)
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, ci = C.c_void_p, C.c_int
    lib.ccsm_create.argtypes = [C.POINTER(Config), C.POINTER(Weights), ci, C.POINTER(vp)]
    lib.ccsm_destroy.argtypes = [vp]
    lib.ccsm_destroy.restype = None
    lib.ccsm_workspace_create.argtypes = [vp, ci, C.POINTER(vp)]
    lib.ccsm_workspace_destroy.argtypes = [vp]
    lib.ccsm_workspace_destroy.restype = None
    lib.ccsm_forward_host.argtypes = [vp, vp, ci, C.POINTER(Batch), C.POINTER(H0), vp, vp, vp]
    lib.ccsm_submit_host.argtypes = [vp, vp, ci, C.POINTER(Batch), C.POINTER(H0), vp]
    lib.ccsm_wait_host.argtypes = [vp, vp, vp]
    lib.ccsm_forward_device.argtypes = [vp, vp, ci, C.POINTER(Batch), C.POINTER(H0), vp, vp, vp]
    lib.ccsm_last_error.restype = C.c_char_p
    lib.ccsm_version.restype = C.c_char_p
    lib.ccsm_model_precision.argtypes = [vp]
    lib.ccsm_model_probe_error.argtypes = [vp]
    lib.ccsm_model_probe_error.restype = C.c_float
    lib.ccsm_model_probe_error_hybrid.argtypes = [vp]
    lib.ccsm_model_probe_error_hybrid.restype = C.c_float
    lib.ccsm_model_probe_error_of.argtypes = [vp, C.c_int]
    lib.ccsm_model_probe_error_of.restype = C.c_float
    lib.ccsm_model_probe_tail.argtypes = [vp, C.c_int]
    lib.ccsm_model_probe_tail.restype = C.c_float
    lib.ccsm_model_probe_q999.argtypes = [vp]
    lib.ccsm_model_probe_q999.restype = C.c_float
    lib.ccsm_model_probe_sites.argtypes = [vp]
    lib.ccsm_model_probe_sites.restype = C.c_int
    lib.ccsm_model_quant_error.argtypes = [vp]
    lib.ccsm_model_quant_error.restype = C.c_float
    lib.ccsm_model_set_precision.argtypes = [vp, ci]
    lib.ccsm_workspace_force_split3.argtypes = [vp]
    lib.ccsm_model_data_probe_add.argtypes = [vp, vp, vp, ci]
    lib.ccsm_model_data_probe_decide.argtypes = [vp]
    lib.ccsm_model_data_probe_error.argtypes = [vp]
    lib.ccsm_model_data_probe_error.restype = C.c_float
    lib.ccsm_model_data_probe_q999.argtypes = [vp]
    lib.ccsm_model_data_probe_q999.restype = C.c_float
    lib.ccsm_model_data_probe_sites.argtypes = [vp]
    lib.ccsm_model_data_probe_verdict.argtypes = [vp]
    lib.ccsm_workspace_bytes.argtypes = [vp]
    lib.ccsm_workspace_bytes.restype = C.c_size_t
    lib.ccsm_workspace_set_timing.argtypes = [vp, ci]
    lib.ccsm_workspace_last_timing.argtypes = [vp, _FP]
    lib.ccsm_selftest_mfma.argtypes = [ci, _FP]
    lib.ccsm_selftest_split_f8.argtypes = [ci, _FP, _FP]
    lib.ccsm_selftest_split_mx.argtypes = [ci, ci, _FP, _FP, C.POINTER(C.c_int)]
    lib.ccsm_measure_mfma_ceiling.argtypes = [ci, ci, C.c_double, _FP, _FP]
    lib.ccsm_debug_read.argtypes = [vp, ci, vp, C.c_size_t]
    lib.ccsm_debug_rows_padded.argtypes = [ci]
    lib.ccsm_debug_fp8_e4m3.argtypes = [C.c_float]
    lib.ccsm_debug_rows_capacity.argtypes = [vp]
    lib.ccsm_group_add_device.argtypes = [vp, vp, ci, C.POINTER(Batch), C.POINTER(H0), vp, vp, vp]
    lib.ccsm_group_run.argtypes = [vp, vp, vp]
    lib.ccsm_group_pending.argtypes = [vp]
    lib.ccsm_forward_reads_host.argtypes = [vp, vp, C.POINTER(Reads), C.POINTER(H0), vp, vp, vp, vp, C.POINTER(C.c_int32), vp]
    lib.ccsm_submit_reads_host.argtypes = [vp, vp, C.POINTER(Reads), vp, C.POINTER(H0), vp]
    lib.ccsm_wait_reads_host.argtypes = [vp, vp, vp, vp, vp, C.POINTER(C.c_int32)]
    lib.ccsm_workspace_timing_mean.argtypes = [vp, _FP, C.POINTER(C.c_int)]
    lib.ccsm_aggr_create.argtypes = [C.POINTER(AggrWeights), ci, C.c_uint64, C.c_int64, C.POINTER(vp)]
    lib.ccsm_aggr_destroy.argtypes = [vp]
    lib.ccsm_aggr_destroy.restype = None
    lib.ccsm_aggr_set_only_close.argtypes = [vp, ci]
    lib.ccsm_aggr_forward_host.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, vp, vp]
    lib.ccsm_aggr_forward_device.argtypes = [vp, C.c_int64, vp, vp, C.c_int64, vp, vp]
    _lib = lib
    return lib


def check(status):
    if status != OK:
        raise CcsmError(status, load().ccsm_last_error().decode())
