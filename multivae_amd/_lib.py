"""ctypes binding of libmvk.so (the C ABI declared in include/mvk.h).

There is NO CPU fallback: if the HIP library is missing or a call returns an error code the product path
raises.  PyTorch only provides device memory (`tensor.data_ptr()`) and the HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))


# MVK_* variables that are NOT experiment switches (read directly, with or without MVK_TUNE)
_ALWAYS_READ = {"MVK_TUNE", "MVK_LIB_PATH", "MVK_DEFER_MB", "MVK_SYNC_DEBUG", "MVK_TRAINER_ALLOW_CPU", "MVK_CPU_THREADS",
                "MVK_BENCH_SAME_GPU", "MVK_FORCE_DIST", "MVK_DIST_BACKEND", "MVK_ADAM_ZERO", "MVK_RCCL", "MVK_BENCH_CHILD", "MVK_OVERLAP", "MVK_GRAPH_ADAM"}


def _warn_ignored_switches():
    if os.environ.get("MVK_TUNE") == "1":
        return
    ignored = sorted(k for k in os.environ if k.startswith("MVK_") and k not in _ALWAYS_READ)
    if ignored:
        import warnings

        warnings.warn(f"{', '.join(ignored)} set without MVK_TUNE=1: experiment switches are ignored unless MVK_TUNE=1 "
                      "(this process runs the shipped configuration)", RuntimeWarning, stacklevel=3)


_warn_ignored_switches()


def tune(name, default):
    """Experiment switches (the MVK_* A/B knobs of DESIGN.md section 9) are honoured only under MVK_TUNE=1: a process without it
    runs ONE configuration, the shipped one (and is told so once, at import, if it sets such a switch)."""
    return os.environ.get(name, default) if os.environ.get("MVK_TUNE") == "1" else default


LIB_PATH = os.environ.get("MVK_LIB_PATH") or os.path.join(_HERE, "libmvk.so")  # MVK_LIB_PATH: A/B builds (tools/)

MVK_OK = 0
DIST = {"normal": 0, "laplace": 1, "bernoulli": 2, "categorical": 3}
ACT = {None: 0, "none": 0, "relu": 1, "sigmoid": 2, "leaky_relu_0.2": 3}
FAMILY = {"normal": 0, "laplace_with_softmax": 1, "normal_with_softplus": 2}  # 2: std kernels only (density = normal)
MAX_MODALITIES = 8

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_i64 = C.c_int64


class ReconDesc(C.Structure):
    _fields_ = [("recon", _p), ("x", _p), ("mask", _p), ("rows", _p), ("drecon", _p), ("rowcoef", _p),
                ("D", _i64), ("dist", C.c_int32), ("scale", _f), ("rescale", _f), ("coef", _f),
                ("n_classes", C.c_int32)]


class PackDesc(C.Structure):  # mvk_pack_desc
    _fields_ = [("Wref", _p), ("Wdown", _p), ("Wup", _p), ("Cv", C.c_int32), ("Cu", C.c_int32),
                ("ld_down", C.c_int32), ("col_off", C.c_int32), ("kind", C.c_int32), ("Fdown", _p), ("Fup", _p),
                ("amax", _p)]


class CopyDesc(C.Structure):  # mvk_copy_desc
    _fields_ = [("dst", _p), ("src", _p), ("bytes", _i64)]


class SeedDesc(C.Structure):  # mvk_seed_desc
    _fields_ = [("buf", _p), ("n", _i64), ("coef", _f), ("fill", C.c_int32)]


class TermDesc(C.Structure):
    _fields_ = [("v", _p), ("mask", _p), ("n", _i64), ("period", _i64), ("coef", _f), ("lossw", _f), ("gfill", _p)]


# name -> argtypes (restype is always int); mirrors include/mvk.h one for one
PROTOTYPES = {
    "mvk_version": [],
    "mvk_mopoe_posterior_fwd": [_p, _p, _i, _p, _i, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mvk_mopoe_posterior_bwd": [_p, _p, _i, _p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p],
    "mvk_mvtcae_posterior_fwd": [_p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "mvk_mvtcae_posterior_bwd": [_p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p],
    "mvk_jmvae_posterior_fwd": [_p, _p, _p, _p, _i, _p, _i, _i, _i, _p, _p, _p, _p],
    "mvk_jmvae_posterior_bwd": [_p, _p, _p, _p, _i, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mvk_recon_nll_fwd": [C.POINTER(ReconDesc), _i, _i, _i, _p],
    "mvk_recon_nll_bwd": [C.POINTER(ReconDesc), _i, _i, _i, _p],
    "mvk_reduce_terms": [C.POINTER(TermDesc), _i, _f, _p, _p, _p],
    "mvk_reduce_terms_ws": [C.POINTER(TermDesc), _i, _f, _p, _p, _p, _i64, _p],
    "mvk_scale_by_device_scalar": [_p, _i64, _p, _p],
    "mvk_linear_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _i64, _p],
    "mvk_linear_bwd_data": [_p, _p, _p, _i, _i, _i, _p, _i, _p, _i, _i, _p, _p, _i64, _p],
    "mvk_linear_bwd_weight": [_p, _p, _p, _p, _i, _i, _i, _p, _i, _p, _i64, _p],
    "mvk_act_bwd_colsum": [_p, _p, _i, _i, _i, _p, _p, _p, _i64, _p],
    "mvk_colsum_acc": [_p, _p, _i, _p, _i, _i, _p, _i64, _p],
    "mvk_nchw_channel_sum_acc": [_p, _p, _i, _p, _i, _i, _i, _p, _i64, _p],
    "mvk_act_bwd": [_p, _p, _i64, _i, _p],
    "mvk_act_bwd_scaled": [_p, _f, _p, _i, _p, _i64, _p],
    "mvk_gemm": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _i, _p, _i, _p, _i, _p, _i64, _p],
    "mvk_pack_conv4s2_weight": [_p, _i, _i, _p, _i, _i, _p, _p],
    "mvk_conv4s2_down": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p, _p, _i64, _i, _p, _p],
    "mvk_conv4s2_up": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i64, _i, _p, _p],
    "mvk_pack_weights": [C.POINTER(PackDesc), _i, _p],
    "mvk_loss_backward_seed": [C.POINTER(SeedDesc), _i, _p, _p],
    "mvk_f32_to_bf3": [_p, _i64, _p, _p],
    "mvk_bf3_to_f32": [_p, _i64, _p, _p],
    "mvk_conv4s2_small_up_fwd_nll": [_p, _p, _p, _p, _i, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_up_fwd_nll_w": [_p, _p, _p, _p, _i, _f, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_up_fwd_nll_s": [_p, _p, _p, _p, _i, _f, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "mvk_conv4s2_small_up_fwd_s": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p],
    "mvk_conv4s2_small_up_fwd_nll_sy": [_p, _p, _p, _p, _i, _f, _f, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "mvk_conv4s2_small_up_bwd_pre_s": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _p, _p, _p],
    "mvk_conv4s2_small_up_bwd_pre": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p],
    "mvk_conv3x3": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _i64, _p],
    "mvk_conv3x3_y": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _i64, _p],
    "mvk_conv3x3_res": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _f, _p, _i64, _p],
    "mvk_conv3x3_wgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _i64, _p],
    "mvk_conv3x3_f": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _f, _p, _i, _f, _p, _i64, _p],
    "mvk_conv3x3_wgrad_f": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _i64, _p],
    "mvk_conv3x3_s": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _f, _p, _i, _f, _p, _p, _p, _p, _i64, _p],
    "mvk_conv3x3_s2": [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p, _p, _i64, _p],
    "mvk_conv3x3_s_part": [_p, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _i, _f, _p, _p, _p, _p],
    "mvk_amax": [_p, _i64, _p, _p],
    "mvk_conv4s2_wgrad_s": [_p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _i64, _p],
    "mvk_gemm_smallk_amax": [_p, _p, _p, _i, _i, _i, _i, _p, _i, _i, _p, _p],
    "mvk_conv4s2_small_up_bwd_pre_y": [_p, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p, _p],
    "mvk_conv4s2_down_s": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i64, _p, _p],
    "mvk_conv4s2_up_s": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _p, _p, _p, _p, _i64, _p, _p],
    "mvk_conv3x3_wgrad_s": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _p, _p, _i64, _p],
    "mvk_avgpool3s2_fwd": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_avgpool3s2_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_upsample2_fwd": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_upsample2_bwd": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_axpby": [_p, _f, _p, _f, _i64, _i, _p, _p],
    "mvk_transpose_act": [_p, _p, _i, _i, _i, _i, _p, _i, _p],
    "mvk_probe_mfma_bf16": [_p, _i, _i, _p],
    "mvk_probe_stream_copy": [_p, _p, _i64, _p],
    "mvk_device_rng": [_p, _i64, _p, _i, _f, _f, _p],
    "mvk_conv4s2_wgrad": [_p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i64, _p],
    "mvk_conv4s2_wgrad_pair": [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _i64, _p],
    "mvk_conv4s2_up_nchw_small": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_up_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_down_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_down_fwd_wref": [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p],
    "mvk_conv4s2_small_up_bwd": [_p, _p, _i, _p, _i, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _i, _i, _p],
    "mvk_pack_unflatten_weight": [_p, _i, _i, _p, _p],
    "mvk_unflatten_wgrad": [_p, _p, _p, _i, _i, _i, _p, _i64, _p],
    "mvk_flatten_wgrad": [_p, _p, _p, _i, _i, _i, _p, _i64, _p],
    "mvk_nchw_to_nhwc": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_nhwc_to_nchw": [_p, _p, _i, _i, _i, _i, _p],
    "mvk_adam_step": [_p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _d, _p],
    "mvk_heads_fwd": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i64, _i64, _p],
    "mvk_heads_bwd": [_p, _i, _p, _p, _p, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i64, _p],
    "mvk_defer_begin": [_p, _i64, _p, _i64],
    "mvk_defer_flush": [_p],
    "mvk_defer_end": [_p],
    "mvk_adam_step_amsgrad": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _d, _p],
    "mvk_adam_step_fused": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _d, _i, _p],
    "mvk_copy_batch": [C.POINTER(CopyDesc), _i, _p],
    "mvk_adam_prepare": [_p, _p, _p],
    "mvk_adam_step_pub": [_p, _p, _p, _p, _p, _i64, _d, _d, _d, _d, _d, _i, _d, _i, _p, _p],
    "mvk_adam_identity": [_p, _p],
    "mvk_adam_step_dev": [_p, _p, _p, _p, _p, _i64, _p, _i, _p],
    "mvk_mmvae_std_fwd": [_p, _i, _i, _i, _p, _p],
    "mvk_mmvae_std_bwd": [_p, _p, _p, _i, _i, _i, _p, _p],
    "mvk_mmvae_latent_fwd": [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _p, _p],
    "mvk_mmvae_objective_fwd": [_p, _p, _p, _p, _i, _i, _i, _i, _p, _p, _p, _p, _p, _f, _p],
    "mvk_mmvae_latent_bwd": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _i, _f, _p],
    "mvk_mmvaeplus_cross_latent_fwd": [_p, _p, _p, _i64, _i, _i, _i, _p, _p],
    "mvk_mmvaeplus_cross_latent_bwd": [_p, _p, _i64, _i, _i, _i, _p, _p, _p],
    "mvk_mvae_posterior_fwd": [_p, _p, _p, _i, _p, _i, _p, _i, _i, _p, _p, _p, _p, _p],
    "mvk_mvae_posterior_bwd": [_p, _p, _p, _i, _p, _i, _p, _p, _i, _i, _p, _p, _p, _p],
    "mvk_gauss_sample_kl_fwd": [_p, _p, _p, _i, _i, _i, _p, _p, _p],
    "mvk_gauss_sample_kl_bwd": [_p, _p, _p, _p, _p, _i, _i, _i, _p, _p, _p],
    "mvk_iwae_sample": [_p, _p, _p, _i, _i, _i, _i, _p, _p],
    "mvk_iwae_logw": [_p, _p, _i, _p, _p, _i, _p, _p, _i, _i, _i, _i, _p, _p],
    "mvk_iwae_reduce": [_p, _i, _i, _i, _p, _p],
    "mvk_poe_fwd": [_p, _p, _i, _i64, _f, _i, _p, _p, _p],
    "mvk_poe_bwd": [_p, _p, _i, _i64, _f, _i, _p, _p, _p, _p, _p],
    "mvk_kl_gauss_fwd": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i, _p, _p],
    "mvk_kl_gauss_bwd": [_p, _i64, _p, _i64, _p, _i64, _p, _i64, _i64, _i, _p, _p, _p, _p, _p, _p],
    "mvk_logprob_fwd": [_p, _p, _i64, _i64, _i, _f, _i, _f, _p, _p],
    "mvk_logprob_bwd": [_p, _p, _i64, _i64, _i, _f, _i, _f, _p, _p, _p],
    "mvk_comm_unique_id": [_p],
    "mvk_comm_init": [C.POINTER(C.c_void_p), _i, _i, _p],
    "mvk_comm_destroy": [_p],
    "mvk_allreduce_avg": [_p, _i64, _i, _p, _p],
    "mvk_comm_size": [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "mvk_allreduce_avg_ranges": [_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), _i, _i64, _p, _p],
    "mvk_event_create": [C.POINTER(C.c_void_p)],
    "mvk_event_destroy": [_p],
    "mvk_event_record": [_p, _i, _p],
    "mvk_stream_wait_event": [_p, _p],
    "mvk_dense16_pack": [_p, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "mvk_dense16_first": [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p],
    "mvk_dense16_fwd_nll": [_p, _p, _p, _p, _p, _p, _p, _p, _i, _p, _f, _f, _p, _p, _p, _p, _p, _i, _i, _i, _p],
    "mvk_dense16_bwd_data": [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i64, _i, _i, _i, _p],
    "mvk_dense16_wgrad": [_p, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _i64, _i, _i, _i, _p],
    "mvk_dense16_unsplit": [_p, _p, _p, _p, _f, _i, _i, _i, _p, _p],
}

_lib = None


class MvkError(RuntimeError):
    pass


def load(path=None):
    """Load libmvk.so and set every prototype.  Raises if the library or any declared symbol is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise MvkError(f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(path)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.mvk_conv4s2_small_up_supported.argtypes = [_i, _i, _i, _i]
    lib.mvk_conv4s2_small_up_supported.restype = C.c_int
    lib.mvk_conv4s2_small_up_nll_supported.argtypes = [_i, _i, _i, _i]
    lib.mvk_conv4s2_small_up_nll_supported.restype = C.c_int
    lib.mvk_conv3x3_fused_ok.argtypes = [_i, _i, _i, _i, _i]
    lib.mvk_conv3x3_fused_ok.restype = C.c_int
    lib.mvk_conv3x3_scaled_ok.argtypes = [_i, _i, _i, _i, _i]
    lib.mvk_conv3x3_scaled_ok.restype = C.c_int
    lib.mvk_conv3x3_wgrad_scaled_ok.argtypes = [_i, _i, _i, _i, _i]
    lib.mvk_conv3x3_wgrad_scaled_ok.restype = C.c_int
    lib.mvk_conv4s2_scaled_ok.argtypes = [_i, _i, _i, _i, _i]
    lib.mvk_conv4s2_scaled_ok.restype = C.c_int
    lib.mvk_conv4s2_wgrad_scaled_ok.argtypes = [_i, _i, _i, _i, _i]
    lib.mvk_conv4s2_wgrad_scaled_ok.restype = C.c_int
    lib.mvk_defer_wanted.argtypes = []
    lib.mvk_defer_wanted.restype = C.c_int64
    lib.mvk_defer_pending.argtypes = []
    lib.mvk_defer_pending.restype = C.c_int
    lib.mvk_prof_enable.argtypes = [_p, _i, _p, _p]
    lib.mvk_prof_enable.restype = C.c_int
    lib.mvk_prof_calibrate.argtypes = [_p, _i, _p]
    lib.mvk_prof_calibrate.restype = C.c_int
    lib.mvk_prof_count.argtypes = []
    lib.mvk_prof_count.restype = C.c_int
    lib.mvk_prof_clock_khz.argtypes = []
    lib.mvk_prof_clock_khz.restype = C.c_int
    lib.mvk_imgconv_frag_bytes.argtypes = [_i, _i]
    lib.mvk_imgconv_frag_bytes.restype = C.c_int64
    lib.mvk_comm_id_bytes.argtypes = []
    lib.mvk_comm_id_bytes.restype = C.c_int
    lib.mvk_comm_available.argtypes = []
    lib.mvk_comm_available.restype = C.c_int
    lib.mvk_dense16_ok.argtypes = [_i, _i, _i]
    lib.mvk_dense16_ok.restype = C.c_int
    lib.mvk_dense16_fwd_nll_rows.argtypes = [_i]
    lib.mvk_dense16_fwd_nll_rows.restype = C.c_int
    lib.mvk_dense16_colsum_rows.argtypes = [_i]
    lib.mvk_dense16_colsum_rows.restype = C.c_int
    lib.mvk_splitk_workspace_floats.argtypes = [_i, _i, _i]
    lib.mvk_splitk_workspace_floats.restype = C.c_int64
    _lib = lib
    return lib


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def ptr_array(tensors):
    """HOST array of device pointers (for the `const float* const*` arguments)."""
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def check(rc, what):
    if rc != MVK_OK:
        raise MvkError(f"{what} failed with code {rc}")


_SYNC_DEBUG = os.environ.get("MVK_SYNC_DEBUG", "")  # debugging aid: "1" = device sync after every launch, or a
# comma list of entry-point names to sync after


# GEMM-shaped entry points -> algorithmic FLOP of one call (2 x multiply-adds) from its positional arguments; used by
# bench.py (COUNT_FLOPS = [0.0] switches the count on for one eager step).  Everything else on the path is
# elementwise / reduction work (a few FLOP per byte) and is not counted.
GEMM_FLOPS = {
    "mvk_linear_fwd": lambda a: 2.0 * a[4] * a[5] * a[6],
    "mvk_linear_bwd_data": lambda a: 2.0 * a[3] * a[4] * a[5],
    "mvk_linear_bwd_weight": lambda a: 2.0 * a[4] * a[5] * a[6],
    "mvk_gemm": lambda a: 2.0 * a[3] * a[4] * a[5],
    "mvk_conv4s2_down": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_up": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_wgrad": lambda a: 2.0 * a[3] * a[4] * a[5] * 16 * a[6] * a[7],
    "mvk_conv4s2_wgrad_pair": lambda a: 2.0 * a[14] * 16 * (a[3] * a[4] * a[5] * a[6] + a[10] * a[11] * a[12] * a[13]),
    "mvk_conv3x3": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_y": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_res": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_wgrad": lambda a: 2.0 * a[3] * a[4] * a[5] * 9 * a[6] * a[7],
    "mvk_conv3x3_f": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_wgrad_f": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_s": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv4s2_down_s": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_wgrad_s": lambda a: 2.0 * a[3] * a[4] * a[5] * 16 * a[6] * a[7],
    "mvk_conv4s2_up_s": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv3x3_wgrad_s": lambda a: 2.0 * a[4] * a[5] * a[6] * 9 * a[7] * a[8],
    "mvk_conv3x3_s2": lambda a: 2.0 * a[5] * a[6] * a[7] * 9 * a[8] * a[9],
    "mvk_conv3x3_s_part": lambda a: 2.0 * a[6] * a[7] * a[8] * 9 * a[9] * a[10],
    "mvk_conv4s2_up_nchw_small": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_small_up_fwd": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_small_down_fwd": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_small_down_fwd_wref": lambda a: 2.0 * a[4] * a[5] * a[6] * 16 * a[7] * a[8],
    "mvk_conv4s2_small_up_bwd": lambda a: 2.0 * 2.0 * a[12] * a[13] * a[14] * 16 * a[15] * a[16],  # data + weight
    "mvk_conv4s2_small_up_fwd_nll": lambda a: 2.0 * a[8] * a[9] * a[10] * 16 * a[11] * a[12],
    "mvk_conv4s2_small_up_fwd_nll_w": lambda a: 2.0 * a[9] * a[10] * a[11] * 16 * a[12] * a[13],
    "mvk_conv4s2_small_up_fwd_nll_s": lambda a: 2.0 * a[9] * a[10] * a[11] * 16 * a[12] * a[13],
    "mvk_conv4s2_small_up_bwd_pre": lambda a: 2.0 * 2.0 * a[11] * a[12] * a[13] * 16 * a[14] * a[15],  # data + weight
    "mvk_conv4s2_small_up_bwd_pre_y": lambda a: 2.0 * 2.0 * a[11] * a[12] * a[13] * 16 * a[14] * a[15],
    "mvk_conv4s2_small_up_fwd_nll_sy": lambda a: 2.0 * a[9] * a[10] * a[11] * 16 * a[12] * a[13],
    "mvk_conv4s2_small_up_bwd_pre_s": lambda a: 2.0 * 2.0 * a[11] * a[12] * a[13] * 16 * a[14] * a[15],
    "mvk_gemm_smallk_amax": lambda a: 2.0 * a[3] * a[4] * a[5],
    "mvk_unflatten_wgrad": lambda a: 2.0 * a[3] * a[4] * 16 * a[5],
    "mvk_flatten_wgrad": lambda a: 2.0 * a[3] * 16 * a[4] * a[5],
    "mvk_heads_fwd": lambda a: 2.0 * (2 if a[4] and a[4].value else 1) * a[7] * a[8] * a[9],
    "mvk_heads_bwd": lambda a: 4.0 * a[15] * a[16] * a[17] * (2 if a[3] else 1),
    "mvk_dense16_first": lambda a: 2.0 * a[7] * a[8] * a[9],
    "mvk_dense16_fwd_nll": lambda a: 2.0 * a[17] * a[18] * a[19],
    "mvk_dense16_bwd_data": lambda a: 2.0 * a[11] * a[12] * a[13],
    "mvk_dense16_wgrad": lambda a: 2.0 * a[12] * a[13] * a[14],
}
COUNT_FLOPS = None


_HIP = None


def new_stream(device):
    """A dedicated HIP stream wrapped as a torch stream.  `torch.cuda.Stream()` hands out the 32 streams of a per-device pool
    round-robin: in a long-lived process two `Stream` objects created far apart are the SAME hipStream_t, and which of the
    package's cached side streams alias torch's capture stream (or each other) depends on how many streams the process made
    before — a captured step whose fork / join edges fold onto one stream crashed hipGraphLaunch
    (hip::Graph::UpdateStreams) for one particular test order.  Streams made here are outside that pool."""
    global _HIP
    if _HIP is None:
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        _HIP = C.CDLL(cand if os.path.exists(cand) else "libamdhip64.so")
        _HIP.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _HIP.hipStreamCreateWithFlags.restype = C.c_int
    h = C.c_void_p()
    with torch.cuda.device(device):
        rc = _HIP.hipStreamCreateWithFlags(C.byref(h), 1)  # hipStreamNonBlocking
    if rc != 0 or not h.value:
        raise MvkError(f"hipStreamCreateWithFlags failed ({rc})")
    return torch.cuda.ExternalStream(h.value, device=device)


def call(name, *args):
    lib = load()
    check(getattr(lib, name)(*args), name)
    if COUNT_FLOPS is not None and name in GEMM_FLOPS:
        COUNT_FLOPS[0] += GEMM_FLOPS[name](args)
    if _SYNC_DEBUG and (_SYNC_DEBUG == "1" or name in _SYNC_DEBUG.split(",")):
        torch.cuda.synchronize()


def require_gpu_tensor(t, name="tensor"):
    if not t.is_cuda:
        raise MvkError(f"{name} must live on the GPU: multivae_amd has no CPU compute path")
    if t.dtype != torch.float32:
        raise MvkError(f"{name} must be float32, got {t.dtype}")
