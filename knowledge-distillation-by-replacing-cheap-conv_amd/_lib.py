"""ctypes binding of libkdcc_hip.so (include/kdcc.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails this module raises.  Build with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C knowledge-distillation-by-replacing-cheap-conv_amd/csrc`.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KDCC_LIB=tuning loads the diagnostics build (`make -C csrc TUNING=1`: timing ablations that alter results, in-kernel
# timestamps) for tools/; everything else -- tests, bench, trainers -- runs the default library, which compiles them out.
# (KDCC_LIB=/path/to/other.so: A/B against another build of the same ABI, tools/bench_*.py.)
_sel = os.environ.get("KDCC_LIB", "")
LIB_PATH = (os.path.join(_HERE, "libkdcc_hip_tuning.so") if _sel == "tuning" else _sel if _sel.endswith(".so")
            else os.path.join(_HERE, "libkdcc_hip.so"))


def build_tuning():
    """Compile the diagnostics build next to the default one (tools/ call this before importing the ops)."""
    import subprocess
    subprocess.check_call(["make", "-s", "-j", str(min(8, os.cpu_count() or 1)), "-C", os.path.join(_HERE, "csrc"), "TUNING=1"])

KD_F32, KD_BF16 = 0, 1
KD_PACK_FWD, KD_PACK_DGRAD = 0, 1

c_int, c_i64, c_f, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class ConvDesc(C.Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "N", "H", "W", "Cin", "Ho", "Wo", "Cout", "kh", "kw", "stride", "pad",
                                     "dil", "ldx")]


class ConvEpilogue(C.Structure):
    _fields_ = [("res_pre", c_vp), ("ld_res_pre", c_int),
                ("mask", c_vp), ("ld_mask", c_int), ("mask_scale", c_vp),
                ("res_post", c_vp), ("ld_res_post", c_int),
                ("out_raw", c_vp), ("ld_raw", c_int), ("raw_f32", c_int),
                ("out_act", c_vp), ("ld_act", c_int), ("act_scale", c_vp), ("act_shift", c_vp), ("act_relu", c_int),
                ("bn_sums", c_vp),
                ("cls_w", c_vp), ("cls_out", c_vp), ("ld_cls", c_int), ("ncls", c_int)]


class DwDesc(C.Structure):
    _fields_ = [(n, c_int) for n in ("dtype", "N", "H", "W", "C", "k", "pad", "dil", "ldx", "ldy")]


class RadamTensor(C.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("exp_avg", c_vp), ("exp_avg_sq", c_vp), ("n", c_i64), ("step", c_int),
                ("lr", c_f), ("beta1", c_f), ("beta2", c_f), ("eps", c_f), ("weight_decay", c_f)]


class DwEpilogue(C.Structure):
    _fields_ = [("res_pre", c_vp), ("ld_res_pre", c_int), ("mask", c_vp), ("ld_mask", c_int), ("mask_scale", c_vp),
                ("res_post", c_vp), ("ld_res_post", c_int)]


class DConvDesc(C.Structure):
    _fields_ = [(n, c_int) for n in ("N", "C", "H", "W", "K", "kh", "kw", "stride", "pad", "dil", "groups")]


class View3(C.Structure):
    _fields_ = [("ptr", c_vp), ("dtype", c_int), ("sN", c_i64), ("sC", c_i64), ("sP", c_i64)]


_P = C.POINTER
_SIGS = {
    "kd_version": (c_int, []),
    "kd_last_error": (C.c_char_p, []),
    "kd_conv2d_fwd": (c_int, [_P(ConvDesc), c_vp, c_vp, _P(ConvEpilogue), c_vp]),
    "kd_conv1x1_dual_supported": (c_int, [_P(ConvDesc), c_int, c_int, _P(ConvEpilogue)]),
    "kd_conv1x1_dual_fwd": (c_int, [_P(ConvDesc), c_vp, c_vp, c_int, c_int, c_vp, _P(ConvEpilogue), c_vp]),
    "kd_conv_set_persist_cus": (c_int, [c_int]),
    "kd_conv2d_bn_sums_rows": (c_int, [_P(ConvDesc), _P(ConvEpilogue)]),
    "kd_conv2d_cls_supported": (c_int, [_P(ConvDesc), _P(ConvEpilogue)]),
    "kd_bn_sums_finish_workspace": (c_sz, [c_int, c_int]),
    "kd_bn_sums_finish": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "kd_pack_conv_weight": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_pw_wgrad_workspace": (c_sz, [c_int, c_int, c_int]),
    "kd_pw_wgrad": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_conv2d_wgrad_workspace": (c_sz, [_P(ConvDesc)]),
    "kd_conv2d_wgrad": (c_int, [_P(ConvDesc), c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_stem_wgrad_workspace": (c_sz, [c_int, c_int, c_int]),
    "kd_stem_wgrad": (c_int, [c_int, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "kd_maxpool3x3s2_bwd_workspace": (c_sz, [c_int, c_int, c_int, c_int]),
    "kd_maxpool3x3s2_bwd": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "kd_upsample_bilinear_ac_bwd_workspace": (c_sz, [c_int] * 6),
    "kd_upsample_bilinear_ac_bwd": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp,
                                            c_sz, c_vp]),
    "kd_zero_insert": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_relu_bn_bwd": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_i64, c_int, c_vp]),
    "kd_channel_sums_workspace": (c_sz, [c_int, c_i64, c_int]),
    "kd_channel_sums": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "kd_bn_eval_param_grads": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_vp]),
    "kd_broadcast_add": (c_int, [c_int, c_vp, c_vp, c_int, c_int, c_i64, c_int, c_f, c_int, c_vp]),
    "kd_conv2d_direct_fwd": (c_int, [_P(DConvDesc), c_vp, c_vp, c_vp, c_vp, c_vp]),
    "kd_conv2d_direct_dgrad": (c_int, [_P(DConvDesc), c_vp, c_vp, c_vp, c_vp]),
    "kd_conv2d_direct_wgrad": (c_int, [_P(DConvDesc), c_vp, c_vp, c_vp, c_vp, c_int, c_vp]),
    "kd_bn2d_fwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f, c_f, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_bn2d_bwd": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_pack_dw_weight": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "kd_dwconv_fwd": (c_int, [_P(DwDesc), c_vp, c_vp, c_vp, _P(DwEpilogue), c_vp, c_vp]),
    "kd_dwconv_fwd_sum": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_vp, c_vp]),
    "kd_dwconv_fwd_fanout": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_vp, c_vp]),
    "kd_dwconv_wgrad_workspace": (c_sz, [_P(DwDesc)]),
    "kd_dwconv_wgrad": (c_int, [_P(DwDesc), c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_dwconv_wgrad_multi_workspace": (c_sz, [_P(DwDesc), c_int]),
    "kd_dwconv_wgrad_multi": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_lattice_rows": (c_i64, [c_int, c_int, c_int, c_int]),
    "kd_dwconv_lattice_ok": (c_int, [_P(DwDesc), c_int]),
    "kd_lattice_rows_move": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_int, c_vp, c_int, c_int, c_vp]),
    "kd_dwconv_fwd_fanout_lattice": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_vp, c_vp]),
    "kd_dwconv_fwd_sum_lattice": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_vp, c_vp]),
    "kd_dwconv_wgrad_multi_lattice": (c_int, [_P(DwDesc), c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_stem_conv": (c_int, [c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "kd_scale_by_device_scalar": (c_int, [c_vp, c_int, c_i64, c_vp, c_vp]),
    "kd_pointwise_small": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_int, c_vp]),
    "kd_conv3x3_small": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_stem_conv_pool": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
    "kd_maxpool3x3s2": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
    "kd_upsample_bilinear_ac": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_upsample_bilinear": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_gated_conv": (c_int, [c_int, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_i64, c_int, c_vp]),
    "kd_edge_attention": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kd_edge_aspp": (c_int, [c_int, c_vp, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp]),
    "kd_canny_workspace": (c_sz, [c_int, c_int, c_int]),
    "kd_canny": (c_int, [c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "kd_canny_continue": (c_int, [c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "kd_aspp_image_pool_workspace": (c_sz, [c_int, c_int, c_int]),
    "kd_aspp_image_pool": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int,
                                   c_vp, c_sz, c_vp]),
    "kd_aspp_image_pool_sums": (c_int, [c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_sz, c_vp]),
    "kd_bn_fold": (c_int, [c_vp, c_vp, c_vp, c_vp, c_f, c_vp, c_vp, c_int, c_vp]),
    "kd_copy_cast": (c_int, [c_vp, c_int, c_i64, c_i64, c_i64, c_vp, c_int, c_i64, c_i64, c_i64, c_int, c_int, c_i64, c_vp]),
    "kd_loss_workspace": (c_sz, [c_int, c_int, c_i64]),
    "kd_kldiv": (c_int, [_P(View3), _P(View3), c_f, c_int, c_int, c_i64, c_vp, _P(View3), c_f, c_vp, c_sz, c_vp]),
    "kd_hint_mse": (c_int, [_P(View3), _P(View3), c_f, c_int, c_int, c_i64, c_vp, _P(View3), c_f, c_vp, c_sz, c_vp]),
    "kd_weighted_hint_mse": (c_int, [_P(View3), _P(View3), c_vp, c_int, c_int, c_int, c_i64, c_vp, _P(View3), c_f, c_vp,
                                     c_sz, c_vp]),
    "kd_ce2d": (c_int, [_P(View3), c_vp, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "kd_ce2d_up": (c_int, [c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "kd_kldiv_up": (c_int, [c_vp, c_vp, c_f, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_sz, c_vp]),
    "kd_ce2d_grad": (c_int, [_P(View3), c_vp, c_int, c_int, c_int, c_i64, _P(View3), c_f, c_vp, c_sz, c_vp]),
    "kd_ce2d_weighted": (c_int, [_P(View3), c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_sz, c_vp]),
    "kd_ce2d_weighted_grad": (c_int, [_P(View3), c_vp, c_vp, c_int, c_int, c_int, c_int, c_i64, _P(View3), c_f, c_vp, c_sz, c_vp]),
    "kd_confusion": (c_int, [_P(View3), c_vp, c_int, c_int, c_i64, c_vp, c_int, c_vp]),
    "kd_radam_step": (c_int, [c_vp, c_vp, c_vp, c_vp, c_i64, c_int, c_f, c_f, c_f, c_f, c_f, c_vp]),
    "kd_radam_step_multi": (c_int, [_P(RadamTensor), c_int, c_vp]),
    "kd_small_linear": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_i64, c_int, c_int, c_vp, c_int, c_vp]),
    "kd_upsample_bilinear_bwd": (c_int, [c_vp, c_int, c_int, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp,
                                         c_sz, c_vp]),
    "kd_small_wgrad_workspace": (c_sz, [c_int, c_int, c_i64]),
    "kd_small_wgrad": (c_int, [c_vp, c_int, c_int, c_int, c_vp, c_int, c_int, c_int, c_i64, c_vp, c_vp, c_int, c_vp, c_sz, c_vp]),
    "kd_gate_mix_bwd": (c_int, [c_vp, c_int, c_int, c_vp, c_vp, c_int, c_vp, c_int, c_vp, c_vp, c_int, c_int, c_i64, c_vp]),
    "kd_edge_attention_bwd": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp]),
    "kd_rank1_add": (c_int, [c_int, c_vp, c_int, c_vp, c_vp, c_int, c_i64, c_int, c_vp]),
    "kd_debug_kernel_log_enable": (c_int, [c_int]),
    "kd_debug_kernel_log_read": (c_i64, [C.c_char_p, c_sz]),
    "kd_debug_last_kernel": (C.c_char_p, []),
}

_lib = None


class KdccError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise KdccError(f"{LIB_PATH} not built: the HIP extension is required (no CPU fallback). "
                            "Run __graft_entry__.build().")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class kernel_log:
    """`with kernel_log() as log: ...; log.counts` -> {device kernel name: launches} for the dispatchers that choose between
    kernels (kd_debug_kernel_log_*, include/kdcc.h).  Host-side counters; nothing about a launch changes."""

    def __enter__(self):
        self.counts = {}
        check(lib().kd_debug_kernel_log_enable(1), "kd_debug_kernel_log_enable")
        return self

    def snapshot(self):
        need = int(lib().kd_debug_kernel_log_read(None, 0)) + 1
        buf = C.create_string_buffer(need)
        lib().kd_debug_kernel_log_read(buf, need)
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt = line.rsplit("\t", 1)
            out[name] = int(cnt)
        return out

    def __exit__(self, *exc):
        self.counts = self.snapshot()
        lib().kd_debug_kernel_log_enable(0)
        return False


def last_kernel():
    """Name of the device kernel the calling thread's last dispatch selected."""
    return lib().kd_debug_last_kernel().decode()


def exported_symbols():
    return sorted(_SIGS)


def check(rc, what=""):
    if rc != 0:
        msg = lib().kd_last_error()
        raise KdccError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")
