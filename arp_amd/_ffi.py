"""ctypes binding of libarp_hip.so (the C ABI in include/arp_hip.h).

There is no CPU fallback: if the shared library is missing the import of any compute module
raises, and if no HIP device is present every compute call fails with the library's error.

Process hygiene: PyTorch-ROCm wheels bundle their own libamdhip64.so and request it by its unversioned
file name.  A process that uses the GPU through libarp_hip.so (bound to /opt/rocm's runtime) and imports
torch AFTERWARDS ends up with two HIP runtimes and aborts in glibc at exit.  If a process needs both,
import torch FIRST (libarp_hip.so then binds to the runtime torch already loaded: same SONAME) -- the
tests, bench.py (N > 1) and __graft_entry__.smoke() do -- or keep torch in a child process (bench.py's
cpu_baseline leg does).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# ARP_LIB: an alternate build of the SAME library (A/B experiments: `make -C arp_amd/csrc ALT=kv0 EXTRA=-DARP_G2_KV=0` writes
# arp_amd/alt/kv0/libarp_hip.so) -- selected here instead of being copied over the product; relative paths are repo-relative
LIB_PATH = os.path.join(_HERE, "libarp_hip.so")
if os.environ.get("ARP_LIB"):
    LIB_PATH = os.environ["ARP_LIB"] if os.path.isabs(os.environ["ARP_LIB"]) else os.path.join(os.path.dirname(_HERE), os.environ["ARP_LIB"])

MODE_F32, MODE_BF16, MODE_F16, MODE_F16X3, MODE_F16C = 0, 1, 2, 3, 4
ACT_NONE, ACT_QGELU, ACT_RELU, ACT_TANH, ACT_GELU_TANH = 0, 1, 2, 3, 4


class ArpError(RuntimeError):
    pass


class ClipCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "patch", "width", "layers", "heads", "embed", "img_res", "txt_width", "txt_layers", "txt_heads", "ctx",
        "vocab", "mode", "device", "max_batch", "attn_impl", "n_streams")]


class DtCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "emb", "depth", "heads", "mlp_ratio", "n_actions", "window", "enc_tokens", "enc_dim", "use_adapter", "mode",
        "device", "world", "rank")] + [(n, C.c_float) for n in ("lambda_ret", "weight_decay", "clip_norm", "b1", "b2", "eps")] + [("alibi_bias", C.c_int32)]


class FtCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("layers", "width_v", "width_t", "embed", "hidden", "n_actions", "mode", "device", "use_vip", "use_id")] + [
        (n, C.c_float) for n in ("gamma", "logit_scale", "weight_decay", "b1", "b2", "eps")] + [("goal_conditioned", C.c_int32)]


class EncCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("patch", "width", "layers", "heads", "mlp_ratio", "img_res", "mode", "device", "max_frames",
                                          "attn_impl")]


# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A labeller uses one stream per
# batch part; with a SECOND handle in the process its streams landed on queues already in use and it ran at the single-stream rate
# (78 k instead of 90 k frames/s, whichever handle came second); with 8 queues both run at 90 k.  Read by the runtime at its first
# HIP call, so it has to be in the environment before that; an explicit setting of the user's wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `make -C arp_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    return C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


lib = _load()

_vp, _i, _f = C.c_void_p, C.c_int, C.c_float
_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)

# name -> (restype, argtypes); also the list tests check against include/arp_hip.h
SIGNATURES = {
    "arp_last_error": (C.c_char_p, []),
    "arp_version": (_i, []),
    "arp_device_count": (_i, []),
    "arp_dev_malloc": (_i, [C.POINTER(_vp), C.c_size_t]),
    "arp_dev_free": (_i, [_vp]),
    "arp_memcpy_h2d": (_i, [_vp, _vp, C.c_size_t]),
    "arp_memcpy_d2h": (_i, [_vp, _vp, C.c_size_t]),
    "arp_set_device": (_i, [_i]),
    "arp_dev_synchronize": (_i, []),
    "arp_host_register": (_i, [_vp, C.c_size_t]),
    "arp_host_unregister": (_i, [_vp]),
    "arp_clip_create": (_i, [C.POINTER(ClipCfg), C.POINTER(_vp)]),
    "arp_clip_destroy": (_i, [_vp]),
    "arp_clip_load_weight": (_i, [_vp, C.c_char_p, _fp, _i64p, _i]),
    "arp_clip_set_fp8_mlp": (_i, [_vp, _i]),
    "arp_clip_finalize_weights": (_i, [_vp]),
    "arp_clip_set_text": (_i, [_vp, _i32p, _i]),
    "arp_clip_get_text_features": (_i, [_vp, _fp]),
    "arp_clip_label": (_i, [_vp, _u8p, _i, _i, _i, _i, _fp]),
    "arp_clip_label_submit": (_i, [_vp, _i, _u8p, _i, _i, _i, _i]),
    "arp_clip_label_collect": (_i, [_vp, _i, _fp]),
    "arp_clip_label_dev_async": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "arp_clip_sync": (_i, [_vp]),
    "arp_clip_set_streams": (_i, [_vp, _i]),
    "arp_clip_encode_image": (_i, [_vp, _u8p, _i, _i, _i, _i, _i, _fp]),
    "arp_preprocess": (_i, [_u8p, _i, _i, _i, _i, _i, _fp]),
    "arp_bicubic_coeffs": (_i, [_i, _i, _i32p, _i32p, _i32p, _i]),
    "arp_clip_clock_probe": (_i, [_vp, _i]),
    "arp_clip_clock_read": (_i, [_vp, C.POINTER(C.c_double)]),
    "arp_clip_profile_enable": (_i, [_vp, _i]),
    "arp_clip_profile_reset": (_i, [_vp]),
    "arp_clip_profile_json": (_i, [_vp, C.c_char_p, _i]),
    "arp_event_create": (_i, [C.POINTER(_vp)]),
    "arp_event_destroy": (_i, [_vp]),
    "arp_clip_event_record": (_i, [_vp, _vp]),
    "arp_event_elapsed_ms": (_i, [_vp, _vp, _fp]),
    "arp_dt_create": (_i, [C.POINTER(DtCfg), C.POINTER(_vp)]),
    "arp_dt_destroy": (_i, [_vp]),
    "arp_dt_num_params": (_i, [_vp, _i64p, _i32p]),
    "arp_dt_param_info": (_i, [_vp, _i, C.c_char_p, _i, _i64p, _i32p]),
    "arp_dt_set_tensor": (_i, [_vp, C.c_char_p, _i, _fp]),
    "arp_dt_get_tensor": (_i, [_vp, C.c_char_p, _i, _fp]),
    "arp_dt_set_step": (_i, [_vp, C.c_int64]),
    "arp_dt_get_step": (_i, [_vp, _i64p]),
    "arp_dt_set_batch": (_i, [_vp, _fp, _i32p, _fp, _i]),
    "arp_dt_forward": (_i, [_vp, _fp, _fp, _fp]),
    "arp_dt_backward": (_i, [_vp]),
    "arp_dt_train_step": (_i, [_vp, _f, _fp]),
    "arp_dt_train_step_async": (_i, [_vp, _f]),
    "arp_dt_val_step": (_i, [_vp, _fp]),
    "arp_dt_upload_batch_async": (_i, [_vp, _i, _fp, _i32p, _fp, _i]),
    "arp_dt_upload_batch_images_async": (_i, [_vp, _i, _fp, _i32p, _fp, _i]),
    "arp_dt_select_batch": (_i, [_vp, _i]),
    "arp_dt_bucket_plan": (_i, [C.POINTER(DtCfg), _i64p, _i64p]),
    "arp_dt_sync": (_i, [_vp]),
    "arp_dt_event_record": (_i, [_vp, _vp]),
    "arp_dt_comm_unique_id": (_i, [_vp]),
    "arp_dt_comm_init": (_i, [_vp, _vp, _i, _i]),
    "arp_dt_broadcast_state": (_i, [_vp]),
    "arp_dt_set_adapter_corrections": (_i, [_vp, _i]),
    "arp_dt_debug_read": (C.c_int64, [_vp, C.c_char_p, _vp, C.c_int64]),
    "arp_dt_comm_info": (_i, [_vp, _i32p]),
    "arp_dt_comm_selfcheck": (_i, [_vp, C.POINTER(C.c_double)]),
    "arp_dt_profile_enable": (_i, [_vp, _i]),
    "arp_dt_profile_reset": (_i, [_vp]),
    "arp_dt_profile_json": (_i, [_vp, C.c_char_p, _i]),
    "arp_clip_encode_image_multiscale": (_i, [_vp, _u8p, _i, _i, _i, _fp, _fp]),
    "arp_clip_encode_image_multiscale_pil": (_i, [_vp, _u8p, _i, _i, _i, _i, _fp, _fp]),
    "arp_clip_set_prompt_reduce": (_i, [_vp, _i]),
    "arp_clip_encode_text_multiscale": (_i, [_vp, _i32p, _i, _fp, _fp]),
    "arp_clip_encode_image_multiscale_dev": (_i, [_vp, _u8p, _i, _i, _i, _vp, _vp]),
    "arp_clip_encode_text_multiscale_dev": (_i, [_vp, _i32p, _i, _vp, _vp]),
    "arp_ft_set_batch_dev": (_i, [_vp, _vp, _vp, _vp, _vp, _fp, _i32p, _i]),
    "arp_ft_create": (_i, [C.POINTER(FtCfg), C.POINTER(_vp)]),
    "arp_ft_destroy": (_i, [_vp]),
    "arp_ft_num_params": (_i, [_vp, _i64p, _i32p]),
    "arp_ft_param_info": (_i, [_vp, _i, C.c_char_p, _i, _i64p, _i32p]),
    "arp_ft_set_tensor": (_i, [_vp, C.c_char_p, _i, _fp]),
    "arp_ft_get_tensor": (_i, [_vp, C.c_char_p, _i, _fp]),
    "arp_ft_set_step": (_i, [_vp, C.c_int64]),
    "arp_ft_get_step": (_i, [_vp, _i64p]),
    "arp_ft_dropped_gradients": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "arp_ft_set_batch": (_i, [_vp, _fp, _fp, _fp, _fp, _fp, _i32p, _i]),
    "arp_ft_forward": (_i, [_vp, _fp, _fp, _fp]),
    "arp_ft_encode": (_i, [_vp, _i, _fp, _fp, _i, _fp]),
    "arp_ft_backward": (_i, [_vp]),
    "arp_ft_train_step": (_i, [_vp, C.c_float, _fp]),
    "arp_ft_train_step_async": (_i, [_vp, C.c_float]),
    "arp_ft_sync": (_i, [_vp]),
    "arp_ft_comm_init": (_i, [_vp, _vp, _i, _i]),
    "arp_ft_broadcast_state": (_i, [_vp]),
    "arp_ft_comm_info": (_i, [_vp, _i32p]),
    "arp_ft_comm_selfcheck": (_i, [_vp, C.POINTER(C.c_double)]),
    "arp_ft_bucket_plan": (_i, [C.POINTER(FtCfg), _i64p, _i64p]),
    "arp_ft_event_record": (_i, [_vp, _vp]),
    "arp_ft_profile_enable": (_i, [_vp, _i]),
    "arp_ft_profile_reset": (_i, [_vp]),
    "arp_ft_profile_json": (_i, [_vp, C.c_char_p, _i]),
    "arp_enc_create": (_i, [C.POINTER(EncCfg), C.POINTER(_vp)]),
    "arp_enc_destroy": (_i, [_vp]),
    "arp_enc_load_weight": (_i, [_vp, C.c_char_p, _fp, _i64p, _i]),
    "arp_enc_finalize_weights": (_i, [_vp]),
    "arp_enc_forward": (_i, [_vp, _fp, _i, _fp]),
    "arp_enc_set_streams": (_i, [_vp, _i, _i, _i]),
    "arp_enc_profile_enable": (_i, [_vp, _i]),
    "arp_enc_profile_json": (_i, [_vp, C.c_char_p, _i]),
    "arp_dt_attach_encoder": (_i, [_vp, _vp]),
    "arp_dt_set_batch_images": (_i, [_vp, _fp, _i32p, _fp, _i]),
    "arp_dt_encode_ahead": (_i, [_vp, _i]),
    "arp_h5_write_rows_deflated": (_i, [_vp, C.c_int64, C.c_int64, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _i, _i]),
    "arp_h5_inflate_last_frames": (_i, [_i, _i, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _u8p, C.c_uint64, C.c_uint64,
                                        C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), _u8p, _i]),
    "arp_op_gemm_nt": (_i, [_i, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i]),
    "arp_op_skinny_gemm": (_i, [_i, _i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _f, _fp]),
    "arp_op_gemm_f16c": (_i, [_i, _fp, _fp, _fp, _fp, _i, _i, _i, _i32p]),
    "arp_op_gemm_fp8": (_i, [_i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _i, _f]),
    "arp_op_gemm_bench": (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _fp]),
    "arp_op_gemm_tn": (_i, [_i, _i, _i, _fp, _fp, _fp, _i, _i, _i, _f]),
    "arp_op_gemm_relu_bwd": (_i, [_i, _fp, _fp, _fp, _fp, _fp, _i, _i, _i]),
    "arp_op_adapter_dy": (_i, [_i, _fp, _fp, _fp, _fp, _f, _fp, _fp, _fp, _i, _i, _i, _i]),
    "arp_op_layernorm": (_i, [_fp, _fp, _fp, _fp, _i, _i, _f]),
    "arp_op_attention": (_i, [_i, _i, _fp, _fp, _i, _i, _i, _i, _i]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = the library does not export a declared symbol
    _fn.restype = _res
    _fn.argtypes = _args


def last_error():
    s = lib.arp_last_error()
    return s.decode("utf-8", "replace") if s else ""


def check(rc):
    if rc < 0:
        raise ArpError(last_error())
    return rc


def device_count():
    return lib.arp_device_count()


def require_gpu():
    if device_count() <= 0:
        raise ArpError("no HIP device visible: libarp_hip.so has no CPU fallback")


def as_ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))
