"""ctypes binding of liblidarnerf_hip.so (C ABI: include/lidarnerf_hip.h).

PyTorch is used only for device memory and streams: every call passes raw device pointers + sizes + the current HIP
stream.  There is NO CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# LNH_LIB_PATH: developer override (A/B builds of the same library, tools/); still no fallback if it is missing
_LIB_PATH = os.environ.get("LNH_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "liblidarnerf_hip.so")

P, U32, I32, F32 = C.c_void_p, C.c_uint32, C.c_int, C.c_float

# name -> argument ctypes (stream is appended automatically)
_SIGS = {
    "lnh_grid_encode_forward": [P, P, P, P, U32, U32, U32, U32, F32, U32, P, U32, I32, U32, I32],
    "lnh_grid_encode_backward": [P, P, P, P, P, U32, U32, U32, U32, F32, U32, P, P, U32, I32, U32, I32],
    "lnh_grid_encode_backward_ws": [P, P, P, P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, P, C.c_uint64],
    "lnh_grid_encode_backward_ws_levels": [P, P, P, P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, P, C.c_uint64,
                                           U32, U32],
    "lnh_grid_encode_backward_ws_begin": [P, P, P, P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, P, C.c_uint64],
    "lnh_grid_encode_backward_ws_finish": [P, P, P, P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, P, C.c_uint64,
                                           U32, U32],
    "lnh_grid_encode_backward_ws_ex": [P, P, P, P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, P, C.c_uint64, U32, U32,
                                       I32, U32],
    "lnh_grad_total_variation": [P, P, P, P, F32, U32, U32, U32, U32, F32, U32, U32, I32, I32],
    "lnh_grid_corner_indices": [P, P, P, U32, U32, U32, U32, F32, U32, U32, I32],
    "lnh_freq_encode_forward": [P, U32, U32, U32, U32, P],
    "lnh_freq_encode_backward": [P, P, U32, U32, U32, U32, P],
    "lnh_sh_encode_forward": [P, P, U32, U32, U32, P],
    "lnh_sh_encode_backward": [P, P, U32, U32, U32, P, P],
    "lnh_mlp_forward": [P, P, U32, U32, U32, U32, U32, U32, U32, P, P],
    "lnh_mlp_backward": [P, P, P, U32, U32, U32, U32, U32, U32, U32, P, P, P, C.c_uint64],
    "lnh_mlp_backward_data": [P, P, P, U32, U32, U32, U32, U32, U32, P, P],
    "lnh_mlp_wgrad": [P, P, U32, U32, U32, P, P, C.c_uint64],
    "lnh_near_far_from_aabb": [P, P, P, U32, F32, P, P],
    "lnh_sph_from_ray": [P, P, F32, U32, P],
    "lnh_morton3D": [P, U32, P],
    "lnh_morton3D_invert": [P, U32, P],
    "lnh_packbits": [P, U32, F32, P],
    "lnh_occupancy_lookup": [P, P, P, F32, U32, U32, U32, P, P],
    "lnh_march_rays_train": [P, P, P, F32, F32, U32, U32, U32, U32, U32, P, P, P, P, P, P, P, P],
    "lnh_composite_rays_train_forward": [P, P, P, P, U32, U32, F32, P, P, P],
    "lnh_composite_rays_train_backward": [P, P, P, P, P, P, P, P, U32, U32, F32, P, P],
    "lnh_lidar_composite_rays_train_forward": [P, P, P, P, P, P, P, U32, U32, U32, F32, P, P, P],
    "lnh_lidar_composite_rays_train_backward": [P, P, P, P, P, P, P, P, P, P, P, P, P, U32, U32, U32, F32, P, P],
    "lnh_march_rays": [U32, U32, P, P, P, P, F32, F32, U32, U32, U32, P, P, P, P, P, P, P],
    "lnh_composite_rays": [U32, U32, F32, P, P, P, P, P, P, P, P],
    "lnh_lidar_weights": [P, P, P, U32, U32, F32, P],
    "lnh_lidar_composite_forward": [P, P, P, P, U32, U32, U32, F32, P, P, P, P],
    "lnh_lidar_composite_backward": [P, P, P, P, P, P, P, U32, U32, U32, F32, P, P],
    "lnh_lidar_resample": [P, P, P, P, U32, U32, U32, F32, U32, P, P, P],
    "lnh_lidar_resample_strided": [P, P, U32, P, P, U32, U32, U32, F32, U32, P, P, P],
    "lnh_lidar_resample_points": [P, P, U32, P, P, U32, U32, U32, F32, P, P, P, P, P, P, F32, P],
    "lnh_lidar_sample_points": [P, P, P, P, F32, U32, U32, U32, U32, P],
    "lnh_lidar_coarse_sample_points": [P, P, P, P, F32, U32, U32, U32, F32, F32, P, P],
    "lnh_grid_encode_forward_mapped": [P, P, P, P, U32, U32, U32, U32, U32, U32, U32, F32, U32, I32],
    "lnh_density_mlp_forward": [P, P, U32, U32, U32, U32, U32, P, P],
    "lnh_density_mlp_backward": [P, P, P, U32, U32, U32, U32, P, P, P, C.c_uint64],
    "lnh_lidar_merge_weights": [P, P, P, P, U32, U32, F32, P, P],
    "lnh_lidar_color_forward": [P, P, P, P, P, U32, U32, P],
    "lnh_lidar_color_composite_forward": [P, P, P, P, P, P, P, U32, U32, F32, P, P, P, P, P, P],
    "lnh_lidar_coarse_samples": [P, U32, U32, F32, F32, P],
    "lnh_lidar_dir_term": [P, P, U32, U32, U32, P, P],
    "lnh_lidar_dir_term_freq": [P, U32, P, U32, U32, P, P],
    "lnh_lidar_dir_term_backward": [P, P, U32, U32, P, P, U32, P, C.c_uint64],
    "lnh_lidar_pack_weights": [P, U32, P, U32, P, U32, U32, P, U32, P, U32, P, P],
    "lnh_lidar_to_pano": [P, U32, U32, U32, F32, F32, F32, P, P, P],
    "lnh_pano_to_lidar": [P, P, U32, U32, F32, F32, P, P],
    "lnh_chamfer_nn": [P, U32, P, U32, P, P],
    "lnh_grad_check_f16": [P, C.c_uint64, P],
    "lnh_adam_table_step": [P, P, P, P, P, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, P, P, P, P],
    "lnh_adam_table_step_dlr": [P, P, P, P, P, C.c_uint64, P, C.c_double, C.c_double, C.c_double, P, P, P, P],
    "lnh_lidar_step_prologue": [P, U32, P, U32, P, U32, U32, P, U32, P, U32, P, P, P, P, P, P, F32, U32, U32, U32, F32, F32, P, P, P,
                                P],
    "lnh_lidar_loss": [P, P, P, U32, F32, F32, F32, P, P, P, P],
    "lnh_lidar_loss_patch": [P, P, P, U32, U32, U32, F32, F32, F32, F32, F32, P, P, P, P],
    "lnh_train_check": [P, P, C.c_uint64, P, P, U32, F32, F32, C.c_double, C.c_double],
    "lnh_train_step": [P, P, P, P, P, P, C.c_uint64, P, P, P, U32, P, P, C.c_double, C.c_double, C.c_double, C.c_double,
                       C.c_double, U32],
    "lnh_zero_regions": [P, P, U32],
    "lnh_lidar_march_prologue": [P, P, P, U32, F32, F32, P, P, P, P, U32],
    "lnh_lidar_color_backward": [P, P, P, P, P, P, P, U32, U32, P, P, P, P, C.c_uint64],
    "lnh_lidar_color_backward_image": [P, P, P, P, P, P, P, U32, U32, P, P, P, P, C.c_uint64],
    "lnh_ragged_color_forward": [P, P, P, P, U32, U32, P],
    "lnh_ragged_color_backward": [P, P, F32, P, P, P, P, U32, U32, P, P, P, P, C.c_uint64],
    "lnh_ragged_points": [P, F32, U32, P],
    "lnh_ragged_pack_weights": [P, U32, P, U32, P, U32, U32, P, U32, P, U32, P, P],
    "lnh_ragged_color_input": [P, P, U32, U32, P],
    "lnh_ragged_color_input_rays": [P, P, P, P, U32, U32, U32, P],
    "lnh_ragged_color_output": [P, U32, P],
    "lnh_ragged_color_output_backward": [P, P, U32, P],
    "lnh_ragged_grad_rows": [P, F32, P, P, U32, U32, P],
}
for _n in ("lnh_mlp_forward", "lnh_mlp_backward", "lnh_mlp_backward_data", "lnh_mlp_wgrad", "lnh_density_mlp_forward", "lnh_density_mlp_backward",
           "lnh_lidar_dir_term", "lnh_lidar_pack_weights", "lnh_lidar_step_prologue", "lnh_lidar_color_forward", "lnh_lidar_color_backward",
           "lnh_lidar_color_composite_forward", "lnh_lidar_color_backward_image", "lnh_lidar_dir_term_freq",
           "lnh_ragged_pack_weights", "lnh_ragged_color_input", "lnh_ragged_color_input_rays", "lnh_ragged_color_output",
           "lnh_ragged_color_output_backward", "lnh_ragged_grad_rows", "lnh_ragged_color_forward", "lnh_ragged_color_backward"):
    _SIGS[_n + "_bf16"] = _SIGS[_n]  # bf16-operand build of the MLP kernels (include/lidarnerf_hip.h, last section)
EXPORTS = sorted(list(_SIGS) + ["lnh_version", "lnh_last_error", "lnh_arch", "lnh_build_variant",
                                 "lnh_grid_backward_workspace_size", "lnh_grid_backward_workspace_size_min",
                                 "lnh_grid_backward_plan_info", "lnh_grid_backward_workspace_clear_bytes",
                                 "lnh_grid_backward_set_slice_entries", "lnh_wgrad_workspace_bytes"])

LNH_F32, LNH_F16 = 0, 1
LNH_BWD_WS_CLEARED, LNH_BWD_TABLE_ZERO = 1, 2

_lib = None


def lib_path():
    return _LIB_PATH


def lib():
    """Load the shared library (fails loudly: the HIP extension IS the product path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"{_LIB_PATH} not found — build it with `python lidar-nerf_amd/build.py` "
                "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
        L = C.CDLL(_LIB_PATH)
        for name, sig in _SIGS.items():
            fn = getattr(L, name)
            fn.argtypes = sig + [P]
            fn.restype = C.c_int
        L.lnh_grid_backward_workspace_size.argtypes = [P, U32, U32, U32, U32, F32, U32, U32, I32, I32]
        L.lnh_grid_backward_workspace_size.restype = C.c_uint64
        L.lnh_grid_backward_workspace_size_min.argtypes = [P, U32, U32, U32, U32, F32, U32, U32, I32, I32]
        L.lnh_grid_backward_workspace_size_min.restype = C.c_uint64
        L.lnh_grid_backward_workspace_clear_bytes.argtypes = [P, U32, U32, U32, U32, F32, U32, U32, I32, U32, I32, C.c_uint64]
        L.lnh_grid_backward_workspace_clear_bytes.restype = C.c_uint64
        L.lnh_grid_backward_plan_info.argtypes = [P, U32, U32, U32, U32, F32, U32, U32, I32, I32, U32, P]
        L.lnh_grid_backward_plan_info.restype = C.c_int
        L.lnh_grid_backward_set_slice_entries.argtypes = [U32]
        L.lnh_grid_backward_set_slice_entries.restype = None
        L.lnh_wgrad_workspace_bytes.argtypes = []
        L.lnh_wgrad_workspace_bytes.restype = C.c_uint64
        L.lnh_last_error.restype = C.c_char_p
        L.lnh_arch.restype = C.c_char_p
        L.lnh_build_variant.restype = C.c_char_p
        L.lnh_version.restype = C.c_int
        # A library that is not the product build (tools/probe_variants.py: phases compiled out, fake cursors, ...) gives
        # WRONG results by construction; LNH_LIB_PATH may point at one only together with LNH_ALLOW_VARIANT=1.
        tag = L.lnh_build_variant().decode()
        if tag != "product" and os.environ.get("LNH_ALLOW_VARIANT") != "1":
            raise RuntimeError(f"{_LIB_PATH} is the timing-probe build '{tag}', not the product library "
                               "(set LNH_ALLOW_VARIANT=1 to time it; its results are wrong by construction)")
        _lib = L
    return _lib


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream():
    """hipStream_t of torch's current stream on the current device.  (torch.cuda.current_stream() builds a Stream object
    per call: 9 us of host time, ~45 times per training step — the occupancy-grid step of BASELINE config 4 was host-bound on
    it; the raw-handle accessor torch itself uses costs a fraction of a microsecond.)"""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


# Optional per-entry-point HIP-event timing (bench.py): TIMERS[name] = [(start_event, end_event, tag), ...].
# Events are recorded on the stream the kernel is launched on (torch's current stream).
TIMERS = None


def enable_timers(names=None):
    """Start collecting HIP events around calls (all entry points, or only `names`)."""
    global TIMERS, _TIMED
    TIMERS, _TIMED = {}, (set(names) if names else None)


def disable_timers():
    global TIMERS
    t, TIMERS = TIMERS, None
    return t


_TIMED = None


def call(name, *args, tag=None, timer=None):
    """Invoke an entry point on the current stream; raise on any non-zero status.  `timer`: the name the optional HIP-event
    timing files this call under (an entry point that serves several roles, e.g. lnh_grid_encode_backward_ws_ex)."""
    L = _lib if _lib is not None else lib()
    tname = timer or name
    timed = TIMERS is not None and (_TIMED is None or tname in _TIMED)
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = getattr(L, name)(*args, stream())
    if timed:
        e1.record()
        TIMERS.setdefault(tname, []).append((e0, e1, tag))
    if rc != 0:
        raise RuntimeError(f"{name} failed ({rc}): {L.lnh_last_error().decode()}")


def zero_regions(tensors):
    """Clear up to 8 device tensors with ONE launch (lnh_zero_regions; zero fills are kernels, never memset nodes)."""
    ts = [t for t in tensors if t is not None and t.numel()]
    if not ts:
        return
    for t in ts:
        if not (t.is_cuda and t.is_contiguous()):
            raise RuntimeError("lidarnerf_hip: zero_regions needs contiguous GPU tensors")
    ptrs = (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
    sizes = (C.c_uint64 * len(ts))(*[t.numel() * t.element_size() for t in ts])
    call("lnh_zero_regions", C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), len(ts))


_WGRAD_WS = {}


def wgrad_ws(device):
    """(pointer, bytes) of the weight-gradient workspace of `device` for the MLP backward entry points (include/
    lidarnerf_hip.h, lnh_wgrad_workspace_bytes): allocated once per device and kept for the life of the process —
    a captured step keeps its address.  The backward launches of a process use it one after the other (one training stream
    per device, as everywhere in this package); a caller that runs backward passes on several streams of one device at the
    same time must hand each its own workspace."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _WGRAD_WS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("lidarnerf_hip: the weight-gradient workspace must exist before a step is captured "
                               "(run one step eagerly first)")
        t = _WGRAD_WS[key] = torch.empty(int(lib().lnh_wgrad_workspace_bytes()), dtype=torch.uint8,
                                         device=torch.device("cuda", key))
    return t.data_ptr(), t.numel()


def ptr_array(values):
    """Host array of device pointers (None -> NULL) for the entry points that take `T *const *`; keep the returned object
    alive until the call has returned."""
    return (C.c_void_p * max(len(values), 1))(*[(v if v else None) for v in values])


def u32_array(values):
    return (C.c_uint32 * max(len(values), 1))(*[int(v) for v in values])


# indices into the optimizer's device-side scalar buffer (include/lidarnerf_hip.h LNH_TS_*)
TS_SCALE, TS_GROWTH, TS_FOUND, TS_INV, TS_INV_TABLE, TS_LAST_SCALE, TS_T, TS_IT, TS_LR, TS_T_NEXT, TS_IT_NEXT, TS_SKIPPED = range(12)
TRAIN_STATE_FLOATS, TRAIN_MAX_SMALL = 16, 16


def mlp_suffix(dt):
    """Entry-point suffix of the MLP kernels for a torch element type (fp16 build: '', bf16 build: '_bf16')."""
    if dt == torch.float16:
        return ""
    if dt == torch.bfloat16:
        return "_bf16"
    raise RuntimeError(f"lidarnerf_hip: unsupported MLP element type {dt} (float16 / bfloat16)")


def dtype_code(dt):
    if dt == torch.float32:
        return LNH_F32
    if dt == torch.float16:
        return LNH_F16
    raise RuntimeError(f"lidarnerf_hip: unsupported table dtype {dt} (float32 / float16 only)")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("lidarnerf_hip: tensor must live on the GPU (no CPU path in this library)")
        if t is not None and not t.is_contiguous():
            raise RuntimeError("lidarnerf_hip: tensor must be contiguous")
