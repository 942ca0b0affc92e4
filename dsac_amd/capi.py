"""ctypes binding of libdsac_hip.so (include/dsac_hip.h).  Thin: one Python function per C export, same
names, same argument order.  Arguments may be numpy arrays (host pointers), torch tensors (host or HIP
device pointers) or raw integer addresses.

There is no fallback: if the shared library is missing this module raises at import time, and
dsac_create fails when no gfx950 device is present.
"""
import ctypes as C
import os

import numpy as np

try:  # torch bundles its own HIP runtime (same SONAME as /opt/rocm's); it must be the one the process loads
    import torch  # noqa: F401  -- first, so that libdsac_hip.so binds to the runtime torch uses for device memory/streams
except ImportError:  # pure C-ABI use without torch is fine: the library then binds to /opt/rocm/lib/libamdhip64.so
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# DSAC_HIP_LIB: an alternative build of the same library (kernel A/B experiments, scripts/r03_k4_ab.sh); never a different implementation
LIB_PATH = os.environ.get("DSAC_HIP_LIB") or os.path.join(_HERE, "libdsac_hip.so")

DSAC_OK = 0
DSAC_ERR_INVALID = -1
DSAC_ERR_NO_DEVICE = -2
DSAC_ERR_HIP = -3
DSAC_ERR_NO_FRAME = -4
DSAC_ERR_ALLOC = -5

DSAC_FRAME_QUANTISE_INT16 = 1
DSAC_FRAME_BORROW = 2
DSAC_BWD_QUIRK_TRANSPOSE = 1
DSAC_BWD_PARITY_FP64 = 2
DSAC_BWD_QUIRK_ROT_WRITEBACK = 4

# every symbol include/dsac_hip.h declares (tests/test_boundary.py checks the header against this list
# and the library against both)
EXPORTS = [
    "dsac_version", "dsac_create", "dsac_destroy", "dsac_last_error", "dsac_set_stream", "dsac_get_stream", "dsac_synchronize",
    "dsac_device_info", "dsac_set_frame", "dsac_sample", "dsac_score_hypotheses", "dsac_sample_ahead", "dsac_score_sampled", "dsac_reproject", "dsac_softmax", "dsac_dpnp", "dsac_score_backward",
    "dsac_soft_score_backward", "dsac_refine", "dsac_refine_fd", "dsac_loss", "dsac_path1_and_softmax_backward", "dsac_set_k2_events", "dsac_profile_enable",
    "dsac_profile_read", "dsac_last_pose_gradients", "dsac_refine_all", "dsac_refine_fd_set", "dsac_refine_fd_sets", "dsac_loss_batch", "dsac_gather_patches", "dsac_set_frames", "dsac_score_hypotheses_frames", "dsac_set_option", "dsac_backward_path1",
    "dsac_loss_frames", "dsac_process_images", "dsac_join_tail", "dsac_select",
    "dsac_device_alloc", "dsac_device_free", "dsac_host_alloc", "dsac_host_free", "dsac_copy_async", "dsac_fill_zero_async", "dsac_tail_wait",
    "dsac_gather_rows",
    "dsac_softmax_frames", "dsac_process_images_begin", "dsac_process_images_finish",
    "dsac_refine_fd_sets_frames", "dsac_loss_batch_frames", "dsac_select_frames", "dsac_soft_score_derr",
    "dsac_refstream_init", "dsac_refstream_discard", "dsac_sample_refstream",
]


class DsacError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("dsac_hip error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "dsac_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C dsac_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, i32, u64, f32, f64, u32 = C.c_void_p, C.c_int, C.c_uint64, C.c_float, C.c_double, C.c_uint
    lib.dsac_version.restype = C.c_char_p
    lib.dsac_version.argtypes = []
    lib.dsac_last_error.restype = C.c_char_p
    lib.dsac_last_error.argtypes = [vp]
    lib.dsac_create.argtypes = [C.POINTER(vp), i32]
    lib.dsac_destroy.argtypes = [vp]
    lib.dsac_destroy.restype = None
    lib.dsac_set_stream.argtypes = [vp, vp]
    lib.dsac_get_stream.argtypes = [vp]
    lib.dsac_get_stream.restype = vp
    lib.dsac_synchronize.argtypes = [vp]
    lib.dsac_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(u64), C.c_char_p]
    lib.dsac_set_frame.argtypes = [vp, vp, vp, i32, i32, f32, f32, f32, f32, u32]
    lib.dsac_sample.argtypes = [vp, i32, u64, vp, f32, i32, vp, vp, vp]
    lib.dsac_reproject.argtypes = [vp, i32, vp, f32, vp, f32, f32, vp]
    lib.dsac_sample_ahead.argtypes = [vp, i32, i32, u64, vp, f32, i32, vp, vp, vp]
    lib.dsac_score_sampled.argtypes = [vp, i32, f32, f32, f32, f64, vp, vp, vp, vp, vp, vp]
    lib.dsac_score_hypotheses.argtypes = [vp, i32, u64, vp, f32, i32, f32, f32, f32, f64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dsac_softmax.argtypes = [vp, i32, vp, f64, vp, vp, vp, vp]
    lib.dsac_dpnp.argtypes = [vp, i32, vp, f32, vp]
    lib.dsac_score_backward.argtypes = [vp, i32, vp, vp, vp, vp, u32, vp]
    lib.dsac_soft_score_backward.argtypes = [vp, i32, vp, vp, vp, f32, f32, f32, vp, u32, vp]
    lib.dsac_refine.argtypes = [vp, i32, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp]
    lib.dsac_refine_fd.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, f32, f32, f32, vp, vp, vp, i32, vp]
    lib.dsac_loss.argtypes = [vp, vp, vp, vp, vp]
    lib.dsac_path1_and_softmax_backward.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.dsac_set_k2_events.argtypes = [vp, vp, vp]
    lib.dsac_profile_enable.argtypes = [vp, i32]
    lib.dsac_profile_read.argtypes = [vp, i32, C.POINTER(f64), C.POINTER(i32), i32]
    lib.dsac_last_pose_gradients.argtypes = [vp, i32, vp]
    lib.dsac_refine_all.argtypes = [vp, i32, vp, vp, i32, i32, i32, f32, vp, vp, vp, vp]
    lib.dsac_refine_fd_set.argtypes = [vp, vp, vp, i32, i32, i32, f32, vp, f32, f32, vp, vp, vp, i32, vp]
    lib.dsac_loss_batch.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.dsac_refine_fd_sets.argtypes = [vp, i32, vp, vp, i32, i32, i32, f32, vp, f32, f32, vp, vp, vp, i32, vp]
    lib.dsac_gather_patches.argtypes = [vp, vp, i32, i32, vp, i32, i32, vp, vp]
    lib.dsac_set_frames.argtypes = [vp, i32, vp, vp, i32, i32, i32, f32, f32, f32, f32, u32]
    lib.dsac_score_hypotheses_frames.argtypes = [vp, i32, u64, f32, i32, f32, f32, f32, f64, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dsac_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.dsac_loss_frames.argtypes = [vp, i32, vp, vp, vp, vp]
    lib.dsac_select.argtypes = [vp, i32, vp, vp, i32, f64, vp, vp, vp]
    lib.dsac_join_tail.argtypes = [vp]
    lib.dsac_tail_wait.argtypes = [vp, vp]
    lib.dsac_device_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.dsac_device_free.argtypes = [vp, vp]
    lib.dsac_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    lib.dsac_host_free.argtypes = [vp, vp]
    lib.dsac_copy_async.argtypes = [vp, vp, vp, C.c_size_t]
    lib.dsac_fill_zero_async.argtypes = [vp, vp, C.c_size_t]
    lib.dsac_gather_rows.argtypes = [vp, vp, vp, C.c_size_t, i32, vp]
    lib.dsac_process_images.argtypes = [vp, i32, u64, f32, i32, f32, f32, f32, f64, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dsac_refine_fd_sets_frames.argtypes = [vp, i32, vp, vp, vp, i32, i32, i32, f32, vp, f32, f32, vp, vp, vp, i32, vp]
    lib.dsac_loss_batch_frames.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    lib.dsac_select_frames.argtypes = [vp, i32, i32, vp, vp, i32, vp, vp, vp, vp]
    lib.dsac_soft_score_derr.argtypes = [vp, i32, vp, vp, f32, f32, f32, vp]
    lib.dsac_softmax_frames.argtypes = [vp, i32, i32, vp, f64, vp, vp, vp, vp]
    lib.dsac_process_images_begin.argtypes = [vp, i32, u64, f32, i32, f32, f32, f32, vp, vp, vp, vp, vp]
    lib.dsac_process_images_finish.argtypes = [vp, i32, vp, f64, vp, i32, i32, i32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.dsac_refstream_init.argtypes = [vp, u32, i32]
    lib.dsac_refstream_discard.argtypes = [vp, i32, C.c_ulonglong]
    lib.dsac_sample_refstream.argtypes = [vp, i32, f32, C.c_longlong, vp, vp, vp, vp, vp]
    lib.dsac_backward_path1.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp, f32, f32, f32, f64, vp, vp, vp, vp, vp]
    return lib


lib = _load()


def ptr(x):
    """Address of x: None -> NULL, int -> itself, numpy array / torch tensor -> data pointer."""
    if x is None:
        return None
    if isinstance(x, int):
        return x
    if isinstance(x, np.ndarray):
        if not x.flags["C_CONTIGUOUS"]:
            raise ValueError("dsac_amd: array arguments must be C-contiguous")
        return x.ctypes.data
    if hasattr(x, "data_ptr"):  # torch.Tensor
        if not x.is_contiguous():
            raise ValueError("dsac_amd: tensor arguments must be contiguous")
        return x.data_ptr()
    raise TypeError("dsac_amd: cannot take the address of %r" % type(x))


def check(ctx, rc):
    if rc != DSAC_OK:
        raise DsacError(rc, lib.dsac_last_error(ctx).decode("utf-8", "replace"))


def version():
    return lib.dsac_version().decode()
