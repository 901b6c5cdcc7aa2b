"""ctypes binding of libmeshraster_hip.so (the C-ABI declared in include/meshraster_hip.h).

The product path has NO fallback: if the shared library is missing, does not export a
symbol of the header, or a call returns non-zero, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (HOC_LIB_PATH: another build of the same ABI -- scripts/build_variant.py makes A / B libraries for one-box kernel comparisons)
LIB_PATH = os.environ.get("HOC_LIB_PATH") or os.path.join(_HERE, "libmeshraster_hip.so")
ABI_VERSION = 8  # MR_ABI_VERSION of include/meshraster_hip.h
FLAG_REFERENCE_ALGO = 1
FLAG_SPARSE_TILES = 2
FLAG_OUTPUT_ZEROED = 4
FLAG_TILE_PER_WORKGROUP = 8
FLAG_TILE_LIST_CLEARED = 16

_c = ctypes
_P, _I, _F, _L = _c.c_void_p, _c.c_int, _c.c_float, _c.c_int64

# name -> (restype, argtypes); must list every MR_API prototype of include/meshraster_hip.h
SIGNATURES = {
    "mr_abi_version": (_I, []),
    "mr_device_ok": (_I, []),
    "mr_selftest_division": (_I, [_P, _P, _P, _P, _L, _P]),
    "mr_forward_face_index_map": (_I, [_P] * 6 + [_I, _I, _I, _F, _F, _I, _I, _I, _P]),
    "mr_forward_texture_sampling": (_I, [_P] * 8 + [_I, _I, _I, _I, _F, _P]),
    "mr_backward_pixel_map": (_I, [_P] * 7 + [_I, _I, _I, _F, _I, _I, _P]),
    "mr_backward_textures": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "mr_backward_depth_map": (_I, [_P] * 7 + [_I, _I, _I, _P]),
    "mr_render_workspace_bytes": (_L, [_I, _I, _I]),
    "mr_render_forward": (_I, [_P, _P, _P, _I] + [_P] * 7 + [_L, _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _P]),
    "mr_face_inv_map": (_I, [_P, _P, _P, _I, _I, _I, _P]),
    "mr_render_backward_workspace_bytes": (_L, [_I, _I, _I]),
    "mr_render_backward_list_workspace_bytes": (_L, [_I, _I]),
    "mr_render_backward": (_I, [_P] * 11 + [_L, _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _P]),
    "mr_pixel_map_terms": (_I, [_P, _I]),
    "mr_render_vc_forward": (_I, [_P, _P, _P, _P, _I] + [_P] * 6 + [_L, _I, _I, _I, _I, _I, _F, _F, _F, _I, _I, _I, _I, _I, _P]),
    "mr_render_vc_backward": (_I, [_P] * 7 + [_I, _I, _I, _I, _I, _F, _I, _I, _P]),
    "mr_render_flow_backward": (_I, [_P] * 11 + [_I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _I, _P, _P]),
    "mr_render_flow_forward": (_I, [_P, _P, _P, _P, _I, _P, _I, _F] + [_P] * 8 + [_L, _I, _I, _I, _I, _I, _F, _F, _F, _I, _P, _I, _P, _P, _L, _I, _P]),
    "mr_flow_vertices_forward": (_I, [_P] * 7 + [_I, _F] + [_P] * 4 + [_I, _I, _P]),
    "mr_flow_vertices_backward": (_I, [_P] * 8 + [_I, _I, _P]),
    "mr_flow_vertices_parts_forward": (_I, [_P] * 4 + [_I, _I] + [_P] * 5 + [_I, _F] + [_P] * 4 + [_I, _P]),
    "mr_flow_vertices_parts_backward": (_I, [_P] * 4 + [_I, _I] + [_P] * 8 + [_I, _P]),
    "mr_stack_pair_faces": (_I, [_P, _I, _P, _I, _P, _I, _I, _I, _P]),
    "mr_flow_pair_prologue_parts": (_I, [_P] * 4 + [_I, _I] + [_P] * 5 + [_I, _F] + [_P] * 5 + [_I, _P, _P, _I, _I, _I, _P, _L, _P]),
    "mr_mano_workspace_floats": (_L, [_I]),
    "mr_mano_forward": (_I, [_P] * 12 + [_I, _I] + [_P] * 3 + [_I, _P]),
    "mr_mano_backward": (_I, [_P] * 10 + [_I, _I] + [_P] * 5 + [_I, _P]),
    "mr_meshreg_post_forward": (_I, [_P] * 6 + [_F] * 5 + [_P] * 5 + [_I, _I, _I, _I, _P]),
    "mr_meshreg_post_backward": (_I, [_P] * 6 + [_F] * 5 + [_P] * 10 + [_I, _I, _I, _I, _P]),
    "mr_warp_forward": (_I, [_P] * 4 + [_I, _I, _I, _I, _F, _I, _P]),
    "mr_warp_backward": (_I, [_P] * 5 + [_I, _I, _I, _I, _F, _I, _P]),
    "mr_occlusion_mask": (_I, [_P] * 4 + [_L, _P, _P, _P, _P, _I, _I, _I, _F, _F, _P]),
    "mr_occlusion_flow": (_I, [_P] * 4 + [_L] + [_P] * 8 + [_I, _I, _I, _I, _I, _F, _F, _P]),
    "mr_flow_mask": (_I, [_P, _P, _P, _I, _F, _P, _I, _I, _P]),
    "mr_flow_finalize_forward": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "mr_flow_finalize_backward": (_I, [_P] * 5 + [_I, _I, _I, _I, _P]),
    "mr_pair_consist_workspace_bytes": (_L, [_I, _I, _I]),
    "mr_pair_consist_forward": (_I, [_P] * 6 + [_I, _P, _L] + [_P] * 11 + [_I, _I, _I, _F, _P, _P, _I, _P]),
    "mr_pair_consist_backward": (_I, [_P] * 6 + [_I] + [_P] * 5 + [_I, _I, _I, _F, _P, _P, _I, _P, _P]),
    "mr_render_tile_list": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "mr_render_clear_bytes": (_L, [_I, _I, _I]),
    "mr_occlusion_flow_tiles": (_I, [_P] * 4 + [_L] + [_P] * 8 + [_I, _I, _I, _I, _F, _F, _P, _P, _L, _L, _P]),
    "mr_pair_consist_tiles_workspace_bytes": (_L, [_I, _I]),
    "mr_pair_consist_forward_tiles": (_I, [_P] * 6 + [_I, _P, _L, _P, _P, _P, _I, _I, _I, _F, _P, _P, _I, _P, _P, _L, _L, _P]),
    "mr_pair_consist_backward_tiles": (_I, [_P] * 6 + [_I] + [_P] * 5 + [_I, _I, _I, _F, _P, _P, _I, _P, _P, _P, _L, _L, _P]),
    "mr_flow_pair_forward_tiles": (_I, [_P] * 4 + [_L] + [_P] * 12 + [_I, _P, _L, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _L, _L, _P]),
    "mr_flow_pair_backward_tiles": (_I, [_P] * 9 + [_I] + [_P] * 8 + [_I, _I, _P, _I, _I, _I, _I, _I, _F, _F, _I, _I, _P]),
    "mr_flow_pair_forward_grad_tiles": (_I, [_P] * 4 + [_L] + [_P] * 12 + [_I, _P, _L, _P, _P, _P, _I, _I, _I, _I, _F, _F, _F, _P, _P, _L, _L, _P, _P, _P, _P, _P]),
    "mr_flow_pair_backward_unit_tiles": (_I, [_P] * 9 + [_I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _I, _P, _P]),
    "mr_flow_pair_scatter_work_bytes": (_L, [_I, _I]),
    "mr_pair_step_struct_bytes": (_L, []),
    "mr_pair_step_field_offsets": (_I, [_P, _I]),
    "mr_pair_step_sizes": (_I, [_P, _P, _P, _P]),
    "mr_pair_step_forward": (_I, [_P, _P]),
    "mr_pair_step_backward": (_I, [_P, _P]),
    "mr_frames_to_batch_workspace_bytes": (_L, [_I, _I, _I]),
    "mr_frames_to_batch": (_I, [_P] * 3 + [_F] * 6 + [_P, _L, _P, _P] + [_I] * 6 + [_P]),
    "mr_bn_act_forward": (_I, [_P] * 6 + [_F, _I, _I, _I, _P, _I, _I, _I, _P]),
    "mr_bn_act_backward_workspace_bytes": (_L, [_I, _I]),
    "mr_bn_act_backward": (_I, [_P] * 8 + [_F, _I, _I, _I] + [_P] * 5 + [_L, _I, _I, _I, _P]),
    "mr_stem_pool_forward": (_I, [_P] * 5 + [_F, _I, _I, _P, _P, _I, _I, _I, _I, _P]),
    "mr_stem_pool_backward_workspace_bytes": (_L, [_I, _I, _I, _I]),
    "mr_stem_pool_backward": (_I, [_P] * 8 + [_F, _I, _I] + [_P] * 4 + [_L, _I, _I, _I, _I, _P]),
}

_lib = None


def load():
    """Load the library (once) and bind every entry point; raise if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python handobjectconsist_amd/build.py`). "
            "There is no CPU / PyTorch fallback for the render + warp path."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH} has ABI {lib.mr_abi_version()}, expected {ABI_VERSION}; rebuild it")
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


class _StreamArg(ctypes.c_void_p):
    """hipStream_t argument that remembers which device it belongs to (see ``call``)."""

    device_index = None


def stream_ptr(device=None):
    """The torch current stream of ``device`` as the C-ABI's stream argument.  Every entry point takes the
    stream of the device its tensors live on; ``call`` switches the thread's current HIP device to that
    device for the duration of the launch when it differs (tensors on cuda:1 while cuda:0 is current)."""
    arg = _StreamArg(torch.cuda.current_stream(device).cuda_stream)
    if device is not None:
        device = torch.device(device)
        arg.device_index = device.index if device.index is not None else torch.cuda.current_device()
    return arg


def check_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise TypeError("handobjectconsist_amd kernels support only cuda (ROCm) tensors")


def call(name, *args):
    """Invoke an entry point; non-zero status -> RuntimeError (the reference's C++ asserts
    surface as RuntimeError too)."""
    fn = getattr(load(), name)
    dev = getattr(args[-1], "device_index", None) if args else None
    if dev is not None and dev != torch.cuda.current_device():
        # HIP launches go to the calling thread's CURRENT device: make it the tensors' device
        with torch.cuda.device(dev):
            rc = fn(*args)
    else:
        rc = fn(*args)
    if rc != 0:
        kind = {-1: "bad argument", -2: "not implemented"}.get(rc, f"hipError_t {rc}")
        raise RuntimeError(f"{name} failed: {kind}")
    return rc


def tile_list(workspace, batch_size, num_faces, image_size):
    """Where mr_render_flow_forward left the tile list in its ``workspace`` tensor: (header pointer, entries pointer,
    capacity) for the *_tiles entry points, or None for raster sizes that build no list."""
    hdr, ents, cap = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
    rc = load().mr_render_tile_list(ptr(workspace), int(batch_size), int(num_faces), int(image_size), ctypes.byref(hdr),
                                    ctypes.byref(ents), ctypes.byref(cap))
    if rc == -2:
        return None
    if rc != 0:
        raise RuntimeError(f"mr_render_tile_list failed: {rc}")
    return hdr, ents, int(cap.value)


def has_tile_list(batch_size, num_faces, image_size):
    """Does mr_render_flow_forward build a tile list for this raster (a bin is a tile; <= 2^31 tiles)?  Sizes only."""
    hdr, ents, cap = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
    return load().mr_render_tile_list(ctypes.c_void_p(256), int(batch_size), int(num_faces), int(image_size), ctypes.byref(hdr),
                                      ctypes.byref(ents), ctypes.byref(cap)) == 0


def contig(t, dtype=torch.float32):
    if t is None:
        return None
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()
