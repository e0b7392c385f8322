"""ctypes binding of libdm_sampler.so (C ABI: include/dm_sampler.h).

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is NO
fallback: if the shared object is missing, loading raises, and every op of this package
fails loudly rather than running some other implementation.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DM_SAMPLER_LIB", os.path.join(_HERE, "libdm_sampler.so"))

_lib = None

_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float

_SIGS = {
    "dm_last_error": (C.c_char_p, []),
    "dm_version": (_i, []),
    "dm_build_key": (C.c_char_p, []),
    "dm_unet_create": (_i, [C.POINTER(_vp), _i, _i, _i, _i]),
    "dm_unet_create_ex": (_i, [C.POINTER(_vp), _i, _i, _i, _i, _i]),
    "dm_unet_num_params": (_i, [_vp]),
    "dm_unet_set_precision": (_i, [_vp, _i]),
    "dm_unet_destroy": (None, [_vp]),
    "dm_unet_packed_floats": (_sz, [_vp]),
    "dm_unet_pack_weights": (_i, [_vp, C.POINTER(_vp), _i, _vp, _vp, _vp]),
    "dm_unet_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i]),
    "dm_unet_forward": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp, _sz, _vp]),
    "dm_time_embed": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "dm_sampler_step": (_i, [_vp, _vp, _vp, C.POINTER(_f), _vp, _sz, _vp]),
    "dm_sampler_coefs": (_i, [_vp, _vp, _vp, _i, _vp, _i, _f, _vp]),
    "dm_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _vp, _f, _vp, _sz, _i, _vp]),
    "dm_sample_chunked": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _vp, _f, _vp, _sz, _i, _vp]),
    "dm_sample_workspace_bytes": (_sz, [_vp, _i, _i, _i, _i, _i, _i]),
    "dm_sample_branches": (_i, [_vp, _i, _i, _i, _i]),
    "dm_stream_create": (_i, [C.POINTER(_vp)]),
    "dm_stream_destroy": (_i, [_vp]),
    "dm_sampler_step_batched": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _sz, _vp]),
    "dm_attention": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "dm_final_conv": (_i, [_vp, _vp, _i, _f, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "dm_init": (_i, []),
    "dm_noalloc_scratch_bytes": (_sz, []),
    "dm_unet_forward_noalloc": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _vp]),
    "dm_postprocess": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "dm_class_embed": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "dm_unet_profile": (_i, [_vp, _vp, _vp, _vp, C.POINTER(_f), _i, _i, _i, _vp, _sz, _i, _vp, _i,
                             C.POINTER(_i), C.POINTER(_f), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dm_conv_packed_floats": (_sz, [_i, _i, _i]),
    "dm_pack_conv_weight": (_i, [_i, _vp, _i, _i, _vp, _vp]),
    "dm_conv_stats_partials": (_i, [_i, _i, _i]),
    "dm_debug_lowres_taps": (None, [_vp]),
    "dm_debug_conv_amax": (None, [_vp, _i, _vp, _i, _vp]),
    "dm_groupnorm_silu": (_i, [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "dm_conv2d": (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _vp, _i, _f, _vp, _vp, _vp, _i, _vp, _vp,
                       _i, _i, _i, _vp]),
}
EXPORTS = tuple(_SIGS)
ABI_VERSION = 4  # include/dm_sampler.h: DM_ABI_VERSION
CHUNK_FN = C.CFUNCTYPE(_vp, _vp, _i, _i)  # dm_chunk_fn


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no fallback path.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        if l.dm_version() != ABI_VERSION:
            raise RuntimeError(f"{LIB_PATH} implements ABI version {l.dm_version()}, this binding needs {ABI_VERSION}: rebuild it "
                               "(python -c 'import __graft_entry__ as g; g.build()')")
        _lib = l
    return _lib


def check(rc):
    """Map a non-zero return code to RuntimeError carrying the library's message."""
    if rc != 0:
        raise RuntimeError(lib().dm_last_error().decode())


def ptr(t):
    """Device/host pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def own_stream(device):
    """A torch stream object over a HIP stream of this library's own (dm_stream_create), one per call, never released:
    torch.cuda.Stream() cycles through a POOL of 32 shared streams per device, so a pooled stream can be the stream some
    other component launches into - fatal for a stream that is being captured into a hipGraph."""
    import torch
    h = _vp()
    with torch.cuda.device(device):
        check(lib().dm_stream_create(C.byref(h)))
    return torch.cuda.ExternalStream(h.value, device=device)
