"""ctypes binding of libsdnative.so (the C ABI declared in include/sdnative.h).

The library is REQUIRED: there is no Python/CPU fallback.  If it is missing the
import of any op raises immediately (build it with `python -m scenedreamer_amd.build`).
`import torch` happens before the dlopen on purpose: libsdnative.so needs
libamdhip64.so.7 and must bind to the HIP runtime instance PyTorch already
loaded, otherwise streams and device pointers would belong to two runtimes.
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must be imported before dlopen, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libsdnative.so")
ABI_VERSION = 5      # include/sdnative.h SDN_ABI_VERSION

SDN_F32, SDN_F16 = 0, 1

_lib = None
_lock = threading.Lock()

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_u = ctypes.c_uint32
c_f = ctypes.c_float
c_i64 = ctypes.c_int64

_SIGNATURES = {
    "sdn_abi_version": (c_i, []),
    "sdn_last_error": (ctypes.c_char_p, []),
    "sdn_rvip": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sdn_rvip_occupancy_bytes": (ctypes.c_size_t, [c_p]),
    "sdn_rvip_build_occupancy": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_rvip_u8": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "sdn_rvip_build_occupancy_u8": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_rvip_debug_counts": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "sdn_volume_compact": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p]),
    "sdn_scene_columns": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_p]),
    "sdn_scene_paste_trees": (c_i, [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_p, c_p, c_p]),
    "sdn_scene_column_tops": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "sdn_posenc_fwd": (c_i, [c_p, c_p, c_i64, c_i64, c_i, c_i, c_p]),
    "sdn_posenc_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_i64, c_i, c_i, c_p]),
    "sdn_grid_encode_fwd": (c_i, [c_p, c_p, c_i, c_p, c_p, c_u, c_u, c_u, c_u, c_f, c_u, c_i, c_p, c_u, c_i, c_p]),
    "sdn_grid_encode_bwd": (c_i, [c_p, c_p, c_p, c_i, c_p, c_p, c_u, c_u, c_u, c_u, c_f, c_u, c_i, c_p, c_p, c_u,
                                  c_i, c_p]),
    "sdn_grid_level_scales": (c_i, [c_u, c_f, c_u, c_p, c_p]),
    "sdn_field_packed_weight_bytes": (ctypes.c_size_t, []),
    "sdn_field_consts_floats": (ctypes.c_size_t, []),
    "sdn_field_const_offset": (c_i, [c_i]),
    "sdn_field_feat_bytes": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "sdn_field_aux_elems": (ctypes.c_size_t, [ctypes.c_int32, ctypes.c_int32]),
    "sdn_field_collapse_table": (c_i, [c_p, c_p, c_u, c_f, c_u, c_p, c_p, c_p]),
    "sdn_field_trunk_shift": (c_i, []),
    "sdn_field_pack_weights": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_field_pack_weights_mx": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_field_encode": (c_i, [c_p, c_p, c_p, c_p, c_p, c_u, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, ctypes.c_int32,
                               ctypes.c_int32, c_f, c_f, c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, c_p]),
    "sdn_sample_depth": (c_i, [c_p, c_p, c_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_f, c_p, c_p, c_p,
                               ctypes.c_int32, c_p]),
    "sdn_field_mlp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                            ctypes.c_float, c_p, ctypes.c_int32, c_p, c_p, c_p, c_p]),
    "sdn_field_render": (c_i, [c_p, c_p, c_p, c_p, c_p, c_u, c_p, c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, ctypes.c_int32,
                               ctypes.c_int32, c_f, c_f, c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, c_f, c_p, ctypes.c_int32, c_p,
                               ctypes.c_int32, c_p, c_p, c_p, c_p]),
    "sdn_render_mlp": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i64, ctypes.c_int32, ctypes.c_int32, c_p, c_p]),
    "sdn_sky_packed_weight_bytes": (ctypes.c_size_t, []),
    "sdn_sky_consts_floats": (ctypes.c_size_t, []),
    "sdn_sky_pack_weights": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_sky_pack_weights_mx": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_sky_partial_rows": (ctypes.c_int32, [ctypes.c_int32, ctypes.c_int32]),
    "sdn_sky_mlp": (c_i, [c_p, c_p, c_p, c_p, c_p, ctypes.c_int32, ctypes.c_int32, c_p, c_p, ctypes.c_int32, ctypes.c_int32, c_p]),
    "sdn_conv_plane_dims": (None, [c_i, c_i, c_p, c_p]),
    "sdn_conv_packed_weight_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "sdn_conv_pack_weights": (c_i, [c_p, c_i, c_i, c_i, c_p, c_p]),
    "sdn_conv_planes_from_f32": (c_i, [c_p, c_i, c_p, c_p, c_i, c_i, c_p]),
    "sdn_conv": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "sdn_conv_head_packed_weight_bytes": (ctypes.c_size_t, []),
    "sdn_conv_head_pack_weights": (c_i, [c_p, c_p, c_p]),
    "sdn_conv_head": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "sdn_conv_chain_packed_weight_bytes": (ctypes.c_size_t, []),
    "sdn_conv_chain_consts_floats": (ctypes.c_size_t, []),
    "sdn_conv_chain_pack_weights": (c_i, [c_p, c_p, c_p, c_p, c_p]),
    "sdn_conv_chain": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_p]),
    "sdn_debug_mfma_probe": (c_i, [c_p, c_p, c_p, c_p]),
}
# entry points added by later kernels register themselves here (name -> (restype, argtypes))
EXTRA_SIGNATURES = {}


class SdnError(RuntimeError):
    pass


class FieldAux(ctypes.Structure):
    """sdn_field_aux (include/sdnative.h): optional device pointers for the other return values of Generator._forward_perpix."""
    _fields_ = [("weights", c_p), ("depth", c_p), ("sigma", c_p), ("colour", c_p), ("sky_blended", c_p), ("nosky", c_p),
                ("colour_passes", c_p), ("flags", ctypes.c_int32)]


FIELD_NO_COLOUR_SKIP = 1     # sdn_field_aux.flags (include/sdnative.h)


def lib():
    """Return the loaded library, loading it on first use.  Raises if absent."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise ImportError(
                        f"{LIB_PATH} not found: the HIP extension is mandatory (no fallback). "
                        "Build it with `python -m scenedreamer_amd.build`.")
                L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
                sigs = dict(_SIGNATURES)
                sigs.update(EXTRA_SIGNATURES)
                for name, (res, args) in sigs.items():
                    fn = getattr(L, name)  # AttributeError if the symbol is not exported
                    fn.restype = res
                    fn.argtypes = args
                if L.sdn_abi_version() != ABI_VERSION:
                    raise ImportError(f"libsdnative ABI version {L.sdn_abi_version()} != {ABI_VERSION} expected by this package: "
                                      "rebuild with `python -m scenedreamer_amd.build --force`")
                _lib = L
    return _lib


def declared_symbols():
    sigs = dict(_SIGNATURES)
    sigs.update(EXTRA_SIGNATURES)
    return sorted(sigs)


def check(code, what=""):
    if code != 0:
        msg = lib().sdn_last_error().decode("utf-8", "replace")
        raise SdnError(msg or f"{what} failed with code {code}")


def current_stream(device=None):
    """hipStream_t of PyTorch's current stream as an integer handle."""
    return torch.cuda.current_stream(device).cuda_stream
