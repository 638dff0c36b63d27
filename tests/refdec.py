"""ctypes wrapper over oracle/_ref/libj40ref.so (the unmodified reference compiled by oracle/Makefile).

TEST INFRASTRUCTURE ONLY: used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(_HERE, "..", "oracle", "_ref", "libj40ref.so")


def err4(code):
    return "".join(chr((code >> s) & 0xff) for s in (24, 16, 8, 0)) if code else ""


class Ref:
    def __init__(self, path=REF_SO):
        self.lib = C.CDLL(os.path.abspath(path))
        L = self.lib
        L.ref_decode_rgba.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ref_decode_rgba.restype = C.c_uint32
        L.ref_decode_into.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.ref_decode_into.restype = C.c_uint32
        L.ref_free.argtypes = [C.c_void_p]
        L.ref_error_string_for.argtypes = [C.c_void_p, C.c_size_t]
        L.ref_error_string_for.restype = C.c_char_p
        L.ref_stage_open.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.ref_stage_open.restype = C.c_void_p
        L.ref_stage_close.argtypes = [C.c_void_p]
        L.ref_stage_frame_info.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_stage_lf_group_info.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.ref_stage_lf_group_plane.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ref_stage_lf_group_plane.restype = C.c_int
        L.ref_stage_varblocks.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.ref_stage_llf.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ref_stage_coeffs.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        L.ref_stage_dq_matrix.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_stage_dq_matrix.restype = C.c_int32
        L.ref_stage_order.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.ref_stage_order.restype = C.c_int32
        L.ref_stage_block_ctx_map.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_stage_block_ctx_map.restype = C.c_int32
        L.ref_stage_combine.argtypes = [C.c_void_p]
        L.ref_stage_combine.restype = C.c_uint32
        L.ref_stage_plane_i16.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_stage_plane_i16.restype = C.c_int
        L.ref_stage_num_planes.argtypes = [C.c_void_p]
        L.ref_stage_plane_size.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.ref_stage_rgba.argtypes = [C.c_void_p, C.c_void_p]
        L.ref_kat_inverse_dct2d.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ref_kat_inverse_by_dctsel.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        L.ref_kat_inverse_dct.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ref_kat_forward_llf.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.ref_kat_natural_order.argtypes = [C.c_int32, C.c_int32, C.c_void_p]
        L.ref_kat_natural_order.restype = C.c_int32
        L.ref_kat_library_dq_matrix.argtypes = [C.c_int, C.c_void_p]
        L.ref_kat_library_dq_matrix.restype = C.c_int32
        L.ref_kat_half_secant.argtypes = [C.c_int]
        L.ref_kat_half_secant.restype = C.c_float
        L.ref_kat_lf2llf_scale.argtypes = [C.c_int]
        L.ref_kat_lf2llf_scale.restype = C.c_float
        L.ref_kat_srgb_i16.argtypes = [C.c_float, C.c_int]
        L.ref_kat_srgb_i16.restype = C.c_int16
        L.ref_kat_cbrtf.argtypes = [C.c_float]
        L.ref_kat_cbrtf.restype = C.c_float
        L.ref_kat_inverse_rct16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]

    def decode(self, data: bytes):
        """returns (err4 string, rgba uint8 [h, w, 4] or None)"""
        buf = C.create_string_buffer(data, len(data))
        out = C.POINTER(C.c_uint8)()
        w, h = C.c_int32(), C.c_int32()
        err = self.lib.ref_decode_rgba(buf, len(data), C.byref(out), C.byref(w), C.byref(h))
        if err or not out:
            return err4(err), None
        arr = np.ctypeslib.as_array(out, shape=(h.value, w.value, 4)).copy()
        self.lib.ref_free(out)
        return "", arr

    def error_string(self, data: bytes):
        buf = C.create_string_buffer(data, len(data))
        return self.lib.ref_error_string_for(buf, len(data)).decode()


class RefStage:
    """staged reference decode: all sections parsed, combine not yet run"""
    INFO = ["width", "height", "is_modular", "num_lf_groups", "num_groups", "num_passes", "nb_block_ctx", "block_ctx_size",
            "num_hf_presets", "global_scale", "quant_lf", "x_qm_scale", "b_qm_scale", "nb_qf_thr", "nb_lf_thr0", "nb_lf_thr1",
            "nb_lf_thr2", "group_size_shift", "bpp", "num_extra_channels", "xyb_encoded"]

    def __init__(self, ref: Ref, data: bytes):
        self.ref = ref
        self.buf = C.create_string_buffer(data, len(data))
        err = C.c_uint32()
        self.h = ref.lib.ref_stage_open(self.buf, len(data), C.byref(err))
        self.err = err4(err.value)
        if not self.h:
            raise RuntimeError("reference rejected stream: " + self.err)
        info = np.zeros(32, np.int64)
        ref.lib.ref_stage_frame_info(self.h, info.ctypes.data)
        self.info = dict(zip(self.INFO, info.tolist()))

    def close(self):
        if self.h:
            self.ref.lib.ref_stage_close(self.h)
            self.h = None

    def lf_group_info(self, gg):
        a = np.zeros(9, np.int32)
        self.ref.lib.ref_stage_lf_group_info(self.h, gg, a.ctypes.data)
        return dict(zip(["left", "top", "width", "height", "width8", "height8", "width64", "height64", "nb_varblocks"], a.tolist()))

    def plane(self, gg, which):
        gi = self.lf_group_info(gg)
        shape, dt = {0: ((gi["height8"], gi["width8"]), np.int32), 1: ((gi["height8"], gi["width8"]), np.uint8),
                     2: ((gi["height64"], gi["width64"]), np.int16), 3: ((gi["height64"], gi["width64"]), np.int16),
                     4: ((gi["height8"], gi["width8"]), np.int16)}[which]
        a = np.zeros(shape, dt)
        assert self.ref.lib.ref_stage_lf_group_plane(self.h, gg, which, a.ctypes.data) == 0
        return a

    def varblocks(self, gg):
        n = self.lf_group_info(gg)["nb_varblocks"]
        a, b = np.zeros(n, np.int32), np.zeros(n, np.float32)
        self.ref.lib.ref_stage_varblocks(self.h, gg, a.ctypes.data, b.ctypes.data)
        return a, b

    def llf(self, gg, c):
        gi = self.lf_group_info(gg)
        a = np.zeros(gi["height8"] * gi["width8"], np.float32)
        self.ref.lib.ref_stage_llf(self.h, gg, c, a.ctypes.data)
        return a

    def coeffs(self, gg, c):
        gi = self.lf_group_info(gg)
        a = np.zeros(gi["height8"] * gi["width8"] * 64, np.float32)
        self.ref.lib.ref_stage_coeffs(self.h, gg, c, a.ctypes.data)
        return a

    def dq_matrix(self, idx):
        a = np.zeros((65536, 3), np.float32)
        n = self.ref.lib.ref_stage_dq_matrix(self.h, idx, a.ctypes.data)
        return a[:n].copy()

    def order(self, p, idx, c):
        a = np.zeros(65536, np.int32)
        n = self.ref.lib.ref_stage_order(self.h, p, idx, c, a.ctypes.data)
        return a[:n].copy()

    def block_ctx_map(self):
        a = np.zeros(4096, np.uint8)
        n = self.ref.lib.ref_stage_block_ctx_map(self.h, a.ctypes.data)
        return a[:n].copy()

    def combine(self):
        return err4(self.ref.lib.ref_stage_combine(self.h))

    def plane_i16(self, c):
        w, h = C.c_int32(), C.c_int32()
        self.ref.lib.ref_stage_plane_size(self.h, c, C.byref(w), C.byref(h))
        a = np.zeros((h.value, w.value), np.int16)
        assert self.ref.lib.ref_stage_plane_i16(self.h, c, a.ctypes.data) == 0
        return a

    def rgba(self):
        a = np.zeros((self.info["height"], self.info["width"], 4), np.uint8)
        self.ref.lib.ref_stage_rgba(self.h, a.ctypes.data)
        return a
