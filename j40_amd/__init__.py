"""j40_amd -- Python host-side mirror of the j40 C API on top of libj40hip.so (MI355X / gfx950).

Thin ctypes plumbing only: the product is build/libj40hip.so (C++ host parser + HIP kernels). The
module fails loudly when the library is missing -- there is no Python or CPU fallback for the hot
path. Function names mirror the reference's public API (j40.h:233-272):

    img = j40_amd.from_memory(data)      # j40_from_memory
    img.output_format(J40_RGBA, J40_U8X4)
    if img.next_frame():                 # j40_next_frame
        rgba = img.frame_pixels_u8x4()   # j40_current_frame + j40_frame_pixels_u8x4 (numpy view copy)
    img.error(), img.error_string(); img.free()

`Frame` exposes the thin C-ABI of include/j40hip.h (parse / upload / decode on a stream / status /
stage dumps) for the parity tests, bench.py and the multi-GPU driver.
"""
import ctypes as C
import os
import numpy as np

J40_RGBA = 0x1755
J40_U8X4 = 0x0F33

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("J40HIP_LIB") or os.path.join(_ROOT, "build", "libj40hip.so")
_lib = None


class J40Error(RuntimeError):
    def __init__(self, code, where=""):
        self.code = code
        super().__init__("j40 error %r %s" % (code, where))


def err4(code):
    return "".join(chr((code >> s) & 0xFF) for s in (24, 16, 8, 0)) if code else ""


class _Image(C.Structure):
    class _U(C.Union):
        _fields_ = [("inner", C.c_void_p), ("err", C.c_uint32), ("saved_errno", C.c_int)]
    _fields_ = [("magic", C.c_uint32), ("u", _U)]


class _FrameHandle(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("reserved", C.c_uint32), ("inner", C.c_void_p)]


class _Pixels(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("stride_bytes", C.c_int32), ("data", C.c_void_p)]


# j40hip_output_alloc (include/j40hip.h): (ctx, width, height, *stride_bytes) -> host memory for the pixels
OUTPUT_ALLOC = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_size_t))


def lib():
    """loads build/libj40hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libj40hip.so is missing at %s: run `make lib` (or __graft_entry__.build())" % LIB_PATH)
    # PyTorch wheels bundle their own libamdhip64; a process that loads both that copy and /opt/rocm's ends up with
    # two HIP runtimes of which only the first sees the GPU. Let torch (the plumbing for device buffers, streams and
    # torch.distributed) load first when it is installed, so that libj40hip.so binds to the runtime already there.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, i32, i64, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_int64, C.c_size_t
    sigs = {
        "j40_error": (u32, [vp]), "j40_error_string": (C.c_char_p, [vp]),
        "j40_from_memory": (u32, [vp, vp, sz, vp]), "j40_from_file": (u32, [vp, C.c_char_p]),
        "j40_output_format": (u32, [vp, i32, i32]), "j40_next_frame": (C.c_int, [vp]),
        "j40_current_frame": (_FrameHandle, [vp]), "j40_frame_pixels_u8x4": (_Pixels, [vp, i32]),
        "j40_row_u8x4": (vp, [_Pixels, i32]), "j40_free": (None, [vp]),
        "j40hip_frame_parse": (vp, [vp, sz, C.c_int, C.POINTER(u32)]), "j40hip_frame_free": (None, [vp]),
        "j40hip_frame_parse_ex": (vp, [vp, sz, C.c_int, u32, C.POINTER(u32)]),
        "j40hip_frame_info": (None, [vp, vp]), "j40hip_frame_codestream_size": (sz, [vp]), "j40hip_frame_num_sections": (i64, [vp]),
        "j40hip_frame_lf_group_info": (None, [vp, i64, vp]), "j40hip_frame_lf_group_plane": (C.c_int, [vp, i64, C.c_int, vp]),
        "j40hip_frame_varblocks": (None, [vp, i64, vp, vp]), "j40hip_frame_llf": (None, [vp, i64, C.c_int, vp]),
        "j40hip_frame_dq_matrix": (i32, [vp, C.c_int, vp]), "j40hip_frame_order": (i32, [vp, C.c_int, C.c_int, C.c_int, vp]),
        "j40hip_frame_block_ctx_map": (i32, [vp, vp]), "j40hip_frame_global_plane": (C.c_int, [vp, C.c_int, vp, C.POINTER(i32), C.POINTER(i32)]),
        "j40hip_kat_natural_order": (i32, [i32, i32, vp]), "j40hip_kat_library_dq_matrix": (i32, [C.c_int, vp]),
        "j40hip_kat_forward_llf": (None, [vp, i32, i32]), "j40hip_kat_half_secant": (C.c_float, [C.c_int]),
        "j40hip_kat_lf2llf_scale": (C.c_float, [C.c_int]),
        "j40hip_device_count": (C.c_int, []), "j40hip_frame_upload": (u32, [vp, C.c_int]),
        "j40hip_frame_set_group_range": (u32, [vp, i64, i64]), "j40hip_frame_decode": (u32, [vp, vp, sz, vp]),
        "j40hip_frame_status": (u32, [vp]), "j40hip_frame_decode_to_host": (u32, [vp, vp, sz]), "j40hip_frame_two_phase_sections": (C.c_int32, [vp]),
        "j40hip_frame_read_coeffs": (u32, [vp, i64, C.c_int, vp]), "j40hip_frame_read_plane_i16": (u32, [vp, C.c_int, vp]),
        "j40hip_frame_decode_timed": (u32, [vp, vp, sz, vp, vp]), "j40hip_frame_force_dense": (None, [vp, C.c_int]),
        "j40hip_kat_device_srgb_u8": (u32, [vp, sz, vp]),
        "j40hip_batch_create": (vp, [vp, i64, C.POINTER(u32)]), "j40hip_batch_free": (None, [vp]),
        "j40hip_batch_decode": (u32, [vp, vp, vp, vp]), "j40hip_batch_decode_timed": (u32, [vp, vp, vp, vp, vp]),
        "j40hip_batch_decode_recorded": (u32, [vp, vp, vp, vp, i32]), "j40hip_batch_elapsed": (u32, [vp, i32, vp]), "j40hip_batch_wait_stage": (u32, [vp, i32, i32, vp]),
        "j40hip_batch_reset": (u32, [vp, vp, i64]), "j40hip_frame_section_sizes": (i64, [vp, vp]), "j40hip_frame_coop_sections": (i32, [vp, vp]), "j40hip_frame_quad_sections": (i32, [vp]), "j40hip_frame_split_sections": (i32, [vp]), "j40hip_frame_lf_bundle": (sz, [vp, vp, sz, vp]), "j40hip_frame_parse_on": (vp, [vp, sz, C.c_int, u32, C.c_int, vp, vp]), "j40hip_frame_lf_on_device": (C.c_int, [vp]), "j40hip_frame_from_lf_bundle": (vp, [vp, sz, vp]),
        "j40hip_frame_upload_on": (u32, [vp, C.c_int, vp]), "j40hip_thread_release": (None, []), "j40hip_shutdown": (None, []),
        "j40hip_frame_status_begin": (u32, [vp, vp]), "j40hip_frame_status_end": (u32, [vp]), "j40hip_frame_mark_idle": (None, [vp]),
        "j40hip_frame_after_frame_status": (u32, [vp]),
        "j40hip_pipeline_create": (vp, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(u32)]), "j40hip_pipeline_free": (None, [vp]), "j40hip_pipeline_create_ex": (vp, [C.c_int, C.c_int, C.c_int, C.c_int, u32, C.POINTER(u32)]), "j40hip_pipeline_lf_device_frames": (i64, [vp]),
        "j40hip_pipeline_submit": (u32, [vp, vp, sz, vp, sz, C.c_int, C.POINTER(i64)]), "j40hip_pipeline_drain": (u32, [vp]),
        "j40hip_pipeline_run": (u32, [vp, vp, sz, OUTPUT_ALLOC, vp]), "j40hip_pipeline_set_max_wait_ms": (None, [vp, C.c_double]),
        "j40hip_stage_dump_create": (vp, [vp, sz, C.c_int, C.c_int, C.POINTER(u32)]), "j40hip_stage_dump_free": (None, [vp]), "j40hip_stage_dump_info": (None, [vp, vp]),
        "j40hip_stage_dump_lf_group_info": (C.c_int, [vp, i64, vp]), "j40hip_stage_dump_plane": (C.c_int, [vp, i64, C.c_int, vp]),
        "j40hip_stage_dump_varblocks": (C.c_int, [vp, i64, vp, vp, vp]), "j40hip_stage_dump_llf": (C.c_int, [vp, i64, C.c_int, vp]),
        "j40hip_stage_dump_group_blocks": (i64, [vp, i64, vp, i64]), "j40hip_stage_dump_sorted_varblocks": (i64, [vp, vp, vp, vp, i64]), "j40hip_stage_dump_rgba": (C.c_int, [vp, vp]),
        "j40hip_copy_engine": (C.c_int, [C.c_int, vp, vp]),
        "j40hip_frame_restoration": (None, [vp, vp]), "j40hip_frame_set_restoration": (None, [vp, C.c_int]), "j40hip_frame_sharpness": (C.c_int, [vp, i64, vp]),
        "j40hip_frame_read_xyb": (u32, [vp, C.c_int, vp]), "j40hip_frame_restoration_ms": (C.c_float, [vp]),
        "j40hip_kat_device_restoration": (u32, [vp, i32, i32, vp, vp, vp, C.c_int, C.c_int, vp]),
        "j40hip_pipeline_result": (u32, [vp, i64]), "j40hip_pipeline_stats": (None, [vp, vp]), "j40hip_pipeline_stats_ex": (None, [vp, vp]), "j40hip_pipeline_lf_stats": (None, [vp, vp]), "j40hip_pipeline_reset_stats": (None, [vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(L, name)  # AttributeError = the library does not export what include/*.h declares
        fn.restype = res
        fn.argtypes = args
    L._declared = sorted(sigs)
    _lib = L
    return L


def device_count():
    return lib().j40hip_device_count()


def copy_engine(device=0):
    """j40hip_copy_engine: the SDMA engine the pipeline's copies back to host memory go to on `device` (-1: hipMemcpyAsync), the measured
    device-to-host GB/s per engine and the runtime's {free, recommended} masks"""
    import numpy as np
    rates = np.zeros(16, np.float64); masks = np.zeros(3, np.uint32)
    e = lib().j40hip_copy_engine(device, rates.ctypes.data, masks.ctypes.data)
    return {"engine": e, "d2h_gb_per_s_per_engine": {str(i): round(float(r), 1) for i, r in enumerate(rates) if r != 0}, "engines_free_mask": hex(int(masks[0])), "engines_recommended_mask": hex(int(masks[1])), "upload_engines_mask": hex(int(masks[2]))}


def shutdown():
    """j40hip_shutdown: stops and joins the library's service threads and gives cached device memory, pooled events and cached tables
    back. Optional; every Frame / Batch / Pipeline of the process must have been closed."""
    if _lib is not None:
        _lib.j40hip_shutdown()


class Image:
    """mirror of a j40_image handle and the public API calls on it"""

    def __init__(self):
        self._img = _Image()
        self._buf = None
        self._live = False

    def error(self):
        return err4(lib().j40_error(C.byref(self._img)))

    def error_string(self):
        return lib().j40_error_string(C.byref(self._img)).decode()

    def output_format(self, channel=J40_RGBA, fmt=J40_U8X4):
        return err4(lib().j40_output_format(C.byref(self._img), channel, fmt))

    def next_frame(self):
        return bool(lib().j40_next_frame(C.byref(self._img)))

    def frame_pixels_u8x4(self, channel=J40_RGBA):
        """returns a numpy copy [height, width, 4] of the rendered frame (or of the error placeholder)"""
        L = lib()
        fr = L.j40_current_frame(C.byref(self._img))
        px = L.j40_frame_pixels_u8x4(C.byref(fr), channel)
        rows = np.ctypeslib.as_array(C.cast(px.data, C.POINTER(C.c_uint8)), shape=(px.height, px.stride_bytes))
        return rows[:, : px.width * 4].reshape(px.height, px.width, 4).copy(), px.stride_bytes, px.data

    def free(self):
        if self._live:
            lib().j40_free(C.byref(self._img))
            self._live = False


def from_memory(data: bytes) -> Image:
    img = Image()
    img._buf = C.create_string_buffer(data, len(data))  # borrowed by the library until free()
    lib().j40_from_memory(C.byref(img._img), img._buf, len(data), None)
    img._live = True
    return img


def from_file(path: str) -> Image:
    img = Image()
    lib().j40_from_file(C.byref(img._img), path.encode())
    img._live = True
    return img


def decode(data: bytes):
    """whole path through the public API; returns (err4, rgba ndarray or None)"""
    img = from_memory(data)
    img.output_format()
    out = None
    if img.next_frame():
        out = img.frame_pixels_u8x4()[0]
    err = img.error()
    img.free()
    return err, out


def decode_timed(buf, size, want_pixels=False):
    """the reference's call sequence (dj40.c) on an existing ctypes buffer, timed from j40_from_memory to the return of
    j40_frame_pixels_u8x4 -- the pixels are then in host memory, in the image's own plane; no copy into numpy inside the clock.
    Returns (err4, milliseconds, pixels or None)"""
    import time
    L = lib()
    img = Image()
    t0 = time.perf_counter()
    L.j40_from_memory(C.byref(img._img), buf, size, None)
    img._live = True
    img.output_format()
    ok = img.next_frame()
    px = None
    if ok:
        fr = L.j40_current_frame(C.byref(img._img))
        px = L.j40_frame_pixels_u8x4(C.byref(fr), J40_RGBA)
    ms = (time.perf_counter() - t0) * 1e3
    err = img.error()
    out = None
    if ok and want_pixels and not err:
        rows = np.ctypeslib.as_array(C.cast(px.data, C.POINTER(C.c_uint8)), shape=(px.height, px.stride_bytes))
        out = rows[:, : px.width * 4].reshape(px.height, px.width, 4).copy()
    img.free()
    return err, ms, out


INFO_FIELDS = ["width", "height", "is_modular", "num_lf_groups", "num_groups", "num_passes", "nb_block_ctx", "block_ctx_size",
               "num_hf_presets", "global_scale", "quant_lf", "x_qm_scale", "b_qm_scale", "nb_qf_thr", "nb_lf_thr0", "nb_lf_thr1",
               "nb_lf_thr2", "group_size_shift", "bpp", "num_extra_channels", "xyb_encoded"]


class Restoration(C.Structure):
    """j40hip_restoration (include/j40hip.h): the frame header's RestorationFilter bundle"""
    _fields_ = [("gab_enabled", C.c_int32), ("gab_weights", C.c_float * 6), ("epf_iters", C.c_int32), ("epf_sharp_lut", C.c_float * 8), ("epf_channel_scale", C.c_float * 3),
                ("epf_quant_mul", C.c_float), ("epf_pass0_sigma_scale", C.c_float), ("epf_pass2_sigma_scale", C.c_float), ("epf_border_sad_mul", C.c_float), ("epf_sigma_for_modular", C.c_float)]

    def as_list(self):
        """the 24 numbers in the order oracle/ref_harness.c's ref_stage_restoration writes them"""
        return ([float(self.gab_enabled)] + list(self.gab_weights) + [float(self.epf_iters)] + list(self.epf_sharp_lut) + list(self.epf_channel_scale)
                + [self.epf_quant_mul, self.epf_pass0_sigma_scale, self.epf_pass2_sigma_scale, self.epf_border_sad_mul, self.epf_sigma_for_modular])

    def params15(self):
        """sharp_lut[8], channel_scale[3], quant_mul, pass0, pass2, border_sad_mul: what the checkers' filter entry points take"""
        return np.array(list(self.epf_sharp_lut) + list(self.epf_channel_scale) + [self.epf_quant_mul, self.epf_pass0_sigma_scale, self.epf_pass2_sigma_scale, self.epf_border_sad_mul], np.float32)


def kat_device_restoration(planes, sharpness, hfmul_inv, r, mode=1, device=0):
    """j40hip_kat_device_restoration: the filter kernels on caller planes [3, h, w] float32 (a copy is filtered and returned with the
    reciprocal-sigma plane); returns (4-char code, planes, sigma)"""
    a = np.ascontiguousarray(planes, np.float32).copy()
    _, h, w = a.shape
    sh = np.ascontiguousarray(sharpness, np.int16); hf = np.ascontiguousarray(hfmul_inv, np.float32)
    sigma = np.zeros(((h + 7) // 8, (w + 7) // 8), np.float32)
    code = lib().j40hip_kat_device_restoration(a.ctypes.data, w, h, sh.ctypes.data, hf.ctypes.data, C.byref(r), mode, device, sigma.ctypes.data)
    return err4(code), a, sigma


class Frame:
    """thin C-ABI (include/j40hip.h): host parse, plan upload, hot path on a HIP stream"""

    def __init__(self, data: bytes, threads: int = 4, lf_device=None):
        """lf_device: a HIP device index -- the LfGroup streams are decoded there instead of on the host (j40hip_frame_parse_on)"""
        L = lib()
        self._buf = C.create_string_buffer(data, len(data))
        err = C.c_uint32()
        if lf_device is None:
            self.h = L.j40hip_frame_parse(self._buf, len(data), threads, C.byref(err))
        else:
            self.h = L.j40hip_frame_parse_on(self._buf, len(data), threads, 1, int(lf_device), None, C.byref(err))
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_frame_parse")
        info = np.zeros(32, np.int64)
        L.j40hip_frame_info(self.h, info.ctypes.data)
        self.info = dict(zip(INFO_FIELDS, info.tolist()))
        self.width, self.height = self.info["width"], self.info["height"]
        self.codestream_size = L.j40hip_frame_codestream_size(self.h)

    @classmethod
    def parse_streamed(cls, data: bytes, step: int = 4096, threads: int = 4, flags: int = 0, log=None):
        """j40hip_frame_parse_streamed over a buffer that fills up as the parser asks (include/j40hip.h): the buffer starts as
        0xAA bytes and every need(n) reveals the stream's bytes up to n rounded up to a multiple of `step` -- a parser that read a
        byte it had not asked for would see rubbish. log (a list) receives every n asked for."""
        L = lib()
        self = cls.__new__(cls)
        n = len(data)
        self._buf = (C.c_uint8 * max(n, 1)).from_buffer(bytearray(b"\xaa" * max(n, 1)))
        have = [0]
        import threading
        lock = threading.Lock()

        def need(_ctx, upto):
            with lock:
                if log is not None:
                    log.append(int(upto))
                want = min(n, (int(upto) + step - 1) // step * step)
                if want > have[0]:
                    C.memmove(C.addressof(self._buf) + have[0], data[have[0]:want], want - have[0])
                    have[0] = want

        def have_now(_ctx):
            with lock:
                return have[0]

        NEED = C.CFUNCTYPE(None, C.c_void_p, C.c_size_t)
        HAVE = C.CFUNCTYPE(C.c_size_t, C.c_void_p)
        self._cb = (NEED(need), HAVE(have_now))
        L.j40hip_frame_parse_streamed.restype = C.c_void_p
        L.j40hip_frame_parse_streamed.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_uint32, NEED, HAVE, C.c_void_p, C.POINTER(C.c_uint32)]
        err = C.c_uint32()
        self.h = L.j40hip_frame_parse_streamed(self._buf, n, threads, flags, self._cb[0], self._cb[1], None, C.byref(err))
        self.revealed = have[0]
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_frame_parse_streamed")
        need(None, n)   # (the rest arrives before anything else looks at the buffer)
        info = np.zeros(32, np.int64)
        L.j40hip_frame_info(self.h, info.ctypes.data)
        self.info = dict(zip(INFO_FIELDS, info.tolist()))
        self.width, self.height = self.info["width"], self.info["height"]
        self.codestream_size = L.j40hip_frame_codestream_size(self.h)
        return self

    @classmethod
    def from_lf_bundle(cls, blob: bytes):
        """a frame handle from the blob Frame.lf_bundle() of another process made (VarDCT frames; nothing is parsed here)"""
        L = lib()
        self = cls.__new__(cls)
        self._buf = C.create_string_buffer(blob, len(blob))
        err = C.c_uint32()
        self.h = L.j40hip_frame_from_lf_bundle(self._buf, len(blob), C.byref(err))
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_frame_from_lf_bundle")
        info = np.zeros(32, np.int64)
        L.j40hip_frame_info(self.h, info.ctypes.data)
        self.info = dict(zip(INFO_FIELDS, info.tolist()))
        self.width, self.height = self.info["width"], self.info["height"]
        self.codestream_size = L.j40hip_frame_codestream_size(self.h)
        return self

    def lf_on_device(self):
        return bool(lib().j40hip_frame_lf_on_device(self.h))

    def lf_bundle(self) -> bytes:
        """the parsed frame (codestream + LF bundle + tables) as one relocatable blob: what rank 0 of a sharded decode can
        broadcast instead of the codestream"""
        L = lib()
        err = C.c_uint32()
        need = L.j40hip_frame_lf_bundle(self.h, None, 0, C.byref(err))
        self._chk(err.value, "in j40hip_frame_lf_bundle")
        out = C.create_string_buffer(need)
        if L.j40hip_frame_lf_bundle(self.h, out, need, C.byref(err)) != need:
            self._chk(err.value or int.from_bytes(b"!mem", "big"), "in j40hip_frame_lf_bundle")
        return out.raw

    def close(self):
        if self.h:
            lib().j40hip_frame_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, code, where):
        if code:
            raise J40Error(err4(code), where)

    def upload(self, device=0):
        self._chk(lib().j40hip_frame_upload(self.h, device), "in j40hip_frame_upload")

    def force_dense(self, dense=True):
        """dense coefficient planes instead of event lists for the next upload (what the library does by itself after "evof")"""
        lib().j40hip_frame_force_dense(self.h, 1 if dense else 0)

    def coop_sections(self):
        """(sections the wave-cooperative Modular kernel takes, sections of the plan); (-1, 0) without a Modular plan"""
        total = C.c_int32(0)
        n = lib().j40hip_frame_coop_sections(self.h, C.byref(total))
        return int(n), int(total.value)

    def quad_sections(self):
        return int(lib().j40hip_frame_quad_sections(self.h))

    def split_sections(self):
        """sections the two-pass Modular decoder takes (position-only MA trees; modular_split.hip)"""
        return int(lib().j40hip_frame_split_sections(self.h))

    def section_sizes(self):
        """bytes of every pass-group section (pass-major), from the TOC"""
        n = lib().j40hip_frame_section_sizes(self.h, None)
        a = np.zeros(n, np.int64)
        lib().j40hip_frame_section_sizes(self.h, a.ctypes.data)
        return a

    def set_group_range(self, first, count):
        self._chk(lib().j40hip_frame_set_group_range(self.h, first, count), "in j40hip_frame_set_group_range")

    def decode(self, rgba_ptr, stride_bytes, stream=0):
        self._chk(lib().j40hip_frame_decode(self.h, rgba_ptr, stride_bytes, stream), "in j40hip_frame_decode")

    def decode_timed(self, rgba_ptr, stride_bytes, stream=0):
        ms = np.zeros(3, np.float32)
        self._chk(lib().j40hip_frame_decode_timed(self.h, rgba_ptr, stride_bytes, stream, ms.ctypes.data), "in j40hip_frame_decode_timed")
        return ms

    def status(self):
        return err4(lib().j40hip_frame_status(self.h))

    def decode_to_host(self):
        out = np.zeros((self.height, self.width, 4), np.uint8)
        code = lib().j40hip_frame_decode_to_host(self.h, out.ctypes.data, self.width * 4)
        return err4(code), out

    def two_phase_sections(self):
        """how decode_to_host last went: k > 0 = two phases with the k longest sections beside the others, 0 = one, -1 = not yet"""
        return lib().j40hip_frame_two_phase_sections(self.h)

    # ---- restoration filters (include/j40hip.h) ----
    def restoration(self):
        """the frame header's RestorationFilter bundle as parsed (j40hip_restoration)"""
        r = Restoration()
        lib().j40hip_frame_restoration(self.h, C.byref(r))
        return r

    def set_restoration(self, mode):
        """-1: as J40HIP_RESTORATION says, 0: off (what j40 does), 1: the filters the frame signals, 2: exactly as j40's routines stand"""
        lib().j40hip_frame_set_restoration(self.h, int(mode))

    def sharpness(self, gg):
        gi = self.lf_group_info(gg)
        a = np.zeros((gi["height8"], gi["width8"]), np.int16)
        assert lib().j40hip_frame_sharpness(self.h, gg, a.ctypes.data) == 0
        return a

    def read_xyb(self, stage):
        """after a decode that ran the filters: stage 0 / 1 -> [3, height, width] float32 (before / after them), 2 -> the reciprocal sigmas [h8, w8]"""
        a = np.zeros((3, self.height, self.width), np.float32) if stage < 2 else np.zeros(((self.height + 7) // 8, (self.width + 7) // 8), np.float32)
        self._chk(lib().j40hip_frame_read_xyb(self.h, stage, a.ctypes.data), "in j40hip_frame_read_xyb")
        return a

    def restoration_ms(self):
        return float(lib().j40hip_frame_restoration_ms(self.h))

    # ---- stage accessors ----
    def lf_group_info(self, gg):
        a = np.zeros(9, np.int32)
        lib().j40hip_frame_lf_group_info(self.h, gg, a.ctypes.data)
        return dict(zip(["left", "top", "width", "height", "width8", "height8", "width64", "height64", "nb_varblocks"], a.tolist()))

    def plane(self, gg, which):
        gi = self.lf_group_info(gg)
        shape, dt = {0: ((gi["height8"], gi["width8"]), np.int32), 1: ((gi["height8"], gi["width8"]), np.uint8),
                     2: ((gi["height64"], gi["width64"]), np.int16), 3: ((gi["height64"], gi["width64"]), np.int16)}[which]
        a = np.zeros(shape, dt)
        assert lib().j40hip_frame_lf_group_plane(self.h, gg, which, a.ctypes.data) == 0
        return a

    def varblocks(self, gg):
        n = self.lf_group_info(gg)["nb_varblocks"]
        a, b = np.zeros(n, np.int32), np.zeros(n, np.float32)
        lib().j40hip_frame_varblocks(self.h, gg, a.ctypes.data, b.ctypes.data)
        return a, b

    def llf(self, gg, c):
        gi = self.lf_group_info(gg)
        a = np.zeros(gi["height8"] * gi["width8"], np.float32)
        lib().j40hip_frame_llf(self.h, gg, c, a.ctypes.data)
        return a

    def dq_matrix(self, idx):
        a = np.zeros((65536, 3), np.float32)
        n = lib().j40hip_frame_dq_matrix(self.h, idx, a.ctypes.data)
        return a[:n].copy()

    def order(self, p, idx, c):
        a = np.zeros(65536, np.int32)
        n = lib().j40hip_frame_order(self.h, p, idx, c, a.ctypes.data)
        return a[:n].copy()

    def block_ctx_map(self):
        a = np.zeros(4096, np.uint8)
        n = lib().j40hip_frame_block_ctx_map(self.h, a.ctypes.data)
        return a[:n].copy()

    def global_plane(self, c):
        w, h = C.c_int32(), C.c_int32()
        if lib().j40hip_frame_global_plane(self.h, c, None, C.byref(w), C.byref(h)) != 0:
            return None
        a = np.zeros((h.value, w.value), np.int16)
        lib().j40hip_frame_global_plane(self.h, c, a.ctypes.data, C.byref(w), C.byref(h))
        return a

    def read_coeffs(self, gg, c):
        gi = self.lf_group_info(gg)
        a = np.zeros(gi["height8"] * gi["width8"] * 64, np.float32)
        self._chk(lib().j40hip_frame_read_coeffs(self.h, gg, c, a.ctypes.data), "in j40hip_frame_read_coeffs")
        return a


class Batch:
    """throughput mode (include/j40hip.h, j40hip_batch_*): uploaded VarDCT frames decoded by one entropy
    launch with one pass-group section per wavefront lane"""

    def __init__(self, frames):
        L = lib()
        self.frames = list(frames)
        arr = (C.c_void_p * len(self.frames))(*[f.h for f in self.frames])
        err = C.c_uint32()
        self.h = L.j40hip_batch_create(arr, len(self.frames), C.byref(err))
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_batch_create")

    def close(self):
        if self.h:
            lib().j40hip_batch_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _args(self, rgba_ptrs, strides):
        n = len(self.frames)
        return (C.c_void_p * n)(*rgba_ptrs), (C.c_size_t * n)(*strides)

    def decode(self, rgba_ptrs, strides, stream=0):
        p, s = self._args(rgba_ptrs, strides)
        code = lib().j40hip_batch_decode(self.h, p, s, stream)
        if code:
            raise J40Error(err4(code), "in j40hip_batch_decode")

    def decode_recorded(self, rgba_ptrs, strides, stream, slot):
        """asynchronous: stage events go to `slot`; read them with elapsed(slot) after synchronising the stream"""
        p, s = self._args(rgba_ptrs, strides)
        code = lib().j40hip_batch_decode_recorded(self.h, p, s, stream, slot)
        if code:
            raise J40Error(err4(code), "in j40hip_batch_decode_recorded")

    def wait_stage(self, slot, stage, stream):
        """`stream` waits for stage 1 (cleared) / 2 (entropy decoded) / 3 (pixels written) of the decode recorded in `slot`"""
        code = lib().j40hip_batch_wait_stage(self.h, slot, stage, stream)
        if code:
            raise J40Error(err4(code), "in j40hip_batch_wait_stage")

    def elapsed(self, slot):
        ms = (C.c_float * 3)()
        code = lib().j40hip_batch_elapsed(self.h, slot, ms)
        if code:
            raise J40Error(err4(code), "in j40hip_batch_elapsed")
        return ms[0], ms[1], ms[2]

    def decode_timed(self, rgba_ptrs, strides, stream=0):
        """returns (entropy ms, pixels ms, clear ms) measured with HIP events on `stream`"""
        p, s = self._args(rgba_ptrs, strides)
        ms = (C.c_float * 3)()
        code = lib().j40hip_batch_decode_timed(self.h, p, s, stream, ms)
        if code:
            raise J40Error(err4(code), "in j40hip_batch_decode_timed")
        return ms[0], ms[1], ms[2]


class StageDump:
    """j40hip_stage_dump_*: one image through the pipeline's device stages, every stage's product copied back (parity tests)"""

    def __init__(self, data, device=0, lf_on_device=True):
        self._buf = C.create_string_buffer(bytes(data), len(data))
        err = C.c_uint32()
        self.h = lib().j40hip_stage_dump_create(self._buf, len(data), device, 1 if lf_on_device else 0, C.byref(err))
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_stage_dump_create")
        a = np.zeros(8, np.uint32)
        lib().j40hip_stage_dump_info(self.h, a.ctypes.data)
        self.verdict, self.flags = err4(int(a[0])), int(a[1])
        self.lf_on_device = bool(a[1] & 4)
        self.num_lf_groups, self.num_groups, self.width, self.height, self.dct_used, self.num_varblocks = (int(v) for v in a[2:8])

    def lf_group_info(self, gg):
        a = np.zeros(10, np.int32)
        assert lib().j40hip_stage_dump_lf_group_info(self.h, gg, a.ctypes.data) == 0
        d = dict(zip(["left", "top", "width", "height", "width8", "height8", "width64", "height64", "nb_varblocks"], a[:9].tolist()))
        d["status"] = err4(int(a[9]) & 0xffffffff)
        return d

    def plane(self, gg, which):
        gi = self.lf_group_info(gg)
        c8, c64 = (gi["height8"], gi["width8"]), (gi["height64"], gi["width64"])
        shape, dt = {0: (c8, np.int32), 1: (c8, np.uint8), 2: (c64, np.int16), 3: (c64, np.int16), 4: (c8, np.int16), 5: (c8, np.int16), 6: (c8, np.int16), 7: (c8, np.int16)}[which]
        a = np.zeros(shape, dt)
        if lib().j40hip_stage_dump_plane(self.h, gg, which, a.ctypes.data) != 0:
            return None
        return a

    def varblocks(self, gg):
        n = self.lf_group_info(gg)["nb_varblocks"]
        a, b, c = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32), np.zeros((max(n, 1), 3), np.int32)
        assert lib().j40hip_stage_dump_varblocks(self.h, gg, a.ctypes.data, b.ctypes.data, c.ctypes.data) == n
        return a[:n], b[:n], c[:n]

    def llf(self, gg, c):
        gi = self.lf_group_info(gg)
        a = np.zeros(gi["height8"] * gi["width8"], np.float32)
        assert lib().j40hip_stage_dump_llf(self.h, gg, c, a.ctypes.data) == 0
        return a

    def group_blocks(self, group):
        a = np.zeros((1024, 3), np.uint32)
        n = lib().j40hip_stage_dump_group_blocks(self.h, group, a.ctypes.data, 1024)
        assert 0 <= n <= 1024
        return a[:n]

    def sorted_varblocks(self):
        cap = max(self.num_varblocks, 1)
        a, f, cs = np.zeros((cap, 8), np.int32), np.zeros((cap, 3), np.float32), np.zeros(28, np.int32)
        n = lib().j40hip_stage_dump_sorted_varblocks(self.h, a.ctypes.data, f.ctypes.data, cs.ctypes.data, cap)
        assert 0 <= n <= cap
        return a[:n], f[:n], cs

    def rgba(self):
        a = np.zeros((self.height, self.width, 4), np.uint8)
        assert lib().j40hip_stage_dump_rgba(self.h, a.ctypes.data) == 0
        return a

    def close(self):
        if self.h:
            lib().j40hip_stage_dump_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """whole-frame throughput pipeline (include/j40hip.h, j40hip_pipeline_*): codestreams in host memory -> RGBA u8x4 in host or
    device memory; host worker threads parse and upload, one thread batches the uploaded frames per entropy launch"""

    def __init__(self, device=0, host_threads=0, batch_frames=32, max_in_flight=2, lf_streams="auto", tune_malloc=False, lf_on_device=None):
        """lf_streams: who decodes the LfGroup streams of the batched frames -- "auto" (decided frame by frame), "device"
        (k_lf_lanes), "host" (the worker threads). `lf_on_device` is the deprecated spelling of rounds 1-2: True = "device",
        False = "host"."""
        if lf_on_device is not None:
            lf_streams = "device" if lf_on_device else "host"
        err = C.c_uint32()
        flags = {"auto": 0, "device": 1, "host": 2}[lf_streams] | (4 if tune_malloc else 0)
        self.h = lib().j40hip_pipeline_create_ex(device, host_threads, batch_frames, max_in_flight, flags, C.byref(err))
        if not self.h:
            raise J40Error(err4(err.value), "in j40hip_pipeline_create")
        self._keep = []

    def lf_device_frames(self):
        return int(lib().j40hip_pipeline_lf_device_frames(self.h))

    def submit(self, data, rgba_ptr, stride_bytes, device_output=False):
        """data: bytes-like (kept alive until close / drain); rgba_ptr: address of the output image; returns the ticket"""
        buf = data if isinstance(data, C.Array) else C.create_string_buffer(bytes(data), len(data))
        self._keep.append(buf)
        t = C.c_int64()
        code = lib().j40hip_pipeline_submit(self.h, buf, len(data), rgba_ptr, stride_bytes, 1 if device_output else 0, C.byref(t))
        if code:
            raise J40Error(err4(code), "in j40hip_pipeline_submit")
        return t.value

    def submit_raw(self, buf, size, rgba_ptr, stride_bytes, device_output=False):
        """as submit, for a ctypes buffer the caller keeps alive until the ticket is done"""
        t = C.c_int64()
        code = lib().j40hip_pipeline_submit(self.h, buf, size, rgba_ptr, stride_bytes, 1 if device_output else 0, C.byref(t))
        if code:
            raise J40Error(err4(code), "in j40hip_pipeline_submit")
        return t.value

    def run(self, data):
        """one image, synchronously (j40hip_pipeline_run): the calling thread sleeps until it is done; any number of threads may call
        this at once and share the pipeline's batches. Returns (err4, rgba ndarray [height, width, 4] or None)."""
        buf = data if isinstance(data, C.Array) else C.create_string_buffer(bytes(data), len(data))
        got = {}

        def alloc(ctx, width, height, stride_out):
            got["a"] = np.empty((int(height), int(width), 4), np.uint8)
            stride_out[0] = int(width) * 4
            return got["a"].ctypes.data

        cb = OUTPUT_ALLOC(alloc)
        code = lib().j40hip_pipeline_run(self.h, buf, len(data), cb, None)
        return err4(code), (got.get("a") if not code else None)

    def set_max_wait_ms(self, ms):
        lib().j40hip_pipeline_set_max_wait_ms(self.h, float(ms))

    def drain(self):
        code = lib().j40hip_pipeline_drain(self.h)
        if code:
            raise J40Error(err4(code), "in j40hip_pipeline_drain")
        self._keep = []

    def result(self, ticket):
        return err4(lib().j40hip_pipeline_result(self.h, ticket))

    def stats(self):
        a = (C.c_double * 12)()
        lib().j40hip_pipeline_stats_ex(self.h, a)
        # (upload_thread_ms: single_thread_ms under the name it had until round 3)
        b = (C.c_double * 5)()
        lib().j40hip_pipeline_lf_stats(self.h, b)
        return dict(parse_thread_ms=a[0], single_thread_ms=a[1], upload_thread_ms=a[1], completed=int(a[2]), wall_ms=a[3], k1_ms=a[4], k2_ms=a[5], launches=int(a[6]), launch_frames=int(a[7]),
                    lf_plan_ms=a[8], lf_device_frames=int(a[9]), single_frames=int(a[10]), k1_kernel_ms=a[11],
                    lf_kernel_ms=b[0], lf_launches=int(b[1]), lf_launch_frames=int(b[2]), lf_launch_sections=int(b[3]), lf_launch_waves=int(b[4]))

    def reset_stats(self):
        lib().j40hip_pipeline_reset_stats(self.h)

    def close(self):
        if self.h:
            lib().j40hip_pipeline_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
