"""libjxl_amd — MI355X-native JPEG XL VarDCT decode path (Python plumbing over the C ABI).

The product is the shared library ``libjxl_amd/_build/libjxl_amd.so`` (HIP kernels + ``extern "C"`` layer declared in
``include/jxl_amd_hip.h``, ``include/jxl_amd.h`` and ``include/jxl/decode.h``).  This module only binds it with
ctypes for tests and ``bench.py``; it contains no decode logic and there is no CPU fallback: if the library or a HIP
device is missing, calls raise.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libjxl_amd.so")
ENC_PATH = os.path.join(_HERE, "_build", "libjxlenc.so")

_lib = None
_enc = None


class JxlAmdError(RuntimeError):
    pass


def build(verbose=False):
    """Compiles the in-tree shared libraries (hipcc --offload-arch=gfx950)."""
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout[-4000:], r.stderr[-4000:])
    if r.returncode:
        raise JxlAmdError("building libjxl_amd failed")


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise JxlAmdError("%s is missing: run libjxl_amd.build() (hipcc) first; there is no fallback path" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        vp, cp, u32p = ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32)
        L.jxlhip_version.restype = cp
        L.jxlhip_device_count.restype = ctypes.c_int
        L.jxlhip_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
        L.jxlhip_ctx_destroy.argtypes = [vp]
        for name in ("jxlhip_run_entropy", "jxlhip_run_transform", "jxlhip_run_filter_color", "jxlhip_run_all", "jxlhip_sync"):
            getattr(L, name).argtypes = [vp]
        for name in ("jxlhip_run_entropy_batch", "jxlhip_run_transform_batch", "jxlhip_run_filter_color_batch"):
            getattr(L, name).argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        L.jxlhip_download_rgb8.argtypes = [vp, vp, ctypes.c_size_t]
        L.jxlhip_set_option.argtypes = [vp, cp, ctypes.c_int]
        L.jxlhip_share_planes.argtypes = [vp, vp]
        L.jxlhip_download_rgb8_rows.argtypes = [vp, vp, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32]
        L.jxlamd_frame_upload_band.argtypes = [vp, vp, ctypes.c_uint32, ctypes.c_uint32]
        L.jxlhip_rgb8_device_ptr.argtypes = [vp]
        L.jxlhip_rgb8_device_ptr.restype = vp
        L.jxlhip_get_errors.argtypes = [vp, u32p, ctypes.c_size_t]
        L.jxlhip_download.argtypes = [vp, cp, vp, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
        L.jxlhip_last_stage_ms.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
        L.jxlamd_frame_parse.argtypes = [cp, ctypes.c_size_t, vp, vp, ctypes.POINTER(vp)]
        L.jxlamd_frame_free.argtypes = [vp]
        L.jxlamd_frame_info.argtypes = [vp, u32p]
        L.jxlamd_frame_out_size.argtypes = [vp, u32p]
        L.jxlamd_frame_upload.argtypes = [vp, vp]
        L.jxlamd_modframe_parse.argtypes = [cp, ctypes.c_size_t, ctypes.POINTER(vp)]
        L.jxlamd_modframe_free.argtypes = [vp]
        L.jxlamd_modframe_info.argtypes = [vp, u32p]
        L.jxlamd_modframe_upload.argtypes = [vp, vp]
        L.jxlamd_modframe_extra_buffer.argtypes = [vp, ctypes.c_uint32]
        L.jxlamd_modframe_extra_buffer.restype = ctypes.c_uint32
        L.jxlhip_modular_run.argtypes = [vp]
        L.jxlhip_modular_run_batch.argtypes = [ctypes.POINTER(vp), ctypes.c_size_t]
        L.jxlhip_modular_status.argtypes = [vp, u32p, u32p, ctypes.c_size_t]
        L.jxlhip_modular_download_buffer.argtypes = [vp, ctypes.c_uint32, vp, ctypes.c_size_t]
        L.jxlhip_set_output_format.argtypes = [vp, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int]
        L.jxlhip_download_pixels.argtypes = [vp, vp, ctypes.c_size_t]
        L.jxlamd_last_error.restype = cp
        L.JxlThreadParallelRunnerCreate.restype = vp
        L.JxlThreadParallelRunnerCreate.argtypes = [vp, ctypes.c_size_t]
        L.JxlThreadParallelRunnerDestroy.argtypes = [vp]
        _lib = L
    return _lib


def _check(r, what):
    if r != 0:
        err = lib().jxlamd_last_error()
        raise JxlAmdError("%s failed: code %d %s" % (what, r, err.decode() if err else ""))


class Frame:
    """Host-parsed frame (headers, DC, tables); AC sections stay compressed."""

    INFO = ("xsize", "ysize", "xsize_blocks", "ysize_blocks", "num_groups", "num_dc_groups", "num_passes", "used_acs",
            "epf_iters", "gab", "coef_bits", "ac_bytes", "log_alpha", "num_clusters", "ctx_map_size")

    def __init__(self, data, threads=0, frame_pos=0, frame_index=0):
        """frame_pos / frame_index: a later frame of an animation (frame_pos = the `end` of the frame before it)."""
        L = lib()
        self._data = bytes(data)  # must outlive upload
        self._h = ctypes.c_void_p()
        L.jxlamd_frame_parse_at.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_void_p,
                                            ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]
        L.jxlamd_frame_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.jxlamd_frame_end.restype = ctypes.c_size_t
        runner = None
        pool = None
        if threads > 0:
            pool = L.JxlThreadParallelRunnerCreate(None, threads)
            runner = ctypes.cast(L.JxlThreadParallelRunner, ctypes.c_void_p)
        try:
            _check(L.jxlamd_frame_parse_at(self._data, len(self._data), frame_pos, frame_index, runner, pool, ctypes.byref(self._h)),
                   "jxlamd_frame_parse_at")
        finally:
            if pool:
                L.JxlThreadParallelRunnerDestroy(pool)
        info = (ctypes.c_uint32 * 16)()
        L.jxlamd_frame_info(self._h, info)
        self.info = dict(zip(self.INFO, list(info)))
        wh = (ctypes.c_uint32 * 2)()
        L.jxlamd_frame_out_size(self._h, wh)
        self.info["out_xsize"], self.info["out_ysize"] = int(wh[0]), int(wh[1])  # the image (upsampled frames: > frame size)
        t = (ctypes.c_uint32 * 3)()
        self.end = int(L.jxlamd_frame_end(self._h, t))  # byte offset behind the frame
        self.duration, self.is_last, self.timecode = int(t[0]), bool(t[1]), int(t[2])

    def close(self):
        if self._h:
            lib().jxlamd_frame_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModFrame:
    """Host-parsed Modular (lossless) frame: headers, trees, histograms, stream descriptors; samples stay compressed."""

    INFO = ("xsize", "ysize", "num_color", "has_alpha", "bits", "num_streams", "num_buffers", "num_ops", "num_extra", "section_bytes",
            "max_table_words", "lz77", "max_tree_nodes")

    def __init__(self, data, frame_pos=0, frame_index=0):
        L = lib()
        self._data = bytes(data)  # must outlive upload
        self._h = ctypes.c_void_p()
        L.jxlamd_modframe_parse_at.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t,
                                               ctypes.POINTER(ctypes.c_void_p)]
        L.jxlamd_modframe_end.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.jxlamd_modframe_end.restype = ctypes.c_size_t
        _check(L.jxlamd_modframe_parse_at(self._data, len(self._data), frame_pos, frame_index, ctypes.byref(self._h)),
               "jxlamd_modframe_parse_at")
        info = (ctypes.c_uint32 * 16)()
        L.jxlamd_modframe_info(self._h, info)
        self.info = dict(zip(self.INFO, list(info)))
        t = (ctypes.c_uint32 * 3)()
        self.end = int(L.jxlamd_modframe_end(self._h, t))
        self.duration, self.is_last, self.timecode = int(t[0]), bool(t[1]), int(t[2])

    def close(self):
        if self._h:
            lib().jxlamd_modframe_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipContext:
    """One HIP stream + the device buffers of one frame (include/jxl_amd_hip.h)."""

    def __init__(self, device=0):
        L = lib()
        if L.jxlhip_device_count() <= 0:
            raise JxlAmdError("no HIP device visible: the decode path has no CPU implementation")
        self._h = ctypes.c_void_p()
        _check(L.jxlhip_ctx_create(device, ctypes.byref(self._h)), "jxlhip_ctx_create")
        self.frame_info = None

    def close(self):
        if self._h:
            lib().jxlhip_ctx_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def debug_color(self, xyb, linear=False):
        """Test entry: the colour stage alone on XYB triples (array [3, n]) -> float RGB [n, 3]."""
        xyb = np.ascontiguousarray(xyb, np.float32)
        n = xyb.shape[1]
        out = np.empty((n, 3), np.float32)
        L = lib()
        L.jxlhip_debug_color.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
        _check(L.jxlhip_debug_color(self._h, xyb.ctypes.data, n, 1 if linear else 0, out.ctypes.data), "jxlhip_debug_color")
        return out

    def check_guards(self):
        """JXLHIP_GUARD=1 debug aid: 0 when no kernel wrote next to one of this context's device buffers."""
        t = ctypes.c_uint32()
        _check(lib().jxlhip_check_guards(self._h, ctypes.byref(t)), "jxlhip_check_guards")
        return int(t.value)

    def set_option(self, name, value):
        _check(lib().jxlhip_set_option(self._h, name.encode(), int(value)), "jxlhip_set_option")

    def share_planes(self, lender):
        """Keep this context's XYB planes in `lender`'s buffer (None: its own again); call before upload()."""
        _check(lib().jxlhip_share_planes(self._h, lender._h if lender is not None else None), "jxlhip_share_planes")
        self._lender = lender  # keeps the lender alive

    def upload(self, frame, band=None):
        """band = (group_row_begin, group_row_end): produce only those rows of 256x256 groups (multi-GPU split)."""
        b0, b1 = band if band else (0, 0)
        _check(lib().jxlamd_frame_upload_band(frame._h, self._h, b0, b1), "jxlamd_frame_upload_band")
        self.frame_info = dict(frame.info)

    def halo_rows(self):
        """Rows of the neighbouring bands the filters of this band read (after an upload)."""
        L = lib()
        L.jxlhip_halo_rows.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        n = ctypes.c_uint32()
        _check(L.jxlhip_halo_rows(self._h, ctypes.byref(n)), "jxlhip_halo_rows")
        return int(n.value)

    def halo_floats(self):
        """Floats of one halo block ([3][rows][padded xsize])."""
        xp = ((self.frame_info["xsize"] + 7) // 8) * 8
        return 3 * self.halo_rows() * xp

    def halo_pack(self, side, device_ptr, nbytes):
        """side 0: this band's first rows (for the band above), 1: its last rows (for the band below) -> device memory."""
        L = lib()
        L.jxlhip_halo_pack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        _check(L.jxlhip_halo_pack(self._h, side, device_ptr, nbytes), "jxlhip_halo_pack")

    def halo_unpack(self, side, device_ptr, nbytes):
        """side 0: the rows just above this band, 1: just below it <- device memory (what the neighbour packed)."""
        L = lib()
        L.jxlhip_halo_unpack.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        _check(L.jxlhip_halo_unpack(self._h, side, device_ptr, nbytes), "jxlhip_halo_unpack")

    def set_output_format(self, data_type=2, num_channels=3, bits=0, big_endian=False):
        """JxlDataType numbering: 0 f32, 2 u8, 3 u16, 5 f16; call before upload."""
        _check(lib().jxlhip_set_output_format(self._h, data_type, num_channels, bits, 1 if big_endian else 0), "jxlhip_set_output_format")
        self._out = (data_type, num_channels)

    def set_output_orientation(self, orientation=1):
        """Undo this image orientation (1..8, EXIF numbering) in the pixel writer; call before upload."""
        L = lib()
        L.jxlhip_set_output_orientation.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        _check(L.jxlhip_set_output_orientation(self._h, int(orientation)), "jxlhip_set_output_orientation")
        self._transposed = orientation > 4

    def enc_rerun(self, times=1):
        """Measurement: the forward-path kernels of the last encode_rgb8_gpu on this context, `times` more times on the
        resident input. Returns (ms of all passes, ms of the transform kernel of the last pass)."""
        L = lib()
        L.jxlhip_enc_forward_rerun.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        L.jxlhip_enc_last_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.jxlhip_enc_last_transform_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        _check(L.jxlhip_enc_forward_rerun(self._h, times), "jxlhip_enc_forward_rerun")
        a, b = ctypes.c_float(), ctypes.c_float()
        _check(L.jxlhip_enc_last_ms(self._h, ctypes.byref(a)), "jxlhip_enc_last_ms")
        _check(L.jxlhip_enc_last_transform_ms(self._h, ctypes.byref(b)), "jxlhip_enc_last_transform_ms")
        return a.value, b.value

    def upload_modular(self, mframe):
        _check(lib().jxlamd_modframe_upload(mframe._h, self._h), "jxlamd_modframe_upload")
        self.frame_info = dict(mframe.info)
        self.frame_info["out_xsize"], self.frame_info["out_ysize"] = mframe.info["xsize"], mframe.info["ysize"]

    def run_modular(self):
        _check(lib().jxlhip_modular_run(self._h), "jxlhip_modular_run")

    def modular_status(self):
        n = self.frame_info["num_streams"]
        st = (ctypes.c_uint32 * max(n, 1))()
        eb = (ctypes.c_uint32 * max(n, 1))()
        r = lib().jxlhip_modular_status(self._h, st, eb, n)
        return r, list(st)[:n], list(eb)[:n]

    def modular_buffer(self, index):
        need = self.frame_info["xsize"] * self.frame_info["ysize"]
        out = np.empty(need, np.int32)
        _check(lib().jxlhip_modular_download_buffer(self._h, index, out.ctypes.data, need), "jxlhip_modular_download_buffer")
        return out.reshape(self.frame_info["ysize"], self.frame_info["xsize"])

    def pixels(self):
        """The interleaved result in the format of set_output_format (default RGB8)."""
        fi = self.frame_info
        dt, nc = getattr(self, "_out", (2, 3))
        np_dt = {0: np.float32, 2: np.uint8, 3: np.uint16, 5: np.float16}[dt]
        oxs, oys = fi["out_xsize"], fi["out_ysize"]
        if getattr(self, "_transposed", False):
            oxs, oys = oys, oxs
        out = np.empty((oys, oxs, nc), np_dt)
        _check(lib().jxlhip_download_pixels(self._h, out.ctypes.data, oxs * nc * out.itemsize), "jxlhip_download_pixels")
        return out

    def rgb8_rows(self, y0, y1):
        fi = self.frame_info
        out = np.empty((y1 - y0, fi["out_xsize"], 3), np.uint8)
        _check(lib().jxlhip_download_rgb8_rows(self._h, out.ctypes.data, fi["out_xsize"] * 3, y0, y1), "jxlhip_download_rgb8_rows")
        return out

    def run_entropy(self):
        _check(lib().jxlhip_run_entropy(self._h), "jxlhip_run_entropy")

    def run_transform(self):
        _check(lib().jxlhip_run_transform(self._h), "jxlhip_run_transform")

    def run_filter_color(self):
        _check(lib().jxlhip_run_filter_color(self._h), "jxlhip_run_filter_color")

    def run_all(self):
        _check(lib().jxlhip_run_all(self._h), "jxlhip_run_all")

    def sync(self):
        _check(lib().jxlhip_sync(self._h), "jxlhip_sync")

    def errors(self):
        n = self.frame_info["num_groups"]
        flags = (ctypes.c_uint32 * n)()
        r = lib().jxlhip_get_errors(self._h, flags, n)
        return r, list(flags)

    def stage_ms(self, which):
        ms = ctypes.c_float()
        _check(lib().jxlhip_last_stage_ms(self._h, which, ctypes.byref(ms)), "jxlhip_last_stage_ms")
        return ms.value

    def rgb8(self):
        fi = self.frame_info
        out = np.empty((fi["out_ysize"], fi["out_xsize"], 3), np.uint8)
        _check(lib().jxlhip_download_rgb8(self._h, out.ctypes.data, fi["out_xsize"] * 3), "jxlhip_download_rgb8")
        return out

    def download(self, name):
        fi = self.frame_info
        need = ctypes.c_size_t()
        _check(lib().jxlhip_download(self._h, name.encode(), None, 0, ctypes.byref(need)), "jxlhip_download")
        buf = np.empty(need.value, np.uint8)
        _check(lib().jxlhip_download(self._h, name.encode(), buf.ctypes.data, need.value, None), "jxlhip_download")
        if name == "coeffs":
            dt = np.int16 if fi["coef_bits"] == 16 else np.int32
            return buf.view(dt).reshape(fi["num_groups"], 3, 65536)
        if name in ("dc", "inv_sigma"):  # block-resolution planes as the transform / filter stages read them
            return buf.view(np.float32)
        return buf.view(np.float32).reshape(3, fi["ysize_blocks"] * 8, fi["xsize_blocks"] * 8)


def run_entropy_batch(ctxs):
    """Entropy stage of several resident frames as one kernel launch (jxlhip_run_entropy_batch)."""
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().jxlhip_run_entropy_batch(arr, len(ctxs)), "jxlhip_run_entropy_batch")


def halo_pack_batch(ctxs, side, device_ptr, block_bytes, transport_stream=0):
    """jxlhip_halo_pack_batch: block i (frame i of the set) at device_ptr + i * block_bytes; whatever is enqueued on
    `transport_stream` (a hipStream_t as an integer; 0 = the null stream) afterwards follows the copies. No host wait."""
    L = lib()
    L.jxlhip_halo_pack_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(L.jxlhip_halo_pack_batch(arr, len(ctxs), side, device_ptr, block_bytes, transport_stream), "jxlhip_halo_pack_batch")


def halo_unpack_batch(ctxs, side, device_ptr, block_bytes, transport_stream=0):
    """jxlhip_halo_unpack_batch: the copies follow what `transport_stream` holds now (the receive); the set's filter launch
    follows the copies. No host wait."""
    L = lib()
    L.jxlhip_halo_unpack_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(L.jxlhip_halo_unpack_batch(arr, len(ctxs), side, device_ptr, block_bytes, transport_stream), "jxlhip_halo_unpack_batch")


def run_transform_batch(ctxs):
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().jxlhip_run_transform_batch(arr, len(ctxs)), "jxlhip_run_transform_batch")


def run_filter_color_batch(ctxs):
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().jxlhip_run_filter_color_batch(arr, len(ctxs)), "jxlhip_run_filter_color_batch")


def run_modular_batch(ctxs):
    """Every stream of several resident Modular frames as one launch (jxlhip_modular_run_batch)."""
    arr = (ctypes.c_void_p * len(ctxs))(*[c._h for c in ctxs])
    _check(lib().jxlhip_modular_run_batch(arr, len(ctxs)), "jxlhip_modular_run_batch")


def decode_lossless(data, num_channels=3, data_type=2, device=0):
    """One-shot helper: Modular (lossless) codestream -> HxWxC samples through the GPU path."""
    f = ModFrame(data)
    c = HipContext(device)
    try:
        c.set_output_format(data_type, num_channels)
        c.upload_modular(f)
        c.run_modular()
        r, status, _ = c.modular_status()
        if r:
            raise JxlAmdError("corrupt Modular streams: %r" % [(i, s) for i, s in enumerate(status) if s])
        return c.pixels()
    finally:
        c.close()
        f.close()


def decode_rgb8(data, device=0, threads=0):
    """One-shot helper: codestream bytes -> HxWx3 uint8 through the GPU path."""
    f = Frame(data, threads)
    c = HipContext(device)
    try:
        c.upload(f)
        c.run_all()
        r, flags = c.errors()
        if r:
            raise JxlAmdError("corrupt AC sections: %r" % [(g, e) for g, e in enumerate(flags) if e])
        return c.rgb8()
    finally:
        c.close()
        f.close()


# ----------------------------------------------------------------------------------------------- synthetic streams
class EncParams(ctypes.Structure):
    _fields_ = [("distance", ctypes.c_float), ("epf_iters", ctypes.c_int32), ("gab", ctypes.c_int32),
                ("strategy_mode", ctypes.c_int32), ("strategy_mask", ctypes.c_uint32), ("seed", ctypes.c_uint32),
                ("max_clusters", ctypes.c_int32), ("skip_dc_smoothing", ctypes.c_int32), ("random_cmap", ctypes.c_int32),
                ("zero_ac", ctypes.c_int32), ("num_histograms", ctypes.c_int32), ("big_coeffs", ctypes.c_int32), ("num_passes", ctypes.c_int32), ("upsampling", ctypes.c_int32), ("custom_orders", ctypes.c_int32), ("custom_bctx", ctypes.c_int32),
                ("custom_cmap", ctypes.c_int32), ("custom_lf", ctypes.c_int32), ("ac_code_mode", ctypes.c_int32),
                ("noise", ctypes.c_int32), ("cfl_fit", ctypes.c_int32), ("color_transform", ctypes.c_int32), ("raw_quant", ctypes.c_int32),
                ("chroma_subsampling", ctypes.c_int32), ("ec_upsampling", ctypes.c_int32)]


def _enc_lib():
    global _enc
    if _enc is None:
        if not os.path.exists(ENC_PATH):
            raise JxlAmdError("%s is missing: run libjxl_amd.build()" % ENC_PATH)
        E = ctypes.CDLL(ENC_PATH)
        pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))
        E.jxlenc_encode_rgb8.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(EncParams), pp,
                                         ctypes.POINTER(ctypes.c_size_t)]
        E.jxlenc_encode_rgba8.argtypes = E.jxlenc_encode_rgb8.argtypes
        E.jxlenc_encode_lossless.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                             ctypes.c_uint32, pp, ctypes.POINTER(ctypes.c_size_t)]
        E.jxlenc_encode_lossless_samples.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                                     ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, pp, ctypes.POINTER(ctypes.c_size_t)]
        E.jxlenc_encode_random.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(EncParams), pp,
                                           ctypes.POINTER(ctypes.c_size_t)]
        E.jxlenc_synth_image.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p]
        E.jxlenc_free.argtypes = [ctypes.POINTER(ctypes.c_uint8)]
        _enc = E
    return _enc


def _params(**kw):
    p = EncParams()
    p.distance, p.epf_iters, p.gab, p.strategy_mode, p.strategy_mask, p.seed = 1.0, -1, -1, 1, 0, 1
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def icc_decode(coded):
    """Coded ICC profile (the stream behind an image's headers) -> (profile bytes, exact length of the coded form in bits)."""
    L = lib()
    L.jxlamd_icc_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t),
                                    ctypes.POINTER(ctypes.c_size_t)]
    n, bits = ctypes.c_size_t(), ctypes.c_size_t()
    _check(L.jxlamd_icc_decode(coded, len(coded), None, 0, ctypes.byref(n), ctypes.byref(bits)), "jxlamd_icc_decode")
    out = ctypes.create_string_buffer(max(1, n.value))
    _check(L.jxlamd_icc_decode(coded, len(coded), out, n.value, ctypes.byref(n), ctypes.byref(bits)), "jxlamd_icc_decode")
    return out.raw[:n.value], bits.value


def set_custom_upsampling(mask=0, seed=1):
    """Test aid: the next VarDCT streams code their own upsampling weights (CustomTransformData, image_metadata.cc:87-214):
    bit k of `mask` = the 2^(k+1)-fold weight matrix is coded (seeded variations of the default weights); 0 = default again."""
    E = _enc_lib()
    E.jxlenc_set_custom_upsampling.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    E.jxlenc_set_custom_upsampling.restype = None
    E.jxlenc_set_custom_upsampling(mask, seed)


def set_embedded_icc(coded=None):
    """Test aid: the synthetic encoders embed this coded ICC profile (want_icc) in the streams they write from now on
    (None: no profile again)."""
    E = _enc_lib()
    E.jxlenc_set_embedded_icc.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_size_t]
    if not coded:
        E.jxlenc_set_embedded_icc(b"", 0, 0)
        return
    _, bits = icc_decode(coded)
    E.jxlenc_set_embedded_icc(coded, len(coded), bits)


def set_color_encoding(white_point=None, primaries=1, transfer_function=13, gamma=None, intent=1, xy=None):
    """Test aid: the next LOSSLESS streams declare this enum colour encoding (values of jxl/color_encoding.h: white point
    1 D65 / 2 custom / 10 E / 11 DCI; primaries 1 sRGB / 2 custom / 9 Rec.2100 / 11 P3; transfer function 1 709 / 8 linear /
    13 sRGB / 16 PQ / 17 DCI / 18 HLG, or gamma as a float; xy = 8 custom chromaticities white, red, green, blue).
    white_point=None: sRGB again."""
    E = _enc_lib()
    E.jxlenc_set_color_encoding.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 6 + [ctypes.POINTER(ctypes.c_int32)]
    E.jxlenc_set_color_encoding.restype = None
    if white_point is None:
        E.jxlenc_set_color_encoding(0, 1, 1, 0, 0, 13, 1, None)
        return
    arr = (ctypes.c_int32 * 8)(*[int(round(v * 1e6)) for v in (xy or [0] * 8)])
    E.jxlenc_set_color_encoding(1, white_point, primaries, 1 if gamma is not None else 0, int(round((gamma or 0) * 1e7)), transfer_function,
                                intent, arr)


def set_frame_name(name=""):
    """Test aid: the frames written from now on carry this name (empty: none again)."""
    E = _enc_lib()
    E.jxlenc_set_frame_name.argtypes = [ctypes.c_char_p]
    E.jxlenc_set_frame_name.restype = None
    E.jxlenc_set_frame_name(name.encode())


def set_orientation(orientation=1):
    """Test aid: the synthetic encoders declare this image orientation (1..8, EXIF numbering) in the streams they write
    from now on (1: none again)."""
    E = _enc_lib()
    E.jxlenc_set_orientation.argtypes = [ctypes.c_uint32]
    E.jxlenc_set_orientation.restype = None
    E.jxlenc_set_orientation(int(orientation))


def set_splines(splines=None, quantization_adjustment=0):
    """Test aid: the next streams carry these quantised splines (None: none again). Each spline is a dict: points (list of
    integer (x, y) control points), color ([3][32] integer DCT coefficients of X, Y, B along the arc) and sigma ([32])."""
    E = _enc_lib()
    E.jxlenc_set_splines.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t]
    E.jxlenc_set_splines.restype = None
    if not splines:
        E.jxlenc_set_splines(None, 0)
        return
    flat = [int(quantization_adjustment), len(splines)]
    for sp in splines:
        pts = [(int(x), int(y)) for x, y in sp["points"]]
        flat += [pts[0][0], pts[0][1], len(pts) - 1]
        pdx = pdy = 0
        for (x0, y0), (x1, y1) in zip(pts, pts[1:]):  # double deltas (splines.cc:411-431)
            dx, dy = x1 - x0, y1 - y0
            flat += [dx - pdx, dy - pdy]
            pdx, pdy = dx, dy
        for c in range(3):
            flat += [int(v) for v in sp["color"][c]]
        flat += [int(v) for v in sp["sigma"]]
    arr = (ctypes.c_int32 * len(flat))(*flat)
    E.jxlenc_set_splines(arr, len(flat))


def encode_animation(frames, durations, tps=(10, 1), num_loops=0, lossless=False, **kw):
    """Test aid: an animation of full-size frames that replace each other (no layers, blending, crops or references):
    frames[i] (HxWx3 or HxWx4 uint8, all of one size) is shown for durations[i] ticks of tps[0] / tps[1] per second."""
    E = _enc_lib()
    E.jxlenc_set_animation.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 4 + [ctypes.c_int]
    E.jxlenc_set_animation.restype = None
    E.jxlenc_last_header_bytes.restype = ctypes.c_size_t
    out = b""
    try:
        for i, (img, dur) in enumerate(zip(frames, durations)):
            E.jxlenc_set_animation(1, tps[0], tps[1], num_loops, dur, 1 if i == len(frames) - 1 else 0)
            if lossless:
                d = encode_lossless(img, **kw)
            elif img.shape[2] == 4:
                d = encode_rgba8(img, **kw)
            else:
                d = encode_rgb8(img, **kw)
            out += d if i == 0 else d[E.jxlenc_last_header_bytes():]
    finally:
        E.jxlenc_set_animation(0, 10, 1, 0, 0, 1)
    return out


def encode_layers(layers, tps=None, num_loops=0, lossless=False, premultiplied=False, **kw):
    """Test aid: a codestream of several frames composed on a canvas (blending.cc): layers[i] is a dict with img (HxWx3 or
    HxWx4 uint8; layers[0] covers the whole canvas at the origin and gives its size) and optionally x0, y0 (crop origin),
    mode / alpha_mode (BlendMode of colour / alpha: 0 replace, 1 add, 2 blend, 3 alpha-weighted add, 4 multiply), source /
    alpha_source (reference slots), clamp, save_as (slot the blended canvas is kept in) and duration. tps = (numerator,
    denominator) makes it an animation (frames with duration > 0 are shown); tps = None a layered still (only the last
    frame is shown)."""
    E = _enc_lib()
    E.jxlenc_set_animation.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 4 + [ctypes.c_int]
    E.jxlenc_set_animation.restype = None
    E.jxlenc_set_layer.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_uint32] * 8 + [ctypes.c_int]
    E.jxlenc_set_layer.restype = None
    E.jxlenc_last_header_bytes.restype = ctypes.c_size_t
    ch, cw = layers[0]["img"].shape[:2]
    out = b""
    E.jxlenc_set_alpha_premultiplied.argtypes = [ctypes.c_int]
    E.jxlenc_set_alpha_premultiplied.restype = None
    E.jxlenc_set_alpha_premultiplied(1 if premultiplied else 0)  # (the alpha channel is declared associated; samples as given)
    try:
        for i, L in enumerate(layers):
            img = L["img"]
            E.jxlenc_set_animation(1, tps[0] if tps else 10, tps[1] if tps else 1, num_loops, L.get("duration", 0),
                                   1 if i == len(layers) - 1 else 0)
            E.jxlenc_set_layer(1, L.get("x0", 0), L.get("y0", 0), cw, ch, L.get("mode", 0), L.get("alpha_mode", 0), L.get("source", 0),
                               L.get("alpha_source", L.get("source", 0)), L.get("clamp", 0), L.get("save_as", 0), 1 if tps else 0)
            if lossless:
                d = encode_lossless(img, **kw)
            elif img.shape[2] == 4:
                d = encode_rgba8(img, **kw)
            else:
                d = encode_rgb8(img, **kw)
            out += d if i == 0 else d[E.jxlenc_last_header_bytes():]
    finally:
        E.jxlenc_set_layer(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1)
        E.jxlenc_set_animation(0, 10, 1, 0, 0, 1)
        E.jxlenc_set_alpha_premultiplied(0)
    return out


def encode_with_preview(img, preview, **kw):
    """Test aid: `img` (HxWx3 uint8, VarDCT) with a VarDCT preview frame (hxwx3 uint8):
    the image header announces the preview's size (headers.cc:155-183) and the preview is the codestream's first frame, a
    regular frame that is not the last (decode.cc:1266-1268)."""
    E = _enc_lib()
    E.jxlenc_set_animation.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 4 + [ctypes.c_int]
    E.jxlenc_set_animation.restype = None
    E.jxlenc_set_preview.argtypes = [ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
    E.jxlenc_set_preview.restype = None
    E.jxlenc_last_header_bytes.restype = ctypes.c_size_t
    E.jxlenc_set_layer.argtypes = [ctypes.c_int, ctypes.c_int32, ctypes.c_int32] + [ctypes.c_uint32] * 8 + [ctypes.c_int]
    E.jxlenc_set_layer.restype = None
    try:
        E.jxlenc_set_animation(1, 10, 1, 0, 0, 0)  # (multi-frame ...
        E.jxlenc_set_layer(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0)  # ... untimed: a frame header with is_last = false, no duration)
        p = encode_rgb8(preview)
        pframe = p[E.jxlenc_last_header_bytes():]
        E.jxlenc_set_animation(0, 10, 1, 0, 0, 1)
        E.jxlenc_set_preview(1, preview.shape[1], preview.shape[0])
        m = encode_rgb8(img, **kw)
        h = E.jxlenc_last_header_bytes()
    finally:
        E.jxlenc_set_animation(0, 10, 1, 0, 0, 1)
        E.jxlenc_set_layer(0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1)
        E.jxlenc_set_preview(0, 0, 0)
    return m[:h] + pframe + m[h:]


def encode_patched(img, atlas, patches, slot=1, atlas_vardct=False, lossless=False, premultiplied=False, lossless_flags=None, **kw):
    """Test aid: a reference-only frame holding `atlas` (HxWx3 uint8; coded as an XYB Modular frame like libjxl's patch
    frames, or as a VarDCT frame) kept in `slot`, then `img` coded with a patch dictionary. patches: list of dicts with
    x0, y0, xsize, ysize (rectangle in the atlas) and positions: list of (x, y, mode, clamp[, alpha mode, alpha clamp]) with
    PatchBlendMode 0 none, 1 replace, 2 add, 3 multiply, 4 / 5 blend above / below, 6 / 7 alpha-weighted add above / below
    (dec_patch_dictionary.h:32-58); the last two apply to the image's alpha channel (RGBA `img` and `atlas`, VarDCT atlas).
    The patches are drawn over the decoded frame; nothing is subtracted when encoding."""
    E = _enc_lib()
    E.jxlenc_set_reference_frame.argtypes = [ctypes.c_int]
    E.jxlenc_set_reference_frame.restype = None
    E.jxlenc_set_image_size.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    E.jxlenc_set_image_size.restype = None
    E.jxlenc_set_patches.argtypes = [ctypes.POINTER(ctypes.c_int32), ctypes.c_size_t]
    E.jxlenc_set_patches.restype = None
    E.jxlenc_set_animation.argtypes = [ctypes.c_int] + [ctypes.c_uint32] * 4 + [ctypes.c_int]
    E.jxlenc_set_animation.restype = None
    E.jxlenc_last_header_bytes.restype = ctypes.c_size_t
    flat = [len(patches)]
    for p in patches:
        flat += [slot, p["x0"], p["y0"], p["xsize"], p["ysize"], len(p["positions"])]
        for pos in p["positions"]:
            flat += list(pos) + [0] * (6 - len(pos))
    E.jxlenc_set_alpha_premultiplied.argtypes = [ctypes.c_int]
    E.jxlenc_set_alpha_premultiplied.restype = None
    try:
        E.jxlenc_set_alpha_premultiplied(1 if premultiplied else 0)  # (the alpha channel is declared associated; samples as given)
        E.jxlenc_set_image_size(img.shape[1], img.shape[0])
        E.jxlenc_set_reference_frame(slot)
        mflags = MODULAR_XYB if lossless_flags is None else lossless_flags  # (lossless_flags: both frames Modular, e.g. plain RGB with an RCT)
        first = (encode_rgba8(atlas, **kw) if atlas.shape[2] == 4 else encode_rgb8(atlas, **kw)) if atlas_vardct else encode_lossless(atlas, mflags)
        E.jxlenc_set_reference_frame(-1)
        arr = (ctypes.c_int32 * len(flat))(*flat)
        E.jxlenc_set_patches(arr, len(flat))
        second = encode_lossless(img, mflags) if lossless else (encode_rgba8(img, **kw) if img.shape[2] == 4 else encode_rgb8(img, **kw))
        return first + second[E.jxlenc_last_header_bytes():]
    finally:
        E.jxlenc_set_alpha_premultiplied(0)
        E.jxlenc_set_reference_frame(-1)
        E.jxlenc_set_patches(None, 0)
        E.jxlenc_set_image_size(0, 0)


def encode_with_dc_frame(img, dc_vardct=False, **kw):
    """Test aid: what `cjxl --progressive_dc` lays out: a kDCFrame of level 1 (the image's 8x8 block means, 1/8 size; coded as
    an XYB Modular frame like libjxl's DC frames, or as a VarDCT frame), then `img` as a VarDCT frame with kUseDcFrame, whose
    DC groups carry no DC stream: its DC image is the DC frame's output (frame_header.h:348, passes_state.cc:62-77)."""
    E = _enc_lib()
    E.jxlenc_set_image_size.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    E.jxlenc_set_image_size.restype = None
    E.jxlenc_set_dc_frame.argtypes = [ctypes.c_int]
    E.jxlenc_set_dc_frame.restype = None
    E.jxlenc_set_use_dc_frame.argtypes = [ctypes.c_int]
    E.jxlenc_set_use_dc_frame.restype = None
    E.jxlenc_last_header_bytes.restype = ctypes.c_size_t
    h, w = img.shape[:2]
    xb, yb = (w + 7) // 8, (h + 7) // 8
    pad = np.pad(img, ((0, yb * 8 - h), (0, xb * 8 - w), (0, 0)), mode="edge")
    small = np.ascontiguousarray(pad.reshape(yb, 8, xb, 8, 3).mean(axis=(1, 3)).round().astype(np.uint8))
    try:
        E.jxlenc_set_image_size(w, h)
        E.jxlenc_set_dc_frame(1)
        first = encode_rgb8(small, **kw) if dc_vardct else encode_lossless(small, MODULAR_XYB)
        E.jxlenc_set_dc_frame(0)
        E.jxlenc_set_use_dc_frame(1)
        second = encode_rgb8(img, **kw)
        return first + second[E.jxlenc_last_header_bytes():]
    finally:
        E.jxlenc_set_dc_frame(0)
        E.jxlenc_set_use_dc_frame(0)
        E.jxlenc_set_image_size(0, 0)


def synth_image(xsize, ysize, seed=177):
    """Deterministic synthetic RGB8 test image (gradient background, rectangles, discs, texture, noise)."""
    a = np.zeros((ysize, xsize, 3), np.uint8)
    _enc_lib().jxlenc_synth_image(xsize, ysize, seed, a.ctypes.data)
    return a


def _finish(E, r, out, n, what):
    if r:
        raise JxlAmdError("%s failed: %d" % (what, r))
    b = ctypes.string_at(out, n.value)
    E.jxlenc_free(out)
    return b


def encode_rgb8(img, **kw):
    """RGB8 image -> VarDCT codestream (distance=1.0, strategy_mode=1 by default)."""
    E = _enc_lib()
    img = np.ascontiguousarray(img, np.uint8)
    p = _params(**kw)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = E.jxlenc_encode_rgb8(img.tobytes(), img.shape[1], img.shape[0], ctypes.byref(p), ctypes.byref(out), ctypes.byref(n))
    return _finish(E, r, out, n, "jxlenc_encode_rgb8")


def encode_rgb8_gpu(img, ctx, timings=None, device_tokens=False, **kw):
    """VarDCT-encodes an RGB8 image with the pixel-domain half (colour, sharpening, transform selection, forward DCT,
    quantisation) on the GPU (jxlhip_enc_forward on `ctx`) and entropy coding / headers on the host. `timings` (a dict)
    receives forward_s (the call, copies included), assemble_s and kernels_ms (HIP events around the launches).
    device_tokens: the coefficients are tokenised on the device too (jxlhip_enc_tokens) and never copied to the host."""
    E, L = _enc_lib(), lib()
    pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))
    E.jxlenc_encode_rgb8_forward.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(EncParams), ctypes.c_void_p,
                                             ctypes.c_void_p, pp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double)]
    L.jxlhip_enc_last_ms.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
    img = np.ascontiguousarray(img, np.uint8)
    p = _params(**kw)
    out, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t()
    secs = (ctypes.c_double * 3)()
    fn = ctypes.cast(L.jxlhip_enc_forward, ctypes.c_void_p)
    if device_tokens:
        E.jxlenc_encode_rgb8_forward_tokens.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(EncParams)] + \
            [ctypes.c_void_p] * 4 + [pp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double)]
        r = E.jxlenc_encode_rgb8_forward_tokens(img.tobytes(), img.shape[1], img.shape[0], ctypes.byref(p), fn,
                                                ctypes.cast(L.jxlhip_enc_token_counts, ctypes.c_void_p),
                                                ctypes.cast(L.jxlhip_enc_tokens, ctypes.c_void_p), ctx._h, ctypes.byref(out), ctypes.byref(n), secs)
    else:
        r = E.jxlenc_encode_rgb8_forward(img.tobytes(), img.shape[1], img.shape[0], ctypes.byref(p), fn, ctx._h, ctypes.byref(out),
                                         ctypes.byref(n), secs)
    data = _finish(E, r, out, n, "jxlenc_encode_rgb8_forward")
    if timings is not None:
        ms = ctypes.c_float()
        _check(L.jxlhip_enc_last_ms(ctx._h, ctypes.byref(ms)), "jxlhip_enc_last_ms")
        timings.update(forward_s=secs[0], assemble_s=secs[1], kernels_ms=ms.value, device_tokens=int(secs[2]))
    return data


def enc_forward_model(img, ctx=None, **kw):
    """Test access: the raw outputs of one forward call (acs, qf, dc, coeffs) from the GPU path on `ctx`, or from the
    CPU stream writer's own model code when ctx is None."""
    E = _enc_lib()
    i32p = ctypes.POINTER(ctypes.c_int32)
    E.jxlenc_forward_model.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(EncParams), ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    img = np.ascontiguousarray(img, np.uint8)
    ys, xs = img.shape[:2]
    xb, yb, ng = (xs + 7) // 8, (ys + 7) // 8, ((xs + 255) // 256) * ((ys + 255) // 256)
    acs, qf = np.zeros((yb, xb), np.uint8), np.zeros((yb, xb), np.int32)
    dc, co = np.zeros((3, yb, xb), np.int32), np.zeros((ng, 3, 65536), np.int32)
    p = _params(**kw)
    fn = ctypes.cast(lib().jxlhip_enc_forward if ctx is not None else E.jxlenc_forward_cpu, ctypes.c_void_p)
    r = E.jxlenc_forward_model(img.tobytes(), xs, ys, ctypes.byref(p), fn, ctx._h if ctx is not None else None, acs.ctypes.data,
                               qf.ctypes.data, dc.ctypes.data, co.ctypes.data)
    if r:
        raise JxlAmdError("jxlenc_forward_model failed (%d)" % r)
    return dict(acs=acs, qf=qf, dc=dc, coeffs=co)


def encode_rgba8(img, **kw):
    """RGBA8 image -> VarDCT codestream with alpha as a (lossless) Modular-coded extra channel."""
    E = _enc_lib()
    img = np.ascontiguousarray(img, np.uint8)
    assert img.ndim == 3 and img.shape[2] == 4
    p = _params(**kw)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = E.jxlenc_encode_rgba8(img.tobytes(), img.shape[1], img.shape[0], ctypes.byref(p), ctypes.byref(out), ctypes.byref(n))
    return _finish(E, r, out, n, "jxlenc_encode_rgba8")


LOSSLESS_PREFIX, LOSSLESS_LZ77, LOSSLESS_WP, LOSSLESS_SQUEEZE, LOSSLESS_RCT, LOSSLESS_ALL_PREDICTORS, LOSSLESS_PREV_CHANNEL = 1, 2, 4, 8, 16, 32, 64
MODULAR_XYB = 128  # the frame codes XYB integers (Y, X, B - Y): "lossy Modular", not lossless any more


def encode_lossless(img, flags=LOSSLESS_RCT, seed=0):
    """HxWxC uint8 image (C = 1..4: grey, grey + alpha, RGB, RGBA) -> lossless Modular codestream. flags: LOSSLESS_*."""
    E = _enc_lib()
    img = np.ascontiguousarray(img, np.uint8)
    if img.ndim == 2:
        img = img[..., None]
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = E.jxlenc_encode_lossless(img.tobytes(), img.shape[1], img.shape[0], img.shape[2], flags, seed, ctypes.byref(out), ctypes.byref(n))
    return _finish(E, r, out, n, "jxlenc_encode_lossless")


def encode_lossless_samples(samples, bits, exp_bits=0, flags=0, seed=0):
    """HxWxC int32 samples (C = 1..4) -> lossless Modular codestream of an image with `bits`-bit integer samples, or
    (exp_bits != 0) float samples of `bits` bits whose bit patterns the integers are (32 / 8: float32 viewed as int32,
    16 / 5: float16 viewed as uint16). An alpha channel (C = 2 or 4) stays 8-bit."""
    E = _enc_lib()
    a = np.ascontiguousarray(samples, np.int32)
    if a.ndim == 2:
        a = a[..., None]
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = E.jxlenc_encode_lossless_samples(a.ctypes.data, a.shape[1], a.shape[0], a.shape[2], flags, seed, bits, exp_bits,
                                         ctypes.byref(out), ctypes.byref(n))
    return _finish(E, r, out, n, "jxlenc_encode_lossless_samples")


def encode_random(xsize, ysize, **kw):
    """Random valid VarDCT codestream exercising the strategies in strategy_mask (0 = all 27)."""
    E = _enc_lib()
    p = _params(**kw)
    out = ctypes.POINTER(ctypes.c_uint8)()
    n = ctypes.c_size_t()
    r = E.jxlenc_encode_random(xsize, ysize, ctypes.byref(p), ctypes.byref(out), ctypes.byref(n))
    return _finish(E, r, out, n, "jxlenc_encode_random")
