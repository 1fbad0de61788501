"""ORACLE bindings (test infrastructure). ctypes wrapper around oracle/libjxlo.so, the scalar CPU restatement.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libjxlo.so")
_lib = None


def build():
    r = subprocess.run(["make", "-C", _HERE], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("building the oracle failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = ctypes.CDLL(LIB_PATH)
        L.jxlo_decode.restype = ctypes.c_void_p
        L.jxlo_decode.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
        L.jxlo_error.restype = ctypes.c_char_p
        L.jxlo_error.argtypes = [ctypes.c_void_p]
        L.jxlo_buffer.restype = ctypes.c_void_p
        L.jxlo_buffer.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_size_t)]
        L.jxlo_info.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.jxlo_animation.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.jxlo_out_size.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]
        L.jxlo_free.argtypes = [ctypes.c_void_p]
        L.jxlo_dc_params.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        L.jxlo_dc_params.restype = None
        L.jxlo_set_threads.argtypes = [ctypes.c_int]
        L.jxlo_set_threads.restype = ctypes.c_int
        _lib = L
    return _lib


INFO = ("xsize", "ysize", "channels", "modular", "xsize_blocks", "ysize_blocks", "xsize_padded", "ysize_padded",
        "num_groups", "num_dc_groups", "epf_iters", "gab", "num_passes", "used_acs", "bits", "ac_symbols")
_DTYPES = {"rgb8": np.uint8, "rgbf": np.float32, "coeffs": np.int32, "nzeros": np.int32, "xyb_idct": np.float32,
           "xyb_filtered": np.float32, "dc": np.float32, "acs": np.uint8, "quant": np.int32, "sharpness": np.uint8,
           "ytox": np.int8, "ytob": np.int8, "inv_sigma": np.float32, "quant_dc": np.uint8, "modular": np.int32,
           "dc_unsmoothed": np.float32, "alphaf": np.float32}


class Decoded:
    def __init__(self, data, dumps=True, frame=0, prefix=0, preview=False):
        """prefix: draw the frame from the first `prefix` bytes of the codestream alone (FrameDecoder::Flush: AC groups
        whose sections are not whole inside the prefix keep all-zero coefficients); 0 = the whole stream."""
        L = lib()
        data = bytes(data)
        L.jxlo_set_flush_prefix.argtypes = [ctypes.c_size_t]
        L.jxlo_set_flush_prefix.restype = None
        L.jxlo_set_flush_prefix(prefix)
        try:
            self._h = L.jxlo_decode(data, len(data), (1 if dumps else 0) | (2 if preview else 0) | frame << 8)  # preview: the preview frame
        finally:
            L.jxlo_set_flush_prefix(0)
        err = L.jxlo_error(self._h)
        if err:
            msg = err.decode()
            L.jxlo_free(self._h)
            self._h = None
            raise RuntimeError("oracle: " + msg)
        info = (ctypes.c_uint32 * 16)()
        L.jxlo_info(self._h, info)
        self.info = dict(zip(INFO, list(info)))
        wh = (ctypes.c_uint32 * 2)()
        L.jxlo_out_size(self._h, wh)
        self.out_size = (int(wh[0]), int(wh[1]))  # the image: frame size ("xsize", "ysize") times its upsampling factor
        a = (ctypes.c_uint32 * 7)()
        L.jxlo_animation(self._h, a)
        self.animation = dict(zip(("have_animation", "tps_numerator", "tps_denominator", "num_loops", "duration", "is_last", "timecode"),
                                  list(a)))

    @property
    def dc_params(self):
        """Scalars of the DC path: the three DC quantisation steps, quant_scale, epf_quant_mul, epf_sharp_lut[8]."""
        a = (ctypes.c_float * 13)()
        lib().jxlo_dc_params(self._h, a)
        v = [float(x) for x in a]
        return {"dc_step": v[0:3], "quant_scale": v[3], "epf_quant_mul": v[4], "epf_sharp_lut": v[5:13]}

    def buffer(self, name):
        n = ctypes.c_size_t()
        p = lib().jxlo_buffer(self._h, name.encode(), ctypes.byref(n))
        if not p or not n.value:
            return None
        # (a view of the oracle's memory, then one copy: ctypes.string_at cannot make objects of 2 GiB and more, and the
        # coefficients of a 16384x16384 frame are 3 GiB)
        raw = (ctypes.c_uint8 * n.value).from_address(p)
        return np.frombuffer(raw, dtype=_DTYPES[name]).copy()

    @property
    def rgb8(self):
        i = self.info
        return self.buffer("rgb8").reshape(self.out_size[1], self.out_size[0], i["channels"])

    def planes(self, name):
        i = self.info
        a = self.buffer(name)
        if name == "xyb_idct":
            return a.reshape(3, i["ysize_padded"], i["xsize_padded"])
        if name == "xyb_filtered":
            return a.reshape(3, i["ysize"], i["xsize_padded"])
        if name == "rgbf":
            return a.reshape(3, self.out_size[1], self.out_size[0])
        if name == "coeffs":
            return a.reshape(i["num_groups"], 3, 65536)
        raise KeyError(name)

    def close(self):
        if self._h:
            lib().jxlo_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
