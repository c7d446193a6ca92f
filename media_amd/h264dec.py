"""ctypes binding of the decoder peer's C ABI (include/mi355x_h264_dec.h): `Decoder` (GPU reconstruction; fails loudly without
a device) and `Parser` (the host-side CAVLC / header parser alone, usable without a GPU: the CPU tests compare what it
recovers from a stream with the side information of the encoder that wrote it)."""
import ctypes as C
import numpy as np
from .capi import lib, EncoderError, MBINFO_DTYPE, LV_STRIDE

E_STREAM = -7
_bound = False


def _bind():
    global _bound
    L = lib()
    if _bound:
        return L
    vp, sz, ip = C.c_void_p, C.c_size_t, C.POINTER(C.c_int)
    L.mi355x_h264_dec_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.mi355x_h264_dec_destroy.argtypes = [vp]; L.mi355x_h264_dec_destroy.restype = None
    L.mi355x_h264_dec_last_error.argtypes = [vp]; L.mi355x_h264_dec_last_error.restype = C.c_char_p
    L.mi355x_h264_dec_decode.argtypes = [vp, vp, sz, ip]
    L.mi355x_h264_dec_picture_info.argtypes = [vp, ip, ip, ip, ip]
    L.mi355x_h264_dec_read_i420.argtypes = [vp, vp, sz]; L.mi355x_h264_dec_read_i420.restype = C.c_int64
    L.mi355x_h264_dec_read_i420_device.argtypes = [vp, vp, sz]; L.mi355x_h264_dec_read_i420_device.restype = C.c_int64
    L.mi355x_h264_dec_debug_plane.argtypes = [vp, C.c_int, vp, sz]; L.mi355x_h264_dec_debug_plane.restype = C.c_int64
    L.mi355x_h264_dec_sync.argtypes = [vp]
    L.mi355x_h264_dec_timing.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.mi355x_h264_parser_create.restype = vp
    L.mi355x_h264_parser_destroy.argtypes = [vp]; L.mi355x_h264_parser_destroy.restype = None
    L.mi355x_h264_parser_parse.argtypes = [vp, vp, sz]
    L.mi355x_h264_parser_error.argtypes = [vp]; L.mi355x_h264_parser_error.restype = C.c_char_p
    L.mi355x_h264_parser_info.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    L.mi355x_h264_parser_read.argtypes = [vp, C.c_int, vp, sz]; L.mi355x_h264_parser_read.restype = C.c_int64
    _bound = True
    return L


class StreamError(EncoderError):
    """the access unit is damaged or outside the supported feature set"""


class Decoder:
    def __init__(self, device=0):
        L = _bind()
        self.h = C.c_void_p()
        rc = L.mi355x_h264_dec_create(device, C.byref(self.h))
        if rc != 0:
            raise EncoderError("mi355x_h264_dec_create -> %d (no HIP device? there is no CPU reconstruction path)" % rc)

    def close(self):
        if self.h:
            lib().mi355x_h264_dec_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def decode(self, au):
        """one access unit (bytes); True when a picture was decoded"""
        got = C.c_int(0)
        buf = (C.c_uint8 * len(au)).from_buffer_copy(au)
        rc = lib().mi355x_h264_dec_decode(self.h, buf, len(au), C.byref(got))
        if rc != 0:
            msg = lib().mi355x_h264_dec_last_error(self.h).decode()
            raise (StreamError if rc == E_STREAM else EncoderError)("decode -> %d: %s" % (rc, msg))
        return bool(got.value)

    def info(self):
        v = [C.c_int(0) for _ in range(4)]
        if lib().mi355x_h264_dec_picture_info(self.h, *[C.byref(x) for x in v]) != 0:
            raise EncoderError("no picture decoded yet")
        return tuple(x.value for x in v)   # width, height, coded width, coded height

    def i420(self):
        w, h, _, _ = self.info()
        a = np.empty(w * h * 3 // 2, np.uint8)
        n = lib().mi355x_h264_dec_read_i420(self.h, a.ctypes.data, a.nbytes)
        if n != a.nbytes:
            raise EncoderError("read_i420 -> %d" % n)
        return a

    def i420_device(self, ptr, cap):
        return lib().mi355x_h264_dec_read_i420_device(self.h, ptr, cap)

    def plane(self, p):
        """coded-size plane p of the last picture"""
        _, _, cw, ch = self.info()
        a = np.empty((ch // (2 if p else 1), cw // (2 if p else 1)), np.uint8)
        n = lib().mi355x_h264_dec_debug_plane(self.h, p, a.ctypes.data, a.nbytes)
        if n != a.nbytes:
            raise EncoderError("debug_plane -> %d" % n)
        return a

    def sync(self):
        """wait for the picture the last decode() launched"""
        rc = lib().mi355x_h264_dec_sync(self.h)
        if rc != 0:
            raise EncoderError("dec_sync -> %d: %s" % (rc, lib().mi355x_h264_dec_last_error(self.h).decode()))

    def timing(self):
        n, a, b = C.c_uint64(0), C.c_double(0), C.c_double(0)
        lib().mi355x_h264_dec_timing(self.h, C.byref(n), C.byref(a), C.byref(b))
        return n.value, a.value, b.value


class Parser:
    INFO = ("mbw", "mbh", "width", "height", "idr", "qp", "slice_rows", "deblock_idc", "num_ref_active", "t8x8_mode", "has_pcm", "kinds",
            "cqo_cb", "cqo_cr", "filter_oa", "filter_ob", "one_qp", "ref_age0", "ref_age1", "ref_age2")

    def __init__(self):
        self.h = C.c_void_p(_bind().mi355x_h264_parser_create())

    def close(self):
        if self.h:
            lib().mi355x_h264_parser_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def parse(self, au):
        buf = (C.c_uint8 * len(au)).from_buffer_copy(au)
        rc = lib().mi355x_h264_parser_parse(self.h, buf, len(au))
        if rc < 0:
            raise StreamError(lib().mi355x_h264_parser_error(self.h).decode())
        return rc == 1

    def info(self):
        v = (C.c_int32 * 20)()
        lib().mi355x_h264_parser_info(self.h, v, 20)
        return dict(zip(self.INFO, list(v)))

    def vectors4(self):
        """(vectors of the sixteen 4x4 blocks (n, 16, 2), raster order; ref_idx_l0 of the four quadrants (n, 4), 255 = intra)"""
        i = self.info()
        n = i["mbw"] * i["mbh"]
        return self._read(5, np.empty((n, 16, 2), np.int16)), self._read(6, np.empty((n, 4), np.uint8))

    def mbqp(self):
        """QP_Y of every macroblock of the last picture (0 for I_PCM)"""
        i = self.info()
        return self._read(4, np.empty(i["mbw"] * i["mbh"], np.uint8))

    def _read(self, what, arr):
        n = lib().mi355x_h264_parser_read(self.h, what, arr.ctypes.data, arr.nbytes)
        if n != arr.nbytes:
            raise EncoderError("parser_read(%d) -> %d" % (what, n))
        return arr

    def arrays(self):
        i = self.info()
        n = i["mbw"] * i["mbh"]
        return (self._read(0, np.empty(n, MBINFO_DTYPE)), self._read(1, np.empty((n, 8), np.int16)),
                self._read(2, np.empty((n, 16), np.uint8)), self._read(3, np.empty((n, LV_STRIDE), np.int16)))
