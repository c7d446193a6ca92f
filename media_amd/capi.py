"""ctypes binding of the C ABI in include/mi355x_h264.h (media_amd/lib/libmi355x_h264.so).

This is plumbing for tests and bench.py; the product's host side is the C++
VideoEncoderMI355X class in media_amd/host/.  There is no fallback: if the HIP
library is missing or no device is usable, loading / creating raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmi355x_h264.so")
if os.environ.get("MI355X_H264_LIB"):   # A/B measurements: another build of the same library (media_amd/csrc/Makefile target `ab`)
    LIB_PATH = os.environ["MI355X_H264_LIB"]

LV_STRIDE = 416
MBINFO_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("type", "u1"), ("i16_mode", "u1"),
                         ("chroma_mode", "u1"), ("cbp", "u1"), ("tc", "u1", (24,))])
FRAME_IDR, FRAME_P = 1, 3
DBG_RECON_Y, DBG_RECON_U, DBG_RECON_V, DBG_MBINFO, DBG_LEVELS, DBG_PRE_Y, DBG_PRE_U, DBG_PRE_V, DBG_MBAUX, DBG_MVQ = range(10)
K_NAMES = ["me", "tq", "intra", "cavlc", "deblock"]   # index = MI355X_H264_K_* (1: the id is still called K_PMB: k_tq / k_tq8 replaced k_pmb2)

EXPORTS = [
    "mi355x_h264_abi_version", "mi355x_h264_default_config", "mi355x_h264_create", "mi355x_h264_destroy",
    "mi355x_h264_encode", "mi355x_h264_encode_device", "mi355x_h264_encode_batch_device",
    "mi355x_h264_force_idr", "mi355x_h264_last_error", "mi355x_h264_coded_width", "mi355x_h264_coded_height",
    "mi355x_h264_debug_keep_pre", "mi355x_h264_debug_read", "mi355x_h264_stats_enable", "mi355x_h264_stats_read",
    "mi355x_h264_set_qp", "mi355x_h264_set_idr_pic_id", "mi355x_h264_encode_nv12", "mi355x_h264_encode_nv12_device",
    "mi355x_h264_encode_gops_device", "mi355x_h264_last_me_cost", "mi355x_h264_encode_rgba", "mi355x_h264_encode_rgba_device",
    "mi355x_h264_stream_open", "mi355x_h264_stream_close", "mi355x_h264_stream_encode", "mi355x_h264_stream_set_qp",
    "mi355x_h264_stream_force_idr", "mi355x_h264_stream_set_idr_pic_id", "mi355x_h264_stream_last_me_cost",
    "mi355x_h264_stream_last_error", "mi355x_h264_stream_debug_read", "mi355x_h264_stream_hub_stats",
]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32), ("fps", C.c_int32),
                ("bitrate", C.c_int32), ("gop", C.c_int32), ("profile_idc", C.c_int32), ("rc_mode", C.c_int32),
                ("qp", C.c_int32), ("device", C.c_int32), ("disable_deblock", C.c_int32),
                ("batch", C.c_int32), ("input_format", C.c_int32), ("slices", C.c_int32), ("band_index", C.c_int32), ("band_count", C.c_int32), ("refs", C.c_int32), ("search", C.c_int32)]


class Stats(C.Structure):
    _fields_ = [("ms", C.c_double * 5), ("launches", C.c_uint64 * 5), ("mbs", C.c_uint64 * 5),
                ("frames", C.c_uint64), ("p_mbs", C.c_uint64), ("me_searched_mbs", C.c_uint64), ("tq_coded_mbs", C.c_uint64)]


_lib = None


def lib():
    """load the HIP library; raises OSError when it has not been built"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("HIP extension missing: %s (run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.mi355x_h264_default_config.argtypes = [C.POINTER(Config)]
        L.mi355x_h264_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.mi355x_h264_destroy.argtypes = [vp]
        L.mi355x_h264_destroy.restype = None
        L.mi355x_h264_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.POINTER(vp),
                                         C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_device.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_nv12.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_nv12_device.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_rgba.argtypes = [vp, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_rgba_device.argtypes = [vp, vp, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_encode_batch_device.argtypes = [vp, vp, C.c_size_t, C.c_int, vp, C.c_size_t, vp,
                                                      C.POINTER(C.c_size_t)]
        L.mi355x_h264_encode_gops_device.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_int, vp, C.c_size_t, vp, vp]
        L.mi355x_h264_last_me_cost.argtypes = [vp, vp]
        L.mi355x_h264_force_idr.argtypes = [vp]
        L.mi355x_h264_set_qp.argtypes = [vp, C.c_int]
        L.mi355x_h264_set_idr_pic_id.argtypes = [vp, C.c_int, C.c_int]
        L.mi355x_h264_band_info.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        L.mi355x_h264_band_halo_export.argtypes = [vp, C.c_int, vp]
        L.mi355x_h264_band_halo_import.argtypes = [vp, C.c_int, vp]
        L.mi355x_h264_last_error.argtypes = [vp]
        L.mi355x_h264_last_error.restype = C.c_char_p
        L.mi355x_h264_coded_width.argtypes = [vp]
        L.mi355x_h264_coded_height.argtypes = [vp]
        L.mi355x_h264_debug_keep_pre.argtypes = [vp, C.c_int]
        L.mi355x_h264_debug_read.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.mi355x_h264_debug_read.restype = C.c_int64
        L.mi355x_h264_stream_open.argtypes = [C.POINTER(Config), C.POINTER(vp)]
        L.mi355x_h264_stream_close.argtypes = [vp]
        L.mi355x_h264_stream_close.restype = None
        L.mi355x_h264_stream_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.mi355x_h264_stream_set_qp.argtypes = [vp, C.c_int]
        L.mi355x_h264_stream_force_idr.argtypes = [vp]
        L.mi355x_h264_stream_set_idr_pic_id.argtypes = [vp, C.c_int]
        L.mi355x_h264_stream_last_me_cost.argtypes = [vp, C.POINTER(C.c_uint32)]
        L.mi355x_h264_stream_last_error.argtypes = [vp]
        L.mi355x_h264_stream_last_error.restype = C.c_char_p
        L.mi355x_h264_stream_coded_width.argtypes = [vp]
        L.mi355x_h264_stream_coded_height.argtypes = [vp]
        L.mi355x_h264_stream_debug_read.argtypes = [vp, C.c_int, vp, C.c_size_t]
        L.mi355x_h264_stream_debug_read.restype = C.c_int64
        L.mi355x_h264_stream_hub_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        L.mi355x_h264_stats_enable.argtypes = [vp, C.c_int]
        L.mi355x_h264_stats_read.argtypes = [vp, C.POINTER(Stats), C.c_int]
        _lib = L
    return _lib


class EncoderError(RuntimeError):
    pass


class Encoder:
    """thin object wrapper; argument meaning follows mi355x_h264_config"""

    def __init__(self, width, height, qp=26, gop=30, fps=30, profile_idc=66, device=0, disable_deblock=0,
                 bitrate=5000000, rc_mode=0, batch=1, input_format=0, slices=0, band_index=0, band_count=0, refs=0, search=1):
        L = lib()
        cfg = Config()
        L.mi355x_h264_default_config(C.byref(cfg))
        cfg.width, cfg.height, cfg.qp, cfg.gop, cfg.fps = width, height, qp, gop, fps
        cfg.profile_idc, cfg.device, cfg.disable_deblock = profile_idc, device, disable_deblock
        cfg.bitrate, cfg.rc_mode, cfg.batch = bitrate, rc_mode, batch
        cfg.input_format = input_format   # 0 I420, 1 NV12: layout of pictures handed over in device memory
        cfg.slices = slices               # > 1: that many bands of macroblock rows, one slice NAL unit each
        cfg.band_index, cfg.band_count = band_index, band_count   # band_count > 1: this instance codes its share of the slices
        cfg.refs = refs                                           # reference frames searched (0 / 1: one, the reference preset)
        cfg.search = search                                       # 0 exhaustive integer search, 1 seeded by the previous picture's vector (the default)
        self.batch = batch
        self.h = C.c_void_p()
        rc = L.mi355x_h264_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise EncoderError("mi355x_h264_create failed: %d" % rc)
        self.width, self.height = width, height
        self.cw, self.ch = L.mi355x_h264_coded_width(self.h), L.mi355x_h264_coded_height(self.h)
        self.nmb = (self.cw // 16) * (self.ch // 16)

    def _check(self, rc):
        if rc != 0:
            raise EncoderError("rc=%d: %s" % (rc, lib().mi355x_h264_last_error(self.h).decode()))

    def encode(self, i420):
        """host I420 (numpy uint8, width*height*3/2) -> (bytes, frame_type)"""
        w, h = self.width, self.height
        f = np.ascontiguousarray(i420, dtype=np.uint8)
        base = f.ctypes.data
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_encode(self.h, base, w, base + w * h, w // 2, base + w * h * 5 // 4, w // 2,
                                             C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def encode_nv12(self, nv12):
        """host NV12 (Y plane then interleaved UV) -> (bytes, frame_type)"""
        w, h = self.width, self.height
        f = np.ascontiguousarray(nv12, dtype=np.uint8)
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_encode_nv12(self.h, f.ctypes.data, w, f.ctypes.data + w * h, w,
                                                  C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def encode_rgba(self, rgba, stride=None):
        """host RGBA (height x width x 4 bytes, or rows `stride` bytes apart) -> (bytes, frame_type)"""
        f = np.ascontiguousarray(rgba, dtype=np.uint8)
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_encode_rgba(self.h, f.ctypes.data, int(stride or 4 * self.width), C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def encode_rgba_device(self, dev_ptr):
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_encode_rgba_device(self.h, C.c_void_p(dev_ptr), C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def encode_device(self, dev_ptr):
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_encode_device(self.h, C.c_void_p(dev_ptr), C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def encode_batch_device(self, dev_ptr, stride, count, out_buf, sizes):
        """out_buf: numpy uint8 host buffer; sizes: numpy uint32[count]; returns total bytes"""
        tot = C.c_size_t()
        self._check(lib().mi355x_h264_encode_batch_device(self.h, C.c_void_p(dev_ptr), stride, count,
                                                          out_buf.ctypes.data, out_buf.size, sizes.ctypes.data,
                                                          C.byref(tot)))
        return tot.value

    def encode_gops_device(self, dev_ptr, frame_stride, gop_stride, frames_per_gop, out_buf, out_cap_per_gop, sizes, gop_bytes):
        """lockstep encode of `batch` closed GOPs; out_buf uint8[batch*out_cap_per_gop], sizes uint32[batch*frames],
        gop_bytes uint64[batch]"""
        self._check(lib().mi355x_h264_encode_gops_device(self.h, C.c_void_p(dev_ptr), frame_stride, gop_stride, frames_per_gop,
                                                         out_buf.ctypes.data, out_cap_per_gop, sizes.ctypes.data,
                                                         gop_bytes.ctypes.data))

    def me_cost(self):
        a = np.zeros(self.batch, np.uint32)
        self._check(lib().mi355x_h264_last_me_cost(self.h, a.ctypes.data))
        return a

    def force_idr(self):
        self._check(lib().mi355x_h264_force_idr(self.h))

    def set_qp(self, qp):
        self._check(lib().mi355x_h264_set_qp(self.h, qp))

    def set_idr_pic_id(self, nxt, step=1):
        self._check(lib().mi355x_h264_set_idr_pic_id(self.h, nxt, step))

    def band_info(self):
        """(first macroblock row, rows, first slice, slices, halo bytes) of the band this instance codes"""
        r0, rows, s0, ns, hb = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_size_t()
        self._check(lib().mi355x_h264_band_info(self.h, C.byref(r0), C.byref(rows), C.byref(s0), C.byref(ns), C.byref(hb)))
        return r0.value, rows.value, s0.value, ns.value, hb.value

    def halo_export(self, edge, d_dst):
        """edge 0: this band's top rows, 1: its bottom rows, of the newest reconstruction -> device buffer"""
        self._check(lib().mi355x_h264_band_halo_export(self.h, edge, d_dst))

    def halo_import(self, edge, d_src):
        """edge 0: rows right above the band (the upper neighbour's bottom rows), 1: rows right below"""
        self._check(lib().mi355x_h264_band_halo_import(self.h, edge, d_src))

    def keep_pre(self, on=True):
        self._check(lib().mi355x_h264_debug_keep_pre(self.h, int(on)))

    def debug_read(self, what):
        ysz = self.cw * self.ch
        if what in (DBG_RECON_Y, DBG_PRE_Y):
            a = np.empty((self.ch, self.cw), np.uint8)
        elif what in (DBG_RECON_U, DBG_RECON_V, DBG_PRE_U, DBG_PRE_V):
            a = np.empty((self.ch // 2, self.cw // 2), np.uint8)
        elif what == DBG_MBINFO:
            a = np.empty(self.nmb, MBINFO_DTYPE)
        elif what == DBG_MBAUX:
            a = np.empty((self.nmb, 16), np.uint8)
        elif what == DBG_MVQ:
            a = np.empty((self.nmb, 8), np.int16)
        else:
            a = np.empty((self.nmb, LV_STRIDE), np.int16)
        n = lib().mi355x_h264_debug_read(self.h, what, a.ctypes.data, a.nbytes)
        if n != a.nbytes:
            raise EncoderError("debug_read(%d) -> %d" % (what, n))
        return a

    def stats_enable(self, on=True):
        self._check(lib().mi355x_h264_stats_enable(self.h, int(on)))

    def stats(self, reset=True):
        s = Stats()
        self._check(lib().mi355x_h264_stats_read(self.h, C.byref(s), int(reset)))
        return {"frames": s.frames, "p_mbs": s.p_mbs, "me_searched_mbs": s.me_searched_mbs, "tq_coded_mbs": s.tq_coded_mbs,
                "kernels": {K_NAMES[i]: {"ms": s.ms[i], "launches": s.launches[i], "mbs": s.mbs[i]} for i in range(5)}}

    def close(self):
        if getattr(self, "h", None):
            lib().mi355x_h264_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Stream:
    """a stream of the shared engine (include/mi355x_h264.h, "streams"): one picture per call, coded together with the pictures
    other streams of the same geometry deliver at about the same time; thread-safe across streams (one thread per stream)"""

    def __init__(self, width, height, qp=26, gop=30, fps=30, profile_idc=66, device=0, disable_deblock=0, slices=0, search=1):
        L = lib()
        cfg = Config()
        L.mi355x_h264_default_config(C.byref(cfg))
        cfg.width, cfg.height, cfg.qp, cfg.gop, cfg.fps = width, height, qp, gop, fps
        cfg.profile_idc, cfg.device, cfg.disable_deblock, cfg.slices, cfg.search = profile_idc, device, disable_deblock, slices, search
        self.h = C.c_void_p()
        rc = L.mi355x_h264_stream_open(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise EncoderError("mi355x_h264_stream_open failed: %d" % rc)
        self.width, self.height = width, height
        self.cw, self.ch = L.mi355x_h264_stream_coded_width(self.h), L.mi355x_h264_stream_coded_height(self.h)

    def _check(self, rc):
        if rc != 0:
            raise EncoderError("rc=%d: %s" % (rc, lib().mi355x_h264_stream_last_error(self.h).decode()))

    def encode(self, i420):
        w, h = self.width, self.height
        f = np.ascontiguousarray(i420, dtype=np.uint8)
        base = f.ctypes.data
        out, n, ft = C.c_void_p(), C.c_uint32(), C.c_int()
        self._check(lib().mi355x_h264_stream_encode(self.h, base, w, base + w * h, w // 2, base + w * h * 5 // 4, w // 2,
                                                    C.byref(out), C.byref(n), C.byref(ft)))
        return C.string_at(out.value, n.value), ft.value

    def set_qp(self, qp):
        self._check(lib().mi355x_h264_stream_set_qp(self.h, qp))

    def force_idr(self):
        self._check(lib().mi355x_h264_stream_force_idr(self.h))

    def me_cost(self):
        c = C.c_uint32()
        self._check(lib().mi355x_h264_stream_last_me_cost(self.h, C.byref(c)))
        return c.value

    def recon(self, p):
        n = self.cw * self.ch // (4 if p else 1)
        a = np.zeros(n, np.uint8)
        got = lib().mi355x_h264_stream_debug_read(self.h, DBG_RECON_Y + p, a.ctypes.data, a.size)
        if got != n:
            raise EncoderError("stream_debug_read -> %d" % got)
        return a.reshape(self.ch // (2 if p else 1), self.cw // (2 if p else 1))

    def hub_stats(self):
        st, pc, mx, op = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int()
        self._check(lib().mi355x_h264_stream_hub_stats(self.h, C.byref(st), C.byref(pc), C.byref(mx), C.byref(op)))
        return {"steps": st.value, "pictures": pc.value, "max_batch": mx.value, "open_streams": op.value}

    def close(self):
        if getattr(self, "h", None):
            lib().mi355x_h264_stream_close(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
