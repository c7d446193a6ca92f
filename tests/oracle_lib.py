"""ctypes binding of oracle/liboracle_h264.so -- the CHECKER, test-side only."""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ODIR = os.path.join(os.path.dirname(_HERE), "oracle")
_SO = os.path.join(_ODIR, "liboracle_h264.so")

LV_STRIDE = 416
LV_LUMA_DC, LV_LUMA, LV_CHROMA_DC, LV_CHROMA_AC = 0, 16, 272, 280


class Config(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("fps", C.c_int32), ("qp", C.c_int32),
                ("gop", C.c_int32), ("profile_idc", C.c_int32), ("disable_deblock", C.c_int32), ("slices", C.c_int32), ("band_index", C.c_int32), ("band_count", C.c_int32), ("refs", C.c_int32), ("search", C.c_int32)]


MBINFO_DTYPE = np.dtype([("mvx", "<i2"), ("mvy", "<i2"), ("type", "u1"), ("i16_mode", "u1"),
                         ("chroma_mode", "u1"), ("cbp", "u1"), ("tc", "u1", (24,))])
assert MBINFO_DTYPE.itemsize == 32


def _build():
    srcs = [os.path.join(_ODIR, f) for f in os.listdir(_ODIR) if f.endswith((".c", ".h"))]
    if not os.path.exists(_SO) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ODIR, "-s"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        _build()
        L = C.CDLL(_SO)
        vp, u8p = C.c_void_p, C.POINTER(C.c_uint8)
        L.h264o_enc_create.restype = vp
        L.h264o_enc_create.argtypes = [C.POINTER(Config)]
        L.h264o_enc_destroy.argtypes = [vp]
        L.h264o_enc_encode.restype = C.c_int64
        L.h264o_enc_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_size_t,
                                       C.POINTER(C.c_int)]
        for n in ("h264o_enc_coded_width", "h264o_enc_coded_height"):
            getattr(L, n).argtypes = [vp]
        for n in ("h264o_enc_recon", "h264o_enc_recon_pre"):
            getattr(L, n).restype = vp
            getattr(L, n).argtypes = [vp, C.c_int]
        L.h264o_enc_mbinfo.restype = vp
        L.h264o_enc_mbinfo.argtypes = [vp]
        L.h264o_enc_mbaux.restype = vp
        L.h264o_enc_mvq.restype = vp
        L.h264o_enc_mvq.argtypes = [vp]
        L.h264o_enc_mbaux.argtypes = [vp]
        L.h264o_enc_levels.restype = vp
        L.h264o_enc_levels.argtypes = [vp]
        L.h264o_enc_set_qp.argtypes = [vp, C.c_int]
        L.h264o_enc_last_me_cost.argtypes = [vp]
        L.h264o_enc_last_me_cost.restype = C.c_uint32
        L.h264o_enc_set_idr_id.argtypes = [vp, C.c_int, C.c_int]
        L.h264o_enc_halo_bytes.argtypes = [vp]
        L.h264o_enc_halo_bytes.restype = C.c_size_t
        L.h264o_enc_halo_export.argtypes = [vp, C.c_int, vp]
        L.h264o_enc_halo_export.restype = None
        L.h264o_enc_halo_import.argtypes = [vp, C.c_int, vp]
        L.h264o_enc_halo_import.restype = None
        L.h264o_enc_random_picture.restype = C.c_int64
        L.h264o_enc_random_picture.argtypes = [vp, C.c_uint32, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int), vp]
        L.h264o_enc_last_slice_bits.restype = C.c_int64
        L.h264o_enc_last_slice_bits.argtypes = [vp]
        L.h264o_dec_create.restype = vp
        L.h264o_dec_destroy.argtypes = [vp]
        L.h264o_dec_decode.argtypes = [vp, vp, C.c_size_t]
        for n in ("h264o_dec_width", "h264o_dec_height", "h264o_dec_coded_width", "h264o_dec_coded_height",
                  "h264o_dec_last_slice_type", "h264o_dec_last_nal_type"):
            getattr(L, n).argtypes = [vp]
        L.h264o_dec_plane.restype = vp
        L.h264o_dec_plane.argtypes = [vp, C.c_int]
        L.h264o_dec_error.restype = C.c_char_p
        L.h264o_dec_error.argtypes = [vp]
        for n in ("h264o_dec_max_mb_bits", "h264o_dec_max_level_prefix"):
            getattr(L, n).argtypes = [vp]
        L.h264o_dec_mb_kind.argtypes = [vp, C.c_int]
        L.h264o_dec_mb_qp.argtypes = [vp, C.c_int]
        L.h264o_dec_mb_mv.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        ip, up = C.POINTER(C.c_int), C.POINTER(C.c_uint)
        L.h264o_dec_table_coeff_token.argtypes = [C.c_int, C.c_int, C.c_int, ip, up]
        L.h264o_dec_table_total_zeros.argtypes = [C.c_int, C.c_int, C.c_int, ip, up]
        L.h264o_dec_table_run_before.argtypes = [C.c_int, C.c_int, ip, up]
        L.h264o_dec_table_misc.argtypes = [C.c_int, C.c_int, C.c_int]
        L.h264o_fdct4x4.argtypes = [vp, vp]
        L.h264o_rgba_to_i420.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.h264o_rgba_to_i420.restype = None
        L.h264o_idct4x4_add.argtypes = [vp, vp, C.c_int]
        L.h264o_quant4x4.argtypes = [vp, C.c_int, C.c_int, vp]
        L.h264o_dequant4x4.argtypes = [vp, C.c_int, vp]
        L.h264o_mc_luma.argtypes = [vp] + [C.c_int] * 9 + [vp, C.c_int]
        L.h264o_mc_chroma.argtypes = [vp] + [C.c_int] * 9 + [vp, C.c_int]
        L.h264o_luma_sample_ref.argtypes = [vp] + [C.c_int] * 7
        for n in ("h264o_sad16x16", "h264o_satd16x16", "h264o_satd8x8"):
            getattr(L, n).argtypes = [vp, C.c_int, vp, C.c_int]
        L.h264o_pred16x16.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.h264o_pred_chroma8x8.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp]
        L.h264o_deblock_picture.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, C.c_int, vp, C.c_int, C.c_int]
        L.h264o_ue_bits.argtypes = [C.c_uint32, C.POINTER(C.c_uint32)]
        L.h264o_se_bits.argtypes = [C.c_int32, C.POINTER(C.c_uint32)]
        L.h264o_cavlc_block.argtypes = [vp, C.c_int, C.c_int, vp]
        L.h264o_nal_escape.restype = C.c_size_t
        L.h264o_nal_escape.argtypes = [vp, C.c_size_t, vp]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def rgba_to_i420(rgba, width, height, stride=None):
    """the oracle's RGBA ingest (oracle/h264_rgba.c): uint8 RGBA rows -> tight I420 as a flat uint8 array"""
    a = np.ascontiguousarray(rgba, dtype=np.uint8)
    out = np.zeros(width * height * 3 // 2, np.uint8)
    lib().h264o_rgba_to_i420(a.ctypes.data, int(stride or 4 * width), width, height, out.ctypes.data)
    return out


class OracleEncoder:
    def __init__(self, width, height, qp=26, gop=30, fps=30, profile_idc=66, disable_deblock=0, slices=0, band_index=0, band_count=0, refs=0, search=1):
        self.cfg = Config(width, height, fps, qp, gop, profile_idc, disable_deblock, slices, band_index, band_count, refs, search)
        self.h = lib().h264o_enc_create(C.byref(self.cfg))
        if not self.h:
            raise ValueError("oracle rejected config")
        self.width, self.height = width, height
        self.cw = lib().h264o_enc_coded_width(self.h)
        self.ch = lib().h264o_enc_coded_height(self.h)
        self.out = np.zeros(width * height * 8 + (1 << 16), dtype=np.uint8)

    def encode(self, i420, force_idr=False):
        w, h = self.width, self.height
        f = np.ascontiguousarray(i420, dtype=np.uint8)
        y, u, v = f[: w * h], f[w * h: w * h * 5 // 4], f[w * h * 5 // 4:]
        idr = C.c_int(0)
        n = lib().h264o_enc_encode(self.h, _ptr(y), w, _ptr(u), w // 2, _ptr(v), w // 2, int(force_idr),
                                   _ptr(self.out), self.out.size, C.byref(idr))
        if n < 0:
            raise RuntimeError("oracle encode failed %d" % n)
        return bytes(self.out[:n]), bool(idr.value)

    RAND_QP, RAND_CHROMA_OFF, RAND_FILTER_OFF, RAND_PCM, RAND_IDC, RAND_SUBPARTS, RAND_SLICES, RAND_REORDER, RAND_OPENH264_HEADERS, RAND_BIG_LEVELS, RAND_CONSTRAINED_INTRA, RAND_ALL = 1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2047

    def random_picture(self, seed, force_idr=False, features=31):
        """one picture of random conforming syntax (h264o_enc_random_picture): (access unit, is_idr, QP_Y per macroblock);
        mbinfo() / mvq() / mbaux() / levels() then hold what was written"""
        idr = C.c_int(0)
        n_mb = (self.cw // 16) * (self.ch // 16)
        mbqp = np.zeros(n_mb, dtype=np.uint8)
        n = lib().h264o_enc_random_picture(self.h, int(seed) & 0xFFFFFFFF, int(force_idr), int(features), _ptr(self.out), self.out.size,
                                           C.byref(idr), _ptr(mbqp))
        if n < 0:
            raise RuntimeError("oracle random picture failed %d" % n)
        return bytes(self.out[:n]), bool(idr.value), mbqp

    def _plane(self, fn, p):
        cw, ch = (self.cw, self.ch) if p == 0 else (self.cw // 2, self.ch // 2)
        addr = fn(self.h, p)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(ch, cw)).copy()

    def recon(self, p):
        return self._plane(lib().h264o_enc_recon, p)

    def recon_pre(self, p):
        return self._plane(lib().h264o_enc_recon_pre, p)

    def mbinfo(self):
        n = (self.cw // 16) * (self.ch // 16)
        addr = lib().h264o_enc_mbinfo(self.h)
        raw = np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n * 32,)).copy()
        return raw.view(MBINFO_DTYPE)

    def mbaux(self):
        """16 bytes per macroblock: Intra4x4PredMode of the blocks of an Intra4x4 macroblock (blkIdx order)"""
        n = (self.cw // 16) * (self.ch // 16)
        addr = lib().h264o_enc_mbaux(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(n, 16)).copy()

    def mvq(self):
        """(x, y) vectors of the four 8x8 quadrants of every macroblock (meaningful for inter macroblocks)"""
        n = (self.cw // 16) * (self.ch // 16)
        addr = lib().h264o_enc_mvq(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_int16)), shape=(n, 8)).copy()

    def levels(self):
        n = (self.cw // 16) * (self.ch // 16)
        addr = lib().h264o_enc_levels(self.h)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_int16)), shape=(n, LV_STRIDE)).copy()

    def set_qp(self, qp):
        if lib().h264o_enc_set_qp(self.h, qp) != 0:
            raise ValueError("bad qp")

    def halo_bytes(self):
        return lib().h264o_enc_halo_bytes(self.h)

    def halo_export(self, edge, ptr):
        lib().h264o_enc_halo_export(self.h, edge, ptr)

    def halo_import(self, edge, ptr):
        lib().h264o_enc_halo_import(self.h, edge, ptr)

    def set_idr_id(self, nxt, step=1):
        lib().h264o_enc_set_idr_id(self.h, nxt, step)

    def me_cost(self):
        return lib().h264o_enc_last_me_cost(self.h)

    def slice_bits(self):
        return lib().h264o_enc_last_slice_bits(self.h)

    def close(self):
        if self.h:
            lib().h264o_enc_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class OracleDecoder:
    def __init__(self):
        self.h = lib().h264o_dec_create()

    def decode(self, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        rc = lib().h264o_dec_decode(self.h, _ptr(buf), buf.size)
        if rc < 0:
            raise RuntimeError("decode error: " + lib().h264o_dec_error(self.h).decode())
        return rc

    def plane(self, p):
        cw, ch = lib().h264o_dec_coded_width(self.h), lib().h264o_dec_coded_height(self.h)
        if p:
            cw, ch = cw // 2, ch // 2
        addr = lib().h264o_dec_plane(self.h, p)
        return np.ctypeslib.as_array(C.cast(addr, C.POINTER(C.c_uint8)), shape=(ch, cw)).copy()

    @property
    def size(self):
        return lib().h264o_dec_width(self.h), lib().h264o_dec_height(self.h)

    # statistics of the last decoded picture
    KIND_I4, KIND_I16, KIND_IPCM, KIND_INTER, KIND_SKIP = 1, 2, 3, 4, 5

    def mb_kinds(self):
        n = (lib().h264o_dec_coded_width(self.h) // 16) * (lib().h264o_dec_coded_height(self.h) // 16)
        return np.array([lib().h264o_dec_mb_kind(self.h, i) for i in range(n)], dtype=np.int32)

    def mb_qps(self):
        n = (lib().h264o_dec_coded_width(self.h) // 16) * (lib().h264o_dec_coded_height(self.h) // 16)
        return np.array([lib().h264o_dec_mb_qp(self.h, i) for i in range(n)], dtype=np.int32)

    def mb_mv(self, addr, blk4=0):
        x, y, r = C.c_int(0), C.c_int(0), C.c_int(0)
        lib().h264o_dec_mb_mv(self.h, addr, blk4, C.byref(x), C.byref(y), C.byref(r))
        return x.value, y.value, r.value

    @property
    def max_mb_bits(self):
        return lib().h264o_dec_max_mb_bits(self.h)

    @property
    def max_level_prefix(self):
        return lib().h264o_dec_max_level_prefix(self.h)

    def close(self):
        if self.h:
            lib().h264o_dec_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()
