"""ctypes driver of the C++ plugin surface (media_amd/lib/libVideoCodec.so) through the
flat shim in media_amd/host/capi_shim.cpp: CreateVideoEncoder -> VideoEncoder virtuals.
Used by tests to exercise the drop-in boundary the way the reference's caller would."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libVideoCodec.so")

# EncoderRetCode (include/VideoCodecApi.h)
SUCCESS, CREATE_FAIL, INIT_FAIL, START_FAIL, ENCODE_FAIL, STOP_FAIL, DESTROY_FAIL, REGISTER_FAIL, RESET_FAIL, \
    FORCE_KEY_FRAME_FAIL, SET_ENCODE_PARAMS_FAIL = range(11)

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("host library missing: %s (run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        L.vc_create.argtypes = [C.POINTER(vp)]
        for n in ("vc_delete", "vc_init", "vc_start", "vc_stop", "vc_reset"):
            getattr(L, n).argtypes = [vp]
            getattr(L, n).restype = C.c_uint32
        L.vc_create.restype = C.c_uint32
        L.vc_destroy.argtypes = [vp]
        L.vc_destroy.restype = None
        L.vc_encode.argtypes = [vp, vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_uint32)]
        L.vc_encode.restype = C.c_uint32
        L.vc_last_qp.argtypes = [vp]
        L.vc_scene_cuts.argtypes = [vp]
        L.vc_scene_cuts.restype = C.c_uint32
        L.vc_debug_recon_y.argtypes = [vp, vp, C.c_uint64, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.vc_debug_recon_y.restype = C.c_int64
        L.vc_prop_set.argtypes = [C.c_char_p, C.c_char_p]
        L.vc_prop_set.restype = None
        L.vc_prop_get_int.argtypes = [C.c_char_p]
        L.vc_prop_get_str.argtypes = [C.c_char_p, C.c_char_p, C.c_int32]
        _lib = L
    return _lib


def prop_set(key, value):
    lib().vc_prop_set(key.encode(), str(value).encode())


def prop_get(key):
    buf = C.create_string_buffer(256)
    lib().vc_prop_get_str(key.encode(), buf, 256)
    return buf.value.decode()


def set_video_mode(width, height, fps=30, bitrate=5000000, gop=30, profile="baseline", fmt=3, qp=None, slices=None):
    """fill the property store the way a 'video' mode cloud phone would"""
    prop_set("ro.vmi.demo.video.encode.format", fmt)
    prop_set("ro.sys.vmi.cloudphone", "video")
    prop_set("ro.hardware.width", width)
    prop_set("ro.hardware.height", height)
    prop_set("ro.hardware.fps", fps)
    prop_set("persist.vmi.video.encode.bitrate", bitrate)
    prop_set("persist.vmi.video.encode.gopsize", gop)
    prop_set("persist.vmi.video.encode.profile", profile)
    prop_set("persist.vmi.video.encode.param_adjusting", "0")
    prop_set("persist.vmi.video.encode.keyframe", "0")
    prop_set("persist.vmi.video.encode.qp", "" if qp is None else qp)
    prop_set("persist.vmi.video.encode.scenedetect", "1")
    prop_set("persist.vmi.video.encode.slices", "" if slices is None else slices)


class VideoEncoder:
    """Create -> Init -> Start -> Encode x N -> Stop -> Destroy -> delete"""

    def __init__(self):
        self.h = C.c_void_p()
        self.rc_create = lib().vc_create(C.byref(self.h))

    def init(self):
        return lib().vc_init(self.h)

    def start(self):
        return lib().vc_start(self.h)

    def encode(self, data, size=None):
        import numpy as np
        a = np.ascontiguousarray(data, dtype=np.uint8)
        out, n = C.c_void_p(), C.c_uint32()
        rc = lib().vc_encode(self.h, a.ctypes.data, a.size if size is None else size, C.byref(out), C.byref(n))
        return rc, (C.string_at(out.value, n.value) if rc == SUCCESS else b"")

    def stop(self):
        return lib().vc_stop(self.h)

    def destroy(self):
        lib().vc_destroy(self.h)

    def reset(self):
        return lib().vc_reset(self.h)

    def last_qp(self):
        return lib().vc_last_qp(self.h)

    def scene_cuts(self):
        return lib().vc_scene_cuts(self.h)

    def recon_y(self):
        """luma reconstruction of the last picture, coded size (measurement hook)"""
        import numpy as np
        cw, ch = C.c_int32(0), C.c_int32(0)
        buf = np.zeros(4096 * 4096, np.uint8)
        n = lib().vc_debug_recon_y(self.h, buf.ctypes.data, buf.size, C.byref(cw), C.byref(ch))
        if n < 0:
            raise RuntimeError("recon read failed")
        return buf[:n].reshape(ch.value, cw.value)

    def delete(self):
        rc = lib().vc_delete(self.h)
        self.h = C.c_void_p()
        return rc
