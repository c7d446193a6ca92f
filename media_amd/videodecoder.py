"""The decoder plugin surface (include/VideoDecoder.h, media_amd/lib/libVideoDecoder.so) driven from Python through the flat
shim in media_amd/host/dec_shim.cpp: CreateVideoDecoder -> CreateDecoder -> InitDecoder -> hooks -> StartDecoder ->
(SendStreamData, RetrieveFrameData) x N -> StopDecoder -> DestroyVideoDecoder, as an OMX component would."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libVideoDecoder.so")
(SUCCESS, CREATE_FAIL, INIT_FAIL, START_FAIL, DECODE_FAIL, STOP_FAIL, DESTROY_FAIL, RESET_FAIL, GET_PARAMS_FAIL, SET_PARAMS_FAIL,
 SET_FUNC_FAIL, WRITE_OVERFLOW, READ_UNDERFLOW, BAD_PIC_SIZE, EOS) = range(15)
STREAM_AVC, STREAM_HEVC = 0, 1
PIXEL_FORMAT_YUV_420P = 1


class Events(C.Structure):
    _fields_ = [("count", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32), ("stride", C.c_uint32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("decoder plugin library missing: %s (run __graft_entry__.build())" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, u32 = C.c_void_p, C.c_uint32
        for name, args in (("vd_create", [C.POINTER(vp)]), ("vd_delete", [vp]), ("vd_create_decoder", [vp, u32]), ("vd_init", [vp]),
                           ("vd_start", [vp]), ("vd_stop", [vp]), ("vd_flush", [vp]), ("vd_send", [vp, vp, u32]),
                           ("vd_retrieve", [vp, vp, u32, C.POINTER(u32)]), ("vd_set_pic_info", [vp, u32, u32, C.c_int32]),
                           ("vd_get_pic_info", [vp, C.POINTER(u32)]), ("vd_get_port_format", [vp, u32, C.POINTER(C.c_int32)]),
                           ("vd_get_align", [vp, C.POINTER(u32)]), ("vd_install_hooks", [vp, C.POINTER(Events)])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = u32
        L.vd_destroy.argtypes = [vp]
        L.vd_destroy.restype = None
        _lib = L
    return _lib


class PluginDecoder:
    def __init__(self):
        self.h = C.c_void_p()
        self.events = Events()
        self.rc_create = lib().vd_create(C.byref(self.h))

    def create_decoder(self, fmt=STREAM_AVC): return lib().vd_create_decoder(self.h, fmt)
    def init(self): return lib().vd_init(self.h)
    def install_hooks(self): return lib().vd_install_hooks(self.h, C.byref(self.events))
    def start(self): return lib().vd_start(self.h)
    def stop(self): return lib().vd_stop(self.h)
    def flush(self): return lib().vd_flush(self.h)
    def set_pic_info(self, w, h, stride=None): return lib().vd_set_pic_info(self.h, w, h, stride if stride is not None else w)

    def pic_info(self):
        v = (C.c_uint32 * 4)()
        lib().vd_get_pic_info(self.h, v)
        return tuple(v)

    def port_format(self, port):
        f = C.c_int32(-1)
        return lib().vd_get_port_format(self.h, port, C.byref(f)), f.value

    def align(self):
        v = (C.c_uint32 * 2)()
        lib().vd_get_align(self.h, v)
        return tuple(v)

    def send(self, au):
        buf = (C.c_uint8 * max(1, len(au))).from_buffer_copy(au if len(au) else b"\x00")
        return lib().vd_send(self.h, buf, len(au))

    def retrieve(self, cap):
        out = np.empty(cap, np.uint8)
        n = C.c_uint32(0)
        rc = lib().vd_retrieve(self.h, out.ctypes.data, cap, C.byref(n))
        return rc, out[: n.value]

    def delete(self):
        if self.h:
            rc = lib().vd_delete(self.h)
            self.h = C.c_void_p()
            return rc
        return SUCCESS
