"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol
include/mi355x_h264.h declares, refuses to run without a device (no CPU fallback), and the
C++ plugin surface reproduces the reference wrapper's behaviour up to the point where a
device is needed."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def built():
    import __graft_entry__ as g
    g.build()


def test_every_declared_symbol_is_exported():
    from media_amd import capi
    hdr = open(os.path.join(ROOT, "include", "mi355x_h264.h")).read() + open(os.path.join(ROOT, "include", "mi355x_h264_dec.h")).read()
    declared = set(re.findall(r"\b(mi355x_h264_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 30 and "mi355x_h264_dec_decode" in declared and "mi355x_h264_parser_parse" in declared
    out = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = set(re.findall(r" T (mi355x_h264_[a-z_0-9]+)", out))
    assert declared <= exported, "missing: %s" % sorted(declared - exported)
    assert set(capi.EXPORTS) <= exported
    L = capi.lib()
    assert L.mi355x_h264_abi_version() == 3   # 2: config.refs (round 2); 3: stats carry the coded-macroblock counts (round 3)
    cfg = capi.Config()
    L.mi355x_h264_default_config(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(capi.Config) and (cfg.width, cfg.height, cfg.gop) == (720, 1280, 30)


def test_plugin_library_exports_factory():
    from media_amd import videocodec
    out = subprocess.check_output(["nm", "-D", "--defined-only", videocodec.LIB_PATH]).decode()
    for sym in ("CreateVideoEncoder", "DestroyVideoEncoder", "SetMediaLogCallback"):
        assert re.search(r" T %s\b" % sym, out), sym
    # vtable-compatible subclass present
    assert "VideoEncoderMI355X" in subprocess.check_output(["nm", "-DC", videocodec.LIB_PATH]).decode()


def test_decoder_plugin_library_exports_factory():
    from media_amd import videodecoder
    out = subprocess.check_output(["nm", "-D", "--defined-only", videodecoder.LIB_PATH]).decode()
    for sym in ("CreateVideoDecoder", "DestroyVideoDecoder"):
        assert re.search(r" T %s\b" % sym, out), sym
    assert "VideoDecoderMI355X" in subprocess.check_output(["nm", "-DC", videodecoder.LIB_PATH]).decode()


def _no_gpu():
    import torch
    return not torch.cuda.is_available()


@pytest.mark.skipif(not _no_gpu(), reason="checks the no-device failure mode")
def test_no_cpu_fallback_without_device():
    from media_amd import capi
    cfg = capi.Config()
    capi.lib().mi355x_h264_default_config(C.byref(cfg))
    h = C.c_void_p()
    assert capi.lib().mi355x_h264_create(C.byref(cfg), C.byref(h)) == -2  # MI355X_H264_E_NODEVICE
    assert not h.value
    with pytest.raises(capi.EncoderError):
        capi.Encoder(320, 240)
    # the decoder peer likewise: no device, no decoder (the host parser alone is not a decoder)
    from media_amd import h264dec, videodecoder as vd
    with pytest.raises(capi.EncoderError):
        h264dec.Decoder()
    d = vd.PluginDecoder()
    assert d.rc_create == vd.SUCCESS and d.create_decoder() == vd.SUCCESS and d.start() == vd.START_FAIL
    assert d.delete() == vd.SUCCESS


def test_create_rejects_bad_config():
    from media_amd import capi
    L = capi.lib()
    h = C.c_void_p()
    for field, val in (("width", 15), ("height", 4098), ("width", 641), ("qp", 9), ("qp", 52), ("gop", 0), ("profile_idc", 88),
                       ("input_format", 2), ("slices", -1), ("slices", 65), ("band_index", -1), ("band_count", -2)):
        cfg = capi.Config()
        L.mi355x_h264_default_config(C.byref(cfg))
        setattr(cfg, field, val)
        assert L.mi355x_h264_create(C.byref(cfg), C.byref(h)) == -1, field
    cfg = capi.Config()
    L.mi355x_h264_default_config(C.byref(cfg))
    cfg.struct_size = 12
    assert L.mi355x_h264_create(C.byref(cfg), C.byref(h)) == -1
    assert L.mi355x_h264_create(None, C.byref(h)) == -1


def test_property_store_semantics():
    from media_amd import videocodec as vc
    vc.prop_set("persist.vmi.video.encode.bitrate", "3000000")
    assert vc.lib().vc_prop_get_int(b"persist.vmi.video.encode.bitrate") == 3000000
    vc.prop_set("x.junk", "abc")
    # same stringstream parse as the reference (Property.cpp:8-20): an unset / empty property leaves the
    # initial -1, while non-numeric text makes operator>> store 0 (C++11 num_get semantics)
    assert vc.lib().vc_prop_get_int(b"x.junk") == 0
    assert vc.lib().vc_prop_get_int(b"x.never.set") == -1
    os.environ["X_FROM_ENV"] = "42"
    assert vc.lib().vc_prop_get_int(b"x.from.env") == 42        # env seeding


def test_factory_and_init_error_paths():
    from media_amd import videocodec as vc
    # unknown / unbuilt backends -> CREATE_FAIL (VideoCodecApi.cpp:36-38)
    for fmt in (7, -1, 1, 2):
        vc.set_video_mode(1280, 720, fmt=fmt)
        e = vc.VideoEncoder()
        assert e.rc_create == vc.CREATE_FAIL
    # type 0 = OpenH264 through libopenh264.so, bound at InitEncoder like the reference (:197-226): with the
    # backend compiled in (OpenH264 ABI headers present at build time) the object is created and InitEncoder fails
    # because no such library exists on this pool (:203-208); without the headers the slot is unbuilt
    vc.set_video_mode(1280, 720, fmt=0)
    e = vc.VideoEncoder()
    assert e.rc_create in (vc.SUCCESS, vc.CREATE_FAIL)
    if e.rc_create == vc.SUCCESS:
        import ctypes.util
        if ctypes.util.find_library("openh264") is None:
            assert e.init() == vc.INIT_FAIL
        e.destroy()
        assert e.delete() == vc.SUCCESS
    assert vc.lib().vc_delete(None) == vc.SUCCESS              # DestroyVideoEncoder(nullptr) (:48-51)
    vc.set_video_mode(1280, 720, fmt=3)
    e = vc.VideoEncoder()
    assert e.rc_create == vc.SUCCESS
    # invalid ro parameters -> INIT_FAIL (VideoEncoderOpenH264.cpp:159-171)
    for w, h, fps in ((8, 720, 30), (1280, 5000, 30), (1280, 720, 25)):
        vc.set_video_mode(w, h, fps=fps, fmt=3)
        assert e.init() == vc.INIT_FAIL
    vc.set_video_mode(1280, 720, fmt=3)
    vc.prop_set("ro.sys.vmi.cloudphone", "bogus")
    assert e.init() == vc.INIT_FAIL
    # invalid persist parameters are NOT an error: last good values are written back (:111-115)
    vc.set_video_mode(1280, 720, fmt=3, bitrate=99, gop=7, profile="weird")
    rc = e.init()
    assert vc.prop_get("persist.vmi.video.encode.bitrate") == "5000000"
    assert vc.prop_get("persist.vmi.video.encode.gopsize") == "30"
    assert vc.prop_get("persist.vmi.video.encode.profile") == "baseline"
    if _no_gpu():
        assert rc == vc.INIT_FAIL                                # no device: fails loudly, no CPU path
    # short input is refused before anything else (:307-310); destroy is idempotent (:381)
    e.destroy()
    e.destroy()
    assert e.start() == vc.SUCCESS and e.stop() == vc.SUCCESS   # log-only no-ops (:298-302, :367-371)
    assert e.delete() == vc.SUCCESS
