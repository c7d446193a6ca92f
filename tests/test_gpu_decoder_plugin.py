"""VideoDecoderMI355X behind the reference's VideoDecoder interface (include/VideoDecoder.h; peer of
/root/reference/video_decoder/VideoDecoderNetint.cpp), driven through media_amd/host/dec_shim.cpp."""
import numpy as np
import pytest
from media_amd import synth, videodecoder as vd
from oracle_lib import OracleEncoder

pytestmark = pytest.mark.gpu


def _i420(enc, w, h):
    return np.concatenate([enc.recon(0)[:h, :w].ravel(), enc.recon(1)[: h // 2, : w // 2].ravel(), enc.recon(2)[: h // 2, : w // 2].ravel()])


def test_plugin_decoder_protocol():
    w, h = 208, 160
    enc = OracleEncoder(w, h, qp=28, gop=4, profile_idc=100, refs=2)
    d = vd.PluginDecoder()
    assert d.rc_create == vd.SUCCESS
    assert d.create_decoder(vd.STREAM_HEVC) == vd.CREATE_FAIL and d.create_decoder(vd.STREAM_AVC) == vd.SUCCESS
    assert d.init() == vd.SUCCESS and d.install_hooks() == vd.SUCCESS
    assert d.pic_info() == (1280, 720, 1280, 720)                         # the reference adapter's defaults
    assert d.port_format(1) == (vd.SUCCESS, vd.PIXEL_FORMAT_YUV_420P) and d.port_format(0) == (vd.SUCCESS, vd.STREAM_AVC)
    assert d.port_format(7)[0] == vd.GET_PARAMS_FAIL and d.align() == (2, 2)
    assert d.send(b"\x00\x00\x00\x01\x67") == vd.DECODE_FAIL                 # before StartDecoder: stop state
    assert d.start() == vd.SUCCESS
    cap = w * h * 3 // 2
    assert d.retrieve(cap)[0] == vd.READ_UNDERFLOW
    frames = synth.sequence("cut", w, h, 6)
    for i, f in enumerate(frames):
        au = enc.encode(f)[0]
        assert d.send(au) == vd.SUCCESS
        assert d.send(au) == vd.WRITE_OVERFLOW                               # one picture waits at the output
        rc, out = d.retrieve(cap)
        if i == 0:   # the configured size is still the default: event + BAD_PIC_SIZE, the picture stays
            assert rc == vd.BAD_PIC_SIZE and d.events.count == 1 and (d.events.width, d.events.height) == (w, h)
            assert d.set_pic_info(w, h) == vd.SUCCESS
            rc, out = d.retrieve(cap)
        assert rc == vd.SUCCESS and np.array_equal(out, _i420(enc, w, h)), "picture %d" % i
    # a stride wider than the picture: the copy hook pads the rows
    assert d.set_pic_info(w, h, w + 32) == vd.SUCCESS
    au = enc.encode(frames[0])[0]
    assert d.send(au) == vd.SUCCESS
    rc, out = d.retrieve((w + 32) * h * 3 // 2)
    assert rc == vd.SUCCESS and len(out) == (w + 32) * h * 3 // 2
    assert np.array_equal(out[: (w + 32) * h].reshape(h, w + 32)[:, :w], enc.recon(0)[:h, :w])
    # damaged input fails the call, and Flush drops the state: the next picture must be an IDR
    assert d.send(au[: len(au) // 2]) == vd.DECODE_FAIL
    assert d.flush() == vd.SUCCESS
    p = enc.encode(frames[1])[0]
    assert d.send(p) == vd.DECODE_FAIL
    assert d.stop() == vd.SUCCESS and d.stop() == vd.SUCCESS and d.send(au) == vd.DECODE_FAIL
    assert d.delete() == vd.SUCCESS


def test_plugin_decoder_on_streams_of_other_encoders():
    """A stream shaped like what the reference's encoder side sends (OpenH264-style headers, a list modification in every P
    slice, QP per macroblock, chroma QP offset, sub-macroblock partitions, intra and I_PCM macroblocks in P pictures; random
    syntax from oracle/h264_enc.c) through the plugin surface: every retrieved picture equals the oracle's independent decoder's,
    cropped to the display size (130 x 98 in 144 x 112 coded samples)."""
    from oracle_lib import OracleDecoder
    w, h = 130, 98
    enc = OracleEncoder(w, h, qp=30, gop=6, profile_idc=66, refs=1)
    ref = OracleDecoder()
    d = vd.PluginDecoder()
    assert d.create_decoder(vd.STREAM_AVC) == vd.SUCCESS and d.init() == vd.SUCCESS and d.install_hooks() == vd.SUCCESS
    assert d.set_pic_info(w, h) == vd.SUCCESS and d.start() == vd.SUCCESS
    cap = w * h * 3 // 2
    for i in range(13):
        au = enc.random_picture(977 * i + 5, features=256 | 1 | 2 | 4 | 8 | 32)[0]
        assert ref.decode(au) == 1
        assert d.send(au) == vd.SUCCESS, "picture %d" % i
        rc, out = d.retrieve(cap)
        want = np.concatenate([ref.plane(0)[:h, :w].ravel(), ref.plane(1)[: h // 2, : w // 2].ravel(), ref.plane(2)[: h // 2, : w // 2].ravel()])
        assert rc == vd.SUCCESS and np.array_equal(out, want), "picture %d" % i
    assert d.stop() == vd.SUCCESS and d.delete() == vd.SUCCESS
