"""GPU tests of the drop-in boundary: the C++ VideoEncoder plugin surface
(CreateVideoEncoder -> VideoEncoderMI355X), driven like the reference's caller would
(SURVEY.md 8b, Appendix A, D), with the CPU oracle replaying the same QP sequence."""
import numpy as np
import pytest
from media_amd import synth
from media_amd import videocodec as vc
from oracle_lib import OracleEncoder, OracleDecoder

pytestmark = pytest.mark.gpu


def _new(width, height, **kw):
    vc.set_video_mode(width, height, **kw)
    e = vc.VideoEncoder()
    assert e.rc_create == vc.SUCCESS
    assert e.init() == vc.SUCCESS and e.start() == vc.SUCCESS
    return e


def test_fixed_qp_sequence_matches_oracle_and_ownership():
    w, h = 320, 240
    e = _new(w, h, qp=26, gop=30)
    orc = OracleEncoder(w, h, qp=26, gop=30)
    frames = synth.sequence("s1", w, h, 4)
    for f in frames:
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS and bs == orc.encode(f)[0]
    # size guard: inputSize < w*h*3/2 -> ENCODE_FAIL, larger is fine (VideoEncoderOpenH264.cpp:307)
    assert e.encode(frames[0], size=w * h)[0] == vc.ENCODE_FAIL
    big = np.concatenate([frames[0], np.zeros(64, np.uint8)])
    assert e.encode(big)[0] == vc.SUCCESS
    assert e.stop() == vc.SUCCESS
    e.destroy()
    e.destroy()
    assert e.delete() == vc.SUCCESS


def test_keyframe_and_param_adjust_handshake():
    w, h = 176, 144
    e = _new(w, h, qp=28, gop=300)
    frames = synth.sequence("s1", w, h, 6)
    kinds = []
    for i, f in enumerate(frames):
        if i == 2:
            vc.prop_set("persist.vmi.video.encode.keyframe", "1")
        if i == 4:   # live re-config: new GOP size -> full reset -> next output starts with SPS/PPS + IDR
            vc.prop_set("persist.vmi.video.encode.gopsize", "60")
            vc.prop_set("persist.vmi.video.encode.param_adjusting", "1")
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS
        kinds.append(bs[4] & 31)
        assert vc.prop_get("persist.vmi.video.encode.keyframe") == "0"
        assert vc.prop_get("persist.vmi.video.encode.param_adjusting") == "0"
    assert kinds == [7, 1, 7, 1, 7, 1]
    # anything but "0"/"1" is reset to "0" with a warning (:320-323)
    vc.prop_set("persist.vmi.video.encode.param_adjusting", "banana")
    assert e.encode(frames[0])[0] == vc.SUCCESS
    assert vc.prop_get("persist.vmi.video.encode.param_adjusting") == "0"
    assert e.reset() == vc.SUCCESS
    rc, bs = e.encode(frames[1])
    assert rc == vc.SUCCESS and (bs[4] & 31) == 7
    e.delete()


def test_scene_change_recodes_as_idr():
    """bEnableSceneChangeDetect = 1 (VideoEncoderOpenH264.cpp:283): a cut in the content makes the motion
    cost explode; the class re-codes that picture as IDR.  The oracle replays the same rule."""
    w, h = 352, 288
    nmb = (w // 16) * (h // 16)
    e = _new(w, h, qp=28, gop=300)
    orc = OracleEncoder(w, h, qp=28, gop=300)
    frames = synth.sequence("s1", w, h, 3) + synth.sequence("s3", w, h, 1) + synth.sequence("s1", w, h, 2, start=500)
    kinds = []
    for f in frames:
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS
        obs, idr = orc.encode(f)
        if not idr and orc.me_cost() > 3000 * nmb:
            obs, idr = orc.encode(f, force_idr=True)
        assert bs == obs
        kinds.append(bs[4] & 31)
    assert kinds == [7, 1, 1, 7, 7, 1] and e.scene_cuts() == 2
    e.delete()


def test_bitrate_mode_tracks_target_and_replays_on_oracle():
    """reference preset: RC_BITRATE_MODE (VideoEncoderOpenH264.cpp:274).  The controller is host
    logic; the oracle replays its QP decisions and must produce the same stream."""
    w, h, fps, bitrate = 640, 368, 30, 1000000
    e = _new(w, h, fps=fps, bitrate=bitrate, gop=30, qp=None)
    orc = OracleEncoder(w, h, qp=30, gop=30)
    dec = OracleDecoder()
    total, qps = 0, []
    frames = synth.sequence("s1", w, h, 60)
    from media_amd.ratecontrol import RateControl, start_qp
    mirror = RateControl(bitrate, fps, qp=start_qp(bitrate, fps, w, h), gop=30)      # the Python statement of the same controller (media_amd/shard.py carries its state)
    for f in frames:
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS
        qp = e.last_qp()
        assert qp == mirror.qp
        mirror.update(len(bs), (bs[4] & 31) == 7)
        qps.append(qp)
        orc.set_qp(qp)
        assert bs == orc.encode(f)[0]
        assert dec.decode(bs) == 1
        total += len(bs)
    achieved = total * 8 * fps / len(frames)
    assert 0.95 * bitrate < achieved < 1.05 * bitrate, (achieved, qps)   # two seconds: within 5 %
    assert min(qps) >= 12 and max(qps) <= 48
    e.delete()


def test_slices_extension_key():
    """persist.vmi.video.encode.slices = 4: every access unit carries four slice NAL units and equals the oracle's
    four-slice stream; without the key the preset's single slice is used (covered by the tests above)"""
    w, h = 320, 240
    e = _new(w, h, qp=28, gop=30, slices=4)
    orc = OracleEncoder(w, h, qp=28, gop=30, slices=4)
    dec = OracleDecoder()
    for f in synth.sequence("s1", w, h, 4):
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS and bs == orc.encode(f)[0]
        assert sum(1 for k in range(len(bs) - 4) if bs[k:k + 4] == b"\x00\x00\x00\x01" and (bs[k + 4] & 31) in (1, 5)) == 4
        assert dec.decode(bs) == 1
    e.destroy()
    assert e.delete() == vc.SUCCESS
    vc.prop_set("persist.vmi.video.encode.slices", "")


def test_content_no_encoder_setting_can_code_never_fails_the_call():
    """black/white noise at QP 10: round 1 refused such a picture (EncodeOneFrame -> ENCODE_FAIL, a behaviour the reference
    does not have: OpenH264 never fails on content).  Now it is coded - as I_PCM where CAVLC could pass the 3200 bits of
    A.3.1 - and the stream stays decodable without any recovery step"""
    w, h = 640, 480
    e = _new(w, h, qp=10, gop=30)
    dec = OracleDecoder()
    frames = synth.sequence("s1", w, h, 3)
    rc, bs = e.encode(frames[0])
    assert rc == vc.SUCCESS and dec.decode(bs) == 1
    rng = np.random.default_rng(3)
    bad = (rng.integers(0, 2, w * h * 3 // 2, dtype=np.uint8) * 255).astype(np.uint8)
    rc, bs = e.encode(bad)
    assert rc == vc.SUCCESS and dec.decode(bs) == 1
    assert (dec.mb_kinds() == dec.KIND_IPCM).sum() > 600 and dec.max_mb_bits <= 3200
    assert np.array_equal(dec.plane(0)[:h, :w][dec.mb_kinds().reshape(h // 16, w // 16).repeat(16, 0).repeat(16, 1) == dec.KIND_IPCM],
                          bad[: w * h].reshape(h, w)[dec.mb_kinds().reshape(h // 16, w // 16).repeat(16, 0).repeat(16, 1) == dec.KIND_IPCM])
    for f in frames[1:]:
        rc, bs = e.encode(f)
        assert rc == vc.SUCCESS and dec.decode(bs) == 1
    e.destroy()
    assert e.delete() == vc.SUCCESS


def test_configs3_four_1080p_streams_on_four_threads():
    """BASELINE.json configs[3], "4 x 1080p30 independent streams", as far as one GPU goes: four encoder
    objects through the plugin surface (CreateVideoEncoder, format 3, fixed-QP key) on four host threads at once, each with its
    OWN 1920x1080 content, three pictures each; every access unit must be the CPU oracle's for that stream (the scene-change
    rule of the class replayed on the oracle).  On four GPUs the same objects sit one per device (media_amd/shard.py maps
    stream k to rank k, tests/test_shard_gloo.py); the kernels they run are these."""
    import threading
    w, h = 1920, 1080
    nmb = (w // 16) * ((h + 15) // 16)
    contents = [synth.sequence("s1", w, h, 3), synth.sequence("scroll", w, h, 3), synth.sequence("split", w, h, 3), synth.sequence("s1", w, h, 3, start=500)]
    vc.set_video_mode(w, h, qp=26, gop=30)
    encs = []
    for _ in range(4):
        e = vc.VideoEncoder()
        assert e.rc_create == vc.SUCCESS and e.init() == vc.SUCCESS and e.start() == vc.SUCCESS
        encs.append(e)
    got = [[] for _ in range(4)]

    def work(k):
        for f in contents[k]:
            got[k].append(encs[k].encode(f))

    ths = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(4):
        orc = OracleEncoder(w, h, qp=26, gop=30)
        for i, f in enumerate(contents[k]):
            obs, idr = orc.encode(f)
            if not idr and orc.me_cost() > 3000 * nmb:
                obs, idr = orc.encode(f, force_idr=True)
            rc, bs = got[k][i]
            assert rc == vc.SUCCESS and bs == obs, "stream %d picture %d" % (k, i)
        orc.close()
    assert len({got[k][2][1] for k in range(4)}) == 4, "four different streams"
    for e in encs:
        assert e.stop() == vc.SUCCESS
        e.destroy()
        assert e.delete() == vc.SUCCESS
