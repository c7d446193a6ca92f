"""Decoder peer on the GPU (include/mi355x_h264_dec.h: host parser + the encoder's reconstruction kernels run from parsed
decisions): every picture it decodes must equal, sample for sample, the reconstruction of the encoder that wrote the stream -
which tests/test_oracle_roundtrip.py ties to the independent spec-literal decoder of oracle/h264_dec.c."""
import numpy as np
import pytest
from media_amd import capi, synth, h264dec
from oracle_lib import OracleEncoder, OracleDecoder

pytestmark = pytest.mark.gpu

CASES = [("s1", 176, 144, 26, 66, 0, 0, 0), ("cut", 208, 160, 28, 66, 0, 0, 0), ("split", 176, 144, 26, 66, 0, 0, 0), ("s3", 96, 80, 30, 66, 0, 0, 0),
         ("s3", 64, 48, 10, 66, 0, 0, 0), ("split", 208, 160, 30, 100, 2, 0, 0), ("cut", 176, 144, 22, 77, 3, 3, 0), ("scroll", 130, 98, 30, 100, 0, 2, 0),
         ("s1", 16, 16, 26, 66, 0, 0, 0), ("s1", 320, 240, 34, 66, 0, 0, 1), ("cut", 352, 288, 40, 100, 3, 4, 0), ("ramp", 128, 96, 38, 66, 0, 0, 0)]


@pytest.mark.parametrize("kind,w,h,qp,prof,refs,slices,nodb", CASES)
def test_decoder_equals_the_encoders_reconstruction(kind, w, h, qp, prof, refs, slices, nodb):
    enc = OracleEncoder(w, h, qp=qp, gop=4, profile_idc=prof, refs=refs, slices=slices, disable_deblock=nodb)
    ref_dec = OracleDecoder()
    dec = h264dec.Decoder()
    for i, f in enumerate(synth.sequence(kind, w, h, 7)):
        au, _ = enc.encode(f)
        assert dec.decode(au), "picture %d" % i
        assert ref_dec.decode(au) == 1
        assert dec.info()[:2] == (w, h)
        for p in range(3):
            got = dec.plane(p)
            assert np.array_equal(got, enc.recon(p)), "picture %d plane %d" % (i, p)
            assert np.array_equal(got, ref_dec.plane(p))
        out = dec.i420()
        assert np.array_equal(out[: w * h].reshape(h, w), enc.recon(0)[:h, :w])
        assert np.array_equal(out[w * h: w * h * 5 // 4].reshape(h // 2, w // 2), enc.recon(1)[: h // 2, : w // 2])
    dec.close()


def test_decoder_on_the_hip_encoders_streams_and_size_change():
    """streams written by the HIP encoder (1080p, then a smaller size: the decoder re-creates its engine at the new IDR);
    a P picture without its reference is refused, as is an access unit with a feature outside the supported set"""
    dec = h264dec.Decoder()
    for (w, h, qp) in ((1920, 1080, 26), (640, 368, 30)):
        enc = capi.Encoder(w, h, qp=qp, gop=30)
        for i, f in enumerate(synth.sequence("s1", w, h, 4)):
            au = enc.encode(f)[0]
            assert dec.decode(au)
            cw, ch = dec.info()[2:]
            for p in range(3):
                assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p)), "%dx%d picture %d plane %d" % (w, h, i, p)
            y = f[: w * h].reshape(h, w)
            assert synth.psnr(y, dec.i420()[: w * h].reshape(h, w)) > 34.0
        enc.close()
    fresh = h264dec.Decoder()
    enc = OracleEncoder(64, 48, qp=26, gop=30)
    aus = [enc.encode(f)[0] for f in synth.sequence("s1", 64, 48, 2)]
    with pytest.raises(h264dec.StreamError):
        fresh.decode(aus[1])          # a P picture first
    assert fresh.decode(aus[0]) and fresh.decode(aus[1])
    fresh.close()
    dec.close()


def test_decoder_never_predicts_from_the_wrong_picture_after_a_loss():
    """a dropped or refused reference picture must not leave parser and reconstruction ring out of step.  With two
    reference pictures in use, one P access unit is dropped: the next P picture is refused (frame_num gap) and so is every one
    after it until the IDR picture, from which on every picture again equals the encoder's reconstruction; a damaged unit that
    the parser refuses has the same effect."""
    w, h = 176, 144
    enc = OracleEncoder(w, h, qp=28, gop=5, refs=2)
    aus, recs = [], []
    for f in synth.sequence("s1", w, h, 12):
        aus.append(enc.encode(f)[0])
        recs.append([enc.recon(p).copy() for p in range(3)])
    dec = h264dec.Decoder()

    def good(i):
        assert dec.decode(aus[i]), "picture %d" % i
        for p in range(3):
            assert np.array_equal(dec.plane(p), recs[i][p]), "picture %d plane %d" % (i, p)

    for i in range(3):
        good(i)
    for i in (4, 4):                                   # picture 3 is lost
        with pytest.raises(h264dec.StreamError):
            dec.decode(aus[i])
    for i in (5, 6, 7):                                # IDR picture 5 and on
        good(i)
    with pytest.raises(h264dec.StreamError):
        dec.decode(aus[8][: len(aus[8]) // 2])
    with pytest.raises(h264dec.StreamError):
        dec.decode(aus[9])
    good(10)
    good(11)
    dec.close()


def test_intra4x4_block_3_0_reads_the_macroblock_above_right():
    """tests/golden/dec_damaged_i4_topright.h264: an IDR picture of the oracle encoder with a few flipped bits, found by
    tools/soak_decoder.py's differential run.  It is still a conforming picture, and one of its Intra4x4 macroblocks predicts
    block (3, 0) with mode 7 (vertical-left), which reads four samples of the macroblock ABOVE-RIGHT - something this
    repository's encoder never produces (its row wavefront lags one macroblock, so it does not offer that block the two modes).
    The decoder must follow the standard, not the encoder: its intra row wavefront waits for the macroblock above-right too."""
    import os
    au = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "dec_damaged_i4_topright.h264"), "rb").read()
    ref, dec = OracleDecoder(), h264dec.Decoder()
    assert ref.decode(au) == 1 and dec.decode(au)
    for p in range(3):
        assert np.array_equal(dec.plane(p), ref.plane(p)), "plane %d" % p
    dec.close()


RANDOM_CASES = [(96, 80, 66, 0, 1, 31), (96, 80, 77, 3, 3, 31), (112, 64, 100, 2, 2, 31), (16, 16, 66, 0, 1, 31), (48, 160, 100, 4, 3, 31),
                (96, 80, 100, 0, 3, 1), (96, 80, 66, 2, 1, 2 | 4), (64, 64, 77, 0, 2, 8 | 16), (96, 80, 66, 0, 1, 0), (352, 288, 100, 3, 3, 31),
                (640, 368, 100, 0, 2, 31), (96, 80, 66, 0, 3, 63), (112, 64, 100, 2, 2, 63), (96, 80, 77, 3, 1, 32), (32, 32, 100, 0, 3, 63),
                (352, 288, 66, 0, 3, 63), (640, 368, 100, 4, 3, 63), (96, 80, 66, 0, 3, 127), (112, 64, 100, 0, 2, 127), (16, 64, 77, 0, 1, 32 | 64),
                (352, 288, 100, 0, 3, 127), (640, 368, 66, 0, 2, 127), (96, 80, 66, 0, 3, 255), (112, 64, 100, 2, 2, 128), (96, 80, 77, 0, 3, 128 | 32),
                (352, 288, 100, 0, 3, 255), (640, 368, 66, 3, 2, 128 | 31), (96, 80, 66, 0, 1, 256 | 1 | 2 | 32), (112, 64, 66, 0, 3, 511),
                (640, 368, 66, 0, 1, 256 | 1 | 2 | 4 | 32), (96, 80, 66, 0, 2, 1 | 512), (112, 64, 100, 2, 3, 1023), (352, 288, 100, 0, 3, 1023),
                (96, 80, 66, 0, 2, 1024 | 8), (112, 64, 100, 0, 3, 2047), (352, 288, 66, 0, 1, 1024 | 256 | 1 | 2 | 32)]


@pytest.mark.parametrize("w,h,prof,slices,refs,features", RANDOM_CASES)
def test_decoder_equals_the_independent_decoder_on_random_streams(w, h, prof, slices, refs, features):
    """Streams of RANDOM syntax (oracle/h264_enc.c h264o_enc_random_picture; tests/test_dec_parser.py lists what they hold):
    every macroblock type / mode / partition shape next to every other, QP changing per slice and per macroblock, chroma QP
    offsets (Cb and Cr apart under High), filter offsets, I_PCM inside filtered pictures, filtering across slice edges, and
    (feature 32) sub-macroblock partitions down to 4x4 with a reference index per partition, (feature 64) slices cut at random
    macroblocks, (feature 128) ref_pic_list_modification commands that permute the reference pictures, (feature 256) headers
    laid out the way OpenH264 writes them, (feature 512) levels that do not fit the byte the upload gives each, (feature 1024) constrained_intra_pred_flag.  No
    encoder reconstruction exists for these; the oracle's spec-literal decoder says what they decode to, and the GPU decoder
    must produce the same samples in every picture (errors would also propagate through the P pictures' references)."""
    enc = OracleEncoder(w, h, qp=30, gop=5, profile_idc=prof, slices=slices, refs=refs)
    ref_dec, dec = OracleDecoder(), h264dec.Decoder()
    for i in range(12):
        au, idr, mbqp = enc.random_picture(104729 * i + w + 3 * prof + features, features=features)
        assert ref_dec.decode(au) == 1
        assert dec.decode(au), "picture %d" % i
        for p in range(3):
            got, want = dec.plane(p), ref_dec.plane(p)
            if not np.array_equal(got, want):
                ys, xs = np.nonzero(got != want)
                s = 16 if p == 0 else 8
                k = (int(ys[0]) // s) * ((w + 15) // 16) + int(xs[0]) // s
                raise AssertionError("picture %d (%s) plane %d: %d samples differ, first in macroblock %d (type %d, QP %d)"
                                     % (i, "IDR" if idr else "P", p, ys.size, k, int(enc.mbinfo()["type"][k]), int(mbqp[k])))
    dec.close()


def test_decoder_against_the_committed_random_stream_vectors():
    """tests/golden/random_streams.json: the decoded planes of every picture hash to the committed value (what the oracle's
    independent decoder produced when the fixture was made; tests/test_dec_parser.py re-checks that on the CPU)."""
    import hashlib, json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "random_streams.json")))
    for c in gold["cases"]:
        enc = OracleEncoder(c["width"], c["height"], qp=30, gop=4, profile_idc=c["profile_idc"], slices=c["slices"], refs=c["refs"])
        dec = h264dec.Decoder()
        for i, fr in enumerate(c["frames"]):
            au = enc.random_picture(20261004 + 31 * i, features=c["features"])[0]
            assert hashlib.sha256(au).hexdigest() == fr["sha256"]
            assert dec.decode(au)
            assert hashlib.sha256(b"".join(dec.plane(p).tobytes() for p in range(3))).hexdigest() == fr["decoded_sha256"], "%s picture %d" % (c["name"], i)
        dec.close()


def test_decoder_long_stream_across_frame_num_wraps():
    """600 pictures of one GOP (8-bit frame_num: it wraps twice), three reference pictures, list modification in most P slices,
    QP per macroblock, sub-partitions: PicNum / FrameNumWrap arithmetic (8.2.4.1) and the reconstruction ring over a long run -
    every picture equals the independent decoder's."""
    w = h = 48
    enc = OracleEncoder(w, h, qp=30, gop=1000, profile_idc=66, refs=3)
    ref_dec, dec = OracleDecoder(), h264dec.Decoder()
    for i in range(600):
        au = enc.random_picture(7 * i + 1, features=1 | 32 | 128)[0]
        assert ref_dec.decode(au) == 1 and dec.decode(au), "picture %d" % i
        for p in range(3):
            assert np.array_equal(dec.plane(p), ref_dec.plane(p)), "picture %d plane %d" % (i, p)
    dec.close()
