"""Oracle pinning, part 2: decode(encode(yuv)) must reproduce the encoder's
reconstruction byte for byte (ties dequant, inverse transform, interpolation, intra
prediction, deblocking and the CAVLC syntax to ITU-T H.264 through an independently
written parser), and the committed golden vectors must still match.  CPU only."""
import hashlib
import json
import os
import numpy as np
import pytest
from media_amd import synth
from oracle_lib import OracleEncoder, OracleDecoder

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_golden_and_roundtrip(case):
    w, h = case["width"], case["height"]
    enc = OracleEncoder(w, h, qp=case["qp"], gop=case["gop"], profile_idc=case["profile_idc"], slices=case.get("slices", 0), refs=case.get("refs", 0), search=case.get("search", 0))
    dec = OracleDecoder()
    for i, (f, g) in enumerate(zip(synth.sequence(case["kind"], w, h, len(case["frames"])), case["frames"])):
        bs, idr = enc.encode(f)
        assert idr == g["idr"] and len(bs) == g["bytes"], "frame %d" % i
        assert hashlib.sha256(bs).hexdigest() == g["sha256"], "frame %d bitstream" % i
        rec = b"".join(enc.recon(p).tobytes() for p in range(3))
        assert hashlib.sha256(rec).hexdigest() == g["recon_sha256"], "frame %d reconstruction" % i
        assert dec.decode(bs) == 1
        assert dec.size == (w, h)
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p)), "frame %d plane %d: decoder != encoder" % (i, p)
        # the stream is a legal Annex-B access unit: 4-byte start codes, SPS+PPS before every IDR
        assert bs[:4] == b"\x00\x00\x00\x01"
        assert (bs[4] & 31) == (7 if idr else 1)


def test_mode_set_intra_in_p_and_pcm_are_legal_streams():
    """what the test decoder saw: intra macroblocks inside P pictures after a cut, I_PCM where CAVLC could pass 3200 bits,
    and never a macroblock_layer() above 3200 bits (A.3.1) or a level_prefix above 15 (A.2, Baseline / Main)"""
    w, h = 176, 144
    enc, dec = OracleEncoder(w, h, qp=28, gop=30), OracleDecoder()
    kinds = []
    for f in synth.sequence("cut", w, h, 4):
        bs, _ = enc.encode(f)
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p))
        kinds.append(dec.mb_kinds())
        assert dec.max_mb_bits <= 3200 and dec.max_level_prefix <= 15
    assert (kinds[1] >= dec.KIND_INTER).all()                       # before the cut: inter only
    n_intra = int(((kinds[2] == dec.KIND_I16) | (kinds[2] == dec.KIND_I4)).sum())
    # (Intra16x16 predicts this sinusoidal texture poorly, so the exhaustive search still wins most macroblocks)
    assert n_intra >= 5 and (kinds[2] >= dec.KIND_INTER).any(), "the P picture after the cut should mix intra and inter (%d of %d intra)" % (n_intra, len(kinds[2]))
    # uniform noise at the lowest QP: every macroblock would pass the limit -> I_PCM, reconstruction == source
    enc, dec = OracleEncoder(w, h, qp=10, gop=30), OracleDecoder()
    for i, f in enumerate(synth.sequence("s3", w, h, 3)):
        bs, _ = enc.encode(f)
        assert dec.decode(bs) == 1
        k = dec.mb_kinds()
        assert (k == dec.KIND_IPCM).all(), "picture %d: %s" % (i, np.bincount(k))
        assert dec.max_mb_bits <= 3200 and dec.max_level_prefix <= 15
        assert np.array_equal(dec.plane(0)[:h, :w], f[: w * h].reshape(h, w))
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p))
    # in between (noise at QP 30): a mix, still legal
    enc, dec = OracleEncoder(200, 120, qp=30, gop=30), OracleDecoder()
    for f in synth.sequence("s3", 200, 120, 2):
        assert dec.decode(enc.encode(f)[0]) == 1
        assert dec.max_mb_bits <= 3200 and dec.max_level_prefix <= 15


@pytest.mark.parametrize("w,h,qp,prof,refs,slices", [(176, 144, 26, 66, 0, 0), (208, 160, 30, 100, 2, 0), (176, 144, 22, 77, 3, 3)])
def test_partitions_16x8_8x16_8x8_roundtrip(w, h, qp, prof, refs, slices):
    """two layers drifting apart by a fraction of a sample in 8-sample stripes: the macroblocks of the upper third split into
    two 16x8 partitions, the middle third into 8x16, the lower third into four 8x8.  The independent decoder - which derives
    every vector from mvd_l0 and ITS OWN statement of 8.4.1.3 (neighbour partitions, directional rules) - must arrive at the
    encoder's vectors and reconstruction; with several reference pictures, the 8x8 transform and slices."""
    enc, dec = OracleEncoder(w, h, qp=qp, gop=30, profile_idc=prof, refs=refs, slices=slices), OracleDecoder()
    seen = np.zeros(8, np.int64)
    for i, f in enumerate(synth.sequence("split", w, h, 4)):
        assert dec.decode(enc.encode(f)[0]) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p)), "picture %d plane %d" % (i, p)
        assert dec.max_mb_bits <= 3200 and dec.max_level_prefix <= 15
        mb, mvq = enc.mbinfo(), enc.mvq()
        if i:
            seen += np.bincount(mb["type"], minlength=8)
            for a in np.where(mb["type"] >= 5)[0][::7]:   # a sample of the partitioned macroblocks, every quadrant
                for q in range(4):
                    x, y, r = dec.mb_mv(int(a), (q >> 1) * 8 + (q & 1) * 2)
                    assert (x, y, r) == (mvq[a, 2 * q], mvq[a, 2 * q + 1], mb["chroma_mode"][a]), (i, a, q)
    assert seen[5] > 0 and seen[6] > 0 and seen[7] > 0, seen


def test_quality_and_skip_behaviour():
    w, h = 320, 240
    enc = OracleEncoder(w, h, qp=26)
    for i, f in enumerate(synth.sequence("s1", w, h, 4)):
        enc.encode(f)
        y = f[: w * h].reshape(h, w)
        assert synth.psnr(y, enc.recon(0)[:h, :w]) > 36.0
    # static content: after the IDR everything collapses to P_Skip and a few bytes per picture
    enc = OracleEncoder(w, h, qp=26)
    sizes = []
    for f in synth.sequence("s2", w, h, 5):
        bs, _ = enc.encode(f)
        sizes.append(len(bs))
    mb = enc.mbinfo()
    assert (mb["type"] == 2).all() and sizes[-1] < 16


def test_forced_idr_and_qp_change():
    w, h = 176, 144
    enc = OracleEncoder(w, h, qp=30, gop=100)
    dec = OracleDecoder()
    fr = synth.sequence("s1", w, h, 5)
    kinds = []
    for i, f in enumerate(fr):
        if i == 3:
            enc.set_qp(36)
        bs, idr = enc.encode(f, force_idr=(i == 2))
        kinds.append(idr)
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p))
    assert kinds == [True, False, True, False, False]


def test_emulation_prevention_inside_slices():
    """the 'ramp' input makes slice payloads that contain 00 00 0x: the escaped stream must round-trip"""
    w, h = 320, 240
    enc = OracleEncoder(w, h, qp=30, gop=100)
    dec = OracleDecoder()
    escaped = 0
    for f in synth.sequence("ramp", w, h, 6):
        bs, _ = enc.encode(f)
        escaped += b"\x00\x00\x03" in bs[bs.rfind(b"\x00\x00\x00\x01") + 5:]
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.recon(p))
    assert escaped >= 2


def test_decoder_rejects_garbage():
    dec = OracleDecoder()
    with pytest.raises(RuntimeError):
        dec.decode(b"\x00\x00\x00\x01\x65\x88\x84\x00\x10")  # slice before parameter sets


def test_access_units_walk_like_the_reference_decoder_adapter_expects():
    """the reference's decoder adapter caches the SPS / PPS in front of a key picture by walking the
    leading non-VCL NAL units (tests/annexb.py restates that walk).  IDR access units must yield exactly [SPS, PPS]
    and then slices only, P access units no parameter sets; one slice NAL per slice band."""
    import annexb
    assert annexb.find_nal_start_code(b"\x11\x00\x00\x01\x65") == 1 and annexb.find_nal_start_code(b"\x00\x00\x00\x01\x67") == 0
    assert annexb.find_nal_start_code(b"\x00\x00\x02\x00\x01") == -1
    assert annexb.find_next_non_vcl_nalu(b"\x00\x00\x00\x01\x65\x88") == (0, 5)
    assert annexb.find_next_non_vcl_nalu(b"\x00\x00\x00\x01\x67\x42\x00\x00\x00\x01\x68") == (6, 7)
    assert annexb.find_next_non_vcl_nalu(b"\x00\x00\x01\x68\xce") == (5, 8)          # last unit: the whole buffer
    for slices in (0, 3):
        w, h = 176, 144
        enc = OracleEncoder(w, h, qp=28, gop=3, slices=slices)
        for i, f in enumerate(synth.sequence("s1", w, h, 5)):
            au, idr = enc.encode(f)
            sets, rest = annexb.leading_parameter_sets(au)
            assert [t for t, _ in sets] == ([7, 8] if idr else []), "picture %d" % i
            units = annexb.split_nal_units(rest)
            assert len(units) == max(slices, 1) and all(t == (5 if idr else 1) for _, t, _ in units)
            assert all(ref == (3 if idr else 2) for ref, _, _ in units)
            assert b"".join(b for _, b in sets) + rest == au
