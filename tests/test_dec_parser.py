"""Decoder peer, host side (media_amd/csrc/h264_parse.h through the C ABI of include/mi355x_h264_dec.h): the parser must
recover from a stream exactly the side information of the encoder that wrote it - macroblock types, modes, reference
indices, coded_block_pattern, TotalCoeff, the quadrant vectors (mvd + its own statement of 8.4.1.3), the Intra4x4 modes
(8.3.1.1) and every level the stream carries.  CPU only: the oracle encoder writes the streams."""
import numpy as np
import pytest
from media_amd import synth, h264dec
from oracle_lib import OracleEncoder, OracleDecoder

CASES = [("s1", 176, 144, 26, 66, 0, 0), ("cut", 208, 160, 28, 66, 0, 0), ("split", 176, 144, 26, 66, 0, 0), ("s3", 96, 80, 30, 66, 0, 0),
         ("s3", 64, 48, 10, 66, 0, 0), ("split", 208, 160, 30, 100, 2, 0), ("cut", 176, 144, 22, 77, 3, 3), ("scroll", 130, 98, 30, 100, 0, 2),
         ("s1", 16, 16, 26, 66, 0, 0)]


@pytest.mark.parametrize("kind,w,h,qp,prof,refs,slices", CASES)
def test_parser_recovers_the_encoders_side_information(kind, w, h, qp, prof, refs, slices):
    enc = OracleEncoder(w, h, qp=qp, gop=30, profile_idc=prof, refs=refs, slices=slices)
    par = h264dec.Parser()
    for i, f in enumerate(synth.sequence(kind, w, h, 5)):
        au, idr = enc.encode(f)
        assert par.parse(au), "picture %d" % i
        info = par.info()
        assert (info["width"], info["height"], info["qp"], bool(info["idr"])) == (w, h, qp, bool(idr))
        mb, mvq, aux, lv = par.arrays()
        omb, omvq, oaux, olv = enc.mbinfo(), enc.mvq(), enc.mbaux(), enc.levels()
        for fld in ("type", "cbp", "chroma_mode", "tc", "mvx", "mvy"):
            assert np.array_equal(mb[fld], omb[fld]), "picture %d: %s" % (i, fld)
        i16 = omb["type"] == 0
        inter = np.isin(omb["type"], (1, 2, 5, 6, 7))
        assert np.array_equal(mb["i16_mode"][i16 | inter], omb["i16_mode"][i16 | inter])
        assert np.array_equal(mvq[inter], omvq[inter]), "picture %d: quadrant vectors" % i
        i4 = omb["type"] == 4
        assert np.array_equal(aux[i4], oaux[i4]), "picture %d: Intra4x4 modes" % i
        # every level list the stream carries (by type and coded_block_pattern)
        read = np.zeros(olv.shape, bool)
        cbp = omb["cbp"].astype(np.int32)
        read[i16, 0:16] = True
        for q in range(4):
            read[(cbp >> q) & 1 == 1, 16 + 64 * q: 16 + 64 * (q + 1)] = True
        read[(cbp >> 4) >= 1, 272:280] = True
        read[(cbp >> 4) == 2, 280:408] = True
        read[np.isin(omb["type"], (2, 3)), :] = False
        assert np.array_equal(np.where(read, lv, 0), np.where(read, olv, 0)), "picture %d: levels" % i
        assert not np.where(~read & (omb["type"] != 3)[:, None], lv, 0).any(), "levels outside what the stream carries must stay 0"
        pcm = omb["type"] == 3
        if pcm.any():   # the 384 samples of an I_PCM macroblock travel as bytes at the start of its level area
            raw = lv[pcm].view(np.uint8)[:, :384]
            k = int(np.where(pcm)[0][0])
            mx, my = k % ((w + 15) // 16), k // ((w + 15) // 16)
            src = enc.recon_pre(0)[16 * my:16 * my + 16, 16 * mx:16 * mx + 16]
            assert np.array_equal(raw[0, :256].reshape(16, 16), src)
    par.close()


def test_parser_refuses_what_it_does_not_support():
    par = h264dec.Parser()
    with pytest.raises(h264dec.StreamError):
        par.parse(b"\x00\x00\x00\x01\x65\x88\x84\x00")          # a slice without parameter sets
    enc = OracleEncoder(64, 48, qp=26, gop=30)
    au = enc.encode(synth.sequence("s1", 64, 48, 1)[0])[0]
    assert par.parse(au)
    with pytest.raises(h264dec.StreamError):
        par.parse(au[: len(au) * 2 // 3])                      # truncated slice data
    par.close()


def test_parser_survives_damaged_streams():
    """3 000 damaged access units (bit flips, byte splices, truncations, duplicated and re-ordered NAL units) of streams that use
    every macroblock type: each is either refused with a message or parsed - never a crash, a hang or an out-of-range array
    (the parser's arrays are re-read after every call)."""
    import random
    rng = random.Random(77)
    streams = []
    for (kind, w, h, qp, prof, refs, slices) in (("cut", 96, 80, 28, 66, 0, 0), ("split", 96, 80, 26, 100, 2, 0), ("s3", 64, 48, 12, 77, 3, 2), ("s1", 48, 32, 40, 66, 0, 0)):
        enc = OracleEncoder(w, h, qp=qp, gop=3, profile_idc=prof, refs=refs, slices=slices)
        streams.append([enc.encode(f)[0] for f in synth.sequence(kind, w, h, 4)])
    par = h264dec.Parser()
    ok = bad = 0
    for case in range(3000):
        aus = rng.choice(streams)
        au = bytearray(rng.choice(aus))
        how = rng.randrange(6)
        if how == 0:
            for _ in range(rng.randint(1, 6)):
                au[rng.randrange(len(au))] ^= 1 << rng.randrange(8)
        elif how == 1:
            au = au[: rng.randrange(1, len(au))]
        elif how == 2:
            a, b = sorted(rng.randrange(len(au)) for _ in range(2))
            au[a:b] = bytes(rng.randrange(256) for _ in range(rng.randint(0, 12)))
        elif how == 3:
            au = au + bytearray(rng.choice(aus))
        elif how == 4:
            au = bytearray(rng.randrange(256) for _ in range(rng.randint(1, 64))) + au
        else:
            k = rng.randrange(4, len(au))
            au[k:k] = b"\\x00\\x00\\x01" + bytes([rng.randrange(256)])
        try:
            got = par.parse(bytes(au))
            ok += 1
            if got:
                mb, mvq, aux, lv = par.arrays()
                assert mb["type"].max() <= 7 and aux.max() <= 8
        except h264dec.StreamError as ex:
            bad += 1
            assert str(ex), "a refusal names its reason"
    assert ok > 100 and bad > 1000, (ok, bad)
    # the parser still works afterwards
    for au in streams[0]:
        assert par.parse(au)
    par.close()


def test_two_independent_parsers_agree_on_damaged_streams():
    """Differential check between the product's host parser (media_amd/csrc/h264_parse.h) and the oracle's independent decoder
    (oracle/h264_dec.c), which share no code: 2 000 damaged P-picture access units are given to both after the same intact IDR
    picture.  Each may refuse a unit (the oracle decoder supports more syntax than the product parser, so it accepts more), but
    whenever BOTH accept one they must have read the same macroblock kinds and the same quadrant vectors out of the same bits."""
    import random
    rng = random.Random(99)
    kind_of = {0: 2, 4: 1, 3: 3, 1: 4, 5: 4, 6: 4, 7: 4, 2: 5}   # product type -> oracle decoder kind (I16, I4, I_PCM, inter, skip)
    both = 0
    for (content, w, h, qp, prof, refs) in (("cut", 96, 80, 28, 66, 0), ("split", 96, 80, 26, 100, 2), ("s1", 64, 48, 34, 77, 0)):
        enc = OracleEncoder(w, h, qp=qp, gop=30, profile_idc=prof, refs=refs)
        aus = [enc.encode(f)[0] for f in synth.sequence(content, w, h, 3)]
        for case in range(700):
            au = bytearray(aus[1 + case % 2])
            for _ in range(rng.randint(1, 3)):
                k = rng.randrange(6, len(au))          # (leave the NAL header alone)
                au[k] ^= 1 << rng.randrange(8)
            dec, par = OracleDecoder(), h264dec.Parser()
            assert dec.decode(aus[0]) == 1 and par.parse(aus[0])
            if case % 2:
                assert dec.decode(aus[1]) == 1 and par.parse(aus[1])
            try:
                ok_dec = dec.decode(bytes(au)) == 1
            except Exception:
                ok_dec = False
            try:
                ok_par = par.parse(bytes(au))
            except h264dec.StreamError:
                ok_par = False
            if ok_dec and ok_par:
                both += 1
                mb, mvq, _, _ = par.arrays()
                kinds = dec.mb_kinds()
                assert [kind_of[t] for t in mb["type"]] == list(kinds), (content, case)
                for a in np.where(np.isin(mb["type"], (1, 2, 5, 6, 7)))[0][::5]:
                    for q in range(4):
                        x, y, r = dec.mb_mv(int(a), (q >> 1) * 8 + (q & 1) * 2)
                        assert (x, y, r) == (mvq[a, 2 * q], mvq[a, 2 * q + 1], mb["chroma_mode"][a]), (content, case, a, q)
            par.close()
    assert both > 150, both


def _compare_side_information(par, enc, label, vectors=True):
    mb, mvq, aux, lv = par.arrays()
    omb, omvq, oaux, olv = enc.mbinfo(), enc.mvq(), enc.mbaux(), enc.levels()
    i16 = omb["type"] == 0
    inter = np.isin(omb["type"], (1, 2, 5, 6, 7))
    for fld in ("type", "cbp", "tc") + (("chroma_mode", "mvx", "mvy") if vectors else ()):
        assert np.array_equal(mb[fld], omb[fld]), "%s: %s" % (label, fld)
    if vectors:
        assert np.array_equal(mb["i16_mode"][i16 | inter], omb["i16_mode"][i16 | inter]), label
        assert np.array_equal(mvq[inter], omvq[inter]), "%s: quadrant vectors" % label
    else:   # the generator wrote mvd_l0 / ref_idx_l0 / sub_mb_type draws as such and holds no vectors
        assert np.array_equal(mb["chroma_mode"][~inter], omb["chroma_mode"][~inter]) and np.array_equal(mb["i16_mode"][i16], omb["i16_mode"][i16]), label
    i4 = omb["type"] == 4
    assert np.array_equal(aux[i4], oaux[i4]), "%s: Intra4x4 modes" % label
    read = np.zeros(olv.shape, bool)
    cbp = omb["cbp"].astype(np.int32)
    read[i16, 0:16] = True
    for q in range(4):
        read[(cbp >> q) & 1 == 1, 16 + 64 * q: 16 + 64 * (q + 1)] = True
    read[(cbp >> 4) >= 1, 272:280] = True
    read[(cbp >> 4) == 2, 280:408] = True
    read[np.isin(omb["type"], (2, 3)), :] = False
    assert np.array_equal(np.where(read, lv, 0), np.where(read, olv, 0)), "%s: levels" % label


RANDOM_CASES = [(96, 80, 66, 0, 1, 31), (96, 80, 77, 3, 3, 31), (112, 64, 100, 2, 2, 31), (16, 16, 66, 0, 1, 31), (48, 160, 100, 4, 3, 31),
                (96, 80, 100, 0, 3, 1), (96, 80, 66, 2, 1, 2 | 4), (64, 64, 77, 0, 2, 8 | 16),
                (96, 80, 66, 0, 3, 63), (112, 64, 100, 2, 2, 63), (96, 80, 77, 3, 1, 32), (32, 32, 100, 0, 3, 63),
                (96, 80, 66, 0, 3, 127), (112, 64, 100, 0, 2, 127), (16, 64, 77, 0, 1, 32 | 64),
                (96, 80, 66, 0, 3, 255), (112, 64, 100, 2, 2, 128), (96, 80, 77, 0, 3, 128 | 32),
                (96, 80, 66, 0, 1, 256 | 1 | 2 | 32), (112, 64, 66, 0, 3, 511), (96, 80, 66, 0, 2, 1 | 512), (112, 64, 100, 2, 3, 1023),
                (96, 80, 66, 0, 2, 1024 | 8), (112, 64, 100, 0, 3, 2047)]


@pytest.mark.parametrize("w,h,prof,slices,refs,features", RANDOM_CASES)
def test_parser_reads_random_streams(w, h, prof, slices, refs, features):
    """Streams of RANDOM syntax (oracle/h264_enc.c h264o_enc_random_picture: every macroblock type, prediction mode and
    partition shape, random vectors, reference indices, levels, I_PCM, and what this repository's encoder never writes -
    mb_qp_delta, slice QPs, chroma_qp_index_offsets, filter offsets, every deblocking idc): the product parser must read back
    exactly what was written, QP_Y of every macroblock included, and the oracle's independent decoder must agree on the QPs.
    Feature 32: sub-macroblock partitions down to 4x4 and a reference index per partition, written as random mvd_l0 / ref_idx_l0
    draws - the parser's vectors (its statement of 8.4.1.3 on the 4x4 grid) must be the oracle decoder's for every 4x4 block.
    Feature 64: slices cut at random macroblocks (neighbour availability per macroblock instead of per row).  Feature 128:
    ref_pic_list_modification commands in the P slices (at least one picture of such a case must come out with a permuted list).
    Feature 256: parameter sets and slice headers laid out the way OpenH264 writes them (15-bit frame_num, POC type 0, VUI, a
    list modification naming the previous picture in every P slice) - with one reference picture, QP per macroblock, chroma
    offset and sub-partitions that is the shape of a stream of the reference's own encoder.  Feature 512: levels beyond a signed
    byte (the parser keeps one byte per level and a list of the exceptions; what it hands out as int16 must be what was written).
    Feature 1024: constrained_intra_pred_flag (an inter neighbour counts as missing when Intra4x4PredMode is predicted)."""
    enc = OracleEncoder(w, h, qp=30, gop=4, profile_idc=prof, slices=slices, refs=refs)
    par, dec = h264dec.Parser(), OracleDecoder()
    seen = set()
    for i in range(10):
        au, idr, mbqp = enc.random_picture(7919 * i + w + 3 * prof + features, features=features)
        assert dec.decode(au) == 1, "the oracle decoder accepts the stream"
        assert par.parse(au), "picture %d" % i
        _compare_side_information(par, enc, "picture %d" % i, vectors=not (features & 32))
        if features & 32:
            mv4, refq = par.vectors4()
            for a in np.nonzero(np.isin(enc.mbinfo()["type"], (1, 2, 5, 6, 7)))[0]:
                for b in range(16):
                    want = dec.mb_mv(int(a), b)
                    got = (int(mv4[a, b, 0]), int(mv4[a, b, 1]), int(refq[a, 2 * (b >> 3) + ((b >> 1) & 1)]))
                    assert got == want, "picture %d macroblock %d block %d: %s, the oracle decoder has %s" % (i, a, b, got, want)
        assert np.array_equal(par.mbqp(), mbqp), "picture %d: QP_Y" % i
        assert np.array_equal(dec.mb_qps(), mbqp.astype(np.int32)), "picture %d: QP_Y (oracle decoder)" % i
        info = par.info()
        assert bool(info["idr"]) == idr
        seen |= set(int(t) for t in enc.mbinfo()["type"])
        permuted = locals().get("permuted", False) or (info["ref_age0"], info["ref_age1"], info["ref_age2"]) != (0, 1, 2)
        if features & 1:
            assert not info["one_qp"] or len(set(mbqp)) == 1
        if features & 512:
            big_levels = locals().get("big_levels", 0) + int((np.abs(par.arrays()[3][~np.isin(enc.mbinfo()["type"], (2, 3))].astype(np.int32)) > 127).sum())
    assert seen >= ({0, 1, 2, 4, 5, 6, 7} | ({3} if features & 8 else set())) or w * h <= 256
    assert permuted == bool(features & 128 and refs > 1), "a permuted reference list where, and only where, one was written"
    assert not (features & 512) or big_levels > 100, "levels beyond a byte were in play"
    par.close()


def test_random_stream_golden_vectors():
    """tests/golden/random_streams.json (made by tests/golden/make_golden.py): the generator still writes the same access units
    and the oracle's independent decoder still decodes them to the same pictures - the fixtures the GPU decoder is held to in
    tests/test_gpu_decoder.py."""
    import hashlib, json, os
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "random_streams.json")))
    for c in gold["cases"]:
        enc = OracleEncoder(c["width"], c["height"], qp=30, gop=4, profile_idc=c["profile_idc"], slices=c["slices"], refs=c["refs"])
        dec, par = OracleDecoder(), h264dec.Parser()
        for i, fr in enumerate(c["frames"]):
            au, idr, _ = enc.random_picture(20261004 + 31 * i, features=c["features"])
            assert (hashlib.sha256(au).hexdigest(), len(au), idr) == (fr["sha256"], fr["bytes"], fr["idr"]), "%s picture %d" % (c["name"], i)
            assert dec.decode(au) == 1 and par.parse(au)
            assert hashlib.sha256(b"".join(dec.plane(p).tobytes() for p in range(3))).hexdigest() == fr["decoded_sha256"], "%s picture %d" % (c["name"], i)
        par.close()


def test_parser_under_address_sanitizer(tmp_path):
    """tools/fuzz_parser.cpp: the parser built with AddressSanitizer + UBSan walks 1 500 access units of random-syntax streams
    (every feature of the generator), two thirds of them damaged (bit flips, truncations, splices, duplicated and injected NAL
    units): no read or write outside an array, no undefined arithmetic - the run must end with its summary line."""
    import os, random, shutil, struct, subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fuzz_parser")
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-I", os.path.join(root, "media_amd", "csrc"), os.path.join(root, "tools", "fuzz_parser.cpp"), "-o", exe], capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build not available: " + r.stderr[-200:])
    rng = random.Random(5)
    units = []
    for (w, h, prof, sl, refs, feat) in ((96, 80, 66, 0, 3, 511), (112, 64, 100, 2, 2, 255), (48, 32, 77, 0, 1, 256 | 1 | 2 | 32)):
        enc = OracleEncoder(w, h, qp=30, gop=4, profile_idc=prof, slices=sl, refs=refs)
        aus = [enc.random_picture(rng.getrandbits(30), features=feat)[0] for _ in range(10)]
        for rep in range(50):
            for au in aus:
                b = bytearray(au)
                how = rng.randrange(8)
                if how == 0:
                    for _ in range(rng.randint(1, 6)):
                        b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
                elif how == 1:
                    b = b[: rng.randrange(1, len(b))]
                elif how == 2:
                    a, c = sorted(rng.randrange(len(b)) for _ in range(2))
                    b[a:c] = bytes(rng.randrange(256) for _ in range(rng.randint(0, 12)))
                elif how == 3:
                    b = b + bytearray(rng.choice(aus))
                elif how == 4:
                    k = rng.randrange(4, len(b))
                    b[k:k] = b"\x00\x00\x01" + bytes([rng.randrange(256)])
                units.append(bytes(b))
    path = str(tmp_path / "units.bin")
    with open(path, "wb") as f:
        for u in units:
            f.write(struct.pack("<I", len(u)))
            f.write(u)
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("parsed "), (r.stdout[-300:], r.stderr[-1500:])
    parsed, refused = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert parsed > 400 and refused > 400, r.stdout
