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
            for i, au in enumerate(aus):
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
                # a refused unit costs the parser its reference pictures (P pictures are then refused until the next IDR picture), so
                # every unit comes behind the intact pictures of its GOP: the damage is met with the references in place
                units.extend(aus[i - i % 4:i])
                units.append(bytes(b))
    units.extend(_crafted_units_with_huge_exp_golomb_values())
    path = str(tmp_path / "units.bin")
    with open(path, "wb") as f:
        for u in units:
            f.write(struct.pack("<I", len(u)))
            f.write(u)
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("parsed "), (r.stdout[-300:], r.stderr[-1500:])
    parsed, refused = int(r.stdout.split()[1]), int(r.stdout.split()[3])
    assert parsed > 1500 and refused > 400, r.stdout


class _Bits:
    """MSB-first bit writer for the crafted slices below"""
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]
        return self

    def ue(self, v):
        x = v + 1
        n = x.bit_length() - 1
        return self.u(n, 0).u(n + 1, x)

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def nal(self, hdr):
        bits = self.bits + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytes(sum(b << (7 - k) for k, b in enumerate(bits[i:i + 8])) for i in range(0, len(bits), 8))
        out, zeros = bytearray(), 0
        for c in raw:
            if zeros >= 2 and c <= 3:
                out.append(3)
                zeros = 0
            out.append(c)
            zeros = zeros + 1 if c == 0 else 0
        return b"\x00\x00\x00\x01" + bytes([hdr]) + bytes(out)


_HUGE = [(1 << 31) - 1, 1 << 31, (1 << 31) + 12345, (1 << 32) - 2]   # code words with 30 / 31 leading zeros


def _crafted_p_slices(field):
    """P slices for the oracle encoder's parameter sets (8-bit frame_num, POC type 2, deblocking control present, one reference)
    in which ONE Exp-Golomb element carries a value of about 2^31: as ue() the cast to int is negative, as se() it is +-2^30."""
    out = []
    for huge in _HUGE:
        def v(name, normal):
            return huge if name == field else normal
        b = _Bits()
        b.ue(v("first_mb_in_slice", 0)).ue(v("slice_type", 5)).ue(v("pic_parameter_set_id", 0)).u(8, 1)
        if field == "num_ref_idx_active":
            b.u(1, 1).ue(huge)
        else:
            b.u(1, 0)
        if field == "modification":
            b.u(1, 1).ue(0).ue(huge).ue(3)
        else:
            b.u(1, 0)
        b.u(1, 0)                                        # adaptive_ref_pic_marking_mode_flag
        b.ue(v("slice_qp_delta", 0))                     # (se: the same code words)
        b.ue(v("disable_deblocking_filter_idc", 1))
        b.ue(v("mb_skip_run", 2))
        b.ue(v("mb_type", 0))                            # P_L0_16x16
        b.ue(v("mvd_x", 0)).ue(v("mvd_y", 0))
        b.ue(v("coded_block_pattern", 0))
        b.ue(v("mb_skip_run_2", 1))
        b.ue(v("mb_type_2", 3))                          # P_8x8
        for k in range(4):
            b.ue(v("sub_mb_type", 0) if k == 2 else 0)
        for k in range(8):
            b.ue(0)
        b.ue(v("coded_block_pattern_2", 0))
        b.ue(v("mb_skip_run_3", 0))
        b.ue(v("mb_type_3", 5 + 1))                      # I_16x16_0_0_0 inside a P slice
        b.ue(v("intra_chroma_pred_mode", 0))
        b.ue(v("mb_qp_delta", 0))
        b.u(1, 1)                                        # coeff_token of the DC block: no coefficients
        out.append(b.nal(0x41))
    return out


_CRAFTED_FIELDS = ("first_mb_in_slice", "slice_type", "pic_parameter_set_id", "num_ref_idx_active", "modification", "slice_qp_delta",
                   "disable_deblocking_filter_idc", "mb_skip_run", "mb_type", "mvd_x", "mvd_y", "coded_block_pattern", "mb_skip_run_2", "mb_type_2",
                   "sub_mb_type", "coded_block_pattern_2", "mb_skip_run_3", "mb_type_3", "intra_chroma_pred_mode", "mb_qp_delta")


def _crafted_units_with_huge_exp_golomb_values():
    """every crafted slice behind an intact IDR picture of the stream whose parameter sets it uses; plus the unmodified form"""
    enc = OracleEncoder(48, 32, qp=30, gop=30)
    idr = enc.encode(synth.sequence("s1", 48, 32, 1)[0])[0]
    units = []
    for field in _CRAFTED_FIELDS + ("none",):
        for au in _crafted_p_slices(field)[:1 if field == "none" else None]:
            units += [idr, au]
    # 32 zero bits where a code word should start (no terminating 1 within reach)
    units += [idr, _Bits().ue(0).ue(5).ue(0).u(8, 1).u(3, 0).se(0).ue(1).u(32, 0).u(8, 0xFF).nal(0x41)]
    return units


def test_exp_golomb_values_of_two_to_the_31_are_refused():
    """mb_skip_run was read as unsigned and checked as int - a code word with 31 leading zeros passed the check
    and skip_mb() wrote far past the picture's arrays.  Every ue() / se() element of a P slice is given such a value in turn: the
    product parser must refuse the unit (or, where the element is legitimately that wide, keep every array in range), and the
    unmodified crafted slice must parse - the crafting itself is sound.  The same units run under AddressSanitizer in
    test_parser_under_address_sanitizer."""
    enc = OracleEncoder(48, 32, qp=30, gop=30)
    idr = enc.encode(synth.sequence("s1", 48, 32, 1)[0])[0]
    par = h264dec.Parser()
    assert par.parse(idr)
    assert par.parse(_crafted_p_slices("none")[0]), "the crafted slice is well-formed"
    mb = par.arrays()[0]
    assert list(mb["type"]) == [2, 2, 1, 2, 7, 0], "two skipped, one 16x16, one skipped, one 8x8, one Intra16x16 macroblock"
    for field in _CRAFTED_FIELDS:
        for au in _crafted_p_slices(field):
            assert par.parse(idr)
            with pytest.raises(h264dec.StreamError):
                par.parse(au)
    par.close()


def test_a_dropped_reference_picture_is_noticed():
    """parser and reconstruction ring must not drift apart.  With one P access unit dropped from a stream, the next P
    picture's frame_num no longer follows the previous reference picture's (7.4.3): it is refused, and so is every further P
    picture until an IDR picture arrives - never predicted from the wrong picture.  A refused unit has the same effect."""
    enc = OracleEncoder(64, 48, qp=28, gop=5, refs=2)
    aus = [enc.encode(f)[0] for f in synth.sequence("s1", 64, 48, 12)]
    par = h264dec.Parser()
    for au in aus[:3]:
        assert par.parse(au)
    with pytest.raises(h264dec.StreamError, match="missing"):
        par.parse(aus[4])                                  # aus[3] was lost on the way
    with pytest.raises(h264dec.StreamError, match="without a reference"):
        par.parse(aus[4])                                  # and nothing is predicted from the ring until ...
    for au in aus[5:8]:                                    # ... the next IDR picture (picture 5)
        assert par.parse(au)
    with pytest.raises(h264dec.StreamError):
        par.parse(aus[8][: len(aus[8]) // 2])              # a damaged reference picture: refused ...
    with pytest.raises(h264dec.StreamError, match="without a reference"):
        par.parse(aus[9])                                  # ... and the picture after it is not decoded against a stale list
    assert par.parse(aus[10]) and par.parse(aus[11])       # picture 10 is an IDR picture
    par.close()
