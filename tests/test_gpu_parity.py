"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the
C ABI of include/mi355x_h264.h, against the CPU oracle on the same seeded inputs --
bit-exact (integer/byte work): access units, motion vectors, modes, levels, TotalCoeff,
reconstruction before and after the loop filter -- plus the committed golden vectors and,
at the BASELINE.json full size, size-independent properties (decode round trip,
determinism, batch == frame-by-frame, GOP sharding == serial)."""
import hashlib
import json
import os
import numpy as np
import pytest
from media_amd import capi, synth
from oracle_lib import OracleEncoder, OracleDecoder

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "oracle_vectors.json")))


def _compare_all(enc, orc, tag):
    mb, omb = enc.debug_read(capi.DBG_MBINFO), orc.mbinfo()
    for f in ("mvx", "mvy", "type", "i16_mode", "chroma_mode", "cbp", "tc"):
        assert np.array_equal(mb[f], omb[f]), "%s: mbinfo.%s" % (tag, f)
    # levels: every block the entropy coder reads (by macroblock type and coded_block_pattern, 7.3.5.3) must agree; what it
    # never reads (blocks of an 8x8 quadrant / chroma parts whose coded_block_pattern bit is clear, skipped macroblocks) is
    # unspecified: the GPU does not spend HBM writes on it
    glv, olv = enc.debug_read(capi.DBG_LEVELS), orc.levels()
    read = np.zeros(olv.shape, dtype=bool)
    i16 = omb["type"] == 0
    cbp = omb["cbp"].astype(np.int32)
    read[i16, 0:16] = True
    for q in range(4):
        read[(cbp >> q) & 1 == 1, 16 + 64 * q: 16 + 64 * (q + 1)] = True
    read[(cbp >> 4) >= 1, 272:280] = True
    read[(cbp >> 4) == 2, 280:408] = True
    read[omb["type"] == 3, :] = False            # I_PCM: samples, no levels
    inter = np.isin(omb["type"], (1, 2, 5, 6, 7))  # the vectors of the four 8x8 quadrants (16x8 / 8x16 / 8x8 partitions)
    if inter.any():
        assert np.array_equal(enc.debug_read(capi.DBG_MVQ)[inter], orc.mvq()[inter]), tag + ": quadrant vectors"
    i4 = omb["type"] == 4                         # Intra4x4: the sixteen modes
    if i4.any():
        assert np.array_equal(enc.debug_read(capi.DBG_MBAUX)[i4], orc.mbaux()[i4]), tag + ": Intra4x4 modes"
    assert np.array_equal(np.where(read, glv, 0), np.where(read, olv, 0)), tag + ": levels"
    for p in range(3):
        assert np.array_equal(enc.debug_read(capi.DBG_PRE_Y + p), orc.recon_pre(p)), "%s: pre-filter plane %d" % (tag, p)
        assert np.array_equal(enc.debug_read(capi.DBG_RECON_Y + p), orc.recon(p)), "%s: recon plane %d" % (tag, p)


@pytest.mark.parametrize("case", GOLD["cases"], ids=[c["name"] for c in GOLD["cases"]])
def test_golden_cases_bit_exact(case):
    w, h = case["width"], case["height"]
    enc = capi.Encoder(w, h, qp=case["qp"], gop=case["gop"], profile_idc=case["profile_idc"], slices=case.get("slices", 0), refs=case.get("refs", 0), search=case.get("search", 0))
    enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=case["qp"], gop=case["gop"], profile_idc=case["profile_idc"], slices=case.get("slices", 0), refs=case.get("refs", 0), search=case.get("search", 0))
    for i, (f, g) in enumerate(zip(synth.sequence(case["kind"], w, h, len(case["frames"])), case["frames"])):
        bs, ft = enc.encode(f)
        obs, idr = orc.encode(f)
        assert (ft == capi.FRAME_IDR) == idr == g["idr"]
        assert hashlib.sha256(bs).hexdigest() == g["sha256"], "frame %d vs golden vector" % i
        assert bs == obs, "frame %d vs oracle" % i
        _compare_all(enc, orc, "%s frame %d" % (case["name"], i))
    enc.close()


@pytest.mark.parametrize("kind,qp", [("s1", 20), ("s1", 34), ("s3", 26), ("s2", 40)])
def test_more_content_and_qps(kind, qp):
    w, h = 352, 288
    enc = capi.Encoder(w, h, qp=qp, gop=5)
    enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=qp, gop=5)
    for i, f in enumerate(synth.sequence(kind, w, h, 7)):
        assert enc.encode(f)[0] == orc.encode(f)[0], "frame %d" % i
        _compare_all(enc, orc, "frame %d" % i)
    enc.close()


@pytest.mark.parametrize("kind,qp,w,h", [("s1", 22, 352, 288), ("s1", 34, 320, 240), ("cut", 27, 352, 288), ("s3", 16, 208, 160), ("scroll", 30, 130, 98)])
def test_high_profile_8x8_transform(kind, qp, w, h):
    """profile_idc 100: transform_8x8_mode_flag = 1, inter macroblocks go through the 8x8 transform (k_tq8: 64 samples per lane,
    CAVLC as four interleaved lists, no 4x4-internal deblocking edges): every stage equals the oracle, whose stream the
    independent decoder reconstructs byte for byte (tests/test_oracle_roundtrip.py)"""
    enc = capi.Encoder(w, h, qp=qp, gop=30, profile_idc=100)
    enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=qp, gop=30, profile_idc=100)
    dec = OracleDecoder()
    n8 = 0
    for i, f in enumerate(synth.sequence(kind, w, h, 5)):
        bs = enc.encode(f)[0]
        assert bs == orc.encode(f)[0], "picture %d" % i
        _compare_all(enc, orc, "high picture %d" % i)
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p))
        mb = orc.mbinfo()
        n8 += int(((mb["type"] == 1) & (mb["i16_mode"] == 1)).sum())
    assert n8 > 0 or kind == "s3"
    enc.close()


@pytest.mark.parametrize("refs,kind,prof,slices,w,h", [(3, "cut", 66, 0, 352, 288), (2, "s1", 66, 0, 320, 240), (3, "s1", 100, 3, 352, 288), (3, "scroll", 77, 0, 208, 160)])
def test_multiple_reference_frames(refs, kind, prof, slices, w, h):
    """config.refs = 2 / 3 (BASELINE.json configs[4]: 3-ref motion search): every available reference picture is searched, the
    cheapest wins, ref_idx_l0 is coded (te(v)), vector prediction and boundary strengths look at the reference indices, the
    reconstruction buffers form a ring (sliding window).  Bit-exact against the oracle, decodable by the test decoder."""
    enc = capi.Encoder(w, h, qp=27, gop=6, profile_idc=prof, slices=slices, refs=refs)
    enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=27, gop=6, profile_idc=prof, slices=slices, refs=refs)
    dec = OracleDecoder()
    used = np.zeros(3, int)
    for i, f in enumerate(synth.sequence(kind, w, h, 9)):      # crosses an IDR: the window restarts
        bs = enc.encode(f)[0]
        assert bs == orc.encode(f)[0], "picture %d" % i
        _compare_all(enc, orc, "%d refs picture %d" % (refs, i))
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p))
        mb = orc.mbinfo()
        inter = (mb["type"] == 1) | (mb["type"] == 2)
        for r in range(3):
            used[r] += int((inter & (mb["chroma_mode"] == r)).sum())
    assert used[0] > 0 and used[refs:].sum() == 0
    enc.close()


def test_forced_idr_qp_change_and_strided_input():
    w, h = 176, 144
    enc = capi.Encoder(w, h, qp=30, gop=100)
    orc = OracleEncoder(w, h, qp=30, gop=100)
    for i, f in enumerate(synth.sequence("s1", w, h, 6)):
        if i == 2:
            enc.force_idr()
        if i == 4:
            enc.set_qp(37)
            orc.set_qp(37)
        a, ft = enc.encode(f)
        b, idr = orc.encode(f, force_idr=(i == 2))
        assert a == b and (ft == capi.FRAME_IDR) == idr
    enc.close()


def test_nv12_ingest_main_profile_1080p60():
    """BASELINE.json configs[2]: 1080p60 NV12, main profile, CABAC off (CAVLC): the kernels read the interleaved
    chroma plane directly and must give exactly the I420 result"""
    w, h = 1920, 1080
    enc = capi.Encoder(w, h, qp=26, gop=30, fps=60, profile_idc=77)
    orc = OracleEncoder(w, h, qp=26, gop=30, fps=60, profile_idc=77)
    for i, f in enumerate(synth.sequence("s1", w, h, 3)):
        y, u, v = f[: w * h], f[w * h: w * h * 5 // 4], f[w * h * 5 // 4:]
        nv12 = np.concatenate([y, np.stack([u, v], axis=1).ravel()])
        bs, _ = enc.encode_nv12(nv12)
        assert bs == orc.encode(f)[0]
        if i == 0:
            assert bs[:8] == bytes([0, 0, 0, 1, 0x67, 77, 0x40, 42])   # main profile, level 4.2 for 1080p60
    enc.close()
    # geometry whose chroma rows are not 8-byte aligned / end inside a macroblock: the clamped per-sample path
    w, h = 130, 98
    enc = capi.Encoder(w, h, qp=28)
    orc = OracleEncoder(w, h, qp=28)
    for f in synth.sequence("s1", w, h, 2):
        y, u, v = f[: w * h], f[w * h: w * h * 5 // 4], f[w * h * 5 // 4:]
        assert enc.encode_nv12(np.concatenate([y, np.stack([u, v], axis=1).ravel()]))[0] == orc.encode(f)[0]
    enc.close()


def _to_nv12(f, w, h):
    y, u, v = f[: w * h], f[w * h: w * h * 5 // 4], f[w * h * 5 // 4:]
    return np.concatenate([y, np.stack([u, v], axis=1).ravel()])


def test_rgba_ingest_host_strided_and_device():
    """RGBA pictures (row f3): the conversion kernel + encoder against the oracle's conversion + encoder, from host memory
    (tight rows, and rows with a stride), and from device memory; odd geometry (width not a multiple of 16)."""
    import torch
    from oracle_lib import rgba_to_i420
    rng = np.random.default_rng(5)
    for (w, h) in ((320, 240), (1920, 1080), (200, 120)):
        enc = capi.Encoder(w, h, qp=26, gop=30)
        orc = OracleEncoder(w, h, qp=26, gop=30)
        # a moving colour texture: smooth gradients + noise, alpha random (must not matter)
        yy, xx = np.mgrid[0:h, 0:w]
        for i in range(3):
            pic = np.empty((h, w, 4), np.uint8)
            pic[..., 0] = (xx * 2 + 3 * i) & 255
            pic[..., 1] = (yy * 3 + xx + 5 * i) & 255
            pic[..., 2] = ((xx ^ yy) + 7 * i) & 255
            pic[..., :3] = np.clip(pic[..., :3].astype(np.int16) + rng.integers(-6, 7, (h, w, 3)), 0, 255).astype(np.uint8)
            pic[..., 3] = rng.integers(0, 256, (h, w))
            want = orc.encode(rgba_to_i420(pic, w, h))[0]
            if i == 0:
                got = enc.encode_rgba(pic)[0]
            elif i == 1:
                wide = np.zeros((h, w + 10, 4), np.uint8)
                wide[:, :w] = pic
                got = enc.encode_rgba(wide, stride=4 * (w + 10))[0]
            else:
                got = enc.encode_rgba_device(torch.from_numpy(pic).cuda().data_ptr())[0]
            assert got == want, "%dx%d picture %d" % (w, h, i)
        enc.close()


def test_nv12_device_pictures_in_lockstep_batch():
    """config.input_format = NV12: device-resident NV12 pictures go through the lockstep batch (and the
    batch-1 device entry points) with no conversion pass and give the I420 stream"""
    import torch
    w, h, gop, G = 352, 288, 4, 3
    frames = synth.sequence("s1", w, h, gop * G)
    fbytes = w * h * 3 // 2
    orc = OracleEncoder(w, h, qp=27, gop=gop)
    want = [orc.encode(f)[0] for f in frames]
    dev = torch.from_numpy(np.stack([_to_nv12(f, w, h) for f in frames])).cuda()
    enc = capi.Encoder(w, h, qp=27, gop=gop, batch=G, input_format=1)
    cap = gop * fbytes
    out, sizes, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
    enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, sizes, gb)
    for g in range(G):
        assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), "GOP %d" % g
    enc.close()
    enc = capi.Encoder(w, h, qp=27, gop=gop, input_format=1)
    for i in range(gop + 1):
        assert enc.encode_device(dev[i].data_ptr())[0] == want[i]
    enc.close()


@pytest.mark.parametrize("kind,w,h,qp,prof,refs,G", [("s1", 352, 288, 20, 66, 0, 1), ("cut", 320, 240, 28, 100, 2, 1), ("split", 208, 160, 26, 66, 0, 3),
                                                     ("s3", 130, 98, 34, 77, 0, 2), ("s1", 720, 720, 30, 66, 0, 1), ("scroll", 16, 48, 30, 66, 0, 1)])
def test_loop_filter_with_two_rows_per_wave(monkeypatch, kind, w, h, qp, prof, refs, G):
    """MI355X_H264_PAIR_FILTER: k_deblock_pairs (two macroblock rows per wave, the lower row two macroblocks behind, hand-off
    through LDS inside the pair and through global granules between pairs) must give the stream and reconstruction of the
    one-row-per-wave form: even and odd numbers of macroblock rows, IDR and P pictures, intra macroblocks in P pictures,
    single pictures and lockstep batches."""
    monkeypatch.setenv("MI355X_H264_PAIR_FILTER", "1")
    frames = synth.sequence(kind, w, h, 3 * max(G, 2))
    orc = OracleEncoder(w, h, qp=qp, gop=3, profile_idc=prof, refs=refs)
    want = [orc.encode(f)[0] for f in frames]
    if G == 1:
        enc = capi.Encoder(w, h, qp=qp, gop=3, profile_idc=prof, refs=refs)
        orc = OracleEncoder(w, h, qp=qp, gop=3, profile_idc=prof, refs=refs)
        for i, f in enumerate(frames):
            assert enc.encode(f)[0] == orc.encode(f)[0], "picture %d" % i
            for p in range(3):
                assert np.array_equal(enc.debug_read(capi.DBG_RECON_Y + p), orc.recon(p)), "picture %d plane %d" % (i, p)
    else:
        import torch
        fbytes, gop = w * h * 3 // 2, 3
        dev = torch.from_numpy(np.stack(frames[: G * gop])).cuda()
        enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, refs=refs, batch=G)
        cap = 2 * gop * fbytes
        out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
        enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
        for g in range(G):
            assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), "GOP %d" % g
    enc.close()


@pytest.mark.parametrize("setting,w,h", [(None, 176, 112), ("0", 176, 112), (None, 64, 16), (None, 16, 64), (None, 18, 18)])
def test_loop_filter_forms_in_a_batch_of_eight(monkeypatch, setting, w, h):
    """from a lockstep batch of 8 pictures on the loop filter takes two macroblock rows per wave by default
    (MI355X_H264_PAIR_FILTER=0: one row per wave); both forms must give the oracle's streams - here with an odd number of
    macroblock rows and intra macroblocks in the P pictures, and on pictures of one macroblock row / column"""
    import torch
    if setting is not None:
        monkeypatch.setenv("MI355X_H264_PAIR_FILTER", setting)
    G, gop = 8, 3
    frames = synth.sequence("cut", w, h, G * gop)
    orc = OracleEncoder(w, h, qp=27, gop=gop)
    want = [orc.encode(f)[0] for f in frames]
    fbytes = w * h * 3 // 2
    dev = torch.from_numpy(np.stack(frames)).cuda()
    enc = capi.Encoder(w, h, qp=27, gop=gop, batch=G)
    cap = 2 * gop * fbytes
    out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
    enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
    for g in range(G):
        assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), "GOP %d" % g
    enc.close()


def test_reference_forms_of_the_kernels(monkeypatch):
    """the simpler first forms of the two row-wavefront kernels (one launch per wavefront step) stay selectable for
    debugging and must give the same stream"""
    monkeypatch.setenv("MI355X_H264_DIAG", "1")
    w, h = 208, 160
    enc = capi.Encoder(w, h, qp=27, gop=3)
    orc = OracleEncoder(w, h, qp=27, gop=3)
    for f in synth.sequence("s1", w, h, 5):
        assert enc.encode(f)[0] == orc.encode(f)[0]
    enc.close()


def test_emulation_prevention_slow_path():
    """slices whose payload needs 00 00 03 escapes: k_pack counts the sites, the host inserts the bytes"""
    w, h = 320, 240
    for qp in (26, 30):
        enc = capi.Encoder(w, h, qp=qp, gop=100)
        orc = OracleEncoder(w, h, qp=qp, gop=100)
        escaped = 0
        for f in synth.sequence("ramp", w, h, 6):
            bs, _ = enc.encode(f)
            assert bs == orc.encode(f)[0]
            escaped += b"\x00\x00\x03" in bs[bs.rfind(b"\x00\x00\x00\x01") + 5:]
        assert escaped >= 2
        enc.close()


def test_scene_change_statistic_matches_oracle():
    w, h = 320, 240
    enc = capi.Encoder(w, h, qp=26, gop=100)
    orc = OracleEncoder(w, h, qp=26, gop=100)
    for f in synth.sequence("s1", w, h, 3) + synth.sequence("s3", w, h, 1):
        assert enc.encode(f)[0] == orc.encode(f)[0]
        assert int(enc.me_cost()[0]) == orc.me_cost()
    enc.close()


def test_no_deblock_variant():
    w, h = 160, 96
    enc = capi.Encoder(w, h, qp=32, disable_deblock=1)
    orc = OracleEncoder(w, h, qp=32, disable_deblock=1)
    for f in synth.sequence("s1", w, h, 3):
        assert enc.encode(f)[0] == orc.encode(f)[0]
    enc.close()


def test_full_size_1080p_properties():
    """BASELINE.json configs[1] size: every picture against the oracle, plus size-independent properties"""
    import torch
    w, h, n = 1920, 1080, 12
    frames = synth.sequence("s1", w, h, n)
    enc = capi.Encoder(w, h, qp=26, gop=30)
    orc = OracleEncoder(w, h, qp=26, gop=30)
    dec = OracleDecoder()
    stream = []
    for i, f in enumerate(frames):
        bs, ft = enc.encode(f)
        stream.append(bs)
        assert bs == orc.encode(f)[0], "frame %d vs oracle" % i   # ~0.25 s of oracle time per 1080p P picture
        # property 1: an independent decoder reproduces the device reconstruction exactly
        assert dec.decode(bs) == 1 and dec.size == (w, h)
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p)), "frame %d plane %d" % (i, p)
        assert synth.psnr(f[: w * h].reshape(h, w), dec.plane(0)[:h, :w]) > 36.0
    enc.close()
    # property 2: determinism + batch (device-resident, pipelined) == frame by frame (host input)
    fbytes = w * h * 3 // 2
    dev = torch.from_numpy(np.stack(frames)).cuda()
    enc2 = capi.Encoder(w, h, qp=26, gop=30)
    out = np.zeros(n * fbytes // 2, np.uint8)
    sizes = np.zeros(n, np.uint32)
    tot = enc2.encode_batch_device(dev.data_ptr(), fbytes, n, out, sizes)
    assert tot == sum(len(s) for s in stream) and out[:tot].tobytes() == b"".join(stream)
    assert [int(s) for s in sizes] == [len(s) for s in stream]
    enc2.close()


def test_gop_sharding_equals_serial():
    """closed GOPs encoded by two encoder instances (as two GPUs / two HIP streams would)
    concatenate to exactly the serial stream"""
    w, h, gop, n_gops = 320, 240, 4, 4
    frames = synth.sequence("s1", w, h, gop * n_gops)
    serial = capi.Encoder(w, h, qp=26, gop=gop)
    want = b"".join(serial.encode(f)[0] for f in frames)
    serial.close()
    parts = {}
    for r in range(2):
        e = capi.Encoder(w, h, qp=26, gop=gop)
        e.set_idr_pic_id(r, 2)
        for k in range(r, n_gops, 2):
            parts[k] = b"".join(e.encode(f)[0] for f in frames[k * gop:(k + 1) * gop])
        e.close()
    assert b"".join(parts[k] for k in sorted(parts)) == want


def test_lockstep_batch_equals_serial():
    """config.batch = 4: four closed GOPs of one stream encoded in lockstep (grid.y = 4) must concatenate to
    exactly the stream a batch-1 encoder (and the oracle) produces"""
    import torch
    w, h, gop, G = 352, 288, 5, 4
    frames = synth.sequence("s1", w, h, gop * G)
    fbytes = w * h * 3 // 2
    orc = OracleEncoder(w, h, qp=27, gop=gop)
    want = [orc.encode(f)[0] for f in frames]
    dev = torch.from_numpy(np.stack(frames)).cuda()
    enc = capi.Encoder(w, h, qp=27, gop=gop, batch=G)
    cap = gop * fbytes
    out = np.zeros(G * cap, np.uint8)
    sizes = np.zeros(G * gop, np.uint32)
    gb = np.zeros(G, np.uint64)
    enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, sizes, gb)
    for g in range(G):
        got = out[g * cap: g * cap + int(gb[g])].tobytes()
        assert got == b"".join(want[g * gop:(g + 1) * gop]), "GOP %d" % g
        assert [int(x) for x in sizes[g * gop:(g + 1) * gop]] == [len(x) for x in want[g * gop:(g + 1) * gop]]
    # a second call continues the idr_pic_id sequence (GOPs G..2G-1 of the same stream)
    more = synth.sequence("s1", w, h, gop * G, start=gop * G)
    want2 = [orc.encode(f)[0] for f in more]
    dev2 = torch.from_numpy(np.stack(more)).cuda()
    enc.encode_gops_device(dev2.data_ptr(), fbytes, gop * fbytes, gop, out, cap, sizes, gb)
    for g in range(G):
        assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want2[g * gop:(g + 1) * gop]), "second call GOP %d" % g
    enc.close()


def test_two_instances_of_sixteen_gops_side_by_side_with_cuts():
    """Two engines of 16 closed GOPs each on two host threads (the bench's arrangement, from the size on where the GPU's
    motion-search lock and the picture-walking grids of k_pintra_rows / k_i4_decide / the bS-4 loop filter are in use): every other
    GOP has a cut after its second picture, so one lockstep step mixes P pictures with and without intra macroblocks, and the
    second call finds the engine's intra statistics above the sparse threshold (all pictures of the step resident).  Every GOP ==
    the oracle's serial stream."""
    import threading
    import torch
    w, h, gop, G = 320, 240, 4, 16
    fbytes = w * h * 3 // 2

    def gops(first):   # GOP k: cut content when k is odd
        out = []
        for k in range(first, first + G):
            out += synth.sequence("cut" if k & 1 else "s1", w, h, gop, start=0 if k & 1 else 3 * k)
        return out

    errors = []

    def run(inst):
        try:
            orc = OracleEncoder(w, h, qp=27 + inst, gop=gop)
            enc = capi.Encoder(w, h, qp=27 + inst, gop=gop, batch=G)
            cap = gop * fbytes
            out = np.zeros(G * cap, np.uint8)
            sizes = np.zeros(G * gop, np.uint32)
            gb = np.zeros(G, np.uint64)
            for call in range(3):
                frames = gops(call * G + inst)
                want = [orc.encode(f)[0] for f in frames]
                dev = torch.from_numpy(np.stack(frames)).cuda()
                enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, sizes, gb)
                for g in range(G):
                    if out[g * cap: g * cap + int(gb[g])].tobytes() != b"".join(want[g * gop:(g + 1) * gop]):
                        errors.append((inst, call, g))
            enc.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((inst, repr(ex)))

    ths = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def test_concurrent_instances_under_uneven_load():
    """the row-wavefront kernels hand samples between workgroups inside one launch; run several encoder
    instances of different geometry concurrently (uneven load on the chip, L1-warm consumers) and check
    every access unit against the oracle"""
    import threading
    jobs = [(352, 288, "s1", 26, 8), (640, 368, "s1", 30, 6), (176, 144, "s3", 24, 10), (1280, 720, "s1", 26, 4)]
    want = {}
    for j in jobs:
        w, h, kind, qp, n = j
        orc = OracleEncoder(w, h, qp=qp, gop=3)
        want[j] = [orc.encode(f)[0] for f in synth.sequence(kind, w, h, n)]
    errors = []

    def run(j, rounds):
        try:
            w, h, kind, qp, n = j
            frames = synth.sequence(kind, w, h, n)
            for _ in range(rounds):
                enc = capi.Encoder(w, h, qp=qp, gop=3)
                for i, f in enumerate(frames):
                    if enc.encode(f)[0] != want[j][i]:
                        errors.append((j, i))
                        return
                enc.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((j, repr(ex)))

    ths = [threading.Thread(target=run, args=(j, 6 if j[0] < 1000 else 3)) for j in jobs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors


def test_4k_one_gop_start():
    """BASELINE.json configs[4] size (3840x2160): IDR + P equal the oracle's and decode"""
    w, h = 3840, 2160
    enc = capi.Encoder(w, h, qp=30, gop=30)
    dec = OracleDecoder()
    orc = OracleEncoder(w, h, qp=30, gop=30)
    for i, f in enumerate(synth.sequence("s1", w, h, 2)):
        bs, _ = enc.encode(f)
        assert bs == orc.encode(f)[0], "picture %d" % i
        assert dec.decode(bs) == 1
        for p in range(3):
            assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p))
    enc.close()


def test_maximum_picture_size_4096():
    """the largest picture the reference accepts (16..4096 on each side, VideoEncoderOpenH264.cpp:16-23,:159-171):
    65 536 macroblocks, IDR + P bit-exact against the oracle (oracle: about a second per picture)"""
    w, h = 4096, 4096
    enc = capi.Encoder(w, h, qp=32, gop=30)
    orc = OracleEncoder(w, h, qp=32, gop=30)
    for f in synth.sequence("s1", w, h, 2):
        bs, _ = enc.encode(f)
        assert bs == orc.encode(f)[0]
    enc.close()


def test_randomized_configurations():
    """seeded sweep over geometry (odd macroblock counts, crops, one-macroblock pictures), QP, GOP length, profile,
    loop filter on/off, I420 / NV12 and lockstep batch size: every access unit against the oracle"""
    import random
    import torch
    rng = random.Random(20261004)
    sizes = [(16, 16), (32, 16), (16, 48), (48, 80), (50, 34), (178, 98), (130, 66), (256, 32), (34, 226), (320, 176), (98, 130), (2048, 16)]
    for case in range(36):
        w, h = sizes[case] if case < len(sizes) else (2 * rng.randint(8, 200), 2 * rng.randint(8, 150))
        qp = rng.choice([14, 22, 26, 31, 37, 44])
        gop = rng.choice([1, 2, 3, 5])
        prof = rng.choice([66, 77, 100])
        nodb = rng.random() < 0.25
        nv12 = rng.random() < 0.5
        G = rng.choice([1, 1, 2, 3])
        kind = rng.choice(["s1", "s1", "s2", "s3", "ramp", "scroll", "scroll"]) if case % 6 != 5 else "split"
        frames = synth.sequence(kind, w, h, gop * G if G > 1 else 5)
        tag = "case %d: %dx%d qp %d gop %d profile %d nodeblock %d nv12 %d batch %d %s" % (case, w, h, qp, gop, prof, nodb, nv12, G, kind)
        orc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb))
        want = [orc.encode(f)[0] for f in frames]
        pics = [(_to_nv12(f, w, h) if nv12 else f) for f in frames]
        dev = torch.from_numpy(np.stack(pics)).cuda()
        fbytes = w * h * 3 // 2
        enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), batch=G, input_format=int(nv12))
        if G == 1:
            for i in range(len(frames)):
                assert enc.encode_device(dev[i].data_ptr())[0] == want[i], tag + " picture %d" % i
        else:
            cap = max(4096, gop * fbytes * 2)
            out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
            enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
            for g in range(G):
                assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), tag + " GOP %d" % g
        enc.close()


@pytest.mark.parametrize("w,h,qp,prof,refs,slices", [(352, 288, 26, 66, 0, 0), (320, 240, 30, 100, 2, 0), (208, 160, 22, 77, 3, 3), (130, 98, 34, 100, 3, 2)])
def test_partitions_16x8_8x16_8x8(w, h, qp, prof, refs, slices):
    """two layers drifting apart by a fraction of a sample in 8-sample stripes ("split" content): the macroblocks split into
    16x8 (upper third), 8x16 (middle) and 8x8 (lower third) partitions.  Quadrant vectors, types, levels, reconstruction and
    bitstream equal the oracle's, whose streams the independent decoder reconstructs (tests/test_oracle_roundtrip.py);
    also through the lockstep batch."""
    enc = capi.Encoder(w, h, qp=qp, gop=30, profile_idc=prof, refs=refs, slices=slices)
    enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=qp, gop=30, profile_idc=prof, refs=refs, slices=slices)
    seen = np.zeros(8, np.int64)
    frames = synth.sequence("split", w, h, 5)
    for i, f in enumerate(frames):
        assert enc.encode(f)[0] == orc.encode(f)[0], "picture %d" % i
        _compare_all(enc, orc, "split picture %d" % i)
        if i:
            seen += np.bincount(orc.mbinfo()["type"], minlength=8)
    enc.close()
    assert seen[5] > 0 and seen[6] > 0 and seen[7] > 0, seen
    import torch
    G, gop, fbytes = 2, 3, w * h * 3 // 2
    dev = torch.from_numpy(np.stack(frames + frames[:1])).cuda()
    enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, refs=refs, slices=slices, batch=G)
    cap = 2 * gop * fbytes
    out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
    enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
    orc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, refs=refs, slices=slices)
    want = [orc.encode(f)[0] for f in frames + frames[:1]]
    for g in range(G):
        assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), "GOP %d" % g
    enc.close()


def test_scrolling_content_takes_the_previous_vector():
    """noise-free scroll by (+4, +2): within a few pictures the macroblocks lock on the scroll vector, which the
    previous-picture-vector test then keeps without a search (P_Skip) - and the stream still equals the oracle's"""
    w, h = 352, 288
    enc = capi.Encoder(w, h, qp=30, gop=30)
    orc = OracleEncoder(w, h, qp=30, gop=30)
    sizes = []
    for f in synth.sequence("scroll", w, h, 8):
        bs, _ = enc.encode(f)
        assert bs == orc.encode(f)[0]
        sizes.append(len(bs))
    mb = orc.mbinfo().reshape(-1)
    moved = (mb["mvx"] == -16) & (mb["mvy"] == -8)    # the content moves by (+4, +2) samples: the reference lies at (-4, -2)
    assert moved.mean() > 0.8 and (mb["type"] == 2).mean() > 0.6, (moved.mean(), (mb["type"] == 2).mean())
    assert sizes[-1] < sizes[1]          # later P pictures cost less than the first one (which had to search)
    enc.close()


def test_large_global_motion_and_noise_pictures():
    """seeded sweep aimed at the motion path's corners: global motion up to +-20 samples per picture (beyond the
    +-16 search range, so vectors pile up at the window edge and previous-picture vectors point outside the picture),
    pure-noise pictures (nothing ever quantises to zero, every search runs to the end, worst-case bit counts), very
    low and very high QP, GOPs that restart mid-sequence"""
    import random
    rng = random.Random(11)
    for case in range(120):
        w, h = 2 * rng.randint(8, 200), 2 * rng.randint(8, 150)
        qp = rng.choice([12, 18, 24, 28, 33, 40, 48])
        mot = (rng.randint(-20, 20), rng.randint(-20, 20))
        noise = rng.choice([0, 0, 1, 3, 8])
        gop = rng.choice([3, 5, 30])
        kind = rng.choice(["s1", "s1", "scroll", "rand"])
        if kind == "s1":
            frames = [synth.frame_s1(w, h, i, noise=noise, motion=mot) for i in range(8)]
        elif kind == "scroll":
            frames = [synth.frame_scroll(w, h, i) for i in range(8)]
        else:
            r = np.random.default_rng(case)
            frames = [r.integers(0, 256, w * h * 3 // 2, dtype=np.uint8) for _ in range(8)]
        enc = capi.Encoder(w, h, qp=qp, gop=gop)
        orc = OracleEncoder(w, h, qp=qp, gop=gop)
        for i, f in enumerate(frames):
            assert enc.encode(f)[0] == orc.encode(f)[0], "case %d: %dx%d qp %d motion %s noise %d gop %d %s picture %d" % (
                case, w, h, qp, mot, noise, gop, kind, i)
        enc.close()


def test_slice_bands_bit_exact_and_decodable():
    """several slices per picture (BASELINE.json configs[4]): bands of whole macroblock rows, one NAL
    unit per band, disable_deblocking_filter_idc 2.  Every access unit equals the oracle's, the oracle's independent
    decoder reproduces the GPU reconstruction, and the number of slice NAL units is what the geometry says"""
    import random
    rng = random.Random(5)
    cases = [(320, 240, 4), (64, 48, 3), (178, 98, 2), (352, 288, 8), (48, 80, 5), (1920, 1080, 8), (16, 16, 4), (32, 32, 2)]
    cases += [(2 * rng.randint(8, 160), 2 * rng.randint(8, 120), rng.randint(2, 9)) for _ in range(24)]
    for w, h, sl in cases:
        qp = rng.choice([18, 26, 33, 40])
        kind = rng.choice(["s1", "s1", "scroll", "s2", "s3"])
        n = 3 if w * h > 500000 else 6
        enc = capi.Encoder(w, h, qp=qp, gop=4, slices=sl)
        orc = OracleEncoder(w, h, qp=qp, gop=4, slices=sl)
        dec = OracleDecoder()
        mbh = (h + 15) // 16
        nb = min(max(sl, 1), max(1, mbh // 2))
        rows = -(-mbh // nb)
        want_slices = -(-mbh // rows)
        for i, f in enumerate(synth.sequence(kind, w, h, n)):
            bs, _ = enc.encode(f)
            tag = "%dx%d slices %d qp %d %s picture %d" % (w, h, sl, qp, kind, i)
            assert bs == orc.encode(f)[0], tag
            nals = [bs[k + 4] & 31 for k in range(len(bs) - 4) if bs[k:k + 4] == b"\x00\x00\x00\x01"] if len(bs) < 200000 else None
            if nals is not None:
                assert sum(1 for t in nals if t in (1, 5)) == want_slices, tag
            assert dec.decode(bs) == 1, tag
            for p in range(3):
                assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p)), tag + " plane %d" % p
        enc.close()


def test_slice_bands_on_several_instances_equal_one_instance():
    """SURVEY.md 8e-3 / BASELINE.json configs[4] (slice-parallel): W encoder instances, each coding its band of whole
    slices of the SAME picture and swapping two macroblock rows of reconstruction with its neighbours after every
    picture (mi355x_h264_band_halo_export / _import; across GPUs the swap is one send/recv pair per neighbour), must
    produce byte for byte the access units ONE instance with the same number of slices makes.  Here the W instances
    share one GPU; tests/test_shard_gloo.py runs the exchange schedule over torch.distributed"""
    import torch
    cases = [(320, 240, 4, 2, "s1"), (352, 288, 6, 3, "scroll"), (178, 98, 3, 3, "s1"), (1920, 1080, 8, 4, "s1"), (640, 368, 8, 8, "s3"),
             (96, 112, 3, 2, "s1")]
    for w, h, slices, W, kind in cases:
        n = 4 if w * h > 500000 else 7
        one = capi.Encoder(w, h, qp=27, gop=5, slices=slices)
        parts = [capi.Encoder(w, h, qp=27, gop=5, slices=slices, band_index=r, band_count=W) for r in range(W)]
        info = [p.band_info() for p in parts]
        assert info[0][0] == 0 and sum(i[1] for i in info) == (h + 15) // 16 and all(info[r][0] + info[r][1] == info[r + 1][0] for r in range(W - 1))
        hb = info[0][4]
        buf = torch.empty(hb, dtype=torch.uint8, device="cuda")
        for i, f in enumerate(synth.sequence(kind, w, h, n)):
            want = one.encode(f)[0]
            got = b"".join(p.encode(f)[0] for p in parts)
            assert got == want, "%dx%d slices %d on %d instances, picture %d" % (w, h, slices, W, i)
            for r in range(W):
                if r > 0:            # my top rows become the rows below my upper neighbour
                    parts[r].halo_export(0, buf.data_ptr())
                    parts[r - 1].halo_import(1, buf.data_ptr())
                if r < W - 1:        # my bottom rows become the rows above my lower neighbour
                    parts[r].halo_export(1, buf.data_ptr())
                    parts[r + 1].halo_import(0, buf.data_ptr())
        for p in parts + [one]:
            p.close()


def test_configs4_4k_three_references_eight_slices_on_band_instances():
    """BASELINE.json configs[4] as built: 4K30 I420, 3-reference motion search, slice-parallel - here 8 slice bands on 4 band
    instances of one GPU (the driver's 8-GPU node runs tools/bench_bands.py with one rank per band), halo swap after every
    picture.  The assembled access units equal the ORACLE's (8 slices, 3 references) for an IDR and three P pictures, so
    ref_idx 1 and 2 are in play, and decode."""
    import torch
    w, h, slices, W = 3840, 2160, 8, 4
    parts = [capi.Encoder(w, h, qp=28, gop=30, slices=slices, refs=3, band_index=r, band_count=W) for r in range(W)]
    orc = OracleEncoder(w, h, qp=28, gop=30, slices=slices, refs=3)
    dec = OracleDecoder()
    from media_amd import h264dec
    gdec = h264dec.Decoder()   # the decoder peer on the same 4K / 8-slice / 3-reference stream
    buf = torch.empty(parts[0].band_info()[4], dtype=torch.uint8, device="cuda")
    for i, f in enumerate(synth.sequence("s1", w, h, 4)):
        got = b"".join(p.encode(f)[0] for p in parts)
        assert got == orc.encode(f)[0], "picture %d" % i
        assert dec.decode(got) == 1
        assert gdec.decode(got)
        for pl in range(3):
            assert np.array_equal(gdec.plane(pl), dec.plane(pl)), "picture %d plane %d: GPU decoder vs the independent decoder" % (i, pl)
        for r in range(W):
            if r > 0:
                parts[r].halo_export(0, buf.data_ptr())
                parts[r - 1].halo_import(1, buf.data_ptr())
            if r < W - 1:
                parts[r].halo_export(1, buf.data_ptr())
                parts[r + 1].halo_import(0, buf.data_ptr())
    gdec.close()
    for p in parts:
        p.close()


def test_slice_bands_with_other_options():
    """slices combined with the rest of the configuration surface: profiles, loop filter off (idc 1 instead of 2),
    NV12 device pictures, lockstep batches of GOPs - every access unit / GOP against the oracle"""
    import random
    import torch
    rng = random.Random(77)
    for case in range(24):
        w, h = 2 * rng.randint(16, 180), 2 * rng.randint(24, 140)
        sl = rng.randint(2, 7)
        qp = rng.choice([20, 27, 35])
        gop = rng.choice([2, 3, 4])
        prof = rng.choice([66, 77, 100])
        nodb = rng.random() < 0.3
        nv12 = rng.random() < 0.5
        G = rng.choice([1, 2, 3])
        kind = rng.choice(["s1", "scroll", "s3"])
        frames = synth.sequence(kind, w, h, gop * G if G > 1 else 5)
        tag = "case %d: %dx%d slices %d qp %d gop %d profile %d nodeblock %d nv12 %d batch %d %s" % (case, w, h, sl, qp, gop, prof, nodb, nv12, G, kind)
        orc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), slices=sl)
        want = [orc.encode(f)[0] for f in frames]
        pics = [(_to_nv12(f, w, h) if nv12 else f) for f in frames]
        dev = torch.from_numpy(np.stack(pics)).cuda()
        fbytes = w * h * 3 // 2
        enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), batch=G, input_format=int(nv12), slices=sl)
        if G == 1:
            for i in range(len(frames)):
                assert enc.encode_device(dev[i].data_ptr())[0] == want[i], tag + " picture %d" % i
        else:
            cap = max(8192, gop * fbytes * 2)
            out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
            enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
            for g in range(G):
                assert out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]), tag + " GOP %d" % g
        enc.close()


def test_long_gop_frame_num_wrap_and_gop_boundary_1080p():
    """(a) the longest GOP the reference accepts (gopsize 3000, VideoEncoderOpenH264.cpp:16-23): 300 P pictures in a
    row, frame_num (8 bits here) wraps past 255; (b) 1080p across a GOP boundary: pictures 28..33 of a 34-picture
    run (IDR at 30) - every access unit against the oracle, the stream decodable"""
    w, h = 64, 48
    enc = capi.Encoder(w, h, qp=30, gop=3000)
    orc = OracleEncoder(w, h, qp=30, gop=3000)
    dec = OracleDecoder()
    idrs = 0
    for i in range(300):
        f = synth.frame_s1(w, h, i)
        bs, ft = enc.encode(f)
        idrs += ft == capi.FRAME_IDR
        assert bs == orc.encode(f)[0], "picture %d" % i
        assert dec.decode(bs) == 1
    assert idrs == 1
    for p in range(3):
        assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p))
    enc.close()
    w, h = 1920, 1080
    enc = capi.Encoder(w, h, qp=26, gop=30)
    orc = OracleEncoder(w, h, qp=26, gop=30)
    for i in range(34):
        f = synth.frame_s1(w, h, i)
        bs, ft = enc.encode(f)
        assert (ft == capi.FRAME_IDR) == (i % 30 == 0)
        if i >= 28 or i < 2:
            assert bs == orc.encode(f)[0], "1080p picture %d" % i
        else:
            orc.encode(f)      # (keeps the oracle's reference in step; ~0.1 s per picture)
    enc.close()


def test_content_that_cannot_be_coded_by_cavlc_goes_i_pcm():
    """black-and-white noise at the lowest QP would code to more than 3200 bits per macroblock (and to more than the payload
    buffer): round 1 refused such a picture (MI355X_H264_E_OVERFLOW).  Now every such macroblock is coded as I_PCM - in
    IDR and in P pictures, with one slice and with several - the access units equal the oracle's, and the test decoder
    finds a legal stream: no macroblock_layer() above 3200 bits (A.3.1), no level_prefix above 15 (A.2)"""
    w, h = 640, 480
    rng = np.random.default_rng(3)
    noise = [(rng.integers(0, 2, w * h * 3 // 2, dtype=np.uint8) * 255).astype(np.uint8) for _ in range(3)]   # black / white noise
    for slices in (0, 4):
        enc = capi.Encoder(w, h, qp=10, gop=30, slices=slices)
        enc.keep_pre(True)
        orc = OracleEncoder(w, h, qp=10, gop=30, slices=slices)
        dec = OracleDecoder()
        for k, f in enumerate(noise):
            if k == 2:                          # and back to a QP where CAVLC fits again: a mix of I_PCM and coded macroblocks
                enc.set_qp(30)
                orc.set_qp(30)
            bs, ft = enc.encode(f)
            assert bs == orc.encode(f)[0], "picture %d (%d slices)" % (k, slices)
            _compare_all(enc, orc, "pcm picture %d" % k)
            assert dec.decode(bs) == 1
            kinds = dec.mb_kinds()
            assert (kinds == dec.KIND_IPCM).sum() > (len(kinds) // 2 if k < 2 else 0)
            assert dec.max_mb_bits <= 3200 and dec.max_level_prefix <= 15
            for p in range(3):
                assert np.array_equal(dec.plane(p), enc.debug_read(capi.DBG_RECON_Y + p))
        enc.close()
    # the lockstep batch path
    import torch
    G, gop, fbytes = 2, 3, w * h * 3 // 2
    dev = torch.from_numpy(np.stack(noise * G)).cuda()
    enc = capi.Encoder(w, h, qp=10, gop=gop, batch=G)
    cap = 4 * gop * fbytes
    out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
    enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
    orc = OracleEncoder(w, h, qp=10, gop=gop)
    assert out[:int(gb[0])].tobytes() == b"".join(orc.encode(f)[0] for f in noise)
    enc.close()


def test_intra_macroblocks_inside_p_pictures():
    """a cut inside a GOP (no scene-change IDR at the C ABI: that is the plugin class's rule): the P picture after it mixes
    intra (16x16 / 4x4) and inter macroblocks; every stage equals the oracle, with one slice and with slice bands"""
    for (w, h, slices) in ((352, 288, 0), (640, 368, 4)):
        enc = capi.Encoder(w, h, qp=28, gop=30, slices=slices)
        enc.keep_pre(True)
        orc = OracleEncoder(w, h, qp=28, gop=30, slices=slices)
        n_intra = 0
        for i, f in enumerate(synth.sequence("cut", w, h, 5)):
            assert enc.encode(f)[0] == orc.encode(f)[0], "picture %d" % i
            _compare_all(enc, orc, "cut picture %d" % i)
            if i >= 2:
                n_intra += int(np.isin(orc.mbinfo()["type"], (0, 3, 4)).sum())   # Intra16x16, I_PCM, Intra4x4
        assert n_intra > 0
        enc.close()
