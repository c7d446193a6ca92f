"""Oracle pinning, part 3: the test decoder (oracle/h264_dec.c) is independent of the encoder side.

(a) it uses no function of h264_common.c / h264_enc.c and does not include h264_tables.h;
(b) its tables, typed from the standard in the standard's printed layout (VLC tables as bit strings), agree entry by
    entry with the encoder's h264_tables.h, typed in (length, bits) form;
(c) mutation tests: an error planted in ONE normative formula or table entry on the encoder side (filter tap, rounding,
    dequantiser entry, tC0 entry, boundary-strength rule, intra plane coefficient, chroma weights) makes the round trip
    decode(encode(x)) == reconstruction FAIL.  With the round-1 decoder, which called the encoder's own functions, every
    one of these mutants passed the round trip.
CPU only."""
import ctypes as C
import os
import re
import shutil
import subprocess
import numpy as np
import pytest
import oracle_lib as ol
from media_amd import synth
from test_oracle_kat import _c_table

ODIR = ol._ODIR


def test_decoder_source_shares_nothing_with_the_encoder_side():
    src = open(os.path.join(ODIR, "h264_dec.c")).read()
    code = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    assert "h264_tables.h" not in code
    used = set(re.findall(r"\bh264o_[a-z0-9_]+", code))
    assert all(n == "h264o_dec" or n.startswith("h264o_dec_") for n in used), sorted(used)
    assert not re.search(r"\bo_[a-z0-9_]+\s*\[", code), "a table of h264_tables.h is referenced"


def _dec_code(fn, *args):
    n, v = C.c_int(0), C.c_uint(0)
    rc = fn(*args, C.byref(n), C.byref(v))
    return None if rc else (n.value, v.value)


def test_decoder_vlc_tables_agree_with_the_encoder_tables():
    L = ol.lib()
    ln, bits = _c_table("o_coeff_token_len"), _c_table("o_coeff_token_bits")
    for col in range(4):
        for tc in range(17):
            for t1 in range(4):
                got = _dec_code(L.h264o_dec_table_coeff_token, col, tc, t1)
                i = col * 68 + 4 * tc + t1
                if t1 > tc or (tc == 0 and t1 > 0):
                    assert got is None
                    continue
                assert got == (ln[i], bits[i]), (col, tc, t1)
    cl, cb = _c_table("o_chroma_dc_token_len"), _c_table("o_chroma_dc_token_bits")
    for tc in range(5):
        for t1 in range(min(tc, 3) + 1):
            assert _dec_code(L.h264o_dec_table_coeff_token, 4, tc, t1) == (cl[4 * tc + t1], cb[4 * tc + t1])
    # total_zeros: the header's initialiser rows are ragged, so walk the text row by row
    txt = open(os.path.join(ODIR, "h264_tables.h")).read()

    def rows(name):
        m = re.search(r"\b%s\s*(?:\[[^\]]*\])+\s*=\s*\{(.*?)\};" % name, txt, re.S)
        return [[int(t) for t in re.findall(r"\d+", r)] for r in re.findall(r"\{([^{}]*)\}", m.group(1))]
    tzl, tzb = rows("o_total_zeros_len"), rows("o_total_zeros_bits")
    assert len(tzl) == len(tzb) == 15
    for k in range(15):
        assert len(tzl[k]) == len(tzb[k]) == 16 - k
        for z in range(16):
            got = _dec_code(L.h264o_dec_table_total_zeros, 0, k + 1, z)
            assert got == ((tzl[k][z], tzb[k][z]) if z < 16 - k else None), (k, z)
    czl, czb = rows("o_cdc_total_zeros_len"), rows("o_cdc_total_zeros_bits")
    for k in range(3):
        for z in range(4 - k):
            assert _dec_code(L.h264o_dec_table_total_zeros, 1, k + 1, z) == (czl[k][z], czb[k][z])
    rl, rb = rows("o_run_len"), rows("o_run_bits")
    for zl in range(1, 8):
        n = 15 if zl == 7 else zl + 1
        assert len(rl[zl - 1]) == n
        for r in range(n):
            assert _dec_code(L.h264o_dec_table_run_before, zl, r) == (rl[zl - 1][r], rb[zl - 1][r])


def test_decoder_constant_tables_agree_with_the_encoder_tables():
    L = ol.lib()
    m = L.h264o_dec_table_misc
    assert [m(0, i, 0) for i in range(48)] == _c_table("o_cbp_code2intra")
    assert [m(0, i, 1) for i in range(48)] == _c_table("o_cbp_code2inter")
    assert [m(1, i, 0) for i in range(16)] == _c_table("o_zigzag4x4")
    assert [m(2, i, 0) for i in range(52)] == _c_table("o_alpha")
    assert [m(3, i, 0) for i in range(52)] == _c_table("o_beta")
    assert [m(4, i, j) for i in range(52) for j in range(3)] == _c_table("o_tc0")
    assert [m(5, i, 0) for i in range(52)] == _c_table("o_chroma_qp")
    assert [m(6, i, j) for i in range(6) for j in range(3)] == _c_table("o_dequant_v")
    # 8x8 zig-zag is a permutation that walks anti-diagonals
    zz8 = [m(7, i, 0) for i in range(64)]
    assert sorted(zz8) == list(range(64))
    assert [(p % 8) + (p // 8) for p in zz8] == sorted((p % 8) + (p // 8) for p in zz8)


# ---- mutation tests ---------------------------------------------------------------------------------------------------
MUTANTS = [
    # (name, file, old text, new text): each plants one error in a normative stage of the ENCODER side
    ("luma_6tap_centre_weight", "h264_common.c", "return a - 5 * b + 20 * c + 20 * d - 5 * e + f;", "return a - 5 * b + 20 * c + 19 * d - 5 * e + f;"),
    ("luma_halfpel_rounding", "h264_common.c", "#define Bh(i, j) clip1((b1[(j) + 2][i] + 16) >> 5)", "#define Bh(i, j) clip1((b1[(j) + 2][i] + 15) >> 5)"),
    ("chroma_bilinear_rounding", "h264_common.c", "dx * dy * D + 32) >> 6);", "dx * dy * D + 31) >> 6);"),
    ("idct_rounding", "h264_common.c", "r[j] = (g0 + g3 + 32) >> 6;", "r[j] = (g0 + g3 + 31) >> 6;"),
    ("idct_half_term", "h264_common.c", "int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);", "int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 2);"),
    ("dequant_table_entry", "h264_tables.h", "{13, 20, 16}, {14, 23, 18}", "{13, 20, 16}, {14, 23, 19}"),
    ("tc0_table_entry", "h264_tables.h", "{1, 1, 2},   {1, 1, 2},   {1, 1, 2},\n    {1, 1, 2},   {1, 2, 3}", "{1, 1, 2},   {1, 1, 2},   {1, 1, 2},\n    {1, 1, 1},   {1, 2, 3}"),
    ("alpha_table_entry", "h264_tables.h", "15, 17, 20, 22, 25, 28, 32, 36, 40", "15, 17, 20, 22, 25, 28, 32, 30, 40"),
    ("deblock_bs_mv_threshold", "h264_common.c", "if (abs(pv[2 * (bp >> 2)] - qv[2 * (bq >> 2)]) >= 4 ||", "if (abs(pv[2 * (bp >> 2)] - qv[2 * (bq >> 2)]) >= 5 ||"),
    ("deblock_strong_filter_tap", "h264_common.c", "pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 2) >> 2);", "pix[-2 * xs] = (uint8_t)((p2 + p1 + p0 + q0 + 1) >> 2);"),
    ("deblock_normal_delta", "h264_common.c", "int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);\n        pix[-xs] = clip1(p0 + d);\n        pix[0] = clip1(q0 - d);\n        if (ap < beta)",
     "int d = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 3) >> 3);\n        pix[-xs] = clip1(p0 + d);\n        pix[0] = clip1(q0 - d);\n        if (ap < beta)"),
    ("intra16_plane_coefficient", "h264_common.c", "int b = (5 * H + 32) >> 6, c = (5 * V + 32) >> 6;", "int b = (5 * H + 32) >> 6, c = (5 * V + 31) >> 6;"),
    ("chroma_plane_coefficient", "h264_common.c", "int b = (34 * H + 32) >> 6, c = (34 * V + 32) >> 6;", "int b = (33 * H + 32) >> 6, c = (34 * V + 32) >> 6;"),
    ("chroma_dc_pred_rule", "h264_common.c", "} else if (bx == 1 && by == 0) { /* prefers top */\n                    if (top) dc = (st + 2) >> 2;\n                    else if (left) dc = (sl + 2) >> 2;",
     "} else if (bx == 1 && by == 0) { /* prefers top */\n                    if (left) dc = (sl + 2) >> 2;\n                    else if (top) dc = (st + 2) >> 2;"),
    ("chroma_qp_table_entry", "h264_tables.h", "18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33,", "18, 19, 20, 21, 22, 23, 24, 25, 26, 27, 28, 29, 29, 30, 31, 33, 32, 33,"),
]


def _build_mutant(tmp, fname, old, new):
    for f in os.listdir(ODIR):
        if f.endswith((".c", ".h")) and f != "openh264_differential.cpp":
            shutil.copy(os.path.join(ODIR, f), os.path.join(tmp, f))
    p = os.path.join(tmp, fname)
    txt = open(p).read()
    assert txt.count(old) == 1, "mutation anchor not found exactly once in %s: %r" % (fname, old[:50])
    open(p, "w").write(txt.replace(old, new))
    so = os.path.join(tmp, "libmut.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-std=gnu11", "-msse4.1", "-shared", "-o", so,
                           os.path.join(tmp, "h264_common.c"), os.path.join(tmp, "h264_enc.c"), os.path.join(tmp, "h264_dec.c")])
    return so


def _roundtrip_ok(so):
    """encode a few pictures with the (mutated) encoder, decode with the decoder of the same library: True if every
    decoded picture equals the encoder's reconstruction"""
    L = C.CDLL(so)
    vp = C.c_void_p
    L.h264o_enc_create.restype = vp
    L.h264o_enc_create.argtypes = [C.POINTER(ol.Config)]
    L.h264o_enc_encode.restype = C.c_int64
    L.h264o_enc_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_int)]
    L.h264o_enc_recon.restype = vp
    L.h264o_enc_recon.argtypes = [vp, C.c_int]
    L.h264o_enc_destroy.argtypes = [vp]
    L.h264o_dec_create.restype = vp
    L.h264o_dec_decode.argtypes = [vp, vp, C.c_size_t]
    L.h264o_dec_plane.restype = vp
    L.h264o_dec_plane.argtypes = [vp, C.c_int]
    L.h264o_dec_destroy.argtypes = [vp]
    ok = True
    # two kinds of content and three QPs: the qp 32..38 cases reach the larger table indices, s1 the sub-pel vectors
    for kind, w, h, qp, n in (("s1", 176, 144, 26, 4), ("s1", 176, 144, 33, 4), ("ramp", 128, 96, 38, 3), ("s1", 64, 64, 30, 2)):
        cfg = ol.Config(w, h, 30, qp, 30, 66, 0, 0, 0, 0, 0)
        e = L.h264o_enc_create(C.byref(cfg))
        d = L.h264o_dec_create()
        out = np.zeros(w * h * 8 + 65536, dtype=np.uint8)
        cw, ch = (w + 15) // 16 * 16, (h + 15) // 16 * 16
        for f in synth.sequence(kind, w, h, n):
            f = np.ascontiguousarray(f)
            idr = C.c_int(0)
            nb = L.h264o_enc_encode(e, ol._ptr(f[: w * h]), w, ol._ptr(f[w * h: w * h * 5 // 4]), w // 2, ol._ptr(f[w * h * 5 // 4:]), w // 2, 0,
                                    ol._ptr(out), out.size, C.byref(idr))
            assert nb > 0
            if L.h264o_dec_decode(d, ol._ptr(out), nb) != 1:
                ok = False
                break
            for p in range(3):
                pw, ph = (cw, ch) if p == 0 else (cw // 2, ch // 2)
                a = np.ctypeslib.as_array(C.cast(L.h264o_dec_plane(d, p), C.POINTER(C.c_uint8)), shape=(ph, pw))
                b = np.ctypeslib.as_array(C.cast(L.h264o_enc_recon(e, p), C.POINTER(C.c_uint8)), shape=(ph, pw))
                if not np.array_equal(a, b):
                    ok = False
        L.h264o_enc_destroy(e)
        L.h264o_dec_destroy(d)
        if not ok:
            break
    return ok


def test_unmutated_library_round_trips(tmp_path):
    so = _build_mutant(str(tmp_path), "h264_common.c", "static inline int tap6(", "static inline int tap6(")
    assert _roundtrip_ok(so)


@pytest.mark.parametrize("mut", MUTANTS, ids=[m[0] for m in MUTANTS])
def test_encoder_side_error_breaks_the_round_trip(mut, tmp_path):
    name, fname, old, new = mut
    so = _build_mutant(str(tmp_path), fname, old, new)
    assert not _roundtrip_ok(so), "mutant %s survived: the decoder does not pin this stage" % name
