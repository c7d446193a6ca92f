#!/usr/bin/env python3
"""Generates tests/golden/oracle_vectors.json from the CPU oracle on seeded synthetic
inputs (media_amd/synth.py).  The reference holds no golden vectors of its own
 and its arithmetic (libopenh264.so) is absent here, so these
vectors pin GPU == oracle and guard the oracle against regressions; they are NOT outputs
of the reference.  Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from media_amd import synth  # noqa: E402
from oracle_lib import OracleEncoder  # noqa: E402

CASES = [
    # name, width, height, kind, qp, gop, frames, profile_idc[, slices]
    ("qcif_s1_qp26", 176, 144, "s1", 26, 30, 6, 66),
    ("qvga_s1_gop4", 320, 240, "s1", 26, 4, 6, 66),
    ("tiny_s2_static", 64, 48, "s2", 26, 30, 4, 66),
    ("odd_s3_random_qp30", 200, 120, "s3", 30, 30, 3, 66),
    ("crop_130x98_s1", 130, 98, "s1", 28, 30, 4, 66),
    ("min_16x16_s1", 16, 16, "s1", 26, 30, 3, 66),
    ("qcif_s1_qp12", 176, 144, "s1", 12, 30, 3, 66),
    ("qcif_s1_qp48", 176, 144, "s1", 48, 30, 3, 66),
    ("qcif_s1_main", 176, 144, "s1", 26, 30, 3, 77),
    ("qcif_s1_high", 176, 144, "s1", 26, 30, 3, 100),
    ("portrait_720x1280_s1", 720, 1280, "s1", 26, 30, 3, 66),
    ("qvga_s1_4slices", 320, 240, "s1", 26, 4, 6, 66, 4),
    ("cif_scroll_6slices_main", 352, 288, "scroll", 30, 30, 5, 77, 6),
    ("qcif_cut_qp28", 176, 144, "cut", 28, 30, 5, 66),             # intra macroblocks inside P pictures
    ("qcif_s3_qp10_pcm", 176, 144, "s3", 10, 30, 3, 66),           # I_PCM fallback (IDR and P)
    ("qvga_cut_3slices", 320, 240, "cut", 26, 30, 4, 66, 3),
    ("qcif_s1_3refs", 176, 144, "s1", 26, 30, 6, 66, 0, 3),          # three reference frames: ref_idx_l0, sliding window
    ("qvga_cut_2refs_high", 320, 240, "cut", 28, 30, 6, 100, 2, 2),
    ("qcif_split_partitions", 176, 144, "split", 26, 30, 4, 66),      # 16x8 / 8x16 / 8x8 partitions
    ("cif_split_3refs_high_2slices", 352, 288, "split", 30, 30, 4, 100, 2, 3),
    # the cases above run the EXHAUSTIVE integer search (search = 0: their vectors are those of rounds 1 and 2); the ones below the
    # seeded search, the default since round 3 (config.search = 1): name, ..., slices, refs, search
    ("seeded_qvga_s1", 320, 240, "s1", 26, 30, 6, 66, 0, 0, 1),
    ("seeded_cif_scroll_main", 352, 288, "scroll", 30, 30, 5, 77, 0, 0, 1),
    ("seeded_qcif_cut_qp28", 176, 144, "cut", 28, 30, 5, 66, 0, 0, 1),
    ("seeded_cif_split_3refs_high_2slices", 352, 288, "split", 30, 30, 4, 100, 2, 3, 1),
    ("seeded_qvga_ramp_qp32", 320, 240, "ramp", 32, 30, 5, 66, 0, 0, 1),
    ("seeded_edge_130x98_s1", 130, 98, "s1", 28, 30, 5, 66, 0, 0, 1),
]


def run_case(name, w, h, kind, qp, gop, n, prof, slices=0, refs=0, search=0):
    enc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, slices=slices, refs=refs, search=search)
    frames = []
    for f in synth.sequence(kind, w, h, n):
        bs, idr = enc.encode(f)
        rec = hashlib.sha256(b"".join(enc.recon(p).tobytes() for p in range(3))).hexdigest()
        frames.append({"idr": bool(idr), "bytes": len(bs), "sha256": hashlib.sha256(bs).hexdigest(), "recon_sha256": rec})
    enc.close()
    return {"name": name, "width": w, "height": h, "kind": kind, "qp": qp, "gop": gop, "profile_idc": prof, "slices": slices, "refs": refs, "search": search, "frames": frames}


# Streams of random syntax for the decoder peer (oracle/h264_enc.c h264o_enc_random_picture): what is pinned is the stream
# (sha256 of every access unit) and what the oracle's independent decoder makes of it (sha256 of its three planes).
RANDOM_CASES = [
    # name, width, height, profile_idc, slices, refs, features (oracle_lib.OracleEncoder.RAND_*), pictures
    ("rand_qcif_all_baseline", 176, 144, 66, 0, 3, 63, 6),
    ("rand_qvga_all_high_3slices", 320, 240, 100, 3, 2, 63, 6),
    ("rand_qcif_qp_offsets_main", 176, 144, 77, 2, 1, 1 | 2 | 4 | 16, 5),
    ("rand_tiny_pcm_subparts", 48, 32, 66, 0, 3, 8 | 32, 5),
    ("rand_qcif_everything", 176, 144, 100, 0, 3, 511, 6),                       # + slices of any shape, list modification, OpenH264-style headers
    ("rand_qcif_openh264_shape", 176, 144, 66, 0, 1, 256 | 1 | 2 | 4 | 32, 6),  # one reference, QP per macroblock, offsets, sub-partitions
    ("rand_qcif_all_features", 176, 144, 100, 0, 3, 2047, 6),                    # + levels beyond a byte, constrained_intra_pred_flag
]


def run_random_case(name, w, h, prof, slices, refs, features, n):
    from oracle_lib import OracleDecoder
    enc = OracleEncoder(w, h, qp=30, gop=4, profile_idc=prof, slices=slices, refs=refs)
    dec = OracleDecoder()
    frames = []
    for i in range(n):
        au, idr, _ = enc.random_picture(20261004 + 31 * i, features=features)
        assert dec.decode(au) == 1
        planes = hashlib.sha256(b"".join(dec.plane(p).tobytes() for p in range(3))).hexdigest()
        frames.append({"idr": bool(idr), "bytes": len(au), "sha256": hashlib.sha256(au).hexdigest(), "decoded_sha256": planes})
    enc.close()
    return {"name": name, "width": w, "height": h, "profile_idc": prof, "slices": slices, "refs": refs, "features": features, "frames": frames}


if __name__ == "__main__":
    rnd = {"generator": "tests/golden/make_golden.py", "source": "CPU oracle (oracle/): random-syntax streams and its independent decoder's output, not the reference",
           "cases": [run_random_case(*c) for c in RANDOM_CASES]}
    with open(os.path.join(HERE, "random_streams.json"), "w") as f:
        json.dump(rnd, f, indent=1)
    print("wrote", len(rnd["cases"]), "random-syntax cases")
    out = {"generator": "tests/golden/make_golden.py", "source": "CPU oracle (oracle/), not the reference",
           "cases": [run_case(*c) for c in CASES]}
    with open(os.path.join(HERE, "oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(out["cases"]), "cases")
