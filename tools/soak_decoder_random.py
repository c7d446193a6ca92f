"""Soak test, GPU box only: `python tools/soak_decoder_random.py <seed> <cases>` - the decoder peer against the oracle's
independent decoder on streams of RANDOM syntax (oracle/h264_enc.c h264o_enc_random_picture: every macroblock type, mode and
partition shape, random vectors / reference indices / levels, QP per slice and per macroblock, chroma QP offsets, filter
offsets, I_PCM in filtered pictures, every deblocking idc, sub-macroblock partitions down to 4x4 with a reference index per
partition, slices cut at random macroblocks, reference list modification): random geometry, profile, slices, references, GOP and feature
set per case, 8 pictures each, every plane of every picture compared."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, random, os, pickle
from media_amd import h264dec
from oracle_lib import OracleEncoder, OracleDecoder
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = random.Random(seed)
bad = 0; pictures = 0; t0 = time.time()
dec = h264dec.Decoder()
for case in range(ncase):
    w, h = 2 * rng.randint(8, 200), 2 * rng.randint(8, 150)
    prof = rng.choice([66, 77, 100]); sl = rng.choice([0, 0, 2, 3, 5]); refs = rng.choice([1, 1, 2, 3]); gop = rng.choice([1, 3, 8, 30])
    feat = rng.choice([2047, 2047, 1024, 1024 | 63, 1023, 1023, 1 | 512, 511, 511, 255, 255, 255, 127, 127, 63, 31 | 128, 31, 32, 32 | 64, 128, 128 | 32, 256, 256 | 1 | 2 | 32, 1, 2, 4, 8, 16, 0, 1 | 8, 2 | 4 | 16, 32 | 8])
    tag = (case, w, h, prof, sl, refs, gop, feat)
    try:
        enc = OracleEncoder(w, h, qp=rng.randint(10, 51), gop=gop, profile_idc=prof, slices=sl, refs=refs, disable_deblock=int(rng.random() < 0.15))
        ref = OracleDecoder()
        aus = []
        for i in range(8):
            au, idr, mbqp = enc.random_picture(rng.getrandbits(31), features=feat)
            aus.append(au)
            if ref.decode(au) != 1: raise RuntimeError("oracle decoder: no picture")
            if not dec.decode(au): raise RuntimeError("no picture")
            pictures += 1
            for p in range(3):
                if not np.array_equal(dec.plane(p), ref.plane(p)):
                    os.makedirs("gpurun_out/r02", exist_ok=True)
                    with open("gpurun_out/r02/decrand_diff_%d_%d.bin" % (seed, case), "wb") as fh:
                        pickle.dump({"aus": aus, "gpu": dec.plane(p), "orc": ref.plane(p), "plane": p, "tag": tag}, fh)
                    raise RuntimeError("plane %d of picture %d differs" % (p, i))
    except Exception as ex:
        bad += 1; print("BAD", tag, ex, flush=True)
        dec.close(); dec = h264dec.Decoder()
    if case % 100 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("soak_decoder_random done seed", seed, "cases", ncase, "pictures", pictures, "bad", bad)
