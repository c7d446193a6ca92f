"""Soak test, GPU box only: `python tools/soak_bands.py <seed> <cases>` - slice-band mode: W instances (random W up to the
number of slices) with the halo swap after every picture against ONE oracle encoder with the same slices; pictures up to
2000x1200.  Round 1: seed 5, 600 cases, 0 mismatches."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, random, torch
torch.cuda.init()
from media_amd import capi, synth
from oracle_lib import OracleEncoder
rng = random.Random(int(sys.argv[1])); ncase = int(sys.argv[2])
bad = 0; t0 = time.time()
for case in range(ncase):
    big = rng.random() < 0.3
    w, h = (2 * rng.randint(200, 1000), 2 * rng.randint(150, 600)) if big else (2 * rng.randint(16, 200), 2 * rng.randint(24, 150))
    qp = rng.randint(14, 44); gop = rng.choice([2, 4, 30])
    mbh = (h + 15) // 16
    sl = rng.randint(2, min(12, max(2, mbh // 2)))
    mot = (rng.randint(-18, 18), rng.randint(-18, 18)); noise = rng.choice([0, 2, 6])
    n = 3 if big else 6
    frames = [synth.frame_s1(w, h, i, noise=noise, motion=mot) for i in range(n)]
    one = OracleEncoder(w, h, qp=qp, gop=gop, slices=sl)
    nsl = -(-mbh // (-(-mbh // min(sl, max(1, mbh // 2)))))
    W = rng.randint(1, nsl)
    parts = [capi.Encoder(w, h, qp=qp, gop=gop, slices=sl, band_index=r, band_count=W if W > 1 else 0) for r in range(W)]
    buf = torch.empty(parts[0].band_info()[4], dtype=torch.uint8, device="cuda")
    tag = (case, w, h, qp, gop, sl, nsl, W, mot, noise)
    try:
        for i, f in enumerate(frames):
            want = one.encode(f)[0]
            got = b"".join(p.encode(f)[0] for p in parts)
            if got != want:
                bad += 1; print("MISMATCH", tag, "frame", i, flush=True); break
            for r in range(W):
                if r > 0: parts[r].halo_export(0, buf.data_ptr()); parts[r - 1].halo_import(1, buf.data_ptr())
                if r < W - 1: parts[r].halo_export(1, buf.data_ptr()); parts[r + 1].halo_import(0, buf.data_ptr())
    except Exception as ex:
        bad += 1; print("EXC", tag, ex, flush=True)
    for p in parts: p.close()
    if case % 50 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("stress3 done cases", ncase, "bad", bad)
