import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from media_amd import synth, capi
for (w, h) in [(1920, 16), (1920, 32), (1920, 64), (1920, 256), (1920, 1080), (3840, 16), (640, 1080)]:
    enc = capi.Encoder(w, h, qp=26, gop=1000)
    fr = synth.sequence("s1", w, h, 6)
    enc.encode(fr[0]); enc.encode(fr[1])
    enc.stats_enable(True); enc.stats(reset=True)
    for f in fr[2:]:
        enc.encode(f)
    st = enc.stats()["kernels"]
    n = 4
    print(w, h, {k: round(v["ms"] / n * 1e3, 1) for k, v in st.items() if v["launches"]}, flush=True)
    enc.close()
