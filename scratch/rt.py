import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from media_amd import synth
from oracle_lib import OracleEncoder, OracleDecoder
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kind = sys.argv[4] if len(sys.argv) > 4 else "s1"
qp = int(sys.argv[5]) if len(sys.argv) > 5 else 26
enc = OracleEncoder(w, h, qp=qp, gop=30); dec = OracleDecoder()
tot = 0
for i, f in enumerate(synth.sequence(kind, w, h, n)):
    t = time.time(); bs, idr = enc.encode(f); dt = time.time() - t
    tot += len(bs)
    rc = dec.decode(bs)
    ok = all(np.array_equal(enc.recon(p), dec.plane(p)) for p in range(3))
    y = f[:w*h].reshape(h, w)
    mb = enc.mbinfo()
    print(i, "IDR" if idr else "P", len(bs), "bytes rc", rc, "match", ok, "psnrY %.2f" % synth.psnr(y, enc.recon(0)[:h,:w]),
          "skip", int((mb["type"] == 2).sum()), "of", mb.size, "%.3fs" % dt)
    if not ok:
        for p in range(3):
            a, b = enc.recon(p), dec.plane(p)
            d = np.argwhere(a != b)
            if len(d): print(" plane", p, "first diff", d[0], "count", len(d))
        break
print("total bytes", tot)
