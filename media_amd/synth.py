"""Synthetic I420 inputs of SURVEY.md section 8(d): deterministic, integer-only
past the sine table, reproducible frame by frame (counter-based PRNG).

S1 "pan+noise": smooth texture translated by (+3,+1) luma px/frame + U[-4,4] noise
S2 "static":    frame 0 of S1 repeated (P_Skip path)
S3 "random":    i.i.d. U[0,255] (worst case, kernel microbenchmarks only)
"""
import numpy as np

SEED_S1 = 20261004


def _hash_u32(seed, n):
    """counter-based PRNG: splitmix64 of (seed, index) -> uint32, vectorised"""
    x = np.arange(n, dtype=np.uint64) + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(32)).astype(np.uint32)


def _sine_table(period, amp):
    i = np.arange(period, dtype=np.float64)
    return np.floor(amp * np.sin(2.0 * np.pi * i / period) + 0.5 + 1e-9).astype(np.int32)


def _texture(w, h, ox, oy, scale):
    """sum of four integer-period sinusoids, about +-60 around 128; (ox,oy) pans it"""
    x = (np.arange(w, dtype=np.int64) + ox)
    y = (np.arange(h, dtype=np.int64) + oy)
    periods = [(64 // scale, 24), (96 // scale, 16), (40 // scale, 12), (28 // scale, 8)]
    t = np.full((h, w), 128, dtype=np.int32)
    (p0, a0), (p1, a1), (p2, a2), (p3, a3) = periods
    t += _sine_table(p0, a0)[x % p0][None, :]
    t += _sine_table(p1, a1)[y % p1][:, None]
    t += _sine_table(p2, a2)[(x[None, :] + y[:, None]) % p2]
    t += _sine_table(p3, a3)[(x[None, :] - y[:, None]) % p3]
    return t


def frame_s1(width, height, index, noise=4, motion=(3, 1)):
    """returns contiguous I420 uint8 array of width*height*3/2 bytes"""
    ox, oy = -motion[0] * index, -motion[1] * index  # content moves by +motion
    y = _texture(width, height, ox, oy, 1)
    cw, chh = width // 2, height // 2
    # chroma: half resolution, half motion (integer division keeps it deterministic)
    u = _texture(cw, chh, ox // 2, oy // 2, 2) // 2 + 64
    v = 255 - (_texture(cw, chh, ox // 2 + 7, oy // 2 + 3, 2) // 2 + 64)
    planes = []
    base = 0
    for p in (y, u, v):
        n = p.size
        if noise:
            r = _hash_u32(SEED_S1 + index, base + n)[base:]
            p = p + (r % np.uint32(2 * noise + 1)).astype(np.int32).reshape(p.shape) - noise
        base += n
        planes.append(np.clip(p, 0, 255).astype(np.uint8).ravel())
    return np.concatenate(planes)


def frame_s2(width, height, index):
    return frame_s1(width, height, 0)


def frame_scroll(width, height, index):
    """noise-free texture scrolling by (+4, +2) luma samples per picture: the moving-but-predictable case (a list
    being scrolled on a cloud-phone screen).  Not one of the SURVEY inputs."""
    return frame_s1(width, height, index, noise=0, motion=(4, 2))


def frame_s3(width, height, index):
    n = width * height * 3 // 2
    return (_hash_u32(1 + index, n) & np.uint32(255)).astype(np.uint8)


def frame_ramp(width, height, index):
    """diagonal ramp whose slope grows with the picture index, flat chroma.  Not one of the SURVEY inputs:
    it exists because its slices contain 00 00 0x byte patterns, i.e. it exercises emulation prevention."""
    y = (np.add.outer(np.arange(height), np.arange(width)) * (index + 1) // 4 % 256).astype(np.uint8).ravel()
    return np.concatenate([y, np.full(width * height // 2, 128, np.uint8)])


def frame_cut(width, height, index):
    """S1 for two pictures, then a cut to different content (the same texture mirrored, inverted and offset) that keeps
    panning: the P picture after the cut finds no match, so macroblocks go intra.  Not one of the SURVEY inputs."""
    if index < 2:
        return frame_s1(width, height, index)
    f = frame_s1(width, height, index + 40)
    ysz, csz = width * height, width * height // 4
    y = 255 - f[:ysz].reshape(height, width)[::-1, ::-1]
    u = f[ysz:ysz + csz].reshape(height // 2, width // 2)[::-1, ::-1]
    v = 255 - f[ysz + csz:].reshape(height // 2, width // 2)[::-1, ::-1]
    return np.concatenate([np.ascontiguousarray(y).ravel(), np.ascontiguousarray(u).ravel(), np.ascontiguousarray(v).ravel()])


def frame_split(width, height, index):
    """two layers of a texture that drift apart by a fraction of a sample per picture - (+3, +1) and (-3, -1) QUARTER samples -
    interleaved in 8-sample stripes: rows of 8 in the upper third of the picture (a macroblock's halves move apart: 16x8
    partitions), columns of 8 in the middle third (8x16), an 8x8 checkerboard in the lower third (8x8).  The texture is
    evaluated on a 4x finer grid, so the motion is a true sub-sample shift.  Light noise.  Not one of the SURVEY inputs: it
    exists for the sub-16x16 partitions (test-size pictures only: the fine grid is 16x the picture)."""
    out = []
    base = 0
    for pi, (w, h, cell, div) in enumerate(((width, height, 8, 1), (width // 2, height // 2, 4, 2), (width // 2, height // 2, 4, 2))):
        la = _texture(4 * w, 4 * h, (-3 * index) // div + 11 * pi, (-1 * index) // div, 1)[::4, ::4]
        lb = _texture(4 * w, 4 * h, (3 * index) // div + 170 + 11 * pi, (1 * index) // div + 90, 1)[::4, ::4]
        yy, xx = np.arange(h)[:, None] // cell, np.arange(w)[None, :] // cell
        third = (np.arange(h)[:, None] * 3) // h
        mask = np.where(third == 0, yy & 1, np.where(third == 1, xx & 1, (xx + yy) & 1)).astype(bool)
        p = np.where(mask, lb, la)
        if pi:
            p = p // 2 + 64
        n = w * h
        r = _hash_u32(SEED_S1 + 7919 + index, base + n)[base:]
        p = p + (r % np.uint32(3)).astype(np.int32).reshape(p.shape) - 1
        base += n
        out.append(np.clip(p, 0, 255).astype(np.uint8).ravel())
    return np.concatenate(out)


def sequence(kind, width, height, count, start=0):
    fn = {"s1": frame_s1, "s2": frame_s2, "s3": frame_s3, "ramp": frame_ramp, "scroll": frame_scroll, "cut": frame_cut, "split": frame_split}[kind]
    return [fn(width, height, start + i) for i in range(count)]


def psnr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    mse = np.mean((a - b) ** 2)
    return 99.0 if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
