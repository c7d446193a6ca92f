#!/usr/bin/env python3
"""bench.py -- encoded frames/s of the MI355X H.264 path on BASELINE.json configs[1]:
1080p30 I420 synthetic (S1 pan+noise), baseline profile, fixed QP 26, one stream per GPU.

A "step" is one pass of the hot path over one batch: G closed 30-picture GOPs (1 IDR +
29 P each, the reference's default uiIntraPeriod, VideoEncoderOpenH264.h:18) of the
stream with all pictures already resident in HBM.  The G GOPs are split over I encoder
instances (--instances, default 2); an instance encodes its G/I GOPs in lockstep (every
kernel launch covers the same picture index of all of them, grid.y = G/I) and the
instances run free beside each other on their own HIP streams, so the dependency-bound
deblocking wavefront of one overlaps the throughput-bound motion search of the other.
Closed GOPs are independent under fixed QP and concatenate to the serial stream.
N GPUs = N independent streams, one process per GPU, no data-path collective (weak
scaling); torch.distributed is used only for the barrier and the max-over-ranks clock.
G = 1 is measured too (single_gop_in_flight_fps), and the kernels of one instance running
alone (roofline_exclusive / kernels_exclusive).

Prints ONE JSON line (rank 0) carrying `roofline` for the transform kernel (k_tq) and
`cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N=1 only), plus `plugin`:
the same engine measured through the actual drop-in boundary in the reference's operating mode
(CreateVideoEncoder -> EncodeOneFrame on HOST I420 pictures, bitrate rate control, one picture at a
time per stream, S streams side by side; PCIe included) - never the headline, always beside it.

`python bench.py --gpus N` without a launcher (no RANK in the environment) starts N child
processes itself, one per GPU, BEFORE anything touches a GPU, and relays rank 0's line.
`python bench.py --mode plugin --streams 1,4,16,64` prints only the plugin measurement.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP maps streams onto this many hardware queues; the default (4) would serialise the
# concurrently encoded GOPs.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

WIDTH, HEIGHT, QP, GOP = 1920, 1080, 26, 30
FRAMES_PER_STEP = GOP
# algorithmic HBM bytes per macroblock, each datum counted once:
# k_tq, a macroblock it CODES: source 384 + prediction 384 in; reconstruction 384 + levels 768 + side info 32 out.
# A macroblock k_me's "nothing left to code" tests settled (or handed to the intra pass) costs k_tq the 2-byte
# {type, mode} probe of its MbInfo and nothing else (k_tq.h mb_to_code): it is priced at 2 bytes, not 1952.
PMB_BYTES_PER_MB = 384 + 384 + 384 + 768 + 32
PMB_PROBE_BYTES = 2
# k_me, every macroblock (settled or searched): source 384 + reference 384 in (no credit for the overlapping
# search windows); prediction 384 + side info 32 out
ME_BYTES_PER_MB = 384 + 384 + 384 + 32
# whole pipeline, P picture: 2 720 B per macroblock
PIPE_BYTES_PER_MB = 2720
HBM_PEAK_GBS = 8000.0


def cpu_limits():
    """what bounds this process's CPU time on the box: hardware threads, affinity mask, and the cgroup's quota (cpu.max:
    "<quota> <period>" in microseconds, or "max"); the quota is what makes 256 threads slower than 16 on a box whose
    share is 16 cores"""
    info = {"hardware_threads": os.cpu_count() or 0}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except AttributeError:
        info["affinity"] = info["hardware_threads"]
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                info["cgroup_cpu_max"] = " ".join(txt)
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                info["cgroup_cfs_quota_us"] = q
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            break
        except Exception:
            continue
    info["cgroup_quota_cores"] = None if quota is None else round(quota, 2)
    return info


def cpu_baseline(frames, thread_counts, gop_frames, search=1):
    """time the CPU oracle (kind 'port') on host cores.  For every thread count T of the sweep: T independent encoder
    instances (one Python thread each; the C code runs outside the GIL) are created and warmed with one picture OUTSIDE the
    clock, then all start together on a barrier and each encodes ONE WHOLE closed GOP of the workload (1 IDR + gop_frames - 1
    P pictures: the real mix); fps = T * gop_frames / (last thread's finish - the barrier).  Returns (rows per T, single-thread
    fps)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import OracleEncoder
    rows = []
    for T in thread_counts:
        # the sweep ends once more threads have clearly stopped paying (below 85 % of the best row so far): past the box's CPU
        # share every further doubling only doubles the time of the run
        if len(rows) >= 2 and rows[-1]["fps"] < 0.85 * max(r["fps"] for r in rows):
            break
        encs = [OracleEncoder(WIDTH, HEIGHT, qp=QP, gop=GOP, search=search) for _ in range(T)]
        for e in encs:            # warm-up: allocations, page faults, the first picture's tables
            e.encode(frames[0])
        start = threading.Barrier(T + 1)
        done = [0.0] * T

        def work(k):
            e = encs[k]
            start.wait()
            for i in range(gop_frames):
                e.encode(frames[i % len(frames)], force_idr=(i == 0))
            done[k] = time.perf_counter()

        ths = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        for t in ths:
            t.start()
        start.wait()
        t0 = time.perf_counter()
        for t in ths:
            t.join()
        dt = max(done) - t0
        rows.append({"threads": T, "fps": round(T * gop_frames / dt, 2), "seconds": round(dt, 2)})
        for e in encs:
            e.close()
    single = next((r["fps"] for r in rows if r["threads"] == 1), None)
    return rows, single


def plugin_bench(streams, frames_per_stream, bitrate, device=0):
    """The drop-in boundary in the reference's operating mode: S threads, each
    CreateVideoEncoder (format 3 = MI355X) -> InitEncoder -> StartEncoder -> EncodeOneFrame x F on HOST I420 pictures
    (tight layout, as the reference's InitSrcPic expects), bitrate rate control (no fixed-QP key), scene detection on,
    one picture at a time per stream; H2D and D2H over PCIe are inside every call.  Returns one record per S."""
    import numpy as np
    from media_amd import synth
    from media_amd import videocodec as vc
    # one pool of consecutive pictures of the S1 sequence, never wrapped: stream k codes pictures k + 1, k + 2, ... (picture k is
    # its warm-up), so every stream has its own phase of the content and nobody meets a cut (the 30-picture source
    # used to wrap, the scene detector fired at every wrap and those pictures were coded twice inside the clock)
    nsrc = frames_per_stream + max(streams) + 1
    frames = [np.ascontiguousarray(f) for f in synth.sequence("s1", WIDTH, HEIGHT, nsrc)]
    out = []
    for S in streams:
        vc.set_video_mode(WIDTH, HEIGHT, fps=30, bitrate=bitrate, gop=GOP, profile="baseline", fmt=3, qp=None)
        vc.prop_set("persist.vmi.video.encode.device", device)
        encs = []
        for _ in range(S):
            e = vc.VideoEncoder()
            if e.rc_create != vc.SUCCESS or e.init() != vc.SUCCESS or e.start() != vc.SUCCESS:
                raise SystemExit("plugin bench: encoder %d of %d could not be opened" % (len(encs), S))
            encs.append(e)
        lat = [[] for _ in range(S)]
        nbytes = [0] * S
        first2s = [0] * S            # bytes of the first 60 pictures (2 s at 30 Hz), warm-up IDR included: how fast the controller settles
        fail = [0] * S
        cuts0 = sum(e.scene_cuts() for e in encs)
        psnr = []

        def work(k):
            e = encs[k]
            for i in range(frames_per_stream):
                f = frames[k + 1 + i]
                t0 = time.perf_counter()
                rc, bs = e.encode(f)
                lat[k].append(time.perf_counter() - t0)
                if rc != vc.SUCCESS:
                    fail[k] += 1
                nbytes[k] += len(bs)
                if i < 59:
                    first2s[k] += len(bs)

        for k in range(S):   # warm-up outside the clock: first IDR, allocations
            first2s[k] += len(encs[k].encode(frames[k])[1])
        ths = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        cuts = sum(e.scene_cuts() for e in encs) - cuts0
        # quality, outside the clock: the last picture of stream 0 against its reconstruction
        f = frames[frames_per_stream]
        psnr.append(synth.psnr(f[:WIDTH * HEIGHT].reshape(HEIGHT, WIDTH), encs[0].recon_y()[:HEIGHT, :WIDTH]))
        for e in encs:
            e.stop(); e.destroy(); e.delete()
        allat = np.sort(np.concatenate([np.asarray(x) for x in lat])) * 1e3
        n = S * frames_per_stream
        out.append({"streams": S, "fps_aggregate": round(n / dt, 1), "fps_per_stream": round(n / dt / S, 1),
                    "latency_ms_p50": round(float(allat[len(allat) // 2]), 3), "latency_ms_p99": round(float(allat[min(len(allat) - 1, int(len(allat) * 0.99))]), 3),
                    "bytes_per_picture": round(sum(nbytes) / n, 1), "bitrate_target": bitrate, "bitrate_achieved": round(sum(nbytes) * 8 * 30 / n),
                    "bitrate_first_2s": round(sum(first2s) * 8 * 30 / (S * min(60, frames_per_stream + 1))), "scene_cut_recodes": cuts,
                    "h2d_GBps": round(n / dt * WIDTH * HEIGHT * 1.5 / 1e9, 2),
                    "psnr_y_db": round(float(np.mean(psnr)), 2) if psnr else None, "pictures": n, "encode_failures": sum(fail)})
    res = {"what": "VideoCodecApi plugin surface (CreateVideoEncoder / EncodeOneFrame), host I420 pictures over PCIe, bitrate mode, scene detection on, "
                   "1080p30 S1, GOP 30, baseline; S encoder objects on S host threads of one process (Python threads over ctypes), %d pictures per stream "
                   "from a non-wrapping source; the objects are streams of one shared engine (pictures of different streams coded in one lockstep step) "
                   "unless MI355X_H264_HUB=0" % frames_per_stream,
           "results": out}
    try:
        native = plugin_bench_native(streams, frames_per_stream, bitrate, device)
        if native is not None:
            res["native"] = native
    except Exception as exc:   # the side measurement must not take the line down
        res["native"] = {"error": str(exc)}
    return res


def plugin_bench_native(streams, frames_per_stream, bitrate, device=0):
    """The same measurement with NO interpreter in the timed region: tools/plugin_bench.cpp (media_amd/lib/plugin_bench), S
    std::threads on the C++ plugin surface.  None when the binary has not been built."""
    import subprocess
    import tempfile
    import numpy as np
    from media_amd import synth
    exe = os.path.join(ROOT, "media_amd", "lib", "plugin_bench")
    if not os.path.exists(exe):
        return None
    nsrc = frames_per_stream + max(streams) + 1
    env = dict(os.environ)
    env.update({"RO_VMI_DEMO_VIDEO_ENCODE_FORMAT": "3", "RO_SYS_VMI_CLOUDPHONE": "video", "RO_HARDWARE_WIDTH": str(WIDTH),
                "RO_HARDWARE_HEIGHT": str(HEIGHT), "RO_HARDWARE_FPS": "30", "PERSIST_VMI_VIDEO_ENCODE_BITRATE": str(bitrate),
                "PERSIST_VMI_VIDEO_ENCODE_GOPSIZE": str(GOP), "PERSIST_VMI_VIDEO_ENCODE_PROFILE": "baseline",
                "PERSIST_VMI_VIDEO_ENCODE_PARAM_ADJUSTING": "0", "PERSIST_VMI_VIDEO_ENCODE_KEYFRAME": "0",
                "PERSIST_VMI_VIDEO_ENCODE_SCENEDETECT": "1", "PERSIST_VMI_VIDEO_ENCODE_DEVICE": str(device), "MEDIA_LOG_QUIET": "1"})
    env.pop("PERSIST_VMI_VIDEO_ENCODE_QP", None)
    with tempfile.NamedTemporaryFile(suffix=".i420") as f:
        for fr in synth.sequence("s1", WIDTH, HEIGHT, nsrc):
            f.write(np.ascontiguousarray(fr).tobytes())
        f.flush()
        r = subprocess.run([exe, f.name, str(WIDTH), str(HEIGHT), str(nsrc), str(frames_per_stream), ",".join(str(s) for s in streams)],
                           env=env, capture_output=True, text=True, timeout=600)
    rows = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not rows:
        return {"error": "plugin_bench exited %d: %s" % (r.returncode, r.stderr[-300:])}
    return {"what": "the same through tools/plugin_bench.cpp: S std::threads on the C++ VideoEncoder surface (no interpreter in the timed region), "
                    "host I420 pictures over PCIe, bitrate mode %d bit/s, scene detection on" % bitrate, "results": rows}


def openh264_differential(frames, count):
    """encode the workload's first pictures with a real libopenh264.so if the box has one
    (oracle/_ref/openh264_differential, the reference's dlopen + preset).  {"oracle": "absent"} otherwise."""
    import subprocess
    import tempfile
    tool = os.path.join(ROOT, "oracle", "_ref", "openh264_differential")
    if not os.path.exists(tool):
        return {"oracle": "absent", "reason": "oracle/_ref/openh264_differential not built"}
    try:
        probe = json.loads(subprocess.run([tool], capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[-1])
        if probe.get("oracle") == "absent":
            return probe
    except Exception:
        pass   # a library was found (the tool then wants arguments): go on
    try:
        with tempfile.NamedTemporaryFile(suffix=".i420") as f:
            n = min(count, len(frames))
            for i in range(n):
                f.write(frames[i].tobytes())
            f.flush()
            r = subprocess.run([tool, f.name, str(WIDTH), str(HEIGHT), "30", "5000000", str(GOP), str(n)],
                               capture_output=True, text=True, timeout=300)
            return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as exc:   # never let the optional differential break the bench line
        return {"oracle": "absent", "reason": "differential tool failed: %s" % exc}


def decode_bench(frames_n, device=0):
    """The decoder peer (row f4): a 1080p stream written by the HIP encoder (S1, QP 26, GOP 30) decoded access unit by access
    unit through include/mi355x_h264_dec.h - host CAVLC parse + upload of the parsed arrays + GPU reconstruction + loop
    filter, one access unit per call with the decoder's one picture of look-ahead (the call returns when the picture is
    launched; the next access unit is parsed while the GPU reconstructs it), with and without the D2H copy of every picture
    (which waits for the picture: no overlap then, the way VideoDecoderMI355X's Send / Retrieve pairs use it).  The last decoded
    picture is compared with the encoder's reconstruction."""
    import numpy as np
    from media_amd import capi, synth, h264dec
    enc = capi.Encoder(WIDTH, HEIGHT, qp=QP, gop=GOP, device=device)
    aus, recs = [], []
    for f in synth.sequence("s1", WIDTH, HEIGHT, frames_n):
        aus.append(enc.encode(np.ascontiguousarray(f))[0])
        recs.append(enc.debug_read(capi.DBG_RECON_Y))
    enc.close()
    out = {"what": "VideoDecoder peer: 1080p S1 stream of the HIP encoder (QP %d, GOP %d, %d pictures, %.0f kB / picture), one access unit per call, "
                   "decode_only: the parse of unit n + 1 overlaps the reconstruction of picture n; decode_and_read_i420: every picture waited for and copied out; "
                   "host parse + H2D of the parsed arrays + GPU reconstruction + loop filter" % (QP, GOP, frames_n, sum(map(len, aus)) / frames_n / 1e3)}
    for label, read in (("decode_only", False), ("decode_and_read_i420", True)):
        dec = h264dec.Decoder(device)
        dec.decode(aus[0])            # engine creation + first launches are not timed
        dec.close()
        dec = h264dec.Decoder(device)
        t0 = time.perf_counter()
        for au in aus:
            dec.decode(au)
            if read:
                dec.i420()
        dec.sync()                    # the last picture is complete inside the timed region
        dt = time.perf_counter() - t0
        n, parse_ms, gpu_ms = dec.timing()
        ok = bool(np.array_equal(dec.plane(0), recs[-1]))
        dec.close()
        out[label] = {"fps": round(frames_n / dt, 1), "ms_per_picture": round(dt / frames_n * 1e3, 3), "host_parse_ms_per_picture": round(parse_ms / n, 3),
                      "launch_and_wait_ms_per_picture": round(gpu_ms / n, 3), "last_picture_equals_encoder_reconstruction": ok}
    # several streams: S decoder objects on S host threads (each parses on its own core; the reconstructions share the GPU)
    out["streams"] = []
    for S in (4, 16):
        decs = [h264dec.Decoder(device) for _ in range(S)]
        for d in decs:
            d.decode(aus[0])
        good = [True] * S

        def work(k):
            for au in aus[1:]:
                decs[k].decode(au)
                decs[k].i420()
            good[k] = bool(np.array_equal(decs[k].plane(0), recs[-1]))

        ths = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        dt = time.perf_counter() - t0
        for d in decs:
            d.close()
        out["streams"].append({"streams": S, "fps_aggregate": round(S * (frames_n - 1) / dt, 1), "all_equal_encoder_reconstruction": all(good)})
    return out


def latest_profile(suffix):
    """the newest committed profiles/rNN?_<suffix> (tools/prof.sh + tools/summarize_prof.py), or None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]?_" + suffix)))
    return files[-1] if files else None


def valu_issue_bound(nmb):
    """What actually bounds the pipeline: VALU instruction issue.  From the committed PMC pass
    (profiles/rNN?_summary.json, SQ_INSTS_VALU of the lockstep launches of this same default workload): instructions per
    macroblock over all kernels, averaged over a GOP of 1 IDR + 29 P pictures -> pictures per second 1 024 SIMDs can issue.
    Priced two ways: every instruction at 4 cycles (`bound_fps_4cycles`, the round-2 figure), and by the MEASURED issue classes
    (tools/ubench_issue.hip: ~2.3 cycles for plain 32-bit-encoded VOP1/VOP2 on VGPR / constant operands, ~4.15 for everything
    else) with each kernel's static class mix from profiles/rNN_valu_classes.json (tools/valu_classes.py) applied to its
    dynamic count - `bound_fps`.  Informational; None when no summary is committed."""
    f = latest_profile("summary.json")
    if f is None:
        return None
    try:
        sq = json.load(open(f)).get("sq", {})
        cf = latest_profile("valu_classes.json")
        classes = json.load(open(cf))["kernels"] if cf else {}
        per, cyc = {}, {}
        for name, v in sq.items():
            k = name.split(" grid=")[0]
            if not k.startswith("k_") or not v.get("SQ_WAVES"):
                continue
            if k not in per or v["SQ_INSTS_VALU"] > per[k]:              # the lockstep (largest) launch of each kernel
                per[k] = v["SQ_INSTS_VALU"]
                cyc[k] = per[k] * classes.get(k, {}).get("cycles_per_valu", 4.15)
        if not per.get("k_me<false>") and not per.get("k_me"):
            return None

        def total(d):
            g = lambda *names: sum(d.get(n, 0) for n in names)
            entropy = g("k_bs<false>", "k_cavlc<false, false>", "k_cavlc<true, false>", "k_bit_scan<false>", "k_pack<false>", "k_skip_scan<false>", "k_mvpred<false>",
                        "k_bs", "k_cavlc<false>", "k_cavlc<true>", "k_bit_scan", "k_pack", "k_skip_scan", "k_mvpred")
            db_p = d.get("k_deblock_pairs<false, false>") or d.get("k_deblock_pairs<false>") or d.get("k_deblock_rows<false, false, false>", 0)
            db_i = d.get("k_deblock_pairs<true, false>") or d.get("k_deblock_pairs<true>") or d.get("k_deblock_rows<true, false, false>", 0)
            p_pic = entropy + db_p + g("k_me<false>", "k_me", "k_tq<false>", "k_tq", "k_pintra_rows<false, false>", "k_pintra_rows<false>")
            idr = entropy + db_i + g("k_i4_decide<false>", "k_i4_decide", "k_intra_rows<false>", "k_intra_rows")
            return p_pic, idr

        mbs = 32 * nmb                                                   # the profiled launches cover 32 pictures
        p_pic, idr = total(per)
        p_cyc, i_cyc = total(cyc)
        tot = ((GOP - 1) * p_pic + idr) / GOP / mbs
        tot_cyc = ((GOP - 1) * p_cyc + i_cyc) / GOP / mbs
        return {"valu_per_macroblock": round(tot, 1), "p_picture": round(p_pic / mbs, 1), "idr_picture": round(idr / mbs, 1),
                "issue_cycles_per_macroblock": round(tot_cyc, 1), "mean_cycles_per_valu": round(tot_cyc / tot, 3),
                "source": os.path.basename(f), "classes_source": os.path.basename(cf) if cf else None,
                "bound_fps_4cycles": round(1024 * 2.4e9 / 4 / (tot * nmb), 1),
                "bound_fps": round(1024 * 2.4e9 / (tot_cyc * nmb), 1),
                "note": "all kernels, GOP average (1 IDR + %d P); 256 CUs x 4 SIMDs at 2.4 GHz; bound_fps prices each kernel's dynamic VALU count by its static "
                        "mix of the two measured issue classes (2.3 / 4.15 cycles per wave-instruction with the SIMD saturated)" % (GOP - 1)}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--gops-in-flight", type=int, default=int(os.environ.get("BENCH_GOPS_IN_FLIGHT", "64")),
                    help="closed GOPs of the stream resident and encoded per step per GPU")
    ap.add_argument("--stagger-ms", type=float, default=float(os.environ.get("BENCH_STAGGER_MS", "0")),
                    help="instance i starts i x this many milliseconds after instance 0 (inside the timed region)")
    ap.add_argument("--instances", type=int, default=int(os.environ.get("BENCH_INSTANCES", "2")),
                    help="encoder instances (HIP streams) the GOPs in flight are split over; each encodes its share in lockstep")
    ap.add_argument("--cpu-frames", type=int, default=30, help="pictures per CPU-baseline thread (8..30: one closed GOP, 1 IDR + the rest P)")
    ap.add_argument("--cpu-threads", default="", help="CPU baseline: thread counts to sweep, comma separated (default 1,16,32,64,128,256 up to the affinity mask)")
    ap.add_argument("--input", default="i420", choices=["i420", "nv12"],
                    help="layout of the resident pictures; nv12 + --profile main --fps 60 is BASELINE.json configs[2]")
    ap.add_argument("--profile", default="baseline", choices=["baseline", "main", "high"])
    ap.add_argument("--fps", type=int, default=30, choices=[30, 60])
    ap.add_argument("--content", default="s1", choices=["s1", "s2", "s3", "scroll"],
                    help="synthetic input of SURVEY.md 8(d); s1 pan+noise is the headline workload")
    ap.add_argument("--search", default="seeded", choices=["seeded", "exhaustive"],
                    help="integer motion search (config.search): seeded by the previous picture's vector with the exhaustive pass as fall-back (default), or always exhaustive")
    ap.add_argument("--slices", type=int, default=0,
                    help="slices per picture (bands of macroblock rows, SURVEY.md 8e-3); 0/1 = one slice, the reference preset and the headline")
    ap.add_argument("--mode", default="gops", choices=["gops", "plugin", "decode"],
                    help="gops: the headline (closed GOPs resident in HBM); plugin: only the measurement through the VideoEncoder plugin surface")
    ap.add_argument("--streams", default="1,4,16,64", help="plugin measurement: numbers of concurrent streams (encoder objects), comma separated")
    ap.add_argument("--plugin-frames", type=int, default=300, help="plugin measurement: pictures per stream (10 s at 30 Hz)")
    ap.add_argument("--bitrate", type=int, default=5000000, help="plugin measurement: target bitrate (the reference accepts 1..10 Mbps)")
    ap.add_argument("--no-plugin", action="store_true", help="leave the plugin measurement out of the default line")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start one child per GPU ourselves - before this process has touched a GPU (nothing below this
        # block has run yet, torch is not even imported) - and relay rank 0's line.  Never a re-exec of a GPU process.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        line = procs[0].stdout.read().decode()
        rcs = [p.wait() for p in procs]
        sys.stdout.write(line)
        sys.stdout.flush()
        raise SystemExit(max(rcs))

    import numpy as np
    import torch
    from media_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    plugin_streams = [int(x) for x in args.streams.split(",") if x.strip()]
    if args.mode == "plugin":
        res = plugin_bench(plugin_streams, args.plugin_frames, args.bitrate, local_rank)
        best = max(res["results"], key=lambda r: r["fps_aggregate"])
        print(json.dumps({"metric": "encoded fps @1080p I420 baseline-profile through the VideoEncoder plugin surface (host pictures, bitrate mode)",
                          "value": best["fps_aggregate"], "unit": "frames/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": res["what"], "streams_at_value": best["streams"]}, "plugin": res}), flush=True)
        return
    if args.mode == "decode":
        res = decode_bench(max(2, args.plugin_frames), local_rank)
        print(json.dumps({"metric": "decoded fps @1080p through the VideoDecoder peer's C ABI (host parse + GPU reconstruction, one picture of look-ahead)",
                          "value": res["decode_only"]["fps"], "unit": "frames/s", "n_gpus": 1, "higher_is_better": True, "dtype": "u8", "data": "synthetic",
                          "config": {"workload": res["what"]}, "decode": res}), flush=True)
        return
    dist = None
    if world > 1 or os.environ.get("BENCH_FORCE_DIST"):   # the second form rehearses the N>1 path with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    # synthetic workload, resident in HBM before any timing: G closed GOPs of 30 pictures.  GOP 0 is the S1
    # sequence; GOP g is the same sequence with every plane rolled horizontally by 16*g luma samples, so each
    # GOP is distinct data (G x 93 MB) without G x the host-side generation time.
    G = max(1, args.gops_in_flight)
    frames = synth.sequence(args.content, WIDTH, HEIGHT, FRAMES_PER_STEP)
    fbytes = WIDTH * HEIGHT * 3 // 2
    stride = (fbytes + 255) // 256 * 256
    gop_stride = stride * FRAMES_PER_STEP
    host = np.zeros((FRAMES_PER_STEP, stride), np.uint8)
    for i, f in enumerate(frames):
        host[i, :fbytes] = f
    dev0 = torch.from_numpy(host).to("cuda:%d" % local_rank)
    dev = torch.empty((G, FRAMES_PER_STEP, stride), dtype=torch.uint8, device=dev0.device)
    ysz, csz = WIDTH * HEIGHT, WIDTH * HEIGHT // 4
    for g in range(G):
        dev[g] = dev0
        if g:
            dev[g, :, :ysz] = torch.roll(dev0[:, :ysz].view(FRAMES_PER_STEP, HEIGHT, WIDTH), 16 * g, dims=2).reshape(FRAMES_PER_STEP, ysz)
            for o in (ysz, ysz + csz):
                dev[g, :, o:o + csz] = torch.roll(dev0[:, o:o + csz].view(FRAMES_PER_STEP, HEIGHT // 2, WIDTH // 2), 8 * g, dims=2).reshape(FRAMES_PER_STEP, csz)
    if args.input == "nv12":   # same samples, chroma interleaved in place (U plane + V plane -> UV plane)
        uv = torch.stack([dev[:, :, ysz:ysz + csz], dev[:, :, ysz + csz:ysz + 2 * csz]], dim=3).reshape(G, FRAMES_PER_STEP, 2 * csz)
        dev[:, :, ysz:ysz + 2 * csz] = uv
        del uv
    torch.cuda.synchronize()
    profile_idc = {"baseline": 66, "main": 77, "high": 100}[args.profile]
    enc_kw = dict(qp=QP, gop=GOP, device=local_rank, fps=args.fps, profile_idc=profile_idc, input_format=1 if args.input == "nv12" else 0, slices=args.slices,
                  search=1 if args.search == "seeded" else 0)

    # I encoder instances (own HIP stream each), each encoding G/I GOPs in lockstep: every kernel launch of an
    # instance covers all its pictures of one time step (grid.y = G/I); instances overlap each other's
    # dependency-bound deblocking wavefront with throughput-bound kernels.
    I = max(1, min(args.instances, G))
    while G % I:
        I -= 1
    B = G // I
    insts = [capi.Encoder(WIDTH, HEIGHT, batch=B, **enc_kw) for _ in range(I)]
    for i, e_ in enumerate(insts):
        e_.set_idr_pic_id(i * B, 1)
    enc = insts[0]
    cap = FRAMES_PER_STEP * fbytes * 2 if args.content == "s3" else FRAMES_PER_STEP * fbytes // 2   # random noise does not compress
    outs = [np.zeros(B * cap, np.uint8) for _ in range(I)]
    sizes = [np.zeros(B * FRAMES_PER_STEP, np.uint32) for _ in range(I)]
    gop_bytes = [np.zeros(B, np.uint64) for _ in range(I)]
    enc1 = capi.Encoder(WIDTH, HEIGHT, **enc_kw) if G > 1 else None
    out1 = np.zeros(cap, np.uint8)
    sizes1 = np.zeros(FRAMES_PER_STEP, np.uint32)

    def run_inst(i):
        insts[i].encode_gops_device(dev.data_ptr() + i * B * gop_stride, stride, gop_stride, FRAMES_PER_STEP, outs[i], cap,
                                    sizes[i], gop_bytes[i])

    def steps(n):
        """n steps: every instance encodes its share of the GOPs n times, free-running beside the others"""
        if I == 1:
            for _ in range(n):
                run_inst(0)
        else:
            def work(i):
                if args.stagger_ms > 0 and i:
                    time.sleep(i * args.stagger_ms * 1e-3)
                for _ in range(n):
                    run_inst(i)
            ths = [threading.Thread(target=work, args=(i,)) for i in range(I)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
        return int(gop_bytes[0][0])

    def step():
        return steps(1)

    def one_gop(_):
        e1 = enc1 if enc1 is not None else enc
        if enc1 is None:
            return step()
        e1.force_idr()
        return e1.encode_batch_device(dev.data_ptr(), stride, FRAMES_PER_STEP, out1, sizes1)

    encs = insts + ([enc1] if enc1 is not None else [])

    for _ in range(args.warmup):
        step()
    for e_ in encs:
        e_.stats_enable(not os.environ.get("BENCH_NO_STATS"))
        e_.stats(reset=True)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    nbytes = steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda:%d" % local_rank)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # per-kernel HIP-event statistics of the timed region, summed over all instances
    st = enc.stats(reset=True)
    for e_ in insts[1:]:
        o = e_.stats(reset=True)
        for name, v in o["kernels"].items():
            for key in ("ms", "launches", "mbs"):
                st["kernels"][name][key] += v[key]
        for key in ("p_mbs", "me_searched_mbs", "tq_coded_mbs"):
            st[key] += o[key]
    if enc1 is not None:
        enc1.stats(reset=True)
    # the same kernels with the chip to themselves: instance 0 alone, one step (only when instances overlap)
    st_ex = None
    if I > 1:
        run_inst(0)
        torch.cuda.synchronize()
        st_ex = enc.stats(reset=True)
    # latency mode for reference: one GOP in flight, kernels alone on the GPU (also gives the
    # MC+DCT kernel's duration without other streams' kernels sharing the chip)
    fence()
    t1 = time.perf_counter()
    lat_steps = 2
    for _ in range(lat_steps):
        one_gop(0)
    fence()
    lat_dt = time.perf_counter() - t1
    st1 = (enc1 if enc1 is not None else enc).stats(reset=True)
    # self-check of the measured path: GOP 0 as the lockstep batch coded it == the same 30 pictures through the
    # batch-1 encoder (a different launch geometry of the same kernels); both are bit-exact against the CPU oracle
    # in tests/test_gpu_parity.py at sizes the oracle finishes quickly
    selfcheck = None
    if enc1 is not None:
        calls0 = args.warmup + args.steps + (1 if I > 1 else 0)      # encode_gops_device calls instance 0 has made
        enc1.set_idr_pic_id(((calls0 - 1) * B) & 255, 1)             # idr_pic_id its GOP 0 carried in the last one
        n1 = one_gop(0)
        a = outs[0][:int(gop_bytes[0][0])]
        selfcheck = bool(int(gop_bytes[0][0]) == n1 and np.array_equal(a, out1[:n1]))
    for e_ in encs:
        e_.close()

    if rank == 0:
        total_frames = world * args.steps * FRAMES_PER_STEP * G
        fps = total_frames / dt
        k = st["kernels"]
        nmb = (WIDTH // 16) * ((HEIGHT + 15) // 16)
        tq = k["tq"]
        pmb_ms = tq["ms"] / max(1, tq["launches"])
        # what the launches of the timed region really coded (counted on the device, mi355x_h264_stats): k_tq loads and stores
        # samples and levels only for macroblocks k_me searched and did not hand to the intra pass
        p_mbs = max(1, st["p_mbs"])
        coded_frac = st["tq_coded_mbs"] / p_mbs
        searched_frac = st["me_searched_mbs"] / p_mbs

        def tq_bytes(mbs_launch, frac=coded_frac):
            coded = frac * mbs_launch
            return coded * PMB_BYTES_PER_MB + (mbs_launch - coded) * PMB_PROBE_BYTES

        bytes_launch = tq_bytes(nmb * B)
        achieved = bytes_launch / (pmb_ms * 1e-3) / 1e9 if pmb_ms > 0 else 0.0
        per_kernel = {}
        for name, v in k.items():
            if v["launches"]:
                per_kernel[name] = {"ms_per_picture": round(v["ms"] * nmb / max(1, v["mbs"]), 5),
                                    "ms_per_launch": round(v["ms"] / v["launches"], 4),
                                    "pictures_per_launch": round(v["mbs"] / nmb / v["launches"], 2)}
        traffic = None
        traffic_src = None
        try:  # HBM bytes per k_tq launch from the newest committed PMC passes (profiles/), valid for the same batch size
            tf = latest_profile("traffic.json")
            tj = json.load(open(tf))
            if tj.get("lockstep_batch") == B and "k_tq" in tj.get("kernel", "k_tq"):
                traffic = tj["traffic_bytes_per_launch"]
                traffic_src = "profiles/" + os.path.basename(tf)
        except Exception:
            traffic = None
        res = {
            "metric": "encoded fps @1080p I420 baseline-profile, 1/2/4/8 MI355X vs OpenH264 CPU",
            "value": round(fps, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "1080p%d %s synthetic S1 pan+noise, %s profile, fixed QP 26, closed GOPs of 30 " % (args.fps, args.input.upper(), args.profile) +
                                   "(1 IDR + 29 P), %s, 1 ref, %s +-16 integer search, deblock on, CAVLC; per GPU one stream, %d of its "
                                   "closed GOPs per step on %d encoder instance(s), each encoding its %d GOPs in lockstep "
                                   "(grid.y) on its own HIP streams, the instances' motion searches taking turns (one at a time per GPU: device-side lock, "
                                   "MI355X_H264_ME_TURNS); pictures resident in HBM" % ("single slice" if args.slices < 2 else "%d slice bands (filter idc 2)" % args.slices, args.search, G, I, B),
                       "content": args.content, "width": WIDTH, "height": HEIGHT, "qp": QP, "gop": GOP, "frames_per_step": FRAMES_PER_STEP * G, "gops_in_flight": G, "instances": I, "lockstep_batch": B, "me_turns": os.environ.get("MI355X_H264_ME_TURNS", "1") != "0", "stagger_ms": args.stagger_ms,
                       "streams": world, "bytes_per_gop": int(nbytes), "selfcheck_batch_equals_single": selfcheck, "parity": "bit-exact vs CPU oracle "
                       "(oracle unpinned vs OpenH264: no libopenh264 available)"},
            "roofline": {"kernel": "k_tq (residual + fDCT + quant + dequant + iDCT + recon, 8 macroblocks per wave)", "bound": "hbm",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "bytes_per_launch": int(round(bytes_launch)), "macroblocks_per_launch": nmb * B,
                         "coded_macroblocks_per_launch": round(coded_frac * nmb * B, 1), "coded_fraction": round(coded_frac, 4),
                         "bytes_per_coded_macroblock": PMB_BYTES_PER_MB, "bytes_per_settled_macroblock": PMB_PROBE_BYTES,
                         "avg_launch_ms": round(pmb_ms, 5),
                         "how": "bytes_per_launch = coded x 1952 + (macroblocks - coded) x 2, coded counted on the device over the timed region "
                                "(k_me's zero tests settle the rest and k_tq does not touch them); achieved = bytes_per_launch / avg_launch_ms "
                                "(HIP events on the launching stream); pricing every macroblock at 1952 B would read %.4f" %
                                (PMB_BYTES_PER_MB * nmb * B / (pmb_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pmb_ms > 0 else 0.0)},
            "kernels": per_kernel,
            "single_gop_in_flight_fps": round(lat_steps * FRAMES_PER_STEP / lat_dt, 2),
        }
        me = k["me"]
        if me["launches"]:
            me_ms = me["ms"] / me["launches"]
            me_b = ME_BYTES_PER_MB * nmb * B
            res["roofline_me"] = {"kernel": "k_me (motion search + the MC half of MC+DCT: writes the winning prediction)", "bound": "hbm",
                                  "achieved": round(me_b / (me_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(me_b / (me_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "bytes_per_launch": me_b, "bytes_per_macroblock": ME_BYTES_PER_MB,
                                  "avg_launch_ms": round(me_ms, 5), "searched_fraction": round(searched_frac, 4),
                                  "note": "every macroblock is read and written by k_me (a settled one gets its prediction = reconstruction here); "
                                          "the kernel is bound by VALU issue (the exhaustive search), not by HBM - the fraction is reported, not a target met"}
        res["roofline_pipeline"] = {"what": "whole P-picture pipeline, SURVEY.md 8d: 2 720 algorithmic bytes per macroblock x macroblocks/s", "bound": "hbm",
                                    "achieved": round(PIPE_BYTES_PER_MB * nmb * fps / world / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": round(PIPE_BYTES_PER_MB * nmb * fps / world / 1e9 / HBM_PEAK_GBS, 4)}
        if st_ex is not None:
            pe = st_ex["kernels"]["tq"]
            ms_e = pe["ms"] / max(1, pe["launches"])
            cf_e = st_ex["tq_coded_mbs"] / max(1, st_ex["p_mbs"])
            a_e = tq_bytes(nmb * B, cf_e) / (ms_e * 1e-3) / 1e9
            res["roofline_exclusive"] = {"kernel": "k_tq, same launches with one instance running alone (no other stream's kernels "
                                                   "sharing the CUs during the launch)", "achieved": round(a_e, 1), "peak": HBM_PEAK_GBS,
                                         "unit": "GB/s", "frac": round(a_e / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms_e, 5), "coded_fraction": round(cf_e, 4)}
            res["kernels_exclusive"] = {name: {"ms_per_launch": round(v["ms"] / v["launches"], 4)}
                                        for name, v in st_ex["kernels"].items() if v["launches"]}
        p1 = st1["kernels"]["tq"]
        if p1["launches"]:
            ms1 = p1["ms"] / p1["launches"]
            cf1 = st1["tq_coded_mbs"] / max(1, st1["p_mbs"])
            a1 = tq_bytes(nmb, cf1) / (ms1 * 1e-3) / 1e9
            res["roofline_isolated"] = {"kernel": "k_tq, one GOP in flight (no other kernels on the chip)",
                                        "achieved": round(a1, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": round(a1 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms1, 5), "coded_fraction": round(cf1, 4)}
        ib = valu_issue_bound(nmb)
        if ib is not None and args.content == "s1" and args.slices < 2:
            ib["achieved_frac"] = round(fps / world / ib["bound_fps"], 3)
            res["valu_issue"] = ib
        if world == 1 and not args.no_plugin and args.content == "s1" and args.slices < 2 and args.input == "i420":
            try:
                res["plugin"] = plugin_bench(plugin_streams, args.plugin_frames, args.bitrate, local_rank)
            except SystemExit as exc:   # never let the side measurement take the headline line down
                res["plugin"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            # N independent encoder instances on N host threads.  The sweep shows where the box's CPU
            # share saturates (a cgroup quota makes more threads than that share SLOWER); the best row is the baseline.
            lim = cpu_limits()
            avail = max(1, min(lim["affinity"], lim["hardware_threads"] or lim["affinity"]))
            ref = openh264_differential(frames, args.cpu_frames)
            sys.stderr.write("oracle: %s\n" % ("absent (no libopenh264.so on this box; own CPU restatement timed instead)"
                                               if ref.get("oracle") != "openh264" else "openh264 found"))
            sys.stderr.write("cpu limits: %s\n" % json.dumps(lim))
            counts = [c for c in (1, 16, 32, 64, 128, 256) if c <= avail]
            if avail not in counts and avail < 16:
                counts.append(avail)
            if args.cpu_threads:
                counts = [int(x) for x in args.cpu_threads.split(",")]
            rows, single = cpu_baseline(frames, counts, max(8, min(args.cpu_frames, GOP)), 1 if args.search == "seeded" else 0)
            best = max(rows, key=lambda r: r["fps"])
            per = max(8, min(args.cpu_frames, GOP))
            res["cpu_baseline"] = {"value": best["fps"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
                                   "sample": "the CPU oracle, T independent encoder instances on T threads, each created and warmed with one picture "
                                             "before the clock, then encoding one closed GOP of the same 1080p S1 workload (1 IDR + %d P pictures) from a "
                                             "common start; swept over T = %s, the best row is `value` / `cores`" % (per - 1, [r["threads"] for r in rows]),
                                   "sweep": rows, "single_core_fps": single, "limits": lim,
                                   "note": "own scalar CPU restatement (exhaustive +-16 search algorithm), not OpenH264: a baseline, not a target",
                                   "openh264": ref}
            if ref.get("oracle") == "openh264" and ref.get("fps"):
                # a real library on the box: the reference's own single-threaded configuration is the baseline
                res["cpu_baseline"].update({"value": round(ref["fps"], 2), "cores": 1, "kind": "reference",
                                            "sample": "libopenh264.so with the reference preset (bitrate mode, 1 thread) on the first "
                                                      "%d pictures of the same workload" % ref.get("frames", 0),
                                            "port_fps_best": best["fps"], "port_threads_best": best["threads"]})
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
