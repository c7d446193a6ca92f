#!/usr/bin/env python3
"""BASELINE.json configs[4]: ONE 4K30 I420 stream, slice-parallel over the GPUs of a node.

    python tools/bench_bands.py                       # 1 GPU: one instance, `--slices` slices per picture
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/bench_bands.py --gpus N                 # N GPUs: rank r codes band r of every picture

Every rank holds the pictures of one closed GOP in HBM and an encoder with band_index = rank, band_count = N.  A step
is the GOP (30 pictures); after every picture the ranks swap two macroblock rows of reconstruction with their
neighbours (RCCL send/recv, media_amd/shard.py) and the slice NAL units are gathered on rank 0.  Total work is fixed
as N grows: "scaling": "strong".  --refs 3 (default) is the 3-reference search configs[4] names (the reference preset
itself uses one reference, VideoEncoderOpenH264.cpp:290: --refs 1).  The default bench.py line
(closed-GOP sharding, 1080p) stays the headline; this is the measurement harness of the slice-parallel path."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--slices", type=int, default=8)
    ap.add_argument("--size", default="4k", choices=["4k", "1080p"])
    ap.add_argument("--refs", type=int, default=3, help="reference frames searched (configs[4]: 3)")
    args = ap.parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: one child per GPU, started before this process touches a GPU; rank 0's line is relayed
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
    from media_amd import capi, shard, synth
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    w, h = (3840, 2160) if args.size == "4k" else (1920, 1080)
    gop, qp = 30, 26
    frames = synth.sequence("s1", w, h, gop)
    d_frames = torch.from_numpy(np.stack(frames)).to(dev)
    enc = capi.Encoder(w, h, qp=qp, gop=gop, device=local_rank, slices=args.slices, refs=args.refs, band_index=rank if world > 1 else 0,
                       band_count=world if world > 1 else 0)
    halo = shard.BandHalo(enc.band_info()[4], dev) if world > 1 else None
    nbytes = [0]

    def step():
        total = 0
        for i in range(gop):
            part = enc.encode_device(d_frames[i].data_ptr())[0]
            if world > 1:
                shard.exchange_band_halos(enc, rank, world, dist, halo)
                part = shard.gather_access_unit(part, rank, world, dist, dev)
            total += len(part)
        nbytes[0] = total

    def sync():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = shard.max_over_ranks(time.perf_counter() - t0, dist, dev)
    if rank == 0:
        print(json.dumps({
            "metric": "encoded fps of ONE %s stream, slice-parallel" % args.size, "value": round(args.steps * gop / dt, 2), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "ms_per_picture": round(dt / args.steps / gop * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%dx%d I420 synthetic S1, baseline profile, fixed QP %d, closed GOP of %d, %d slice bands per picture "
                                   "(disable_deblocking_filter_idc 2), %d reference picture(s); band r of every picture on GPU r, halo swap per picture" % (w, h, qp, gop, args.slices, max(1, args.refs)),
                       "bytes_per_gop": nbytes[0]}}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
