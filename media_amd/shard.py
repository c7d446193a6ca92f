"""Multi-GPU sharding of the encode path: independent streams (or closed GOPs of one
stream) go to ranks with no data-path collective; torch.distributed carries only the
barrier and the max-over-ranks clock (BASELINE.json configs[3]) - and, when
one stream's GOPs are sharded in BITRATE mode, the rate-control state: two integers
broadcast once per round of `world` GOPs (RCCL over xGMI on GPUs, gloo in the CPU tests).
Fixed-QP sharding needs no exchange at all."""


def gops_for_rank(n_gops, rank, world):
    """closed GOP k of a stream -> rank k mod world (round robin keeps ranks balanced)"""
    return list(range(rank, n_gops, world))


def streams_for_rank(n_streams, rank, world):
    return list(range(rank, n_streams, world))


def max_over_ranks(seconds, dist, device=None):
    """whole-job time = slowest rank; dist is torch.distributed or None"""
    if dist is None or not dist.is_initialized():
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def reassemble(parts):
    """parts: {gop_index: bytes} from all ranks -> the stream in display order"""
    return b"".join(parts[k] for k in sorted(parts))


def broadcast_rc_state(rc, src, dist, device=None):
    """every rank adopts rank `src`'s rate-control state (qp, virtual buffer, GOP budget left, pictures left, mean P picture: five integers): the one exchange step of
    bitrate-mode GOP sharding.  dist is torch.distributed (backend nccl = RCCL on GPUs, gloo on CPU) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rc.state()
    import torch
    t = torch.tensor(list(rc.state()), dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    rc.set_state(tuple(int(x) for x in t.tolist()))
    return rc.state()


def encode_gops_bitrate(encode_gop, n_gops, rank, world, rc, dist=None, device=None):
    """Bitrate-mode encode of one stream whose closed GOPs are sharded round-robin.

    Round j = GOPs j*world .. j*world + world - 1, one per rank.  Every rank starts the round from the same
    controller state, drives its own GOP picture by picture (`encode_gop(k, rc)` must call rc.update() per
    picture and returns the GOP's bytes), and the state reached by the rank that holds the LAST GOP of the round
    (stream order) is broadcast as the start of the next round.  With world == 1 this is the serial controller.
    Returns {gop_index: bytes} of this rank."""
    mine = {}
    rounds = (n_gops + world - 1) // world
    for j in range(rounds):
        k = j * world + rank
        if k < n_gops:
            mine[k] = encode_gop(k, rc)
        last = min(n_gops, (j + 1) * world) - 1          # last GOP of this round in stream order
        broadcast_rc_state(rc, last - j * world, dist, device)
    return mine


# ---------------------------------------------------------------------------------------------------------
# Slice bands of ONE picture over the ranks (BASELINE.json configs[4]).  Every rank holds an
# encoder created with the full picture geometry, the same `slices` and band_index = rank, band_count = world:
# it codes its run of whole slices.  Slices do not depend on one another inside a picture (no prediction and, with
# disable_deblocking_filter_idc 2, no loop filtering across them); BETWEEN pictures motion search and compensation
# reach up to 19 sample rows beyond the band, so after every picture each rank swaps two macroblock rows of
# reconstruction with each neighbour: one send/recv pair per neighbour, point to point (on GPUs: RCCL over the direct
# xGMI link of the two devices - no ring, no collective), then the slice NAL units are gathered on rank 0.
# ---------------------------------------------------------------------------------------------------------
class BandHalo:
    """the four exchange buffers of one rank (torch uint8 tensors on `device`: cuda for nccl, cpu for gloo)"""

    def __init__(self, halo_bytes, device=None):
        import torch
        mk = lambda: torch.zeros(halo_bytes, dtype=torch.uint8, device=device)
        self.send_up, self.send_down, self.recv_up, self.recv_down = mk(), mk(), mk(), mk()


def exchange_band_halos(engine, rank, world, dist, halo):
    """engine: .halo_export(edge, ptr) / .halo_import(edge, ptr) (capi.Encoder on a GPU, the oracle in the CPU test).
    After the call the reference rows right above and below this rank's band are the neighbours' newest rows."""
    if world == 1:
        return
    ops = []
    if rank > 0:
        engine.halo_export(0, halo.send_up.data_ptr())
        ops += [dist.P2POp(dist.isend, halo.send_up, rank - 1), dist.P2POp(dist.irecv, halo.recv_up, rank - 1)]
    if rank < world - 1:
        engine.halo_export(1, halo.send_down.data_ptr())
        ops += [dist.P2POp(dist.isend, halo.send_down, rank + 1), dist.P2POp(dist.irecv, halo.recv_down, rank + 1)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    if halo.recv_up.is_cuda:
        import torch
        torch.cuda.current_stream().synchronize()     # the received rows are copied on the encoder's own stream next
    if rank > 0:
        engine.halo_import(0, halo.recv_up.data_ptr())
    if rank < world - 1:
        engine.halo_import(1, halo.recv_down.data_ptr())


def gather_access_unit(part, rank, world, dist, device=None, cap=None, sizes_out=None):
    """slice NAL units of every band -> the access unit on rank 0 (b'' elsewhere): sizes by all_gather, then the
    payloads as equally sized uint8 tensors (cap = upper bound of one band's bytes, default: the largest size).
    sizes_out (a list) receives every band's byte count on EVERY rank - what a rate controller needs."""
    if world == 1:
        if sizes_out is not None:
            sizes_out[:] = [len(part)]
        return part
    import torch
    n = torch.tensor([len(part)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    if sizes_out is not None:
        sizes_out[:] = sizes
    cap = max(sizes) if cap is None else cap
    mine = torch.zeros(cap, dtype=torch.uint8, device=device)
    mine[:len(part)] = torch.frombuffer(bytearray(part), dtype=torch.uint8).to(mine.device)
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    if rank != 0:
        return b""
    return b"".join(parts[r][:sizes[r]].cpu().numpy().tobytes() for r in range(world))


SCENE_CUT_COST_PER_MB = 3000   # = VideoEncoderMI355X::Rc::kSceneCutCostPerMb (media_amd/host/VideoEncoderMI355X.h)


def encode_picture_bands(engine, frame, rank, world, dist, halo, device=None, rc=None, scene_detect=False, picture_mbs=None,
                         cut_cost_per_mb=SCENE_CUT_COST_PER_MB):
    """one picture of a band-sharded stream: code this rank's slices, swap halos, gather the access unit on rank 0.
    rc (media_amd.ratecontrol.RateControl, the same initial state on every rank): bitrate mode.  Every rank sets the
    picture QP from its own copy of the controller and feeds it the picture's TOTAL size, which the gather's size
    exchange already delivers to every rank - the copies stay identical without any further message.
    scene_detect (the plugin class's rule, bEnableSceneChangeDetect of the reference preset): the bands' motion costs of a P
    picture are summed over the ranks (one all_reduce of one integer); when the mean exceeds SCENE_CUT_COST_PER_MB per
    macroblock of the PICTURE (picture_mbs) every rank drops its band of that P picture and codes it again as an IDR - the
    same decision on every rank, before any halo is swapped."""
    if rc is not None:
        engine.set_qp(rc.qp)
    part, idr = engine.encode(frame)[:2]
    if scene_detect and not (idr == 1 or idr is True):
        import torch
        cost = engine.me_cost()
        cost = int(cost[0]) if hasattr(cost, "__len__") else int(cost)
        if world > 1:
            t = torch.tensor([cost], dtype=torch.int64, device=device)
            dist.all_reduce(t)
            cost = int(t.item())
        if cost > cut_cost_per_mb * int(picture_mbs):
            if hasattr(engine, "force_idr"):
                engine.force_idr()
                part, idr = engine.encode(frame)[:2]
            else:
                part, idr = engine.encode(frame, force_idr=True)[:2]
    exchange_band_halos(engine, rank, world, dist, halo)
    sizes = []
    au = gather_access_unit(part, rank, world, dist, device, sizes_out=sizes)
    if rc is not None:
        rc.update(sum(sizes), idr == 1)      # (capi: frame type 1 = IDR; the oracle engine: True)
    return au
