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
    """every rank adopts rank `src`'s rate-control state (qp, virtual buffer): the one exchange step of
    bitrate-mode GOP sharding.  dist is torch.distributed (backend nccl = RCCL on GPUs, gloo on CPU) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return rc.state()
    import torch
    t = torch.tensor(list(rc.state()), dtype=torch.int64, device=device)
    dist.broadcast(t, src=src)
    rc.set_state((int(t[0].item()), int(t[1].item())))
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
