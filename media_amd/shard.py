"""Multi-GPU sharding of the encode path: independent streams (or closed GOPs of one
stream) go to ranks with no data-path collective; torch.distributed carries only the
barrier and the max-over-ranks clock (BASELINE.json configs[3])."""


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
