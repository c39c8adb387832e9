"""Channel sharding across the GPUs of a node.

Channels are independent units: a channel's down-converter and demodulator state never touches
another channel's, so the multi-GPU path is a pure partition -- rank r owns channels
{c : c mod world == r} (BASELINE.json configs[3]: "sharded round-robin"), generates/receives its
own input and keeps its own state.  There is NO data-path collective.  torch.distributed (RCCL on
GPUs, gloo in the CPU tests) carries only the trivial exchanges: the scatter of per-channel
configuration from rank 0, the barrier around the timed region and the reduction/gather of
counts, timings and decoded blocks.
"""
import numpy as np


def owned_channels(nch_total, rank, world):
    """Global channel ids owned by `rank` (round-robin)."""
    return np.arange(rank, nch_total, world, dtype=np.int64)


def owner_of(channel, world):
    return int(channel) % int(world)


def local_index(channel, world):
    return int(channel) // int(world)


def scatter_channel_config(cfg_rows, world, rank, dist=None, src=0):
    """Rank `src` holds one config row per GLOBAL channel (e.g. [offset_hz, phase, seed]); every
    rank gets the rows of the channels it owns.  cfg_rows may be None on the other ranks."""
    if world == 1 or dist is None:
        return np.asarray(cfg_rows)[owned_channels(len(cfg_rows), rank, world)]
    import torch
    if rank == src:
        rows = np.asarray(cfg_rows, dtype=np.float64)
        parts = [torch.from_numpy(np.ascontiguousarray(rows[owned_channels(len(rows), r, world)])) for r in range(world)]
        shapes = [list(p.shape) for p in parts]
    else:
        parts, shapes = None, None
    box = [shapes]
    dist.broadcast_object_list(box, src=src)
    shapes = box[0]
    out = torch.empty(shapes[rank], dtype=torch.float64)
    if dist.get_backend() == "nccl":            # RCCL moves device tensors
        dev = torch.device("cuda", torch.cuda.current_device())
        out = out.to(dev)
        parts = [p.to(dev) for p in parts] if parts is not None else None
    dist.scatter(out, scatter_list=parts if rank == src else None, src=src)
    return out.cpu().numpy()


def gather_blocks(local_blocks, own_ids, world, rank, dist=None, dst=0):
    """Blocks decoded on this rank carry LOCAL channel indices; returns on `dst` the merged list
    with GLOBAL channel ids, ordered by (channel, end_bit) like the single-GPU drain."""
    fixed = [(int(own_ids[b[0]]),) + tuple(b[1:]) for b in local_blocks]
    if world == 1 or dist is None:
        return sorted(fixed, key=lambda b: (b[0], b[-1]))
    box = [None] * world if rank == dst else None
    dist.gather_object(fixed, box, dst=dst)
    if rank != dst:
        return None
    merged = [b for part in box for b in part]
    return sorted(merged, key=lambda b: (b[0], b[-1]))


def reduce_timing(seconds, count, world, dist=None, device=None):
    """max over ranks of the timed-region duration, sum of a per-rank count."""
    if world == 1 or dist is None:
        return float(seconds), float(count)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    c = torch.tensor([float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())
