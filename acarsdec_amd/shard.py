"""Channel sharding across the GPUs of a node.

Channels are independent units: a channel's down-converter and demodulator state never touches
another channel's, so the multi-GPU path is a pure partition -- rank r owns channels
{c : c mod world == r} (BASELINE.json configs[3]: "sharded round-robin"), generates/receives its
own input and keeps its own state.  There is NO data-path collective.  torch.distributed (RCCL on
GPUs, gloo in the CPU tests) carries only the trivial exchanges: the per-channel configuration from
rank 0 (ONE BROADCAST of the whole table, 32 B per channel, from which every rank keeps its own rows --
not an ncclSend/Recv scatter: 0.5 MB at 16 384 channels is not worth one), the barrier around the
timed region and the reduction/gather of counts, timings and decoded blocks.
"""
import numpy as np


def owned_channels(nch_total, rank, world):
    """Global channel ids owned by `rank` (round-robin)."""
    return np.arange(rank, nch_total, world, dtype=np.int64)


def owner_of(channel, world):
    return int(channel) % int(world)


def local_index(channel, world):
    return int(channel) // int(world)


def scatter_channel_config(cfg_rows, world, rank, dist=None, src=0, device=None, force=False):
    """Rank `src` holds one config row per GLOBAL channel (e.g. [offset_hz, phase, track, id]); every
    rank ends up with the rows of the channels it owns.  The name says what the caller gets; the transport is a BROADCAST:
    the table is tiny (32 B per channel), so it is sent whole with the most basic collective and sliced locally; the same code
    runs over gloo (CPU tests) and RCCL (device tensors).  cfg_rows may be None on other ranks.
    force: go through the collective even with world == 1 (bench.py --rccl-selftest: the RCCL path on a one-GPU box)."""
    if dist is None or (world == 1 and not force):
        rows = np.asarray(cfg_rows, dtype=np.float64)
        return rows[owned_channels(len(rows), rank, world)]
    import torch
    dev = device if device is not None else torch.device("cpu")
    shape = torch.zeros(2, dtype=torch.int64, device=dev)
    if rank == src:
        rows = np.ascontiguousarray(np.asarray(cfg_rows, dtype=np.float64))
        shape = torch.tensor(list(rows.shape), dtype=torch.int64, device=dev)
    dist.broadcast(shape, src=src)
    n, k = int(shape[0].item()), int(shape[1].item())
    table = torch.from_numpy(rows).to(dev) if rank == src else torch.empty((n, k), dtype=torch.float64, device=dev)
    dist.broadcast(table, src=src)
    return table.cpu().numpy()[owned_channels(n, rank, world)]


def gather_blocks(local_blocks, own_ids, world, rank, dist=None, dst=0):
    """Blocks decoded on this rank carry LOCAL channel indices; returns on `dst` the merged list
    with GLOBAL channel ids, ordered by (channel, end_bit) like the single-GPU drain."""
    fixed = [(int(own_ids[b[0]]),) + tuple(b[1:]) for b in local_blocks]
    if dist is None:
        return sorted(fixed, key=lambda b: (b[0], b[-1]))
    box = [None] * world if rank == dst else None
    dist.gather_object(fixed, box, dst=dst)
    if rank != dst:
        return None
    merged = [b for part in box for b in part]
    return sorted(merged, key=lambda b: (b[0], b[-1]))


def reduce_timing(seconds, count, world, dist=None, device=None):
    """max over ranks of the timed-region duration, sum of a per-rank count.  With a `dist` handle the collectives run
    whatever the world size (a world of one still exercises the backend)."""
    if dist is None:
        return float(seconds), float(count)
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    c = torch.tensor([float(count)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), float(c.item())


def gather_scalars(x, world, dist=None, device=None):
    """every rank's scalar, in rank order, on every rank (per-GPU timings of the bench report)."""
    if dist is None:
        return [float(x)]
    import torch
    mine = torch.tensor([float(x)], dtype=torch.float64, device=device)
    box = [torch.zeros(1, dtype=torch.float64, device=device) for _ in range(world)]
    dist.all_gather(box, mine)
    return [float(b.item()) for b in box]
