"""Multi-GPU partition of the framebuffer.

Every (iteration, pixel) sample is independent and its RNG stream is keyed by the GLOBAL
pixel index (reference: makeSeededRandomEngine(iter, idx, depth), src/pathtrace.cu:203-207,368),
so the frame is cut by rows; each rank renders its rows for all iterations with no
communication, and the float tiles are collected once, at image write-out, with a single
gather (RCCL over xGMI when the backend is "nccl").  The assembled image is bit-identical to
the single-GPU image: nothing is summed across GPUs.

Two partitions:
  * striped (default for N > 1): rank r owns rows r, r+N, r+2N, ...  Work per row varies smoothly
    down the frame (cornell 1080p: the top block of 135 rows costs 1.8x the bottom block because of
    the light), so contiguous blocks give a projected 8-GPU efficiency of only 0.71; interleaved rows
    give every rank the same mix.  64-pixel wave groups still lie inside one row.
  * contiguous: one block of rows per rank (`tile_for_rank`), kept for A/B runs.
"""
from __future__ import annotations

from typing import List, Tuple


def tile_rows(height: int, rank: int, world: int) -> Tuple[int, int]:
    """Rows [r0, r1) owned by `rank`: the first (height % world) ranks get one extra row."""
    base, extra = divmod(height, world)
    r0 = rank * base + min(rank, extra)
    return r0, r0 + base + (1 if rank < extra else 0)


def tile_for_rank(width: int, height: int, rank: int, world: int) -> Tuple[int, int]:
    """(pixel_begin, pixel_count) of the rank's tile in global pixel indices (idx = x + y*W)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    if world > height:
        raise ValueError(f"more ranks ({world}) than image rows ({height})")
    r0, r1 = tile_rows(height, rank, world)
    return r0 * width, (r1 - r0) * width


def striped_tile_for_rank(width: int, height: int, rank: int, world: int) -> dict:
    """Renderer options of the rank's row-interleaved tile (rows rank, rank+world, ...)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    if world > height:
        raise ValueError(f"more ranks ({world}) than image rows ({height})")
    rows = (height - rank + world - 1) // world
    return dict(pixel_begin=rank * width, pixel_count=rows * width, stripe_pixels=width if world > 1 else 0,
                stripe_stride=world * width if world > 1 else 0)


def striped_rows(height: int, rank: int, world: int) -> List[int]:
    return list(range(rank, height, world))


def gather_tiles(tile, width: int, height: int, rank: int, world: int, dst: int = 0, striped: bool = False,
                 always_collective: bool = False):
    """Collect per-rank tiles ([count, 3] float32 tensors, on the GPU for nccl) on rank `dst`
    and return the assembled [H*W, 3] image there (None elsewhere).  One collective.
    `always_collective`: go through the process group even for one rank (tests: the nccl = RCCL
    gather on a one-GPU box)."""
    if world == 1 and not always_collective:
        return tile
    import torch
    import torch.distributed as dist

    if striped:
        counts = [striped_tile_for_rank(width, height, r, world)["pixel_count"] for r in range(world)]
    else:
        counts = [tile_for_rank(width, height, r, world)[1] for r in range(world)]
    assert tile.shape[0] == counts[rank], (tile.shape, counts[rank])
    if dist.get_backend() == "gloo" and tile.is_cuda:  # rehearsal mode: several ranks on one card
        tile = tile.cpu()
    pad = max(counts)
    send = tile
    if tile.shape[0] != pad:  # gather needs equal-sized buffers
        send = torch.zeros((pad, 3), dtype=tile.dtype, device=tile.device)
        send[: tile.shape[0]] = tile
    if rank == dst:
        bufs: List = [torch.empty((pad, 3), dtype=tile.dtype, device=tile.device) for _ in range(world)]
        dist.gather(send, gather_list=bufs, dst=dst)
        if not striped:
            return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
        full = torch.empty((height, width, 3), dtype=tile.dtype, device=tile.device)
        for r, (b, c) in enumerate(zip(bufs, counts)):
            full[r::world] = b[:c].view(-1, width, 3)  # rows r, r+world, ... in the rank's own order
        return full.view(height * width, 3)
    dist.gather(send, gather_list=None, dst=dst)
    return None
