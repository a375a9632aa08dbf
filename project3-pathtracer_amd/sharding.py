"""Pixel sharding of one frame across the GPUs of a node (host logic, no GPU needed).

The path shards trivially: one path per pixel per iteration, exclusive pixel ownership, RNG streams keyed on
the GLOBAL pixel index, so any partition renders the same bits.  Each rank owns one
contiguous band of rows -- a contiguous slice of the reference's row-major framebuffer
(index = x + y*W, ref: src/raytraceKernel.cu:98) -- renders it with no communication, and the bands are
gathered to rank 0 once, when the host wants the image.  The gather is the only exchange step; over xGMI
every peer has its own direct link to the root, so it is a set of concurrent point-to-point sends
(torch.distributed.gather on the nccl = RCCL backend), not a ring.
"""


def band_rows(height, world, rank):
    """Rows [r0, r1) owned by `rank`: as even as possible, every row owned exactly once."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("rank/world")
    return height * rank // world, height * (rank + 1) // world


def max_band_rows(height, world):
    return max(band_rows(height, world, r)[1] - band_rows(height, world, r)[0] for r in range(world))


def weak_scaled_frame(width, height, world):
    """Frame for `world` GPUs at fixed per-GPU pixel count and fixed aspect (world=4 -> 2x in both dimensions)."""
    if world == 1:
        return width, height
    s = float(world) ** 0.5
    return int(round(width * s / 16.0)) * 16, int(round(height * s / 2.0)) * 2


def gather_bands(local_band, height, world, rank, dist=None, dst=0):
    """Gather the per-rank bands (tensors [max_band_rows, W, 3], only the first rows of each are valid) to
    `dst` and assemble the [height, W, 3] frame there.  Returns the frame on dst, None elsewhere.
    `dist` is torch.distributed (any backend: nccl on GPUs, gloo in the CPU tests)."""
    import torch
    if world == 1:
        r0, r1 = band_rows(height, 1, 0)
        return local_band[: r1 - r0]
    bufs = [torch.empty_like(local_band) for _ in range(world)] if rank == dst else None
    dist.gather(local_band, bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        r0, r1 = band_rows(height, world, r)
        parts.append(bufs[r][: r1 - r0])
    return torch.cat(parts, dim=0)
