"""Pixel sharding of one frame across the GPUs of a node (host logic, no GPU needed).

The path shards trivially: one path per pixel per iteration, exclusive pixel ownership, RNG streams keyed on
the GLOBAL pixel index, so any partition renders the same bits.  Each rank owns one
contiguous band of rows -- a contiguous slice of the reference's row-major framebuffer
(index = x + y*W, ref: src/raytraceKernel.cu:98) -- or, to balance the ranks when path lengths vary down the
frame, interleaved strips of STRIP_ROWS rows (strip k belongs to rank k % world); renders them with
no communication, and the tiles are gathered to rank 0 once, when the host wants the image.  The gather is the only exchange step; over xGMI
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


STRIP_ROWS = 8     # stripH = 8


def strip_local_rows(height, world, rank, strip_rows=STRIP_ROWS):
    """Number of frame rows `rank` owns under interleaved strips (== pt_strip_local_rows of the C-ABI)."""
    if world < 1 or not 0 <= rank < world or strip_rows < 1:
        raise ValueError("rank/world/strip_rows")
    nstrips = (height + strip_rows - 1) // strip_rows
    return sum(min(strip_rows, height - k * strip_rows) for k in range(rank, nstrips, world))


def strip_global_rows(height, world, rank, strip_rows=STRIP_ROWS):
    """Frame rows of the rank's local rows 0, 1, ... (its strips packed in order)."""
    nstrips = (height + strip_rows - 1) // strip_rows
    rows = []
    for k in range(rank, nstrips, world):
        rows.extend(range(k * strip_rows, min(height, (k + 1) * strip_rows)))
    return rows


def max_strip_rows(height, world, strip_rows=STRIP_ROWS):
    return max(strip_local_rows(height, world, r, strip_rows) for r in range(world))


def gather_buffers(local_tile, height, world, rank, dst=0):
    """Receive buffers and frame for gather_strips / gather_bands, allocated ONCE by the caller (bench.py: before its timed
    region) and passed to every gather: (bufs, frame) on `dst`, (None, None) elsewhere."""
    import torch
    if world == 1 or rank != dst:
        return None, None
    bufs = [torch.empty_like(local_tile) for _ in range(world)]
    frame = torch.empty((height,) + tuple(local_tile.shape[1:]), dtype=local_tile.dtype, device=local_tile.device)
    return bufs, frame


def gather_strips(local_tile, height, world, rank, strip_rows=STRIP_ROWS, dist=None, dst=0, bufs=None, frame=None):
    """Gather the per-rank strip tiles (tensors [max_strip_rows, W, 3], the first strip_local_rows rows valid) to
    `dst` and put every row at its place in the [height, W, 3] frame.  Returns the frame on dst, None elsewhere.
    bufs / frame: preallocated by gather_buffers (nothing is allocated here then)."""
    import torch
    if world == 1:
        return local_tile[:height]
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(local_tile) for _ in range(world)]
    dist.gather(local_tile, bufs if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    if frame is None:
        frame = torch.empty((height,) + tuple(local_tile.shape[1:]), dtype=local_tile.dtype, device=local_tile.device)
    for r in range(world):
        rows = _strip_rows_tensor(height, world, r, strip_rows, local_tile.device)
        frame.index_copy_(0, rows, bufs[r][: rows.numel()])
    return frame


_ROWS_CACHE = {}


def _strip_rows_tensor(height, world, r, strip_rows, device):
    """Global row numbers of rank r's strips as an index tensor on `device`, built once per (frame, rank): the gather of a
    frame sits inside bench.py's timed region, the host-side list and its upload need not."""
    import torch
    key = (height, world, r, strip_rows, str(device))
    t = _ROWS_CACHE.get(key)
    if t is None:
        t = torch.as_tensor(strip_global_rows(height, world, r, strip_rows), dtype=torch.long, device=device)
        _ROWS_CACHE[key] = t
    return t


def weak_scaled_frame(width, height, world):
    """Frame for `world` GPUs at fixed per-GPU pixel count and fixed aspect (world=4 -> 2x in both dimensions).  The height
    is a multiple of STRIP_ROWS, so every rank's strips are whole and its pixel count a multiple of 64 (the renderer's
    camera-ray tables need chunks that are 64-pixel spans)."""
    if world == 1:
        return width, height
    s = float(world) ** 0.5
    return int(round(width * s / 16.0)) * 16, int(round(height * s / STRIP_ROWS)) * STRIP_ROWS


def scaled_frame(width, height, world, scaling="weak"):
    """Frame rendered by `world` GPUs: "strong" = the configuration's own frame, cut into `world` tiles (BASELINE
    configs[3] and [4] are quoted this way: one 3840x2160 / 1920x1080 frame over the 8 GPUs of a node); "weak" = a frame
    that grows with the GPU count so that every rank keeps ~width x height pixels."""
    if scaling not in ("weak", "strong"):
        raise ValueError("scaling")
    if scaling == "strong" or world == 1:
        return width, height
    return weak_scaled_frame(width, height, world)


def gather_bands(local_band, height, world, rank, dist=None, dst=0, bufs=None, frame=None):
    """Gather the per-rank bands (tensors [max_band_rows, W, 3], only the first rows of each are valid) to
    `dst` and assemble the [height, W, 3] frame there.  Returns the frame on dst, None elsewhere.
    `dist` is torch.distributed (any backend: nccl on GPUs, gloo in the CPU tests).
    bufs / frame: preallocated by gather_buffers (nothing is allocated here then)."""
    import torch
    if world == 1:
        r0, r1 = band_rows(height, 1, 0)
        return local_band[: r1 - r0]
    if rank == dst and bufs is None:
        bufs = [torch.empty_like(local_band) for _ in range(world)]
    dist.gather(local_band, bufs if rank == dst else None, dst=dst)
    if rank != dst:
        return None
    if frame is None:
        frame = torch.empty((height,) + tuple(local_band.shape[1:]), dtype=local_band.dtype, device=local_band.device)
    for r in range(world):
        r0, r1 = band_rows(height, world, r)
        frame[r0:r1].copy_(bufs[r][: r1 - r0])
    return frame
