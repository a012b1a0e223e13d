"""Multi-GPU rendering of one frame: one process per GPU (torch.distributed; backend "nccl" is
RCCL over xGMI on ROCm, "gloo" on CPU for the tests).

The reference splits the image into 64 row bands, renders them on a thread pool and stitches the
bands on the main thread through an mpsc channel (camera.rs:79-123).  Here the frame is split into
8x8 tiles dealt round-robin to ranks (tile t -> rank t % world: fine-grained, so every rank gets the
same mix of cheap sky tiles and expensive ground tiles), each rank renders its tiles into a
tile-major buffer with the HIP path, and the only exchange is ONE gather of those buffers to rank 0
(the channel), followed by the stitch.  RNG streams are keyed by (pixel, sample), so the stitched
image is bit-identical for every world size.
"""
import numpy as np

TILE = 8
TILE_PIX = 64


class TileLayout:
    """Ownership and addressing of 8x8 tiles; mirrors rt_tiles_total / rt_tiles_owned / assemble_kernel."""

    def __init__(self, width, height, world):
        self.width, self.height, self.world = int(width), int(height), int(world)
        self.tiles_x = (self.width + TILE - 1) // TILE
        self.tiles_y = (self.height + TILE - 1) // TILE
        self.tiles_total = self.tiles_x * self.tiles_y
        self.stride = self.owned(0)  # rank 0 owns the most; every rank pads its buffer to this for the gather

    def owned(self, rank):
        return (self.tiles_total - rank + self.world - 1) // self.world

    def tiles_of(self, rank):
        return range(rank, self.tiles_total, self.world)

    def tile_rect(self, t):
        tx, ty = t % self.tiles_x, t // self.tiles_x
        return tx * TILE, ty * TILE, min(self.width, tx * TILE + TILE), min(self.height, ty * TILE + TILE)

    def gather_index_map(self):
        """index (into the rank-major gathered buffer, in pixels) of every frame pixel, shape [H, W]."""
        y, x = np.mgrid[0:self.height, 0:self.width]
        tile = (y >> 3) * self.tiles_x + (x >> 3)
        r = tile % self.world
        lt = tile // self.world
        pix = ((y & 7) << 3) | (x & 7)
        return (r * self.stride + lt) * TILE_PIX + pix


def stitch_host(gathered_pixels, layout):
    """camera.rs:115-123 on the host: gathered [world*stride*64, 3] -> frame [H, W, 3] (numpy)."""
    g = np.asarray(gathered_pixels).reshape(-1, 3)
    return g[layout.gather_index_map()]


class TileGather:
    """The one exchange step (the mpsc channel of camera.rs:77,108,115): every rank's padded tile buffer -> rank dst.

    The receive side is ONE rank-major buffer [world, stride*192] allocated once; `dist.gather` writes straight into its
    rows and `gathered` is handed to the stitch as it is (no per-frame allocation, no torch.cat copy).
    host_staged=True moves the payload through host memory -- only for rehearsing world > 1 on a box whose ranks share
    one GPU (RCCL refuses two ranks on one device, so the rehearsal runs over gloo).
    force=True runs the exchange even at world 1 (a one-row gather into the receive buffer): the way to execute the
    RCCL calls of the N-rank path -- communicator init, gather of f64 device rows -- on a box with a single GPU."""

    def __init__(self, layout, rank, dist, like, dst=0, host_staged=False, force=False):
        import torch
        self.layout, self.rank, self.dist, self.dst, self.host_staged = layout, rank, dist, dst, host_staged
        self.exchange = layout.world > 1 or (force and dist is not None)
        self.gathered = None
        self.rows = None
        self._host_in = None
        n = layout.stride * TILE_PIX * 3
        if self.exchange and rank == dst:
            dev = torch.device("cpu") if host_staged else like.device
            self.gathered = torch.empty((layout.world, n), dtype=like.dtype, device=dev)
            self.rows = list(self.gathered.unbind(0))
            self._dev_out = torch.empty((layout.world, n), dtype=like.dtype, device=like.device) if host_staged else None
        if self.exchange and host_staged:
            self._host_in = torch.empty(n, dtype=like.dtype, device="cpu")

    def __call__(self, local_tiles):
        if not self.exchange:
            return local_tiles
        src = local_tiles
        if self.host_staged:
            self._host_in.copy_(local_tiles)
            src = self._host_in
        self.dist.gather(src, self.rows if self.rank == self.dst else None, dst=self.dst)
        if self.rank != self.dst:
            return None
        if self.host_staged:
            self._dev_out.copy_(self.gathered)
            return self._dev_out.view(-1)
        return self.gathered.view(-1)


def gather_tiles(local_tiles, layout, rank, dist, dst=0):
    """One-shot form of TileGather (allocates its receive buffer per call; loops should keep a TileGather)."""
    return TileGather(layout, rank, dist, local_tiles, dst)(local_tiles)


def render_frame(render_tiles_fn, layout, rank, dist=None, stitch_fn=None, dst=0):
    """Render one frame over `layout.world` ranks.

    render_tiles_fn(rank) -> torch tensor [stride*64*3] f64 holding this rank's tiles (tile-major).  In the
    product this is World.render_tiles_device on the rank's GPU; tests inject a CPU renderer to exercise the
    partition / gather / stitch logic under gloo.
    stitch_fn(gathered) -> frame; defaults to the host stitch.  Returns the frame on rank dst, None elsewhere."""
    local = render_tiles_fn(rank)
    gathered = gather_tiles(local, layout, rank, dist, dst) if layout.world > 1 else local
    if rank != dst:
        return None
    if stitch_fn is not None:
        return stitch_fn(gathered)
    return stitch_host(gathered.cpu().numpy(), layout)
