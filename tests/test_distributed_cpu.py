"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run the product's partition / gather /
stitch logic (rtamd.distributed) with the ORACLE injected as the tile renderer (tests may use the
oracle; the product never does).  The stitched image must be bit-identical to a single-rank render."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path

W, H, SPP = 40, 27, 2   # ragged: 5x4 tiles with partial right/bottom tiles


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    for p in (os.path.join(ROOT, "rust-raytracer_amd"), os.path.join(ROOT, "oracle")):
        sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    import oracle
    from rtamd.distributed import TileLayout, render_frame
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    layout = TileLayout(W, H, world)
    sc = oracle.load_scene_file(scene_path("scene_10.json"), aspect=W / H)

    def render_tiles(r):
        buf = np.zeros((layout.stride, 64, 3))
        for lt, t in enumerate(layout.tiles_of(r)):
            x0, y0, x1, y1 = layout.tile_rect(t)
            img, _ = sc.render(W, H, SPP, seed=1, window=(x0, y0, x1, y1), n_jobs=1, n_workers=1)
            tile = np.zeros((8, 8, 3))
            tile[: y1 - y0, : x1 - x0] = img
            buf[lt] = tile.reshape(64, 3)
        return torch.from_numpy(buf.reshape(-1))

    frame = render_frame(render_tiles, layout, rank, dist)
    if rank == 0:
        np.save(out_path, frame)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_partition_gather_stitch_is_bit_identical(world, tmp_path):
    import torch.multiprocessing as mp
    import oracle
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    ref, _ = oracle.load_scene_file(scene_path("scene_10.json"), aspect=W / H).render(W, H, SPP, seed=1)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_layout_matches_the_c_abi():
    import rtamd
    from rtamd.distributed import TileLayout
    for (w, h, world) in [(1200, 1200, 8), (40, 27, 3), (8, 8, 1), (17, 9, 5), (64, 64, 7)]:
        lay = TileLayout(w, h, world)
        owned = 0
        for r in range(world):
            p = rtamd.default_params(width=w, height=h, rank=r, world=world)
            assert rtamd.tiles_total(p) == lay.tiles_total
            assert rtamd.tiles_owned(p) == lay.owned(r) == len(lay.tiles_of(r))
            owned += lay.owned(r)
        assert owned == lay.tiles_total
        m = lay.gather_index_map()
        assert m.shape == (h, w) and len(np.unique(m)) == w * h    # every pixel has its own slot
