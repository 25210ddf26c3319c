"""world_size-2 gloo rehearsal of the N>1 path: row-tile partition, per-rank tile render, one gather,
assembly on rank 0. On the CPU the per-rank renderer is the oracle (test-only); on the GPU bench.py
runs the same plumbing with rtw_render_device and the nccl (RCCL) backend."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_path, h, w, interleaved, scene=0, estimator=0):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    from raytracing_weekend_amd import abi
    from raytracing_weekend_amd.dist import gather_interleaved, gather_tiles, interleaved_shard, partition_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    blob = abi.build_scene(scene, w, h)
    if interleaved:
        row0, row1, stride, n = interleaved_shard(h, world, rank)
        max_rows = max(interleaved_shard(h, world, g)[3] for g in range(world))
        p = abi.make_params(w, h, 3, 5, row0=row0, row1=row1, row_stride=stride, estimator=estimator)
    else:
        rows = partition_rows(h, world)
        max_rows = max(rows[g + 1] - rows[g] for g in range(world))
        p = abi.make_params(w, h, 3, 5, row0=rows[rank], row1=rows[rank + 1], estimator=estimator)
    img, _ = oracle.render(blob, p, threads=2)
    tile = torch.zeros((max_rows, w, 4), dtype=torch.float32)
    tile[: img.shape[0]] = torch.from_numpy(img)
    full = gather_interleaved(tile, h, rank, world) if interleaved else gather_tiles(tile, rows, rank, world)
    dist.barrier()
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,interleaved,scene,estimator", [(2, 37, True, 0, 0), (3, 20, True, 0, 0), (2, 21, False, 0, 0),
                                                                  (2, 19, True, 4, 1)])  # textures, tree, media, corrected estimator
def test_row_shard_gather_matches_single_process(tmp_path, world, h, interleaved, scene, estimator):
    w = 48
    for p in (ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import oracle
    from raytracing_weekend_amd import abi
    out = str(tmp_path / "full.npy")
    mp.spawn(_worker, args=(world, _free_port(), out, h, w, interleaved, scene, estimator), nprocs=world, join=True)
    got = np.load(out)
    want, _ = oracle.render(abi.build_scene(scene, w, h), abi.make_params(w, h, 3, 5, estimator=estimator), threads=2)
    assert got.shape == (h, w, 4)
    assert np.array_equal(got, want)


def test_partition_rows_covers_the_image():
    from raytracing_weekend_amd.dist import partition_rows
    for h in (1, 7, 1080, 4320):
        for n in (1, 2, 3, 4, 8):
            r = partition_rows(h, n)
            assert r[0] == 0 and r[-1] == h and all(0 <= r[i + 1] - r[i] <= -(-h // n) for i in range(n))
