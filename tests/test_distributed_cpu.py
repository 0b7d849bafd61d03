"""World-size-2 gloo tests (CPU) of the multi-GPU sharding logic in raytracer_amd/distributed.py, with the CPU oracle
standing in for the per-rank renderer: the partitioning, seeding and the reduce step are backend independent."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracer_amd import scenes, distributed


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class _HostTiles(object):
    """The tile entry points of the C-ABI (rt_tile_pack_bytes / rt_pack_tiles / rt_unpack_tiles, include/rt_abi.h) over a
    host accumulator: local tile j of rank r is tile r + j * world of the image, 32x32 pixels of 16 bytes each, row-major
    inside the tile, pixels beyond the image zero.  Stands in for the product so that distributed.gather_tiles -- the
    exchange of the tile split -- runs under gloo on CPU."""

    def __init__(self, acc):  # acc: (h, w, 4) float32, modified in place by unpack_tiles
        self.acc = acc
        self.h, self.w = acc.shape[:2]
        self.tx, self.ty = (self.w + 31) // 32, (self.h + 31) // 32

    def tile_pack_bytes(self, world):
        return ((self.tx * self.ty + world - 1) // world) * 1024 * 16

    def _view(self, ptr, world):
        import ctypes
        n = self.tile_pack_bytes(world) // 4
        return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctypes.c_float)), shape=(n,)).reshape(-1, 32, 32, 4)

    def _tiles(self, rank, world):
        for j, t in enumerate(range(rank, self.tx * self.ty, world)):
            y0, x0 = (t // self.tx) * 32, (t % self.tx) * 32
            yield j, y0, x0, min(32, self.h - y0), min(32, self.w - x0)

    def pack_tiles(self, rank, world, ptr):
        v = self._view(ptr, world)
        v[...] = 0.0
        for j, y0, x0, hh, ww in self._tiles(rank, world):
            v[j, :hh, :ww] = self.acc[y0:y0 + hh, x0:x0 + ww]

    def unpack_tiles(self, rank, world, ptr):
        v = self._view(ptr, world)
        for j, y0, x0, hh, ww in self._tiles(rank, world):
            self.acc[y0:y0 + hh, x0:x0 + ww] = v[j, :hh, :ww]

    def sync(self):
        pass


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    o = oracle.Oracle(brute_force=False, threads=2)
    if mode == "tile_gather":
        scene = scenes.cornell_bunny(96, 70, spp=2, bounces=2, sample_radius=0.06)
        scene.setup(o)
        o.set_tile_split(rank, world)
        for f in range(2):
            distributed.render_frame(o, scene.globals, f, 77)
        acc = o.read_accumulator().copy()
        nbytes = distributed.gather_tiles(_HostTiles(acc), dist, rank, world, torch.device("cpu"), dst=0, via_host=True)
        assert nbytes == ((3 * 3 + world - 1) // world) * 16384   # 96x70 pixels = 3x3 tiles
        if rank == 0:
            np.save(os.path.join(out_dir, "acc.npy"), acc)
        dist.barrier()
        dist.destroy_process_group()
        return
    if mode == "tile":
        scene = scenes.cornell_bunny(96, 70, spp=2, bounces=2, sample_radius=0.06)
        scene.setup(o)
        o.set_tile_split(rank, world)
        for f in range(2):
            distributed.render_frame(o, scene.globals, f, 77)
    else:
        scene = scenes.cornell(64, 36, spp=1, bounces=3)
        scene.setup(o)
        for f in distributed.sample_split_frames(rank, world, 5):
            distributed.render_frame(o, scene.globals, f, 77)
    acc = torch.from_numpy(o.read_accumulator().copy())
    distributed.reduce_accumulator(acc, dist, dst=0)
    if rank == 0:
        np.save(os.path.join(out_dir, "acc.npy"), acc.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _points_worker(rank, world, port, n_points, out_dir):
    """distributed.exchange_point_blocks under gloo: rank r has the fresh payloads of block r of the points only"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    block = (n_points + world - 1) // world
    rng = np.random.RandomState(5)
    truth = rng.rand(n_points * 6).astype(np.float32)            # what one GPU would hold after the pass
    stale = rng.rand(n_points * 6).astype(np.float32)            # the other blocks before the exchange
    lo, hi = min(rank * block, n_points), min((rank + 1) * block, n_points)
    mine = stale.copy(); mine[lo * 6:hi * 6] = truth[lo * 6:hi * 6]
    pts = torch.from_numpy(mine)
    for _ in range(2):                                           # twice: the staging tensors of a caller are reused
        distributed.exchange_point_blocks(pts, n_points, block, rank, world, dist)
    np.save(os.path.join(out_dir, "pts%d.npy" % rank), pts.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_points", [(2, 759), (3, 10), (3, 2), (2, 1)])
def test_sample_point_blocks_are_all_gathered(tmp_path, world, n_points):
    """the exchange of the partitioned sample-point pass (distributed.install_probe_exchange): equal,
    zero-padded blocks -- the last one short, or empty when there are fewer points than ranks -- through one all_gather;
    every rank ends with every block's fresh payloads"""
    mp.spawn(_points_worker, args=(world, _free_port(), n_points, str(tmp_path)), nprocs=world, join=True)
    truth = np.random.RandomState(5).rand(n_points * 6).astype(np.float32)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "pts%d.npy" % r))
        assert np.array_equal(got.view(np.uint32), truth.view(np.uint32)), r


def _run(mode, tmp_path, world=2):
    mp.spawn(_worker, args=(world, _free_port(), mode, str(tmp_path)), nprocs=world, join=True)
    return np.load(os.path.join(str(tmp_path), "acc.npy"))


def test_tile_split_is_bit_exact(tmp_path, oracle_mod):
    got = _run("tile", tmp_path)
    o = oracle_mod.Oracle(brute_force=False, threads=4)
    scene = scenes.cornell_bunny(96, 70, spp=2, bounces=2, sample_radius=0.06)
    scene.setup(o)
    for f in range(2):
        distributed.render_frame(o, scene.globals, f, 77)
    want = o.read_accumulator()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    o.close()


@pytest.mark.parametrize("world", [2, 3])
def test_tile_split_by_gather_of_owned_tiles_is_bit_exact(tmp_path, oracle_mod, world):
    """distributed.gather_tiles, the exchange bench.py --split tile uses: every rank packs the tiles it owns, rank 0 gathers
    and unpacks them (9 tiles over 2 or 3 ranks: unequal shares, padded tiles at the right and bottom edges)"""
    got = _run("tile_gather", tmp_path, world)
    o = oracle_mod.Oracle(brute_force=False, threads=4)
    scene = scenes.cornell_bunny(96, 70, spp=2, bounces=2, sample_radius=0.06)
    scene.setup(o)
    for f in range(2):
        distributed.render_frame(o, scene.globals, f, 77)
    want = o.read_accumulator()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    o.close()


def test_sample_split_matches_sequential_sum(tmp_path, oracle_mod):
    got = _run("sample", tmp_path)
    o = oracle_mod.Oracle(brute_force=False, threads=4)
    scene = scenes.cornell(64, 36, spp=1, bounces=3)
    scene.setup(o)
    for f in range(5):
        distributed.render_frame(o, scene.globals, f, 77)
    want = o.read_accumulator()
    assert np.allclose(got, want, rtol=1e-5, atol=1e-6)        # equal up to float re-association of the frame sum
    assert got.sum() > 0
    o.close()


def test_frame_seed_chain():
    from raytracer_amd import host
    s = 0x12345678
    assert distributed.frame_seed(s, 0) == host.next_frame_rng(s)
    assert distributed.frame_seed(s, 2) == host.next_frame_rng(host.next_frame_rng(host.next_frame_rng(s)))
    assert distributed.sample_split_frames(1, 4, 10) == [1, 5, 9]
    assert sorted(sum([distributed.sample_split_frames(r, 3, 8) for r in range(3)], [])) == list(range(8))
