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


def _worker(rank, world, port, mode, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle
    o = oracle.Oracle(brute_force=False, threads=2)
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


def _run(mode, tmp_path):
    mp.spawn(_worker, args=(2, _free_port(), mode, str(tmp_path)), nprocs=2, join=True)
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
