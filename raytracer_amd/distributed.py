"""Multi-GPU sharding of the path: one process per GPU, torch.distributed (RCCL on GPUs, gloo on CPU
for tests) for the single exchange step.  The reference is single-GPU; these are the two partitions its data flow
allows without changing any pixel's sample stream:

  tile split    (C4)  rank r renders the 32x32 tiles t with t % world == r; a pixel's RNG stream depends only on
                      (px, py, frame seed) (raytracing.hlsl:130), so each pixel is bit-identical to the 1-GPU render.
                      Other ranks' pixels stay zero in the accumulator; the exchange is a gather of the tiles each rank
                      owns (gather_tiles: 1 / world of the frame per rank, copied, not added), which reproduces it exactly
                      (so would a reduce(sum) of the zero-filled frames, at world times the bytes).
                      The probe pass (translucent_rgen) is replicated on every rank by default: it is ~N_points*spp paths
                      against W*H*spp camera paths, and replication keeps sample-point payloads bit-identical without an
                      exchange.  install_probe_exchange partitions it instead: rank r traces block r of the
                      points and an all-gather of the payload blocks follows every pass, on the library's own sample-point
                      stream, beside the frame's camera pass -- the same payloads bit for bit (each point is traced by
                      exactly one rank, with the one-GPU arithmetic).
  sample split  (C5)  rank r renders frames r, r+world, ... of the frame_rng chain (main.cpp:443-454) into its own
                      accumulator; one reduce(sum) of the RGBA32F buffer at the end.  Equal to the sequential sum up to
                      float re-association -- for scenes WITHOUT translucent sample points: the sample-point payloads are
                      a running sum over all earlier frames (raytracing.hlsl:243-263), which a rank that skips frames does
                      not have (require_no_sample_points).  After the reduce the 8-bit image is resolved with the TOTAL
                      frame count (resolve_reduced), not the rank's own.
"""
import numpy as np

from . import host


def frame_seed(seed0, frame):
    """frame_rng used by frame `frame` (0-based): the main loop hashes before every dispatch (main.cpp:443-454)."""
    s = seed0
    for _ in range(frame + 1):
        s = host.next_frame_rng(s)
    return s


class FrameChain(object):
    """incremental frame_seed for monotonically increasing frame indices"""

    def __init__(self, seed0):
        self.seed, self.frame = seed0, -1

    def seed_for(self, frame):
        assert frame > self.frame
        while self.frame < frame:
            self.seed = host.next_frame_rng(self.seed)
            self.frame += 1
        return self.seed


def render_frame(backend, base_globals, frame, seed0):
    """Dispatches global frame `frame`: seeds as the sequential loop would and adds into this backend's accumulator
    (which must still be zero where no earlier frame of this backend has written)."""
    g = base_globals.copy()
    cur = backend.get_globals()
    g.frame_rng = seed0.seed_for(frame) if isinstance(seed0, FrameChain) else frame_seed(seed0, frame)
    g.accumulator_count = frame                       # frame 0 seeds with z = 0; later frames add
    g.translucent_accumulator_count = cur.translucent_accumulator_count
    backend.set_globals(g)
    backend.dispatch_rays()


def sample_split_frames(rank, world, total_frames):
    return list(range(rank, total_frames, world))


def require_no_sample_points(backend):
    """The sample-set split is only equal to the sequential render when no frame depends on the frames before it:
    scenes with translucent sample points (probe payloads accumulate frame by frame) must use the tile split."""
    n = sum(backend.sample_point_count(i) for i in range(backend.translucent_instance_count()))
    if n:
        raise ValueError("sample-set split: the scene has %d translucent sample points whose payloads accumulate over "
                         "consecutive frames; use the tile split (set_tile_split) for it" % n)


def resolve_reduced(backend, total_frames):
    """raytracing.hlsl:161 on an accumulator that now holds `total_frames` frames (after the reduce over ranks): the
    rank's own accumulator_count is one rank's share, so the count is set before the resolve."""
    if hasattr(backend, "resolve_render_target_frames"):   # the product: explicit count, the context's globals stay untouched
        backend.resolve_render_target_frames(int(total_frames))
        return
    g = backend.get_globals()
    own = g.accumulator_count
    g.accumulator_count = int(total_frames)
    backend.set_globals(g)
    backend.resolve_render_target()
    g.accumulator_count = own          # later dispatches of this rank go on from its own count (raytracing.hlsl:155-161)
    backend.set_globals(g)


def device_accumulator(rt, device):
    """zero-copy torch view of the product's accumulator (plumbing for the collective only)"""
    import torch

    class _Buf(object):
        pass
    ptr, nbytes = rt.accumulator_device_ptr()
    b = _Buf()
    b.__cuda_array_interface__ = {"shape": (nbytes // 4,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(b, device=device)


def gather_tiles(rt, dist, rank, world, device, dst=0, via_host=False):
    """Tile split: the one exchange step as a gather of the tiles each rank OWNS (1 / world of the frame per rank) instead
    of a reduce(sum) of whole, mostly untouched frames.  Over xGMI every rank sends its share straight to
    `dst` on its own link: 7/8 of one frame arrives at dst in all, against the 2 (N-1)/N frames a ring reduce moves
    through every link.  Pixels are copied, not added: bit-identical to the sum of zero-filled frames.
    via_host: stage through host memory (gloo rehearsal on one GPU: gloo has no device gather)."""
    import torch
    nbytes = rt.tile_pack_bytes(world)
    mine = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
    rt.pack_tiles(rank, world, mine.data_ptr())
    rt.sync()                                   # the pack ran on the library's stream; the collective runs on torch's
    if via_host:
        parts = [torch.empty(nbytes // 4, dtype=torch.float32) for _ in range(world)] if rank == dst else None
        dist.gather(mine.cpu(), parts, dst=dst)
        parts = [p.to(device) for p in parts] if rank == dst else None
    else:
        parts = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
        dist.gather(mine, parts, dst=dst)
        torch.cuda.synchronize()
    if rank == dst:
        for r in range(world):
            if r != rank:
                rt.unpack_tiles(r, world, parts[r].data_ptr())
        rt.sync()                               # `parts` must outlive the copies
    return nbytes


def reduce_accumulator(tensor, dist, dst=0, all_ranks=False):
    """the one exchange step: sum of the float accumulation buffers (RCCL over xGMI / gloo)"""
    if dist is None or dist.get_world_size() == 1:
        return tensor
    if all_ranks:
        dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    else:
        dist.reduce(tensor, dst=dst, op=dist.ReduceOp.SUM)
    return tensor


def exchange_point_blocks(points, n_points, block, rank, world, dist, stage_mine=None, stage_all=None):
    """The exchange step of the partitioned sample-point pass on a flat float32 tensor `points` of n_points x 6 values
    (RtSamplePoint: position, payload): rank r holds the fresh payloads of block r = points [r * block, min((r + 1) * block, n))
    and receives every other block.  Equal-sized, zero-padded blocks through one all_gather (RCCL / gloo); runs on whatever
    device and stream the tensors live on.  stage_*: reusable staging tensors (block * 6 and world * block * 6 values)."""
    import torch
    if stage_mine is None:
        stage_mine = torch.zeros(block * 6, dtype=torch.float32, device=points.device)
    if stage_all is None:
        stage_all = torch.empty(world * block * 6, dtype=torch.float32, device=points.device)
    lo = min(rank * block, n_points); hi = min(lo + block, n_points)
    if hi > lo:
        stage_mine[:(hi - lo) * 6].copy_(points[lo * 6:hi * 6])
    if dist is not None:   # (a communicator of one rank still makes the call: bench.py --with-collectives rehearses it on a one-GPU box)
        dist.all_gather_into_tensor(stage_all, stage_mine)
    else:
        stage_all[:block * 6].copy_(stage_mine)
    points[:n_points * 6].copy_(stage_all[:n_points * 6])   # (blocks are contiguous: block r starts at r * block)
    return points


def install_probe_exchange(rt, dist, rank, world, device, via_host=False):
    """Partitions the sample-point pass of `rt` over the ranks (rt_set_probe_split) and installs the all-gather of the payload
    blocks behind every pass (rt_set_probe_exchange, include/rt_abi.h).  The library calls back with the HIP stream the pass was
    enqueued on; the copies and the collective are enqueued on that stream (torch runs them on it as an external stream: RCCL's
    own stream waits for it and it waits for RCCL), so nothing blocks the host and the exchange overlaps the frame's camera pass.
    via_host: one-GPU rehearsal with gloo -- the stream is drained and the blocks go through host memory."""
    import torch
    stage = {}

    def view(ptr, n):
        class _Buf(object):
            pass
        b = _Buf()
        b.__cuda_array_interface__ = {"shape": (n * 6,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
        return torch.as_tensor(b, device=device)

    def exchange(stream_ptr, points_ptr, n_points, block):
        if n_points == 0:
            return 0
        ext = torch.cuda.ExternalStream(stream_ptr, device=device)
        with torch.cuda.stream(ext):
            pts = view(points_ptr, n_points)
            if via_host:
                ext.synchronize()
                host_pts = pts.cpu()
                exchange_point_blocks(host_pts, n_points, block, rank, world, dist)
                pts.copy_(host_pts)
                ext.synchronize()
            else:
                key = (n_points, block)
                if key not in stage:   # (allocated once per point set: the tensors are reused by every frame, on this one stream)
                    stage[key] = (torch.zeros(block * 6, dtype=torch.float32, device=device),
                                  torch.empty(world * block * 6, dtype=torch.float32, device=device))
                exchange_point_blocks(pts, n_points, block, rank, world, dist, *stage[key])
        return 0

    rt.set_probe_split(rank, world)
    rt.set_probe_exchange(exchange)
    return exchange
