#!/usr/bin/env python3
"""bench.py -- headline benchmark: Mrays/s (primary + secondary rays) of the HIP path tracer on MI355X.

Default workload = BASELINE.json configs[1] (C2): data/bunny.obj on a floor under a luminaire, 1920x1080, 1 spp,
bounces_per_sample = 1 ("primary + shadow" rays).  One step = one Raytracing::dispatch_rays frame
(raytracing.cpp:652-747) with the frame_rng chain of main.cpp:443-454 advancing, so every step traces a different
sample set into the progressive accumulator.  Inputs (scene, BVH, path buffers) are resident in HBM before timing.

N > 1 (launched by torch.distributed.run, one rank per GPU), full scene replica per GPU, two partitions:
  --split sample (default; C5)  rank r renders frames r, r+N, ... of the chain into a private accumulator; per-GPU work
                                is fixed => "scaling": "weak".
  --split tile   (default for --workload c4)  every rank renders every frame, but only the 32x32 tiles t with
                                t % N == r (rt_set_tile_split); total work is fixed => "scaling": "strong".
Either way no collective runs during the K steps; one exchange closes the job -- RCCL reduce(sum) of the RGBA32F
accumulators to rank 0 (sample split) or a gather of the tiles each rank owns (tile split) -- and is part of the timed job
(`reduce_ms`).  When a block of K steps is shorter than 50 ms, R blocks are timed and the median reported (`timed_blocks`);
the job is then R x K frames with ONE exchange, and a block carries reduce_ms / R (`exchange.ms_charged_per_block`).

The JSON line also carries
  roofline:     the dominant kernel (k_traverse).  What bounds it is vector-instruction issue, not HBM (the scenes are
                cache resident): `achieved` = wave64 vector instructions per launch (SQ_INSTS_VALU of the committed PMC
                pass of this command, profiles/pmc_traffic.json) / mean launch time (live HIP events on the library's
                streams); `peak` = 1,024 SIMDs x 2.4 GHz / the cycles a SIMD needs per instruction at best
                (tools/valu_issue_bench.hip -> profiles/r02_valu_issue.txt: v_fma_f32 / v_mul_f32 with 8 waves per SIMD),
                so frac <= 1 by construction; `frame_issue_frac_at_mix_cost` prices the vector instructions of ALL kernels of
                a frame at the measured cost of the node step's instruction blend against the frame's wall time (c2 only).  `traffic` / `traffic_frac` are the HBM bytes per launch of the same PMC
                passes against 8 TB/s; `model` keeps SURVEY.md 8(d)'s bytes-per-ray figure (not an HBM ceiling).
  cpu_baseline: the CPU oracle (oracle/, test infrastructure) timed on this host on a bounded sample of the same
                workload; it also supplies the per-ray node/triangle/instance counts of the bytes-per-ray model.

`preroll_steps`: untimed frames rendered before the W warm-up steps (at least 50 ms of them) so that short runs start at steady clocks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(1, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402  (source_hash: which kernel sources a committed PMC entry belongs to)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s
SIMDS = 1024           # 256 CUs x 4 SIMDs
CLOCK_GHZ = 2.4        # MI355X_MICROARCH.md: max clock (the chip holds 2.1-2.4 GHz under these loads: frac is conservative)


def valu_issue_cycles():
    """Cycles one SIMD needs per wave64 vector instruction, measured by tools/valu_issue_bench.hip on MI355X and committed
    as profiles/r02_valu_issue.txt (column `wall` at 8 waves per SIMD): `best` = the cheapest class (v_mul_f32 /
    v_fma_f32), `mix` = the select / fma_mix / max / fma blend of k_traverse's node step."""
    best, mix = None, None
    try:
        for line in open(os.path.join(ROOT, "profiles", "r02_valu_issue.txt")):
            f = line.split()
            if len(f) == 6 and f[1] == "8":
                if f[0] in ("v_fma_f32", "v_mul_f32"):
                    best = float(f[4]) if best is None else min(best, float(f[4]))
                if f[0].startswith("mix("):
                    mix = float(f[4])
    except (OSError, ValueError):
        pass
    return best, mix


class _DevBuf(object):
    """zero-copy view of a device pointer for torch (plumbing for torch.distributed only)"""
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 3,
                                         "strides": None}


# kernels of the scene set-up (BVH builds, blue-noise points): in the PMC passes, not in a frame
_SETUP_KERNELS = ("k_sah", "k_ploc", "k_morton", "k_sort_hist", "k_sort_scan", "k_sort_scatter", "k_pack_nodes", "k_flat_", "k_emit_single",
                  "k_instance_boxes", "k_scene_box", "k_triangle_boxes", "k_trirecs", "k_invert_perm", "k_bn_")


def pmc_frame_insts(ks):
    """Vector instructions (wave64) of ONE pipelined frame from the per-kernel, per-launch means of a PMC entry: every render
    kernel's mean x its launches, over the frames of the pass (= launches of k_frame_resolve, one per pipelined frame).  The
    pass also holds bench.py's one counting frame: its k_traverse<true, ...> launches are left out, and so is the shade launch
    that follows each of them (k_shade<.., 0>; unfused when the frame gathers, which its k_camera_resolve shows) and its share
    of the once-per-frame kernels.  None when the entry has no pipelined frames."""
    frames = ks.get("k_frame_resolve", {}).get("launches_sampled", 0)
    if not frames:
        return None
    count_q = sum(v["launches_sampled"] for n, v in ks.items() if n.startswith("k_traverse<true, 1, false"))
    count_p = sum(v["launches_sampled"] for n, v in ks.items() if n.startswith("k_traverse<true, 1, true"))
    flag = "false" if "k_camera_resolve" in ks else "true"
    total = 0.0
    for name, v in ks.items():
        if name.startswith(_SETUP_KERNELS) or name.startswith("k_traverse<true,"):
            continue
        if "<" not in name and any(m.startswith(name + "<") for m in ks):
            continue    # the sum over a kernel's instantiations: those are listed by themselves
        n = float(v["launches_sampled"])
        if count_p:
            if name == "k_shade<false, %s, 0>" % flag: n -= count_q
            elif name == "k_shade<true, %s, 0>" % flag: n -= count_p
            elif name == "k_camera_resolve": n = 0.0
            elif name.startswith(("k_gather", "k_probe", "k_sort_payload")): n -= n / (frames + 1.0)
        if n > 0 and v.get("SQ_INSTS_VALU") is not None:
            total += v["SQ_INSTS_VALU"] * n
    return total / frames


def pmc_key(args):
    """entry of profiles/pmc_traffic.json for this command: the workload, with the size unless it is 1080p"""
    return args.workload if (args.width, args.height) == (1920, 1080) else "%s_%dx%d" % (args.workload, args.width, args.height)


def build_scene(name, args):
    from raytracer_amd import scenes
    if name == "c2":
        return scenes.bunny(args.width, args.height, spp=1, bounces=1), "c2: bunny.obj 1920x1080 1spp bounces=1"
    if name == "c3":
        return scenes.cornell(args.width, args.height, spp=args.spp or 64, bounces=4), "c3: cornell %dspp bounces=4" % (args.spp or 64)
    if name == "c3t":
        return scenes.cornell(args.width, args.height, spp=args.spp or 1, bounces=4, translucent_boxes=True), "c3t: cornell as shipped (dipole BSSRDF)"
    if name == "c4":
        return scenes.cornell_bunny(args.width, args.height, spp=args.spp or 16, bounces=4, sample_radius=args.sample_radius), "c4: cornell + translucent bunny skin_0 %dspp" % (args.spp or 16)
    if name == "c5":
        return scenes.bunny_field(args.width, args.height, spp=1, bounces=4, count=args.instances), "c5: %d instanced bunnies" % args.instances
    raise SystemExit("unknown workload " + name)


def cpu_baseline(scene, budget_s, threads):
    """Oracle (CPU BVH, all threads) on whole frames of the same workload until ~budget_s of CPU time is spent.  A frame of
    many samples per pixel (C3: 64, C4: 16) is cut to 4 samples per pixel for this leg -- the samples of a pixel are
    statistically alike, Mrays/s is a per-ray figure -- and the sample says so."""
    from oracle import oracle
    from raytracer_amd import host
    cpu = oracle.Oracle(brute_force=False, threads=threads)
    scene.setup(cpu)
    g = scene.globals.copy()
    spp_full = g.samples_per_pixel
    g.samples_per_pixel = min(spp_full, 4)
    frames, t0 = 0, time.time()
    while True:
        g.frame_rng = host.next_frame_rng(g.frame_rng)
        cur = cpu.get_globals()
        g.accumulator_count = cur.accumulator_count
        g.translucent_accumulator_count = cur.translucent_accumulator_count
        cpu.set_globals(g)
        cpu.dispatch_rays()
        frames += 1
        if time.time() - t0 > budget_s:
            break
    dt = time.time() - t0
    c = cpu.counters()
    cpu.close()
    rays = max(c["rays"], 1)
    return {"value": c["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": "%d full frame(s) of the same workload%s (%d rays, %.1f s)" % (
                frames, "" if g.samples_per_pixel == spp_full else " at %d of its %d samples per pixel" % (g.samples_per_pixel, spp_full), c["rays"], dt)}, \
           {"n_node": c["nodes"] / rays, "n_tri": c["tris"] / rays, "n_inst": c["insts"] / rays}


def out_frames_in_flight(args):
    """frames the library overlaps on one GPU for this run (its default is 2; the hierarchical gather runs one at a time)"""
    return 1 if args.hg is not None else max(1, min(2, int(os.environ.get("RT_FRAMES_IN_FLIGHT", "2"))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--instances", type=int, default=52743)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-alone", action="store_true",
                    help="skip the one-frame-at-a-time leg (roofline.one_frame_in_flight), e.g. under rocprofv3, so that the "
                         "kernel statistics hold only launches of the timed regime")
    ap.add_argument("--split", choices=["sample", "tile"], default=None,
                    help="N > 1: sample-set split (frames r, r+N, ... per rank; default) or tile split (32x32 tiles t % N == r "
                         "of every frame; default for --workload c4)")
    ap.add_argument("--probe-split", action="store_true",
                    help="tile split: partition the sample-point pass over the ranks too and all-gather the payload blocks after every pass "
                         "(distributed.install_probe_exchange; SURVEY 8e-i).  Default: every rank runs the whole pass, no exchange inside a frame")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a one-GPU box: every rank uses GPU 0 and the collectives go through gloo (checks the "
                         "multi-rank code path; the numbers mean nothing)")
    ap.add_argument("--with-collectives", action="store_true",
                    help="N = 1: initialise the process group (RCCL unless --rehearse) and run the exchange step anyway, as a "
                         "one-rank job: exercises the backend calls of the N > 1 path on a one-GPU box (needs the torch.distributed "
                         "environment: launch through torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--dump-accumulator", default=None, metavar="FILE.npy",
                    help="rank 0 saves the float accumulator as it stands when the timed region ends (after the reduce): "
                         "lets a test compare the multi-rank image with the one-process image")
    ap.add_argument("--sample-radius", type=float, default=0.025, help="blue-noise rejection radius of c4 (main.cpp:50)")
    ap.add_argument("--hg", type=float, default=None, metavar="STERADIANS",
                    help="hierarchical BSSRDF gather with this solid-angle limit (rt_set_gather_mode; not reference behaviour)")
    args = ap.parse_args()
    if args.split is None:
        args.split = "tile" if args.workload == "c4" else "sample"

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or args.with_collectives:
        import torch.distributed as dist
        if args.rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from raytracer_amd import api, host, _abi, distributed
    scene, label = build_scene(args.workload, args)
    if args.hg is not None:
        scene.gather_mode = (True, args.hg)
        label += " [hierarchical gather, limit %g sr]" % args.hg
    rt = api.Raytracer(local_rank)
    points = scene.setup(rt)
    tile = dist is not None and args.split == "tile"
    if tile:
        rt.set_tile_split(rank, world)
        if args.probe_split and points:   # the sample-point pass in blocks, the payloads all-gathered behind every pass
            distributed.install_probe_exchange(rt, dist, rank, world, torch.device("cuda", local_rank), via_host=args.rehearse)
    elif dist is not None and points:
        distributed.require_no_sample_points(rt)   # payloads accumulate frame by frame: such scenes shard by tiles
    acc_ptr, acc_bytes = rt.accumulator_device_ptr()
    acc = torch.as_tensor(_DevBuf(acc_ptr, acc_bytes // 4), device=torch.device("cuda", local_rank))
    del acc_ptr

    chain = distributed.FrameChain(scene.globals.frame_rng)
    step_index = [0]

    def step(timed):
        # sample-set split: this rank owns frames rank, rank + world, ... of the frame_rng chain;
        # tile split: every rank renders every frame (its own tiles of it)
        distributed.render_frame(rt, scene.globals, step_index[0] if tile else step_index[0] * world + rank, chain)
        step_index[0] += 1

    # steady clocks first: a GPU that has just built the scene is still at idle clocks, and `--warmup 5` frames of C2 are 0.7 ms
    # (measured: 20 timed steps read 7 % low after 5 warm-up steps, 1 % low after 100).  At least 50 ms of untimed frames
    # go before the W warm-up steps; their number is reported as `preroll_steps`.
    preroll_steps, t_pre = 0, time.perf_counter()
    while not args.dump_accumulator and time.perf_counter() - t_pre < 0.05:  # (a dumped image is the W + K frames alone)
        for _ in range(8):
            step(False)
        rt.sync()
        preroll_steps += 8
    for _ in range(args.warmup):
        step(False)
    rt.sync()
    if dist is not None:
        # the first collective builds the communicator (RCCL rings over xGMI): untimed, like the other warm-up work
        scratch = torch.zeros_like(acc)
        if args.rehearse:
            dist.all_reduce(scratch, op=dist.ReduceOp.SUM)
        else:
            dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        del scratch
    # every 8th frame of the timed region has its traverse / gather launches bracketed by HIP events on the library's
    # stream (bracketing all of them costs 13 us of a 0.28 ms frame in event packets)
    # (a short run times every frame, so that it has launches to report at all)
    timing_flags = _abi.RT_FLAG_TIME_KERNELS | (_abi.RT_FLAG_TIME_SAMPLED if args.steps >= 16 else 0)
    rt.set_flags(scene.flags | timing_flags)
    rt.reset_stats()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        rt.sync()

    # The contract times exactly K steps between two fences.  K steps of C2 are 2.6 ms at the driver's K = 20: one stray
    # clock dip moves such a window by several per cent, so when a block of K steps is shorter than 50 ms, R blocks of K
    # steps are timed one after the other, each between its own fences, and the MEDIAN block is reported
    # (`timed_blocks` = R, `block_ms` = all of them; `steps` stays K).
    def timed_block():
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        rt.sync()
        t1 = time.perf_counter()   # this rank's K steps are done; the closing fence's own latency (a barrier over N ranks) is not a step
        fence()
        d = torch.tensor([t1 - t0], dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(d, op=dist.ReduceOp.MAX)     # the slowest rank's block (outside the timed window)
        return float(d.item())

    block_s = [timed_block()]
    if not args.dump_accumulator and block_s[0] < 0.05:
        for _ in range(min(int(0.05 / max(block_s[0], 1e-6)), 63)):
            block_s.append(timed_block())
    elif not args.dump_accumulator and block_s[0] < 1.0:
        # longer blocks of many-launch frames (C3: 450 launches per frame and lane) still spread by ~3 % from block to block with how
        # the two lanes' launches happen to interleave: the median of three
        block_s += [timed_block(), timed_block()]
    dt_block = sorted(block_s)[len(block_s) // 2] if len(block_s) > 1 else block_s[0]
    # the one exchange step of the job, after its last frame: timed between fences of its own and ADDED to the block
    fence()
    t_x = time.perf_counter()
    exchange_bytes = 0
    if dist is not None:
        if tile:
            # tile split: every rank sends the tiles it owns (1 / N of the frame) straight to rank 0
            exchange_bytes = distributed.gather_tiles(rt, dist, rank, world, torch.device("cuda", local_rank), dst=0, via_host=args.rehearse)
        elif args.rehearse:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM)  # gloo has no GPU reduce
            exchange_bytes = acc.numel() * 4
        else:
            dist.reduce(acc, dst=0, op=dist.ReduceOp.SUM)   # RCCL over xGMI: sum of the float accumulation buffers
            exchange_bytes = acc.numel() * 4
    fence()
    reduce_ms = (time.perf_counter() - t_x) * 1e3 if dist is not None else 0.0   # this rank's wait for the slowest rank included
    # The job that ran is len(block_s) blocks of K frames and ONE exchange after its last frame; a block's share of it is
    # reduce_ms / blocks (with one block -- every run of 50 ms or more per block -- the whole exchange is in `dt`)
    dt = dt_block + reduce_ms * 1e-3 / len(block_s)
    if args.dump_accumulator and rank == 0:
        import numpy as np
        np.save(args.dump_accumulator, acc.cpu().numpy().reshape(args.height, args.width, 4))

    stats = rt.get_stats()
    n_blocks = len(block_s)
    rays = (stats["rays_camera"] + stats["rays_probe"]) / n_blocks         # per block of K steps (the blocks are statistically alike)
    # the tile split replicates the (small) sample-point pass on every rank: those rays are counted once
    # (with --probe-split the pass is partitioned: every rank counts its own block's rays)
    probe_blocks = tile and args.probe_split and points
    job_rays = (stats["rays_camera"] + (stats["rays_probe"] if (not tile or rank == 0 or probe_blocks) else 0)) / n_blocks
    t_rays = torch.tensor([float(job_rays), dt], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = t_rays.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t_rays, op=dist.ReduceOp.SUM)
        total_rays, dt_max = float(t_rays[0].item()), float(tmax[1].item())
    else:
        total_rays, dt_max = float(job_rays), dt

    if probe_blocks:   # what follows is rank 0's own business: no collective may hide in it
        rt.set_probe_exchange(None)
        rt.set_probe_split(0, 1)
    if rank == 0:
        # traversal counters of this GPU's own BVH (one extra untimed frame with the counting kernel variant)
        rt.set_flags(scene.flags | _abi.RT_FLAG_COUNT_TRAVERSAL)
        rt.reset_stats()
        step(False)
        cs = rt.get_stats()
        crays = max(cs["rays_camera"] + cs["rays_probe"], 1)
        gpu_counts = {"n_node": cs["nodes_visited"] / crays, "n_tri": cs["tris_tested"] / crays, "n_inst": cs["instances_entered"] / crays}

        # the same workload one frame at a time (a second context created with RT_FRAMES_IN_FLIGHT=1): the duration of
        # a traversal launch that has the GPU to itself, next to the overlapped figure of the timed region above
        alone = None
        if world == 1 and out_frames_in_flight(args) > 1 and not args.no_alone:
            saved = os.environ.get("RT_FRAMES_IN_FLIGHT")
            os.environ["RT_FRAMES_IN_FLIGHT"] = "1"
            rt1 = api.Raytracer(local_rank)
            if saved is None: del os.environ["RT_FRAMES_IN_FLIGHT"]
            else: os.environ["RT_FRAMES_IN_FLIGHT"] = saved
            try:
                scene.setup(rt1)
                n1 = max(8, min(args.steps, 48))
                chain1 = distributed.FrameChain(scene.globals.frame_rng)
                for k in range(2): distributed.render_frame(rt1, scene.globals, k, chain1)
                rt1.set_flags(scene.flags | _abi.RT_FLAG_TIME_KERNELS | (_abi.RT_FLAG_TIME_SAMPLED if n1 >= 16 else 0))
                rt1.sync(); rt1.reset_stats()
                t1 = time.perf_counter()
                for k in range(n1): distributed.render_frame(rt1, scene.globals, 2 + k, chain1)
                rt1.sync()
                dt1 = time.perf_counter() - t1
                s1 = rt1.get_stats()
                alone = {"steps": n1, "Mrays_s": (s1["rays_camera"] + s1["rays_probe"]) / dt1 / 1e6, "ms_per_step": dt1 / n1 * 1e3,
                         "mean_launch_ms": s1["traverse_ms"] / max(s1["traverse_launches_timed"], 1)}
            finally:
                rt1.close()

        cpu_line, model = None, None
        if not args.no_cpu and world == 1:  # the CPU baseline is reported at N = 1 only
            threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
            cpu_line, model = cpu_baseline(scene, args.cpu_seconds, threads)
        if model is None:
            model = gpu_counts
        # B_ray = 200 + 64 N_node + 48 N_tri + 64 N_inst; the traverse kernel's share of the 200 B of
        # stream traffic is the 32 B ray it reads and the 20 B hit it writes.  Kept as `roofline.model`: these are bytes
        # that L1 / L2 serve, not HBM bytes, so they are not a ceiling (round 1 reported 1.18 of the HBM peak with them).
        b_trav = 52.0 + 64.0 * model["n_node"] + 48.0 * model["n_tri"] + 64.0 * model["n_inst"]
        b_ray = 200.0 + 64.0 * model["n_node"] + 48.0 * model["n_tri"] + 64.0 * model["n_inst"]
        launches = max(stats["traverse_launches"], 1)
        timed = max(stats["traverse_launches_timed"], 1)
        mean_launch_ms = stats["traverse_ms"] / timed
        rays_all = rays * n_blocks                       # the stats cover all timed blocks
        rays_per_launch = rays_all / launches
        rays_timed = rays_all * timed / launches  # frames are statistically alike: the timed launches carry their share of the rays
        model_gbs = (b_trav * rays_per_launch) / (mean_launch_ms * 1e-3) / 1e9 if mean_launch_ms > 0 else 0.0
        # queued (scattered / sample-point: incoherent) launches apart from the bounce-0 launches: north_star's
        # ">= 1 Grays/s incoherent secondary rays" is read off `secondary_Mrays_s`
        sec_launches = stats["traverse_launches"] - stats["primary_launches"]
        sec_timed = stats["traverse_launches_timed"] - stats["primary_launches_timed"]
        sec_ms = stats["traverse_ms"] - stats["primary_ms"]
        sec_rays = rays_all - stats["rays_primary"]
        secondary = primary = None
        if sec_timed > 0 and sec_ms > 0 and sec_launches > 0:
            secondary = (sec_rays * sec_timed / sec_launches) / (sec_ms * 1e-3) / 1e6
        if stats["primary_launches_timed"] > 0 and stats["primary_ms"] > 0:
            primary = (stats["rays_primary"] * stats["primary_launches_timed"] / stats["primary_launches"]) / (stats["primary_ms"] * 1e-3) / 1e6
        # per-launch counters from the committed PMC passes of this same command (tools/pmc_passes.sh + tools/pmc_summary.py
        # -> profiles/pmc_traffic.json, one entry per workload and size; HBM bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per the
        # gfx950 note of MI355X_MICROARCH.md).  An entry is used only while the kernel sources are the ones it was taken on
        # (`source_hash`): otherwise the counter-based figures are null and `pmc_note` says why -- never a fraction that mixes
        # another build's instruction count with this build's timing.
        pmc_all, pmc, pmc_note = None, None, None
        key = pmc_key(args)
        # (N > 1: under the sample-set split rank 0 runs the one-GPU frames, so the one-GPU passes describe its launches; the
        # tile split's launches are 1 / N of them, and ranks that rehearse on one GPU share it)
        if world > 1 and (tile or args.rehearse):
            pmc_note = "the committed PMC passes are of one-GPU launches: not applicable to %s" % ("ranks sharing one GPU (--rehearse)" if args.rehearse else "the tile split's 1/N-size launches")
        else:
            try:
                entry = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))[key]
                if entry.get("source_hash") != pmc_summary.source_hash(ROOT):
                    pmc_note = "profiles/pmc_traffic.json[%r] was taken on other kernel sources (%s): rerun tools/pmc_passes.sh" % (key, entry.get("source_hash"))
                else:
                    pmc_all = entry["kernels"]
                    pmc = pmc_all.get("k_traverse")
            except (OSError, KeyError, ValueError):
                pmc_note = "no PMC passes committed for %r" % key
        cyc_best, cyc_mix = valu_issue_cycles()
        traffic = insts = achieved = peak = frac = traffic_frac = mix_frac = valu_busy = None
        # Flat one-instance scenes walk AND shade bounce 0 in one kernel (k_shade<true, true, N>: RtStats.fused_bounce0): the launches
        # of k_traverse are then the queued ones alone, and the roofline's launch time is theirs; the bounce-0 kernel gets a line
        # of its own (`bounce0_kernel`)
        fused0 = bool(stats.get("fused_bounce0"))
        bounce0 = None
        if fused0:
            mean_launch_ms = sec_ms / sec_timed if sec_timed > 0 else 0.0
            rays_per_launch = sec_rays / max(sec_launches, 1)
            model_gbs = (b_trav * rays_per_launch) / (mean_launch_ms * 1e-3) / 1e9 if mean_launch_ms > 0 else 0.0
            k0 = [k for k in (pmc_all or {}) if k.startswith("k_shade<true, true, ") and not k.endswith(", 0>")]
            if k0 and stats["primary_launches_timed"] > 0 and cyc_best:
                k0 = max(k0, key=lambda k: pmc_all[k]["launches_sampled"])
                ms0 = stats["primary_ms"] / stats["primary_launches_timed"]
                bounce0 = {"kernel": k0, "valu_insts_per_launch": pmc_all[k0]["SQ_INSTS_VALU"], "mean_launch_ms": ms0,
                           "achieved": pmc_all[k0]["SQ_INSTS_VALU"] / (ms0 * 1e-3) / 1e9, "unit": "Ginst/s",
                           "frac": pmc_all[k0]["SQ_INSTS_VALU"] / (ms0 * 1e-3) / 1e9 / (SIMDS * CLOCK_GHZ / cyc_best),
                           "traffic": pmc_all[k0].get("traffic_bytes_per_launch")}
        # Flat one-instance scenes also run the queued bounces but the last as ONE kernel each (k_shade<false, true, N>: walk + shade,
        # RT_FUSE_QUEUED): the library times those launches with the traversal launches, so the roofline's counters are the
        # per-launch means over both kinds of launch, weighted by how often the passes saw each
        queued_fused = [k for k in (pmc_all or {}) if k.startswith("k_shade<false, true, ") and not k.endswith(", 0>")]
        if queued_fused and fused0:
            entries = [pmc_all[k] for k in queued_fused] + ([pmc] if pmc else [])
            n_all = sum(e["launches_sampled"] for e in entries)
            merged = {"launches_sampled": n_all}
            for f in set().union(*[set(e) for e in entries]):
                if f != "launches_sampled" and all(isinstance(e.get(f), (int, float)) for e in entries):
                    merged[f] = sum(e[f] * e["launches_sampled"] for e in entries) / n_all
            pmc = merged
        if pmc is not None:
            traffic = pmc.get("traffic_bytes_per_launch")
            insts = pmc.get("SQ_INSTS_VALU")
        if traffic is not None and mean_launch_ms > 0:
            traffic_frac = traffic / (mean_launch_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if insts is not None and cyc_best is not None and mean_launch_ms > 0:
            achieved = insts / (mean_launch_ms * 1e-3) / 1e9            # G wave64 vector instructions per second
            peak = SIMDS * CLOCK_GHZ / cyc_best
            frac = achieved / peak
            mix_frac = achieved * cyc_mix / (SIMDS * CLOCK_GHZ) if cyc_mix else None
            if frac > 1.0:  # cannot be: more instructions per second than the SIMDs can issue
                pmc_note = "roofline.frac %.3f > 1: the PMC pass in profiles/ does not fit this run; counter-based figures dropped" % frac
                achieved = frac = mix_frac = None
            elif pmc.get("SQ_ACTIVE_INST_VALU") and pmc.get("GRBM_GUI_ACTIVE"):
                # the hardware's own tally, from the PMC run itself: quad-cycles in which a SIMD executes a vector
                # instruction over the SIMD quad-cycles of the launch (GRBM_GUI_ACTIVE sums the 8 XCDs)
                valu_busy = 4.0 * pmc["SQ_ACTIVE_INST_VALU"] / (SIMDS * pmc["GRBM_GUI_ACTIVE"] / 8.0)
        # the other kernels of a step, from the same passes: vector instructions and HBM bytes per launch, the launch time
        # under the profiler (one kernel at a time on the GPU) and what that makes of the two ceilings
        other = None
        if pmc_all is not None:
            other = {}
            for name, k in sorted(pmc_all.items()):
                if name.startswith("k_traverse") or "<" in name and name.split("<")[0] in pmc_all and not name.startswith(("k_shade", "k_gather")):
                    continue
                ns = k.get("duration_ns")
                e = {"launches_sampled": k.get("launches_sampled"), "valu_insts_per_launch": k.get("SQ_INSTS_VALU"),
                     "traffic_bytes_per_launch": k.get("traffic_bytes_per_launch"), "launch_us_under_pmc": ns / 1e3 if ns else None}
                if ns and k.get("traffic_bytes_per_launch") is not None:
                    e["hbm_frac"] = k["traffic_bytes_per_launch"] / ns / HBM_PEAK_GBS
                if ns and k.get("SQ_INSTS_VALU") is not None and cyc_best:
                    e["valu_issue_frac"] = k["SQ_INSTS_VALU"] / ns / (SIMDS * CLOCK_GHZ / cyc_best)
                other[name] = e
        # the whole frame: vector instructions of ALL its kernels (the same PMC passes) against the issue slots of the frame's wall
        # time, at the measured cost of the node step's blend.  With two frames in flight the kernels overlap, so this -- not the
        # per-launch figure -- says how full the machine is.  (Kernels other than the traversal have other blends -- the gather's is
        # cheaper -- so for them the price is an upper bound.)
        frame_insts = frame_issue_frac = None
        if pmc_all is not None and cyc_mix:
            frame_insts = pmc_frame_insts(pmc_all)
            if frame_insts:
                frame_issue_frac = frame_insts * cyc_mix / (dt_max / args.steps * SIMDS * CLOCK_GHZ * 1e9)
        split_label = ("tile split x%d" if tile else "sample-set split x%d") % world
        out = {
            # BASELINE.json's metric; `value` is the whole job: per GPU at N = 1, the sum over the N GPUs otherwise
            "metric": ("Mrays/s per GPU (primary+secondary) at 1080p" if world == 1 else "Mrays/s over %d GPUs (primary+secondary) at 1080p" % world)
                      if (args.width, args.height) == (1920, 1080) else "Mrays/s (primary+secondary)",
            "value": total_rays / dt_max / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll_steps,
            "timed_blocks": n_blocks, "block_ms": [round(b * 1e3, 4) for b in block_s],   # value: the median block (+ its share of the one exchange)
            "ms_per_step": dt_max / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if tile else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": label, "width": args.width, "height": args.height, "triangles": scene.triangle_count(),
                       "sample_points": points, "rays_per_step_per_gpu": rays / args.steps,
                       "parallelism": (split_label + " + RCCL reduce" if world > 1 else "single GPU") +
                                      (", sample-point pass in blocks + all-gather" if (tile and args.probe_split and points) else ""),
                       # consecutive steps overlap pairwise on the GPU (the library's default, RT_FRAMES_IN_FLIGHT)
                       "frames_in_flight": out_frames_in_flight(args)},
            "primary_Mrays_s": primary,        # bounce-0 launches alone (coherent), from their launch durations
            "secondary_Mrays_s": secondary,    # queued launches alone (scattered / sample-point rays: incoherent)
            "reduce_ms": reduce_ms if dist is not None else None,   # the one exchange step, part of the timed job (rank 0's view)
            "exchange": None if dist is None else {"kind": "gather of owned tiles" if tile else "reduce(sum) of the accumulator",
                                                 "bytes_per_rank": exchange_bytes,
                                                 # one exchange per job of timed_blocks x steps frames: a block carries its share
                                                 "per_job_of_frames": n_blocks * args.steps, "ms_charged_per_block": reduce_ms / n_blocks},
            "roofline": {"bound": "valu_issue", "kernel": "k_traverse", "achieved": achieved, "peak": peak, "unit": "Ginst/s",
                         "frac": frac, "traffic": traffic, "traffic_frac": traffic_frac, "pmc_entry": key, "pmc_note": pmc_note,
                         "valu_insts_per_launch": insts, "cycles_per_inst_best": cyc_best, "cycles_per_inst_node_step_mix": cyc_mix,
                         # the same instruction rate priced at the measured cost of the node step's instruction blend
                         # (selects, v_fma_mix, min / max run at half the fma rate on gfx950): how full the issue slots are
                         "frac_at_mix_cost": mix_frac, "valu_busy_pmc": valu_busy,
                         "frame_valu_insts": frame_insts, "frame_issue_frac_at_mix_cost": frame_issue_frac,
                         "launches_of": ("queued launches: %s and the last bounce's k_traverse (bounce 0 runs in k_shade<true, true, N>: bounce0_kernel)" % ", ".join(sorted(queued_fused)) if queued_fused
                                         else "queued k_traverse launches (bounce 0 runs in k_shade<true, true, N>: bounce0_kernel)") if fused0 else "all k_traverse launches",
                         "bounce0_kernel": bounce0,
                         "mean_launch_ms": mean_launch_ms, "rays_per_launch": rays_per_launch,
                         "traverse_Mrays_s": rays_timed / (stats["traverse_ms"] * 1e-3) / 1e6 if stats["traverse_ms"] > 0 else None,
                         "launches_timed": sec_timed if fused0 else stats["traverse_launches_timed"], "launches": sec_launches if fused0 else stats["traverse_launches"],
                         # SURVEY 8(d)'s algorithmic byte model (64-byte nodes, every byte priced as if it came from HBM)
                         "model": {"bytes_per_ray_traverse": b_trav, "bytes_per_ray_pipeline": b_ray,
                                   "algorithmic_bytes_per_launch": b_trav * rays_per_launch, "GBs": model_gbs,
                                   "frac_of_hbm_peak": model_gbs / HBM_PEAK_GBS,
                                   # what this implementation's records actually are (32-byte packed nodes, 48-byte triangles,
                                   # 96-byte instance records) times its own trees' counts
                                   "bytes_per_ray_traverse_this_layout": 52.0 + 32.0 * gpu_counts["n_node"] + 48.0 * gpu_counts["n_tri"] + 96.0 * gpu_counts["n_inst"],
                                   "model_counts": model, "gpu_bvh_counts": gpu_counts},
                         # launches of the timed region share the GPU with the other frame in flight; this is the same
                         # kernel with the GPU to itself (frames one at a time), and what the job then runs at
                         "one_frame_in_flight": None if alone is None else dict(
                             alone, achieved=None if insts is None else insts / (alone["mean_launch_ms"] * 1e-3) / 1e9,
                             frac=None if (insts is None or peak is None) else insts / (alone["mean_launch_ms"] * 1e-3) / 1e9 / peak)},
            "cpu_baseline": cpu_line,
            "other_kernels": other,
        }
        # acceleration-structure builds are outside the timed loop: reported separately
        out["build"] = {"last_blas_ms": stats["blas_build_ms"], "blas_triangles_total": stats["blas_triangles"],
                        "tlas_ms": stats["tlas_build_ms"], "tlas_instances": stats["tlas_instances"],
                        "tlas_depth": stats["tlas_depth"], "max_blas_depth": stats["max_blas_depth"], "stack_in_lds": stats["stack_in_lds"], "flat_depth": stats["flat_depth"],
                        "tlas_Minstances_per_s": stats["tlas_instances"] / (stats["tlas_build_ms"] * 1e-3) / 1e6 if stats["tlas_build_ms"] > 0 else None}
        if points:
            # blue-noise sample-point generation (rt_generate_translucent_samples blocks: host wall time), at the scene's
            # radius and at a finer one (the reference's slow, unprofiled step, todo.txt:3-4)
            out["build"]["bluenoise_ms"] = stats["bluenoise_ms"]
            out["build"]["bluenoise_points"] = stats["bluenoise_points"]
            out["build"]["bluenoise_radius"] = scene.sample_radius
            if world == 1:
                n_fine = rt.generate_translucent_samples(0.01, scene.rng_seed, scene.phase_seed)
                fine = rt.get_stats()
                out["build"]["bluenoise_fine"] = {"radius": 0.01, "ms": fine["bluenoise_ms"], "points": n_fine}
        if stats["gather_launches"] and stats["gather_ms"] > 0:
            # second hot loop: (hit, sample point) pairs of translucent_chit; 24 B per pair from LDS
            gtimed = max(stats["gather_launches_timed"], 1)
            pairs_s = stats["gather_pairs"] * gtimed / stats["gather_launches"] / (stats["gather_ms"] * 1e-3)
            out["gather"] = {"kernel": "k_gather", "pairs_per_s": pairs_s, "lds_GB_per_s": 24.0 * pairs_s / 1e9,
                             "mean_launch_ms": stats["gather_ms"] / gtimed,
                             "share_of_step": stats["gather_ms"] / gtimed * stats["gather_launches"] / (sum(block_s) * 1e3),
                             # 1 with the batched gather (one launch per frame over the items of all its gathering shade launches, with its
                             # numbering and apply kernels inside the timed bracket), else one per gathering bounce (RT_GATHER_BATCH)
                             "launches_per_step": stats["gather_launches"] / max(len(block_s) * args.steps, 1)}
            gp = (pmc_all or {}).get("k_gather")
            if gp and gp.get("SQ_INSTS_VALU") and cyc_best:
                # second roofline: the gather is bound by vector-instruction issue too (sqrt / filter arithmetic per pair)
                g_ach = gp["SQ_INSTS_VALU"] / (out["gather"]["mean_launch_ms"] * 1e-3) / 1e9
                out["gather"]["roofline"] = {"bound": "valu_issue", "achieved": g_ach, "peak": SIMDS * CLOCK_GHZ / cyc_best, "unit": "Ginst/s",
                                             "frac": g_ach / (SIMDS * CLOCK_GHZ / cyc_best), "valu_insts_per_launch": gp["SQ_INSTS_VALU"],
                                             "traffic": gp.get("traffic_bytes_per_launch"),
                                             "lds_bank_conflict_share": (gp["SQ_LDS_BANK_CONFLICT"] / gp["SQ_LDS_ACTIVE"]) if gp.get("SQ_LDS_ACTIVE") else None}
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    rt.close()


if __name__ == "__main__":
    main()
