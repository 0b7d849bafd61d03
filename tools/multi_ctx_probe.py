#!/usr/bin/env python3
"""Probe: N tile-split contexts on ONE GPU (each renders tiles t % N == r of every frame, on its own streams) against one
context rendering whole frames.  Tells what more independent launch chains in flight are worth for a workload.
With a 4th argument "whole" every context renders whole frames of its own instead (more frames in flight, launches of full size).
usage: python tools/multi_ctx_probe.py <workload> <contexts> [frames] [whole]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import bench
from raytracer_amd import api, distributed

workload, n_ctx = sys.argv[1], int(sys.argv[2])
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 20
whole = len(sys.argv) > 4 and sys.argv[4] == 'whole'
args = argparse.Namespace(width=1920, height=1080, spp=0, instances=52743, sample_radius=0.025)
scene, label = bench.build_scene(workload, args)
ctxs = []
for r in range(n_ctx):
    rt = api.Raytracer(0); scene.setup(rt)
    if n_ctx > 1 and not whole: rt.set_tile_split(r, n_ctx)
    ctxs.append((rt, distributed.FrameChain(scene.globals.frame_rng)))
def run(n, first):
    for f in range(first, first + n):
        for rt, chain in ctxs:
            distributed.render_frame(rt, scene.globals, f, chain)
    for rt, _ in ctxs: rt.sync()
run(2, 0)
for rt, _ in ctxs: rt.reset_stats()
t0 = time.time(); run(frames, 2); dt = time.time() - t0
rays = sum(rt.get_stats()["rays_camera"] + rt.get_stats()["rays_probe"] for rt, _ in ctxs)
print("%s contexts %d%s: %.3f ms per frame, %.0f Mrays/s" % (workload, n_ctx, " (whole frames each)" if whole else "", dt / frames / (n_ctx if whole else 1) * 1e3, rays / dt / 1e6), flush=True)
for rt, _ in ctxs: rt.close()
