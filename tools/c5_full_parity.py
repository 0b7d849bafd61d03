#!/usr/bin/env python3
"""One-off: C5 at full size (52,743 instances, 50 M instanced triangles) against the oracle, every ray.  Too slow for
the test suite (the CPU side takes minutes); prints progress.  usage: tools/c5_full_parity.py [width height]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import api, scenes, _abi
from oracle import oracle
import parity_util as pu
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
t0 = time.time()
def log(msg): print("[%6.1f s] %s" % (time.time() - t0, msg), flush=True)
scene = scenes.bunny_field(w, h, spp=1, bounces=4)
log("scene: %d instances, %d triangles" % (len(scene.instances), scene.triangle_count()))
gpu = api.Raytracer(0); scene.setup(gpu, _abi.RT_FLAG_CAPTURE_HITS); log("gpu scene built")
cpu = oracle.Oracle(brute_force=False, threads=16); scene.setup(cpu, _abi.RT_FLAG_CAPTURE_HITS); log("oracle scene built")
gpu.dispatch_rays(); gpu.sync(); log("gpu frame done")
cpu.dispatch_rays(); log("oracle frame done")
rays = 0
for b in range(scene.globals.bounces_per_sample + 1):
    rays += pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b); log("bounce %d compared" % b)
pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
log("C5 %dx%d: %d rays bit-exact (ids, t, u, v, rng), accumulator within 1e-4" % (w, h, rays))
