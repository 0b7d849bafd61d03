#!/usr/bin/env python3
"""Generates the committed golden fixtures from the CPU oracle (run from the repo root after `make`):

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot run here, so these fixtures pin the *oracle's*
behaviour: tests/test_golden.py checks that the oracle still reproduces them (CPU) and that the HIP path matches them
(GPU).  Contents follow the list in SURVEY.md 8(c): RNG known answers, the config-1 images (hit ids, t, accumulator at
256x256, 1 spp, accumulator_count 0), BSSRDF values at 8 radii, a small blue-noise point set, schlick values."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle  # noqa: E402
from raytracer_amd import scenes, _abi  # noqa: E402


def main():
    kat = {"hash1": {str(x): oracle.hash1(x) for x in list(range(16)) + [0x12345678, 0xFFFFFFFF]},
           "hash3": {"%d,%d,%d" % p: oracle.hash3(*p) for p in [(0, 0, 0), (1, 2, 3), (1919, 1079, 0), (100, 200, 0xDEADBEEF)]},
           "draws": {}}
    for seed in (1, 0x12345678, 0xDEADBEEF, 0xC0A9496A):
        s, d01, d11 = seed, [], []
        for _ in range(8):
            v, s = oracle.random01(s); d01.append(int(np.float32(v).view(np.uint32)))
        s = seed
        for _ in range(8):
            v, s = oracle.random11(s); d11.append(int(np.float32(v).view(np.uint32)))
        kat["draws"][str(seed)] = {"random01_bits": d01, "random11_bits": d11}
    kat["schlick_bits"] = {"%g,%g" % (n, c): int(oracle.schlick(n, c).view(np.uint32)) for n in (1.33, 1.75) for c in (0.0, 0.25, 0.5, 1.0)}
    kat["sincos_bits"] = {repr(p): [int(x.view(np.uint32)) for x in oracle.sincos(p)] for p in (0.0, 0.5, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0, 6.25)}
    o = oracle.Oracle()
    s4 = scenes.cornell_bunny(32, 18)
    s4.setup(o)
    radii = [0.0, 0.001, 0.005, 0.02, 0.05, 0.1, 0.2, 0.39]
    kat["bssrdf_tabulated_skin0_scale0.4"] = {"%g" % r: [float(x) for x in o.bssrdf_tabulated(r)] for r in radii}
    g = s4.globals.copy(); g.translucent_bssrdf_scale = 0.0
    o.set_globals(g)
    kat["bssrdf_dipole_default"] = {"%g" % r: [float(x) for x in o.bssrdf_dipole(r)] for r in radii}
    o.close()
    json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=1, sort_keys=True)

    # config 1: debug_cube 256x256, 1 spp, bounces 1, accumulator_count 0
    o = oracle.Oracle(brute_force=True, threads=8)
    s1 = scenes.debug_cube(256, 256, spp=1, bounces=1)
    s1.setup(o, _abi.RT_FLAG_CAPTURE_HITS)
    o.dispatch_rays()
    out = {"accumulator": o.read_accumulator(), "render_target": o.read_render_target()}
    for b in range(2):
        h = o.read_hit_records(b)
        out["b%d_instance" % b] = h["instance"].astype(np.int8)
        out["b%d_geometry" % b] = h["geometry"].astype(np.int8)
        out["b%d_primitive" % b] = h["primitive"].astype(np.int16)
        out["b%d_t" % b] = h["t"]
        out["b%d_valid" % b] = h["valid"].astype(np.uint8)
        out["b%d_rng" % b] = h["rng_after"]
    np.savez_compressed(os.path.join(HERE, "c1_debug_cube_256.npz"), **out)
    o.close()

    # a small blue-noise set: the debug cube as a translucent mesh
    o = oracle.Oracle()
    sb = scenes.debug_cube(32, 32)
    sb.blas[0][0]["shader"] = _abi.RT_SHADER_TRANSLUCENT
    sb.sample_radius = 0.09
    n = sb.setup(o)
    pts, nrm = o.read_sample_points(0)
    np.savez_compressed(os.path.join(HERE, "bluenoise_cube_r0.09.npz"), position=pts["position"], normal=nrm)
    print("bluenoise points:", n)
    o.close()


if __name__ == "__main__":
    main()
