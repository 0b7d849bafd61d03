#!/usr/bin/env python3
"""Times rt_generate_translucent_samples (the reference's slow, unprofiled step, todo.txt:3-4) on
the C4 scene for a few radii; run under `rocprofv3 --kernel-trace --stats` for the per-kernel breakdown."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracer_amd import api, scenes  # noqa: E402


def main():
    radii = [float(x) for x in sys.argv[1:]] or [0.025, 0.01, 0.005]
    rt = api.Raytracer(0)
    scene = scenes.cornell_bunny(256, 144)
    scene.sample_radius = None
    scene.setup(rt)
    out = []
    for r in radii:
        best = None
        for rep in range(3):
            t0 = time.perf_counter()
            n = rt.generate_translucent_samples(r, 1, 2)
            dt = (time.perf_counter() - t0) * 1e3
            best = dt if best is None else min(best, dt)
        out.append({"radius": r, "points": n, "ms_best_of_3": best, "ms_library": rt.get_stats()["bluenoise_ms"]})
    print(json.dumps(out))
    rt.close()


if __name__ == "__main__":
    main()
