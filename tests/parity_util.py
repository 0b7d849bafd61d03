"""Helpers shared by the GPU parity tests: run one scene description on the HIP product and on the CPU oracle and
compare hit records (bit-exact ids, exact t/u/v), RNG streams, accumulators and render targets."""
import os
import numpy as np

from raytracer_amd import _abi


def compare_hits(gpu_hits, cpu_hits, bounce, t_exact=True):
    assert np.array_equal(gpu_hits["valid"], cpu_hits["valid"]), "bounce %d: alive paths differ (%d pixels)" % (
        bounce, int((gpu_hits["valid"] != cpu_hits["valid"]).sum()))
    live = cpu_hits["valid"] == 1
    for field in ("instance", "geometry", "primitive"):
        bad = live & (gpu_hits[field] != cpu_hits[field])
        assert not bad.any(), "bounce %d: %s differs at %d of %d rays, first %s gpu=%s cpu=%s" % (
            bounce, field, int(bad.sum()), int(live.sum()), np.argwhere(bad)[0], gpu_hits[bad][0], cpu_hits[bad][0])
    hit = live & (cpu_hits["instance"] >= 0)
    for field in ("t", "u", "v"):
        a, b = gpu_hits[field][hit], cpu_hits[field][hit]
        if t_exact:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), "bounce %d: %s not bit-exact (max rel %g)" % (
                bounce, field, float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-30))))
        else:
            assert np.allclose(a, b, rtol=1e-4, atol=0)   # north_star: max relative error < 1e-4 per ray
    assert np.array_equal(gpu_hits["rng_after"][live], cpu_hits["rng_after"][live]), "bounce %d: RNG streams differ" % bounce
    return int(live.sum())


def compare_images(gpu_acc, cpu_acc, rel=1e-4, rmse=1e-3):
    """north_star tolerances: radiance max relative error < 1e-4, image RMSE < 1e-3."""
    assert gpu_acc.shape == cpu_acc.shape
    assert np.array_equal(np.isnan(gpu_acc), np.isnan(cpu_acc))
    g, c = np.nan_to_num(gpu_acc.astype(np.float64)), np.nan_to_num(cpu_acc.astype(np.float64))
    err = np.abs(g - c)
    tol = rel * np.maximum(np.abs(c), 1e-6)
    if os.environ.get("RT_PARITY_REPORT"):  # margin report: worst relative error of every comparison, one per line
        with open(os.environ["RT_PARITY_REPORT"], "a") as f:
            f.write("%s max_rel %.3e\n" % (os.environ.get("PYTEST_CURRENT_TEST", "?"), float((err / np.maximum(np.abs(c), 1e-6)).max())))
    assert (err <= tol).all(), "accumulator max rel err %g at %s" % (float((err / np.maximum(np.abs(c), 1e-6)).max()),
                                                                     np.argwhere(err > tol)[0])
    assert float(np.sqrt(np.mean((g - c) ** 2))) < rmse


def run_pair(scene, gpu, cpu, frames=1, capture=True, reseed=True):
    """Applies `scene` to both backends and dispatches `frames` frames.  Returns per-frame stats."""
    from raytracer_amd import host
    extra = _abi.RT_FLAG_CAPTURE_HITS if capture else 0
    n_gpu = scene.setup(gpu, extra)
    n_cpu = scene.setup(cpu, extra)
    assert n_gpu == n_cpu, "sample point counts differ: gpu %d cpu %d" % (n_gpu, n_cpu)
    g = scene.globals.copy()
    for f in range(frames):
        if reseed:
            g.frame_rng = host.next_frame_rng(g.frame_rng)   # main.cpp:443-454
        for backend in (gpu, cpu):
            cur = backend.get_globals()
            g.accumulator_count = cur.accumulator_count
            g.translucent_accumulator_count = cur.translucent_accumulator_count
            backend.set_globals(g)
            backend.dispatch_rays()
    return n_gpu
