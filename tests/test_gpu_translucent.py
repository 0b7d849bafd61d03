"""GPU parity of the BSSRDF path: blue-noise sample points (bit-exact vs the oracle), probe-ray irradiance
collection (translucent_rgen) and the hits x points gather of translucent_chit."""
import numpy as np
import pytest

from raytracer_amd import scenes, _abi
import parity_util as pu

pytestmark = pytest.mark.gpu


def _pair(oracle_mod):
    from raytracer_amd import api
    return api.Raytracer(0), oracle_mod.Oracle(brute_force=True, threads=8)


def _compare_points(gpu, cpu, payload_rel=None):
    n = gpu.translucent_instance_count()
    assert n == cpu.translucent_instance_count() and n > 0
    total = 0
    for i in range(n):
        gp, gn = gpu.read_sample_points(i)
        cp, cn = cpu.read_sample_points(i)
        assert len(gp) == len(cp) and len(gp) > 0
        assert np.array_equal(gp["position"].view(np.uint32), cp["position"].view(np.uint32)), "sample positions differ"
        assert np.array_equal(gn.view(np.uint32), cn.view(np.uint32)), "sample normals differ"
        if payload_rel is not None:
            a, b = gp["payload"].astype(np.float64), cp["payload"].astype(np.float64)
            assert np.all(np.abs(a - b) <= payload_rel * np.maximum(np.abs(b), 1e-6)), "payload max rel err %g" % (
                np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-6)))
        total += len(gp)
    return total


@pytest.mark.parametrize("radius,flags", [(0.05, 0), (0.025, 0), (0.08, _abi.RT_FLAG_ALL_PHASE_GROUPS)])
def test_bluenoise_points_bit_exact(oracle_mod, radius, flags):
    gpu, cpu = _pair(oracle_mod)
    try:
        s = scenes.cornell_bunny(64, 36, spp=1, bounces=1, sample_radius=radius)
        s.flags |= flags
        n_gpu, n_cpu = s.setup(gpu), s.setup(cpu)
        assert n_gpu == n_cpu > 50
        assert _compare_points(gpu, cpu) == n_gpu
    finally:
        gpu.close(); cpu.close()


def test_bluenoise_two_translucent_boxes(oracle_mod):
    """the scene as shipped (main.cpp:311-340): two translucent geometries in one BLAS -> two point sets"""
    gpu, cpu = _pair(oracle_mod)
    try:
        s = scenes.cornell(64, 36, spp=1, bounces=1, translucent_boxes=True, sample_radius=0.03)
        assert s.setup(gpu) == s.setup(cpu)
        assert gpu.translucent_instance_count() == 2
        _compare_points(gpu, cpu)
    finally:
        gpu.close(); cpu.close()


def _check_translucent(scene, oracle_mod, frames):
    gpu, cpu = _pair(oracle_mod)
    try:
        pu.run_pair(scene, gpu, cpu, frames=frames)
        _compare_points(gpu, cpu, payload_rel=1e-4)
        for b in range(scene.globals.bounces_per_sample + 1):
            pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b)
        pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
        assert gpu.get_stats()["gather_pairs"] > 0
    finally:
        gpu.close(); cpu.close()


def test_c4_tabulated_skin0(oracle_mod):
    _check_translucent(scenes.cornell_bunny(128, 72, spp=2, bounces=3, profile=0, sample_radius=0.04), oracle_mod, frames=2)


def test_c4_tabulated_skin1(oracle_mod):
    _check_translucent(scenes.cornell_bunny(96, 54, spp=1, bounces=2, profile=1, sample_radius=0.05), oracle_mod, frames=2)


def test_c4_dipole(oracle_mod):
    _check_translucent(scenes.cornell_bunny(96, 54, spp=2, bounces=2, tabulated=False, sample_radius=0.05), oracle_mod, frames=2)


def test_c3_as_shipped_dipole_boxes(oracle_mod):
    _check_translucent(scenes.cornell(128, 72, spp=1, bounces=4, translucent_boxes=True, sample_radius=0.03), oracle_mod, frames=3)


def test_double_dispatch_quirk(oracle_mod):
    s = scenes.cornell_bunny(64, 36, spp=1, bounces=2, sample_radius=0.06)
    s.flags |= _abi.RT_FLAG_DOUBLE_DISPATCH
    _check_translucent(s, oracle_mod, frames=2)


def test_sss_disabled_and_collection_disabled(oracle_mod):
    s = scenes.cornell_bunny(64, 36, spp=1, bounces=2, sample_radius=0.06)
    s.flags = _abi.RT_FLAG_COLLECT_SAMPLES      # SSS off: fudge forced to 0 (raytracing.cpp:653-657)
    gpu, cpu = _pair(oracle_mod)
    try:
        pu.run_pair(s, gpu, cpu, frames=2)
        pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
        assert gpu.get_stats()["gather_pairs"] == 0
    finally:
        gpu.close(); cpu.close()


# ---- hierarchical gather (rt_set_gather_mode) --------------------------------
@pytest.mark.parametrize("tabulated,limit", [(True, 0.05), (False, 0.05), (True, 0.5)])
def test_hierarchical_gather_matches_the_oracle(oracle_mod, tabulated, limit):
    """same tree, same walk, same arithmetic on both sides: to the north-star tolerance (in practice bit for bit)"""
    s = scenes.cornell_bunny(96, 54, spp=2, bounces=2, tabulated=tabulated, sample_radius=0.03)
    s.gather_mode = (True, limit)
    _check_translucent(s, oracle_mod, frames=2)


def test_hierarchical_gather_two_instances_double_dispatch(oracle_mod):
    s = scenes.cornell(96, 54, spp=1, bounces=3, translucent_boxes=True, sample_radius=0.02)
    s.gather_mode = (True, 0.1)
    s.flags |= _abi.RT_FLAG_DOUBLE_DISPATCH
    _check_translucent(s, oracle_mod, frames=2)


def test_hierarchical_gather_against_the_exact_sum():
    """limit 0 visits every point (only the order of the float sum differs from the reference loop: 1e-4); a useful
    limit stays close to the exact image and evaluates several times fewer terms"""
    from raytracer_amd import api
    images, terms = {}, {}
    for name, mode in [("exact", None), ("all", (True, 0.0)), ("approx", (True, 0.5))]:
        s = scenes.cornell_bunny(128, 72, spp=2, bounces=2, sample_radius=0.02)
        s.gather_mode = mode
        rt = api.Raytracer(0)
        try:
            s.setup(rt)
            from raytracer_amd import distributed
            for f in range(2):
                distributed.render_frame(rt, s.globals, f, 7)
            images[name] = rt.read_accumulator()
            terms[name] = rt.get_stats()["gather_pairs"]
        finally:
            rt.close()
    pu.compare_images(images["all"], images["exact"])
    assert terms["all"] == terms["exact"]
    err = images["approx"].astype(np.float64) - images["exact"].astype(np.float64)
    rmse = float(np.sqrt(np.mean(err ** 2)))
    print("hierarchical gather, limit 0.5 sr: %.1fx fewer terms, image RMSE %.2e" % (terms["exact"] / terms["approx"], rmse))
    assert rmse < 1e-2
    assert terms["approx"] * 2 < terms["exact"]


def test_dipole_with_per_channel_coefficients(oracle_mod):
    """the shipped defaults share one (sigma_s, sigma_a) across r, g, b (main.cpp:420-422), which the gather kernel
    evaluates once; distinct coefficients take the per-channel path"""
    s = scenes.cornell_bunny(96, 54, spp=1, bounces=2, tabulated=False, sample_radius=0.05)
    s.globals.translucent_scattering[:] = (15.0, 9.0, 4.0)
    s.globals.translucent_absorption[:] = (0.1, 0.3, 0.9)
    _check_translucent(s, oracle_mod, frames=2)
    s.gather_mode = (True, 0.3)
    _check_translucent(s, oracle_mod, frames=1)


# ---- batched gather (RT_GATHER_BATCH; rt_api.hip run_pass) --------------------------------------------
def _render_c4(monkeypatch, env, frames=3, **scene_kw):
    from raytracer_amd import api, distributed
    for k in ("RT_GATHER_BATCH", "RT_GATHER_BATCH_MB", "RT_FRAMES_IN_FLIGHT", "RT_GATHER_GROUPS", "RT_GATHER_BLOCK"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)        # read at rt_create
    s = scenes.cornell_bunny(160, 96, spp=3, bounces=3, sample_radius=0.04, **scene_kw)
    rt = api.Raytracer(0)
    try:
        s.setup(rt)
        for f in range(frames):
            distributed.render_frame(rt, s.globals, f, 11)   # no capture flags: pipelined frames, the production path
        return rt.read_accumulator(), rt.get_stats()
    finally:
        rt.close()


@pytest.mark.parametrize("tabulated", [True, False])
def test_batched_gather_equals_the_launch_by_launch_gather(monkeypatch, tabulated):
    """one gather launch per frame (or per few shade launches when the rows run out: a 2 MB budget holds 3 of the 6 rows of this image)
    adds the same terms as a launch per bounce, after instead of between the emission of a path's later bounces: equal to
    rounding, far inside the radiance tolerance; every setting evaluates the same pairs and is reproducible bit for bit"""
    kw = dict(tabulated=tabulated)
    ref, st_ref = _render_c4(monkeypatch, {"RT_GATHER_BATCH": "0"}, **kw)
    images = {}
    for name, env in [("frame", {}), ("three_rows", {"RT_GATHER_BATCH_MB": "2"}), ("one_at_a_time", {"RT_FRAMES_IN_FLIGHT": "1"}),
                      ("big_blocks", {"RT_GATHER_BLOCK": "1024", "RT_GATHER_GROUPS": "1"})]:
        img, st = _render_c4(monkeypatch, env, **kw)
        assert st["gather_pairs"] == st_ref["gather_pairs"] > 0
        assert st["gather_launches"] < st_ref["gather_launches"]
        a, b = img.astype(np.float64), ref.astype(np.float64)
        rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
        assert rel.max() < 2e-6, "%s: batched and launch-by-launch gather differ by %g" % (name, rel.max())
        images[name] = img
    again, _ = _render_c4(monkeypatch, {}, **kw)
    assert np.array_equal(again.view(np.uint32), images["frame"].view(np.uint32)), "the batched gather is not reproducible"
    assert np.array_equal(images["big_blocks"].view(np.uint32), images["frame"].view(np.uint32)), "the result depends on the launch shape"


def _life_of_a_context(monkeypatch, env):
    """one context through changes of resolution, samples per pixel, emitting bounces and the SSS switch; the accumulator
    after each stage"""
    from raytracer_amd import api, distributed, _abi
    for k in ("RT_GATHER_BATCH", "RT_FRAMES_IN_FLIGHT", "RT_PROBE_SIDE", "RT_GATHER_FUSED"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    s = scenes.cornell_bunny(160, 96, spp=2, bounces=3, sample_radius=0.04)
    rt = api.Raytracer(0)
    out = []
    try:
        s.setup(rt)
        g = s.globals
        for f in range(3):
            distributed.render_frame(rt, g, f, 5)
        out.append(rt.read_accumulator())
        rt.update_resolution(224, 128)                    # larger: the path buffers and the rows of the batched gather are allocated again
        g.camera_aspect = 224.0 / 128.0
        for f in range(3):
            distributed.render_frame(rt, g, f, 5)         # (frame 0 starts a new accumulation)
        out.append(rt.read_accumulator())
        g.samples_per_pixel, g.bounces_per_sample, g.translucent_emission_bounces = 3, 2, 2   # three rows per sample instead of two
        for f in range(3, 5):
            distributed.render_frame(rt, g, f, 5)
        out.append(rt.read_accumulator())
        flags = s.flags
        rt.set_flags(flags & ~_abi.RT_FLAG_ENABLE_SSS)   # no gather, sample collection goes on (on the frame's lane)
        distributed.render_frame(rt, g, 5, 5)
        rt.set_flags(flags)
        g.samples_per_pixel = 1                           # one sample per pixel: the frame's miss / record masks
        for f in range(6, 10):
            distributed.render_frame(rt, g, f, 5)
        out.append(rt.read_accumulator())
        out.append(np.concatenate([rt.read_sample_points(i)[0]["payload"] for i in range(rt.translucent_instance_count())]))
        return out
    finally:
        rt.close()


def test_batched_gather_through_changes_of_the_frame_parameters(monkeypatch):
    ref = _life_of_a_context(monkeypatch, {"RT_GATHER_BATCH": "0"})
    one = _life_of_a_context(monkeypatch, {"RT_FRAMES_IN_FLIGHT": "1"})
    two = _life_of_a_context(monkeypatch, {})
    lane = _life_of_a_context(monkeypatch, {"RT_PROBE_SIDE": "0"})
    unfused = _life_of_a_context(monkeypatch, {"RT_GATHER_FUSED": "0"})
    for k in range(len(ref)):
        assert ref[k].any()
        assert np.array_equal(one[k].view(np.uint32), two[k].view(np.uint32)), "stage %d: frames in flight change the image" % k
        assert np.array_equal(lane[k].view(np.uint32), two[k].view(np.uint32)), "stage %d: the stream of the sample-point pass changes the image" % k
        for name, img in (("fused", two[k]), ("unfused", unfused[k])):
            a, b = img.astype(np.float64), ref[k].astype(np.float64)
            rel = np.abs(a - b) / np.maximum(np.abs(b), 1e-6)
            assert rel.max() < 2e-6, "stage %d: batched (%s) and launch-by-launch gather differ by %g" % (k, name, rel.max())
    assert np.array_equal(ref[-1].view(np.uint32), two[-1].view(np.uint32)), "the sample points' payloads do not depend on the gather"
