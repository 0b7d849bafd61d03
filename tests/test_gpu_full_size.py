"""GPU parity at BASELINE.json's full sizes (1920x1080).  The CPU oracle with its own BVH traces a 1080p frame in well
under a second per million rays on the GPU box's host cores, so the full-size configurations are compared ray by ray
like the small ones; on top of that come the size-independent properties: run-to-run determinism, and the tile split
summing bit-exactly to the single-context image."""
import numpy as np
import pytest

from raytracer_amd import scenes
import parity_util as pu

pytestmark = pytest.mark.gpu


def _full_parity(scene, oracle_mod, frames):
    from raytracer_amd import api
    gpu = api.Raytracer(0)
    cpu = oracle_mod.Oracle(brute_force=False, threads=16)
    try:
        pu.run_pair(scene, gpu, cpu, frames=frames)
        rays = 0
        for b in range(scene.globals.bounces_per_sample + 1):
            rays += pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b)
        pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
        return rays
    finally:
        gpu.close(); cpu.close()


def test_c2_bunny_1080p_every_ray(oracle_mod):
    rays = _full_parity(scenes.bunny(1920, 1080, spp=1, bounces=1), oracle_mod, frames=2)
    assert rays > 3000000


def test_c3_cornell_1080p_every_ray(oracle_mod):
    rays = _full_parity(scenes.cornell(1920, 1080, spp=2, bounces=4), oracle_mod, frames=1)
    assert rays > 4500000   # hit records are tapped for sample 0 of the dispatch; the image covers both samples


@pytest.mark.parametrize("mode", ["3", "0"])
def test_persistent_and_dispatched_traversal_agree_with_the_oracle(oracle_mod, monkeypatch, mode):
    """k_traverse has two ways of handing 64-ray tasks to waves: one workgroup per task (RT_PERSISTENT=0), or one
    persistent wave per slot that takes tasks from partitioned counters with stealing (bit 0: queued launches, bit 1:
    primary launches; off by default).  Both are forced here on a frame large enough for the persistent
    grid (more than 5,120 tasks), with ray grouping on, and compared ray by ray; two frames, so that the counters are
    seen to be left at zero by the first."""
    monkeypatch.setenv("RT_PERSISTENT", mode)
    rays = _full_parity(scenes.bunny(1280, 720, spp=1, bounces=2), oracle_mod, frames=2)
    assert rays > 1200000


def test_1080p_determinism_and_tile_split():
    from raytracer_amd import api, distributed
    scene = scenes.bunny(1920, 1080, spp=1, bounces=1)
    ctx = []
    for split in (None, None, (0, 3), (1, 3), (2, 3)):
        rt = api.Raytracer(0); scene.setup(rt)
        if split: rt.set_tile_split(*split)
        ctx.append(rt)
    try:
        for f in range(3):
            for rt in ctx:
                distributed.render_frame(rt, scene.globals, f, 11)
        imgs = [rt.read_accumulator() for rt in ctx]
        assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))              # same work, same bits
        assert np.array_equal((imgs[2] + imgs[3] + imgs[4]).view(np.uint32), imgs[0].view(np.uint32))
        assert np.array_equal(ctx[0].read_render_target(), ctx[1].read_render_target())
    finally:
        for rt in ctx: rt.close()


def test_c5_instanced_field_4k_every_ray(oracle_mod):
    """C5 as BASELINE.json states it: 52,743 bunny instances (50 M instanced triangles), 3840x2160, bounces 4:
    21.8 M rays compared one by one (the oracle's own two-level BVH needs ~7 s for the frame on 16 host threads)"""
    scene = scenes.bunny_field(3840, 2160, spp=1, bounces=4)
    assert scene.triangle_count() > 49000000
    assert _full_parity(scene, oracle_mod, frames=1) > 20000000


def test_frames_in_flight_do_not_change_the_image(monkeypatch):
    """Dispatches are pipelined over two streams with two sets of path buffers (RT_FRAMES_IN_FLIGHT, default 2); only a
    frame's resolve kernel, and the sample-point pass with its copy, wait for the frame before.  Same bits as one frame
    at a time: for one and for several samples per pixel, with a tile split, with the sample-point pass and both BSSRDF
    gathers, across a read in the middle of the sequence (which drains the pipeline and starts it again) and across a
    change of flags that stops the sample collection for two frames (odd number of collecting frames before it: the
    newest sample points then sit in lane 1's copy)."""
    from raytracer_amd import api, distributed, _abi
    cases = [(scenes.bunny(1280, 720, spp=1, bounces=2), None), (scenes.cornell(640, 360, spp=3, bounces=3), None),
             (scenes.bunny(1920, 1080, spp=1, bounces=1), (1, 2)),
             (scenes.cornell(480, 270, spp=1, bounces=3, translucent_boxes=True), None),
             (scenes.cornell_bunny(480, 270, spp=2, bounces=3), None)]
    results = {}
    for fif in ("1", "2"):
        monkeypatch.setenv("RT_FRAMES_IN_FLIGHT", fif)
        for k, (scene, split) in enumerate(cases):
            rt = api.Raytracer(0)
            try:
                scene.setup(rt)
                if split: rt.set_tile_split(*split)
                mid = None
                flags = _abi.RT_FLAG_COLLECT_SAMPLES | _abi.RT_FLAG_ENABLE_SSS
                for f in range(10):
                    if f == 7: rt.set_flags(flags & ~_abi.RT_FLAG_COLLECT_SAMPLES)
                    if f == 9: rt.set_flags(flags)
                    distributed.render_frame(rt, scene.globals, f, 23)
                    if f == 3: mid = rt.read_render_target()
                pts = [rt.read_sample_points(i)[0] for i in range(rt.translucent_instance_count())]
                results[(fif, k)] = (rt.read_accumulator(), rt.read_render_target(), mid, pts)
            finally:
                rt.close()
    for k in range(len(cases)):
        a, b = results[("1", k)], results[("2", k)]
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32)), k
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), k
        assert len(a[3]) == len(b[3]) and all(x.tobytes() == y.tobytes() for x, y in zip(a[3], b[3])), k
        assert a[0].any()
    assert len(results[("2", 3)][3]) > 0 and len(results[("2", 4)][3]) > 0


def test_scheduling_switches_do_not_change_the_image(monkeypatch):
    """Where a ray sits in its queue, which XCD traces its tile, whether the floor is inlined or the tree flat: none of it may
    change a bit of the image.  Production path (no hit tap, frames in flight), five frames each: the bunny on its floor
    and Cornell + translucent bunny (two-instance flat scenes: RT_SORT_BOX's chord classes, fused and unfused shade), the
    Cornell box alone (one instance: classes of the way through the scene's box), the
    instance field (two levels: chord classes, inlined floor) -- every switch off, one at a time, against all defaults."""
    from raytracer_amd import api, distributed
    cases = [scenes.bunny(960, 540, spp=2, bounces=3), scenes.cornell_bunny(480, 270, spp=2, bounces=3),
             scenes.bunny_field(640, 360, spp=1, bounces=3, grid=24, count=500), scenes.cornell(480, 270, spp=2, bounces=3)]
    switches = [None, "RT_SORT_BOX", "RT_RAY_SORT", "RT_XCD_BANDS", "RT_INLINE", "RT_FLAT_INLINE", "RT_FLAT", "RT_WAVE_MAJOR"]
    images = {}
    for sw in switches:
        if sw: monkeypatch.setenv(sw, "0")
        for k, scene in enumerate(cases):
            rt = api.Raytracer(0)
            try:
                scene.setup(rt)
                for f in range(5):
                    distributed.render_frame(rt, scene.globals, f, 41)
                images[(sw, k)] = (rt.read_accumulator(), rt.read_render_target())
            finally:
                rt.close()
        if sw: monkeypatch.delenv(sw)
    for k in range(len(cases)):
        ref = images[(None, k)]
        assert ref[0].any()
        for sw in switches[1:]:
            got = images[(sw, k)]
            assert np.array_equal(ref[0].view(np.uint32), got[0].view(np.uint32)), (sw, k)
            assert np.array_equal(ref[1], got[1]), (sw, k)


def test_top_level_rebuilds_between_pipelined_frames(monkeypatch):
    """The reference rebuilds the top level every frame (raytracing.cpp:652-676).  A rebuild, like every entry point other
    than a dispatch, first drains both lanes of the frame pipeline: frames rendered with a moving instance, with and
    without frames in flight, give the same bits, and so do statistics calls dropped into a run of dispatches."""
    from raytracer_amd import api, distributed
    scene = scenes.bunny(960, 540, spp=1, bounces=2)
    images = []
    for fif in ("1", "2"):
        monkeypatch.setenv("RT_FRAMES_IN_FLIGHT", fif)
        rt = api.Raytracer(0)
        try:
            rt.update_resolution(scene.width, scene.height)
            handles = [rt.build_blas(geos) for geos in scene.blas]
            rt.set_flags(scene.flags); rt.set_globals(scene.globals)
            for f in range(8):
                if f < 4 or f == 6:   # frames 4, 5 and 7 reuse the top level of the frame before: those overlap
                    inst = []
                    for m, b in scene.instances:
                        m = np.array(m, dtype=np.float32, copy=True)
                        if b == 0: m[3, 0] += 0.01 * f   # row vectors: translation in row 3
                        inst.append((m, handles[b]))
                    rt.build_tlas(inst)
                distributed.render_frame(rt, scene.globals, f, 31)
                if f == 5: assert rt.get_stats()["rays_camera"] > 0
            images.append((rt.read_accumulator(), rt.read_render_target()))
        finally:
            rt.close()
    assert np.array_equal(images[0][0].view(np.uint32), images[1][0].view(np.uint32))
    assert np.array_equal(images[0][1], images[1][1]) and images[0][0].any()


def _double_dispatch(scene):
    from raytracer_amd import _abi
    scene.flags |= _abi.RT_FLAG_DOUBLE_DISPATCH   # raytracing.cpp:688-695: one frame at a time, the fused pass with its own resolve kernel
    return scene


def _emit_at_every_bounce(scene):
    scene.globals.translucent_emission_bounces = scene.globals.bounces_per_sample  # the last bounce's translucent hits gather too
    return scene


@pytest.mark.parametrize("case", ["bunny_b1", "bunny_b3", "cornell_spp3", "cornell_bunny", "ragged_spp3", "cornell_bunny_emit_all",
                                  "cornell_as_shipped", "cornell_as_shipped_emit_all_spp2", "cornell_bunny_double_dispatch"])
def test_images_without_the_hit_tap_match_the_oracle(oracle_mod, case):
    """The ray-by-ray parity tests run with RT_FLAG_CAPTURE_HITS, which keeps the dispatches one at a time and the last
    shade of a frame complete.  The production path -- frames in flight, deferred resolve, lazy render target, and the
    emission-only shade of the last bounce -- is compared here through what it produces: accumulator (north_star
    tolerances; NaN pattern included) and 8-bit render target (at most one step apart where the float differs)
    against the oracle after several frames."""
    from raytracer_amd import api
    scene = {"bunny_b1": lambda: scenes.bunny(960, 540, spp=1, bounces=1), "bunny_b3": lambda: scenes.bunny(640, 360, spp=1, bounces=3),
             "cornell_spp3": lambda: scenes.cornell(480, 270, spp=3, bounces=4),
             "cornell_bunny": lambda: scenes.cornell_bunny(320, 180, spp=2, bounces=3),
             # the batched gather in the fused pass: a gathering last bounce (its shade is emission-only),
             # and the one-instance scene as shipped -- bounce 0 as one kernel, the last bounce shaded by its traversal kernel
             "cornell_bunny_emit_all": lambda: _emit_at_every_bounce(scenes.cornell_bunny(320, 180, spp=1, bounces=2, sample_radius=0.04)),
             "cornell_bunny_double_dispatch": lambda: _double_dispatch(scenes.cornell_bunny(256, 144, spp=2, bounces=2, sample_radius=0.05)),
             "cornell_as_shipped": lambda: scenes.cornell(320, 180, spp=1, bounces=4, translucent_boxes=True, sample_radius=0.03),
             "cornell_as_shipped_emit_all_spp2": lambda: _emit_at_every_bounce(scenes.cornell(256, 144, spp=2, bounces=2, translucent_boxes=True, sample_radius=0.04)),
             # tiles that are only partly inside the image, several samples per pixel, frames in flight: the per-thread
             # records and the miss / record masks of the deferred resolve on ragged tiles
             "ragged_spp3": lambda: scenes.bunny(333, 181, spp=3, bounces=2)}[case]()
    gpu = api.Raytracer(0)
    cpu = oracle_mod.Oracle(brute_force=False, threads=16)
    try:
        pu.run_pair(scene, gpu, cpu, frames=5, capture=False)
        ga, ca = gpu.read_accumulator(), cpu.read_accumulator()
        pu.compare_images(ga, ca)
        gr, cr = gpu.read_render_target().astype(np.int32), cpu.read_render_target().astype(np.int32)
        assert np.abs(gr - cr).max() <= 1
        if "cornell_bunny" not in case and "as_shipped" not in case:   # no tolerance-covered arithmetic without the gather: the same floats, the same bytes
            assert np.array_equal(ga.view(np.uint32), ca.view(np.uint32)) and np.array_equal(gr, cr)
    finally:
        gpu.close(); cpu.close()


def _tile_subset_mask(width, height, world):
    """pixels of the 32x32 tiles t with t % world == 0 (rt_set_tile_split(0, world))"""
    tiles_x = (width + 31) // 32
    ty, tx = np.divmod(np.arange(height)[:, None] // 32 * tiles_x + np.arange(width)[None, :] // 32, tiles_x)
    return ((ty * tiles_x + tx) % world) == 0


@pytest.mark.parametrize("case", ["c4_skin0", "c4_skin1", "c3_64spp"])
def test_baseline_sizes_on_the_production_path(oracle_mod, case):
    """BASELINE.json's C4 (1920x1080, 16 spp, tabulated skin_0 / skin_1 BSSRDF) and C3 (1920x1080, 64 spp) exactly as the
    bench runs them: no hit tap, two frames in flight, deferred resolve, three frames (accumulator_count 0 and non-zero).
    The oracle renders an interleaved 1/64 subset of the 32x32 tiles (its own tile split; the sample-point pass is
    complete in it) and those pixels are compared with the GPU's full-frame accumulator: north_star tolerances for C4
    (the gather's radiance arithmetic), bit for bit for C3."""
    from raytracer_amd import api, distributed
    world = 64
    if case == "c3_64spp":
        scene = scenes.cornell(1920, 1080, spp=64, bounces=4)
    else:
        scene = scenes.cornell_bunny(1920, 1080, spp=16, bounces=4, profile=int(case[-1]))
    gpu = api.Raytracer(0)
    cpu = oracle_mod.Oracle(brute_force=False, threads=16)
    try:
        assert scene.setup(gpu) == scene.setup(cpu)
        cpu.set_tile_split(0, world)
        for f in range(3):
            for backend in (gpu, cpu):
                distributed.render_frame(backend, scene.globals, f, scene.globals.frame_rng)
        mask = _tile_subset_mask(scene.width, scene.height, world)
        assert 30000 < int(mask.sum()) < 40000
        ga, ca = gpu.read_accumulator(), cpu.read_accumulator()
        assert not ca[~mask].any() and ca[mask].any()
        if case == "c3_64spp":
            assert np.array_equal(ga[mask].view(np.uint32), ca[mask].view(np.uint32))
        else:
            pu.compare_images(ga[mask], ca[mask])
            pts_g, pts_c = gpu.read_sample_points(0)[0], cpu.read_sample_points(0)[0]
            assert np.array_equal(pts_g["position"], pts_c["position"])
            assert np.allclose(pts_g["payload"], pts_c["payload"], rtol=1e-4, atol=1e-7)
        gr, cr = gpu.read_render_target().astype(np.int32), cpu.read_render_target().astype(np.int32)
        assert np.abs(gr[mask] - cr[mask]).max() <= 1
    finally:
        gpu.close(); cpu.close()
