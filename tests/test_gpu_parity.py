"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest

from raytracer_amd import scenes, _abi
import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    from raytracer_amd import api
    rt = api.Raytracer(0)
    yield rt
    rt.close()


def _fresh_gpu():
    from raytracer_amd import api
    return api.Raytracer(0)


def _check_scene(scene, oracle_mod, frames=1, brute_force=True, threads=8):
    gpu = _fresh_gpu()
    cpu = oracle_mod.Oracle(brute_force=brute_force, threads=threads)
    try:
        pu.run_pair(scene, gpu, cpu, frames=frames)
        rays = 0
        for b in range(scene.globals.bounces_per_sample + 1):
            rays += pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b)
        pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
        rt_g, rt_c = gpu.read_render_target(), cpu.read_render_target()
        assert np.abs(rt_g.astype(int) - rt_c.astype(int)).max() <= 1
        return rays
    finally:
        gpu.close(); cpu.close()


def test_c1_debug_cube_256(oracle_mod):
    rays = _check_scene(scenes.debug_cube(256, 256, spp=1, bounces=1), oracle_mod)
    assert rays > 65536


def test_c1_debug_cube_multi_frame_spp(oracle_mod):
    _check_scene(scenes.debug_cube(96, 64, spp=3, bounces=3), oracle_mod, frames=3)


def test_c2_bunny_small(oracle_mod):
    _check_scene(scenes.bunny(256, 144, spp=1, bounces=1), oracle_mod)


def test_c2_bunny_deep_bounces(oracle_mod):
    _check_scene(scenes.bunny(128, 72, spp=2, bounces=4), oracle_mod, frames=2)


def test_c3_cornell_small(oracle_mod):
    _check_scene(scenes.cornell(192, 108, spp=4, bounces=4), oracle_mod, frames=2)


def test_odd_resolution_not_multiple_of_tile(oracle_mod):
    _check_scene(scenes.cornell(77, 45, spp=1, bounces=2), oracle_mod)


def test_tile_split_sums_to_the_full_image(oracle_mod):
    """two tile-split contexts (ranks 0 and 1 of 2) on one GPU: accumulators are disjoint and their sum is the
    single-context image bit for bit"""
    from raytracer_amd import api, distributed
    scene = scenes.cornell_bunny(100, 70, spp=2, bounces=2, sample_radius=0.06)
    full = api.Raytracer(0); scene.setup(full)
    parts = []
    for r in range(2):
        rt = api.Raytracer(0); scene.setup(rt); rt.set_tile_split(r, 2); parts.append(rt)
    for f in range(2):
        for rt in [full] + parts:
            distributed.render_frame(rt, scene.globals, f, 99)
    a, b, whole = parts[0].read_accumulator(), parts[1].read_accumulator(), full.read_accumulator()
    assert not np.any((a != 0) & (b != 0))
    assert np.array_equal((a + b).view(np.uint32), whole.view(np.uint32))
    cpu = oracle_mod.Oracle(brute_force=True, threads=8); scene.setup(cpu)
    for f in range(2):
        distributed.render_frame(cpu, scene.globals, f, 99)
    pu.compare_images(whole, cpu.read_accumulator())
    for rt in [full] + parts:
        rt.close()
    cpu.close()


def test_c5_instanced_field_small(oracle_mod):
    """C5 at reduced size: 300 rotated/scaled bunny instances (284k triangles) through the two-level BVH"""
    scene = scenes.bunny_field(160, 90, spp=1, bounces=2, grid=20, count=300)
    _check_scene(scene, oracle_mod, frames=2, brute_force=False)


def test_instance_count_that_fills_the_build_workspace_exactly(oracle_mod):
    """2,048 bunnies + the inlined floor: rt_build_tlas writes 2,049 instance boxes while the top level holds 2,048
    primitives, and 2,048 boxes of 24 bytes end exactly on a page boundary -- the workspace must hold all n boxes
    (it was sized for n - 1: the last box landed past the allocation)"""
    scene = scenes.bunny_field(160, 90, spp=1, bounces=2, grid=48, count=2048)
    _check_scene(scene, oracle_mod, frames=1, brute_force=False)


def test_many_instances_of_tiny_blas(oracle_mod):
    """TLAS-heavy case: 1500 instances of the 12-triangle cube, brute-force oracle"""
    from raytracer_amd import host
    s = scenes.debug_cube(96, 96, spp=1, bounces=2)
    rs = np.random.RandomState(11)
    s.instances = [(host.affine_transform(float(rs.uniform(0.02, 0.06)), float(rs.uniform(0, 6.28)),
                                          tuple(rs.uniform(-0.7, 0.7, 3))), 0) for _ in range(1500)] + [s.instances[1]]
    _check_scene(s, oracle_mod, frames=1, brute_force=True)


@pytest.mark.parametrize("pool", ["0", "2"])
def test_pooled_and_per_lane_traversal_agree_with_the_oracle(oracle_mod, monkeypatch, pool):
    """queued rays of two-level scenes are walked by pools of 256 rays per workgroup that are re-sorted by what they need
    next (RT_POOL=2: rt_dev_pool.h) or one ray per lane to the end (RT_POOL=0, the default): the closest hit does not depend on
    the order of the walk (S5), so both give the oracle's ids, t, u, v and RNG streams bit for bit.  Shallow trees (17 stack
    rows) and deep ones (33), with and without the inlined instance, several frames so that pools see ragged segments."""
    monkeypatch.setenv("RT_POOL", pool)       # read at rt_create
    _check_scene(scenes.bunny_field(160, 90, spp=1, bounces=3, grid=20, count=300), oracle_mod, frames=2, brute_force=False)
    _check_scene(scenes.bunny_field(96, 54, spp=2, bounces=2, grid=8, count=40), oracle_mod, brute_force=False)
    monkeypatch.setenv("RT_INLINE", "0")
    _check_scene(scenes.bunny_field(128, 72, spp=1, bounces=2, grid=8, count=50), oracle_mod, brute_force=False)


@pytest.mark.parametrize("grid", ["0", "2"])
def test_grid_and_tree_top_levels_agree_with_the_oracle(oracle_mod, monkeypatch, grid):
    """two-level scenes walk their top level as a tree (RT_TLAS_GRID=0) or as a uniform grid over the instances' world boxes
    (2: wherever the bottom-level stacks allow; 1, the default, only where the instances suit it): the closest hit does not
    depend on the order of the walk (S5), so both give the oracle's ids, t, u, v and RNG streams bit for bit.  Instances of
    one size on a lattice, with and without the inlined floor; 1,500 small cubes of unequal sizes scattered in a volume
    (three grid axes, cells with many entries, instances spanning several cells), against the brute-force oracle."""
    monkeypatch.setenv("RT_TLAS_GRID", grid)       # read at rt_create
    from raytracer_amd import api, host
    rt = api.Raytracer(0)
    sc = scenes.bunny_field(96, 54, spp=2, bounces=2, grid=8, count=40)
    sc.setup(rt)
    assert (rt.get_stats()["tlas_grid_cells"] > 0) == (grid == "2")
    rt.close()
    _check_scene(scenes.bunny_field(96, 54, spp=2, bounces=2, grid=8, count=40), oracle_mod, brute_force=False)
    _check_scene(scenes.bunny_field(160, 90, spp=1, bounces=3, grid=20, count=300), oracle_mod, frames=2, brute_force=False)
    s = scenes.debug_cube(96, 96, spp=1, bounces=2)
    rs = np.random.RandomState(11)
    s.instances = [(host.affine_transform(float(rs.uniform(0.02, 0.06)), float(rs.uniform(0, 6.28)),
                                          tuple(rs.uniform(-0.7, 0.7, 3))), 0) for _ in range(1500)] + [s.instances[1]]
    _check_scene(s, oracle_mod, frames=1, brute_force=True)
    monkeypatch.setenv("RT_INLINE", "0")
    _check_scene(scenes.bunny_field(128, 72, spp=1, bounces=2, grid=8, count=50), oracle_mod, brute_force=False)


def test_grid_top_level_chosen_by_itself_for_a_field_of_instances(oracle_mod, monkeypatch):
    """RT_TLAS_GRID=1: the library decides -- 4,500 meshes on a lattice get the grid, 300 do not (too few to pay), and the
    Cornell box with its room-sized walls never does (a flat scene); hits as the oracle's either way"""
    monkeypatch.setenv("RT_TLAS_GRID", "1")
    from raytracer_amd import api
    for sc, want in ((scenes.bunny_field(160, 90, spp=1, bounces=2, grid=70, count=4500), True),
                     (scenes.bunny_field(96, 54, spp=1, bounces=2, grid=20, count=300), False),
                     (scenes.cornell(64, 36, spp=1, bounces=2), False)):
        rt = api.Raytracer(0)
        sc.setup(rt)
        assert (rt.get_stats()["tlas_grid_cells"] > 0) == want
        rt.close()
    _check_scene(scenes.bunny_field(160, 90, spp=1, bounces=2, grid=70, count=4500), oracle_mod, frames=1, brute_force=False)


@pytest.mark.parametrize("builder", ["0", "1", "2", "3", "4"])
def test_every_bvh_builder_gives_the_same_hits(oracle_mod, monkeypatch, builder):
    """the closest hit does not depend on the tree (S5): Karras LBVH (0), PLOC (1), binned SAH (2), the
    SAH-BLAS / PLOC-TLAS mix (3), the cheaper of PLOC and SAH (4, default) against brute force"""
    monkeypatch.setenv("RT_BVH_BUILDER", builder)       # read at rt_create
    _check_scene(scenes.bunny(96, 54, spp=1, bounces=3), oracle_mod)
    _check_scene(scenes.bunny_field(96, 54, spp=1, bounces=2, grid=8, count=40), oracle_mod, brute_force=False)


def test_corner_instance_boxes_still_work(oracle_mod, monkeypatch):
    """TLAS boxes from the corners of the BLAS box (the fallback for very large vertex counts)"""
    monkeypatch.setenv("RT_EXACT_INSTANCE_BOXES", "0")
    _check_scene(scenes.bunny_field(96, 54, spp=1, bounces=2, grid=8, count=40), oracle_mod, brute_force=False)


@pytest.mark.parametrize("flat", ["0", "1"])
def test_flat_and_two_level_walks_agree_with_the_oracle(oracle_mod, monkeypatch, flat):
    """scenes of one or two instances are traced in ONE world-space tree (RT_FLAT=1, the default: rt_dev_scene.h
    trace_flat) or through the two levels like every other scene (RT_FLAT=0): same hits, t, u, v, RNG, bit for bit.
    Covers one instance (Cornell), two (bunny on a floor: rotated Y-up mesh; translucent bunny in the room with the
    sample-point pass) and a one-triangle instance (its tree is a single node)."""
    from raytracer_amd import api
    monkeypatch.setenv("RT_FLAT", flat)                   # read at rt_create
    rt = api.Raytracer(0)
    try:
        scenes.bunny(64, 36, spp=1, bounces=1).setup(rt)
        assert (rt.get_stats()["flat_depth"] > 0) == (flat == "1")
    finally:
        rt.close()
    _check_scene(scenes.cornell(96, 54, spp=2, bounces=3), oracle_mod)
    _check_scene(scenes.bunny(128, 72, spp=1, bounces=3), oracle_mod, frames=2)
    _check_scene(scenes.cornell_bunny(96, 54, spp=2, bounces=2, sample_radius=0.06), oracle_mod)
    s = scenes.debug_cube(64, 64, spp=1, bounces=2)
    from raytracer_amd import host
    tri_v = np.zeros(3, dtype=_abi.VERTEX_DTYPE)
    tri_v["position"] = [(-0.4, -0.3, 0.5), (0.4, -0.3, 0.5), (0.0, 0.5, 0.6)]
    tri_v["normal"] = [(0.0, 0.0, -1.0)] * 3
    s.blas[0] = [scenes._geo(tri_v, np.array([0, 2, 1], dtype=np.uint16), _abi.RT_SHADER_LAMBERT, (0.8, 0.8, 0.8))]
    s.instances[0] = (host.affine_transform(1.3, 0.7, (0.1, -0.05, 0.0)), 0)
    _check_scene(s, oracle_mod)


def test_three_instances_keep_the_two_levels():
    """the flat tree is for scenes of at most two instances; a third keeps the two-level walk"""
    from raytracer_amd import api, host
    s = scenes.debug_cube(64, 64, spp=1, bounces=1)
    s.instances.append((host.affine_transform(0.2, 1.0, (0.5, 0.5, 0.3)), 0))
    rt = api.Raytracer(0)
    try:
        s.setup(rt)
        assert rt.get_stats()["flat_depth"] == 0
    finally:
        rt.close()


def test_top_level_rebuilds_switch_between_the_flat_tree_and_the_two_levels(oracle_mod):
    """one context, three top levels in a row: two instances (flat tree), three (two levels: the flat tree is freed), one
    (flat tree again, single subtree) -- after every rebuild the hits equal the oracle's on the same instances"""
    from raytracer_amd import api, host
    scene = scenes.bunny(96, 54, spp=1, bounces=2)
    extra = (host.affine_transform(0.35, 2.0, (0.6, -0.4, 0.1)), 0)
    gpu = api.Raytracer(0)
    cpu = oracle_mod.Oracle(brute_force=True, threads=8)
    try:
        pu.run_pair(scene, gpu, cpu, frames=1)
        for k, (instances, flat) in enumerate([(scene.instances, True), (scene.instances + [extra], False), (scene.instances[:1], True)]):
            if k:  # the same BLAS handles (0 and 1 on both backends: built in the same order by run_pair)
                for backend in (gpu, cpu):
                    backend.build_tlas([(m, b) for m, b in instances])
                    g = backend.get_globals(); g.accumulator_count = 0; backend.set_globals(g)
                    backend.dispatch_rays()
            assert (gpu.get_stats()["flat_depth"] > 0) == flat
            for b in range(scene.globals.bounces_per_sample + 1):
                pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b)
            pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
    finally:
        gpu.close(); cpu.close()


@pytest.mark.parametrize("inline", ["0", "1"])
def test_inlined_instance_and_plain_two_levels_agree_with_the_oracle(oracle_mod, monkeypatch, inline):
    """scenes of more than two instances keep the two levels; the small instance that reaches furthest (at most 8
    triangles: the floor with its luminaire) is then tested by every ray before the walk instead of sitting in the top
    level (RT_INLINE=1, the default: rt_dev_scene.h SceneView::inl_inst) -- same hits, bit for bit, as with every instance
    in the top level (RT_INLINE=0).  Covers the inlined instance last in the array (the field), first and in the middle
    (the top level then holds the last instance in its slot), and two candidates of which one is taken."""
    from raytracer_amd import host
    monkeypatch.setenv("RT_INLINE", inline)               # read at rt_create
    _check_scene(scenes.bunny_field(128, 72, spp=1, bounces=2, grid=8, count=50), oracle_mod, brute_force=False)
    s = scenes.bunny(96, 54, spp=1, bounces=3)            # [bunny, floor quad]
    cube = scenes.debug_cube(8, 8).blas[0]
    s.blas.append(cube)
    s.instances = [s.instances[1], s.instances[0], (host.affine_transform(0.3, 1.0, (0.7, -0.5, 0.2)), 2)]   # floor first
    _check_scene(s, oracle_mod, frames=2)
    s.instances = [s.instances[1], s.instances[0], s.instances[2],
                   (host.affine_transform(0.05, 0.3, (-0.6, 0.4, 0.4)), 1)]                              # floor in the middle + a small copy of it
    _check_scene(s, oracle_mod)


@pytest.mark.parametrize("flat_inline", ["0", "1"])
def test_small_instance_of_a_flat_pair_tested_first(oracle_mod, monkeypatch, flat_inline):
    """two instances of which one has at most 8 triangles (the bunny on its floor, the cube under its light): the flat tree holds
    the big one alone and every ray tests the small one's triangles before the walk (RT_FLAT_INLINE=1, the default:
    rt_dev_scene.h SceneView::flat_inl_inst) -- same hits as with both in the tree (RT_FLAT_INLINE=0).  Both orders of the
    pair, so that the tree's instance is index 0 and index 1; a pair of two small instances keeps both in the tree."""
    from raytracer_amd import host
    monkeypatch.setenv("RT_FLAT_INLINE", flat_inline)      # read at rt_create
    s = scenes.bunny(128, 72, spp=1, bounces=3)
    _check_scene(s, oracle_mod, frames=2)
    s.instances = [s.instances[1], s.instances[0]]
    _check_scene(s, oracle_mod)
    _check_scene(scenes.debug_cube(96, 96, spp=2, bounces=2), oracle_mod)
    q = scenes.debug_cube(64, 64, spp=1, bounces=2)
    q.instances = [q.instances[1], (host.affine_transform(0.5, 0.4, (0.1, 0.0, -0.3)), q.instances[1][1])]   # two quads
    _check_scene(q, oracle_mod)
