"""GPU parity on the corner cases of the path: tiny and ragged resolutions, degenerate and duplicated triangles
(the S5 tie-break), shared edges (watertightness, S4), the 16-bit index limit, one-triangle meshes, zero bounces,
mirrored / non-uniform instance transforms, and the status codes of the C-ABI."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import scenes, _abi, host
import parity_util as pu

pytestmark = pytest.mark.gpu

LAMBERT, LIGHT = _abi.RT_SHADER_LAMBERT, _abi.RT_SHADER_LIGHT


def _check(scene, oracle_mod, frames=1):
    from raytracer_amd import api
    gpu = api.Raytracer(0)
    cpu = oracle_mod.Oracle(brute_force=True, threads=8)
    try:
        pu.run_pair(scene, gpu, cpu, frames=frames)
        rays = 0
        for b in range(scene.globals.bounces_per_sample + 1):
            rays += pu.compare_hits(gpu.read_hit_records(b), cpu.read_hit_records(b), b)
        pu.compare_images(gpu.read_accumulator(), cpu.read_accumulator())
        return rays, gpu.read_hit_records(0)
    finally:
        gpu.close(); cpu.close()


def _mesh(tris):
    """tris: (n, 3, 3) positions -> flat-shaded vertices (normal = normalised cross product, +z for degenerate)"""
    tris = np.asarray(tris, dtype=np.float32)
    v = np.zeros(len(tris) * 3, dtype=_abi.VERTEX_DTYPE)
    for k, t in enumerate(tris):
        n = np.cross(t[1] - t[0], t[2] - t[0]).astype(np.float64)
        ln = np.linalg.norm(n)
        n = n / ln if ln > 0 else np.array([0.0, 0.0, 1.0])
        for j in range(3):
            v["position"][3 * k + j] = t[j]
            v["normal"][3 * k + j] = n
    return v, np.arange(len(tris) * 3, dtype=np.uint16)


def _light_and_floor(s):
    lv, li = host.quad((-0.5, -0.5, 0.9), (-0.5, 0.5, 0.9), (0.5, 0.5, 0.9), (0.5, -0.5, 0.9))
    fv, fi = host.quad((-1.5, -1.5, -0.3), (1.5, -1.5, -0.3), (1.5, 1.5, -0.3), (-1.5, 1.5, -0.3))
    s.blas.append([scenes._geo(fv, fi, LAMBERT, (0.8, 0.8, 0.8)), scenes._geo(lv, li, LIGHT, (0.0, 0.0, 0.0))])
    s.instances.append((np.eye(4, dtype=np.float32), len(s.blas) - 1))


@pytest.mark.parametrize("w,h", [(1, 1), (33, 1), (1, 35), (31, 33), (64, 2)])
def test_tiny_and_ragged_resolutions(oracle_mod, w, h):
    rays, _ = _check(scenes.debug_cube(w, h, spp=2, bounces=2), oracle_mod, frames=2)
    assert rays >= w * h


def test_zero_bounces_and_many_samples(oracle_mod):
    _check(scenes.cornell(40, 30, spp=7, bounces=0), oracle_mod, frames=2)


def test_duplicate_and_degenerate_triangles_tie_break(oracle_mod):
    """the same wall three times in one BLAS (geometry 0 prims 0-1, geometry 1 prims 0-1) and again as a second
    instance, plus zero-area triangles: S5 says the lowest (instance, primitive) wins, degenerates never hit"""
    s = scenes.SceneDesc("dupes", 48, 40)
    a, b, c, d = (-0.6, 0.3, -0.3), (0.6, 0.3, -0.3), (0.6, 0.3, 0.6), (-0.6, 0.3, 0.6)  # faces the camera at -y
    wall = [[a, b, c], [a, c, d]]
    degenerate = [[a, a, b], [a, b, b], [c, c, c], [a, (0.0, 0.3, -0.3), b]]             # repeated / collinear vertices
    v0, i0 = _mesh(wall + degenerate)
    v1, i1 = _mesh(wall)
    s.blas.append([scenes._geo(v0, i0, LAMBERT, (0.9, 0.2, 0.2)), scenes._geo(v1, i1, LAMBERT, (0.2, 0.9, 0.2))])
    s.instances.append((np.eye(4, dtype=np.float32), 0))
    s.instances.append((np.eye(4, dtype=np.float32), 0))
    _light_and_floor(s)
    s.globals.samples_per_pixel = 1
    s.globals.bounces_per_sample = 2
    _, hits = _check(s, oracle_mod)
    on_wall = hits["instance"] == 0
    assert on_wall.sum() > 100 and not (hits["instance"] == 1).any()      # the second copy never wins a tie
    assert (hits["geometry"][on_wall] == 0).all() and (hits["primitive"][on_wall] <= 1).all()


def test_shared_edges_are_watertight(oracle_mod):
    """a fan of 64 thin triangles around a point straight ahead of the camera: every primary ray through the disc
    hits exactly one of them (no leaks along the shared edges, no double hits), ids equal to the oracle's"""
    s = scenes.SceneDesc("fan", 96, 96)
    n = 64
    ring = [(0.5 * np.cos(2 * np.pi * k / n), 0.4, 0.2 + 0.5 * np.sin(2 * np.pi * k / n)) for k in range(n)]
    centre = (0.0, 0.4, 0.2)
    tris = [[centre, ring[k], ring[(k + 1) % n]] for k in range(n)]
    v, i = _mesh(tris)
    # make it face the camera whichever way the winding came out: add the mirrored copy as a second geometry
    v2, i2 = _mesh([[t[0], t[2], t[1]] for t in tris])
    s.blas.append([scenes._geo(v, i, LAMBERT, (0.9, 0.9, 0.9)), scenes._geo(v2, i2, LAMBERT, (0.9, 0.9, 0.9))])
    s.instances.append((np.eye(4, dtype=np.float32), 0))
    _light_and_floor(s)
    s.globals.samples_per_pixel = 1
    s.globals.bounces_per_sample = 1
    _, hits = _check(s, oracle_mod)
    assert (hits["instance"] == 0).sum() > 500


def test_sixteen_bit_index_limit(oracle_mod):
    """a BLAS of exactly 65,535 vertices: the indices of all geometries are rebased into one concatenated vertex
    buffer (raytracing.cpp:360-366) that 16-bit indices must address (:367); one vertex more is refused"""
    rs = np.random.RandomState(5)
    n = 21844
    c = rs.uniform(-0.5, 0.5, (n, 1, 3)) * np.array([1.0, 1.0, 0.6]) + np.array([0.0, 0.0, 0.25])
    tris = c + rs.uniform(-0.03, 0.03, (n, 3, 3))
    v, i = _mesh(tris)
    assert len(v) == 65532
    s = scenes.SceneDesc("max_indices", 64, 48)
    v2, i2 = _mesh([[(-0.9, -0.9, -0.25), (0.9, -0.9, -0.25), (0.0, 0.9, -0.25)]])
    s.blas.append([scenes._geo(v, i, LAMBERT, (0.7, 0.7, 0.9)), scenes._geo(v2, i2, LAMBERT, (0.9, 0.7, 0.7))])
    s.instances.append((np.eye(4, dtype=np.float32), 0))
    _light_and_floor(s)
    s.globals.samples_per_pixel = 1
    s.globals.bounces_per_sample = 2
    _, hits = _check(s, oracle_mod)
    assert (hits["geometry"] == 1).any()  # the rebased triangle of the second geometry is seen
    from raytracer_amd import api
    rt = api.Raytracer(0)
    try:
        v3, i3 = _mesh([[(0, 0, 0), (1, 0, 0), (0, 1, 0)], [(0, 0, 1), (1, 0, 1), (0, 1, 1)]])
        with pytest.raises(RuntimeError, match="16-bit"):
            rt.build_blas([scenes._geo(v, i, LAMBERT, (1, 1, 1)), scenes._geo(v3, i3, LAMBERT, (1, 1, 1))])
    finally:
        rt.close()


def test_single_triangle_meshes_and_mirrored_transforms(oracle_mod):
    """BLAS of one triangle (root with an absent child), instanced with non-uniform scale, shear-free rotation and a
    mirror (negative determinant flips the facing in world space; culling is decided in object space)"""
    s = scenes.SceneDesc("singles", 80, 60)
    v, i = _mesh([[(-0.5, 0.0, -0.2), (0.5, 0.0, -0.2), (0.0, 0.0, 0.6)]])
    vb, ib = _mesh([[(-0.5, 0.0, -0.2), (0.0, 0.0, 0.6), (0.5, 0.0, -0.2)]])
    s.blas.append([scenes._geo(v, i, LAMBERT, (0.9, 0.5, 0.1))])
    s.blas.append([scenes._geo(vb, ib, LAMBERT, (0.1, 0.5, 0.9))])
    def xf(sx, sy, sz, angle, t):
        m = np.eye(4, dtype=np.float32)
        ca, sa = np.float32(np.cos(angle)), np.float32(np.sin(angle))
        m[0, 0], m[0, 1], m[1, 0], m[1, 1] = ca * sx, sa * sx, -sa * sy, ca * sy
        m[2, 2] = sz
        m[3, :3] = t
        return m
    s.instances += [(xf(1.0, 1.0, 1.0, 0.0, (0.0, 0.5, 0.0)), 0), (xf(0.5, 2.0, 0.7, 0.4, (-0.5, 0.2, 0.1)), 1),
                    (xf(-1.0, 1.0, 1.0, 0.0, (0.5, 0.3, 0.0)), 0), (xf(0.6, 0.6, -1.2, 1.1, (0.2, 0.0, 0.3)), 1)]
    _light_and_floor(s)
    s.globals.samples_per_pixel = 2
    s.globals.bounces_per_sample = 3
    _check(s, oracle_mod, frames=2)


def test_far_from_the_origin_and_offset_meshes(oracle_mod):
    """the whole scene and the camera moved to (3000, -2000, 1500), one mesh stored 500 units away from its object-space
    origin and brought back by its instance transform: the box test runs in each tree's own frame with planes rounded
    to half floats, and its padding scales with the coordinates; no hit may be lost to either"""
    s = scenes.debug_cube(72, 56, spp=2, bounces=3)
    shift = np.array([3000.0, -2000.0, 1500.0], dtype=np.float32)
    geos = s.blas[0]
    moved = geos[0]["vertices"].copy()
    moved["position"] += np.float32(500.0)
    s.blas[0] = [scenes._geo(moved, geos[0]["indices"], geos[0]["shader"], geos[0]["color"])]
    m0, b0 = s.instances[0]
    m0 = m0.copy(); m0[3, :3] -= (np.float32(500.0) * (m0[0, :3] + m0[1, :3] + m0[2, :3]))  # p' = (p + 500) M -> p M
    s.instances[0] = (m0, b0)
    s.instances = [(np.concatenate([m[:3], (m[3:] + np.append(shift, np.float32(0.0))).astype(np.float32)]).astype(np.float32), b) for m, b in s.instances]
    for k in range(3):
        s.globals.camera_to_world[12 + k] += float(shift[k])
    rays, hits = _check(s, oracle_mod, frames=2)
    assert (hits["instance"] == 0).sum() > 200 and (hits["instance"] == 1).sum() > 200


def test_status_codes_instead_of_abort():
    """prelude.h:185 aborts; the C-ABI returns a status and a message and stays usable"""
    from raytracer_amd import api
    rt = api.Raytracer(0)
    try:
        lib, h = rt.lib, rt.h
        lib.rt_last_error.restype = C.c_char_p
        assert lib.rt_dispatch_rays(h) != _abi.RT_OK                       # nothing built yet
        assert len(lib.rt_last_error()) > 0
        assert lib.rt_update_resolution(h, C.c_uint32(0), C.c_uint32(16)) != _abi.RT_OK
        inst = _abi.instance_array([(np.eye(4, dtype=np.float32), 7)])
        assert lib.rt_build_tlas(h, inst, C.c_int(1)) != _abi.RT_OK        # BLAS handle 7 does not exist
        out = C.c_int32(-1)
        assert lib.rt_build_blas(h, None, C.c_int(0), C.byref(out)) != _abi.RT_OK
        v, i = _mesh([[(0, 0, 0), (1, 0, 0), (0, 1, 0)]])
        bad = i.copy(); bad[2] = 9                                         # index past the vertex array
        arr, keep = _abi.geometry_array([scenes._geo(v, bad, LAMBERT, (1, 1, 1))])
        assert lib.rt_build_blas(h, arr, C.c_int(1), C.byref(out)) != _abi.RT_OK
        arr, keep = _abi.geometry_array([scenes._geo(v, i[:2], LAMBERT, (1, 1, 1))])   # not a multiple of 3
        assert lib.rt_build_blas(h, arr, C.c_int(1), C.byref(out)) != _abi.RT_OK
        # and the context still works afterwards
        s = scenes.debug_cube(16, 16, 1, 1)
        s.setup(rt)
        rt.dispatch_rays()
        assert np.isfinite(rt.read_accumulator()).all()
        buf = np.empty(16 * 16, dtype=_abi.HIT_DTYPE)
        assert lib.rt_read_hit_records(h, C.c_uint32(0), C.c_void_p(buf.ctypes.data)) != _abi.RT_OK  # capture flag not set
    finally:
        rt.close()


@pytest.mark.parametrize("split", ["tile", "sample", "tile+probe"])
def test_two_ranks_of_the_product_reproduce_the_one_process_image(split, tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), rehearsed on this box's
    one GPU (--rehearse: both ranks on GPU 0, gloo for the collective): rank 0's reduced accumulator against the image
    one process renders from the same frames.  Tile split (C4, with the sample-point pass and the gather): bit for bit.
    Sample-set split (C2): equal up to the re-association of the frame sum."""
    import os, subprocess, sys
    import numpy as np
    from raytracer_amd import api, scenes, distributed
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "acc.npy")
    probe = split == "tile+probe"   # the sample-point pass in two blocks, the payload blocks all-gathered behind every pass (--probe-split)
    if probe: split = "tile"
    workload = "c4" if split == "tile" else "c2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29600 + os.getpid() % 300), os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse",
           "--workload", workload, "--split", split, "--width", "256", "--height", "144", "--spp", "2", "--steps", "3",
           "--warmup", "0", "--no-cpu", "--dump-accumulator", out] + (["--probe-split"] if probe else [])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    import json
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["scaling"] == ("strong" if split == "tile" else "weak") and rec["reduce_ms"] > 0
    assert ("all-gather" in rec["config"]["parallelism"]) == probe
    got = np.load(out)
    scene = scenes.cornell_bunny(256, 144, spp=2, bounces=4) if split == "tile" else scenes.bunny(256, 144, spp=1, bounces=1)
    rt = api.Raytracer(0)
    try:
        scene.setup(rt)
        chain = distributed.FrameChain(scene.globals.frame_rng)
        for f in range(3 if split == "tile" else 6):
            distributed.render_frame(rt, scene.globals, f, chain)
        want = rt.read_accumulator()
    finally:
        rt.close()
    assert want.any()
    if split == "tile":
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("workload,split,size", [("c4", "tile", (1920, 1080)), ("c5", "sample", (3840, 2160))])
def test_the_scale_run_commands_at_baseline_size(workload, split, size, tmp_path):
    """The exact commands a SCALE run issues for BASELINE.md's C4 (tile split, 1080p, 16 spp) and C5 (sample-set split,
    4K, all 52,743 instances) rows, rehearsed with two ranks on this box's one GPU (--rehearse: gloo): they run to the
    end, the line has the fields the driver reads, and the exchange step is the one DESIGN.md section 6 describes --
    a gather of the owned tiles (1 / N of the frame per rank) for the tile split, a sum of the accumulators otherwise."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse",
           "--workload", workload, "--split", split, "--width", str(size[0]), "--height", str(size[1]), "--steps", "2",
           "--warmup", "1", "--no-cpu"]
    r = subprocess.run(cmd, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=400)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["reduce_ms"] > 0 and rec["timed_blocks"] >= 1
    frame_bytes = size[0] * size[1] * 16
    if split == "tile":
        assert rec["scaling"] == "strong" and rec["exchange"]["kind"].startswith("gather")
        tiles = ((size[0] + 31) // 32) * ((size[1] + 31) // 32)
        assert rec["exchange"]["bytes_per_rank"] == (tiles + 1) // 2 * 1024 * 16 < 0.51 * frame_bytes   # half of the tiles, not the frame
    else:
        assert rec["scaling"] == "weak" and rec["exchange"]["bytes_per_rank"] == frame_bytes


@pytest.mark.parametrize("workload,split", [("c2", "sample"), ("c4", "tile"), ("c4", "tile+probe")])
def test_the_rccl_calls_of_the_multi_gpu_path_as_a_one_rank_job(workload, split, tmp_path):
    """A one-GPU box cannot hold two RCCL ranks (one rank per device), so the backend calls of the N > 1 path -- process group
    on the "nccl" backend with a device id, reduce(sum) of the product's accumulator through its zero-copy view, gather of
    packed tiles, the MAX / SUM all-reduces of the timing, the barriers -- are run here as a job of ONE rank
    (bench.py --with-collectives): same calls, same tensors, a communicator of size 1.  The image must be the plain
    one-process image bit for bit (a reduce or a gather over one rank changes nothing)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    probe = split == "tile+probe"   # and the all-gather of the sample points' payload blocks, on the library's sample-point stream
    if probe: split = "tile"
    for tag, extra in (("plain", []), ("rccl", ["--with-collectives", "--split", split] + (["--probe-split"] if probe else []))):
        dump = str(tmp_path / (tag + ".npy"))
        base = [os.path.join(root, "bench.py"), "--gpus", "1", "--workload", workload, "--width", "320", "--height", "180", "--steps", "3",
                "--warmup", "1", "--no-cpu", "--no-alone", "--dump-accumulator", dump] + extra
        cmd = [sys.executable] + base if tag == "plain" else \
              [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
               "--master-port", str(29800 + os.getpid() % 90)] + base
        r = subprocess.run(cmd, cwd=root, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out[tag] = (json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1]), np.load(dump))
    rec, img = out["rccl"]
    assert rec["n_gpus"] == 1 and rec["reduce_ms"] is not None and rec["exchange"]["kind"].startswith("gather" if split == "tile" else "reduce")
    assert rec["config"]["parallelism"].startswith("tile split" if split == "tile" else "sample-set split") or rec["n_gpus"] == 1
    assert np.array_equal(img.view(np.uint32), out["plain"][1].view(np.uint32))
