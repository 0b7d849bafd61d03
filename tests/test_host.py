"""The CPU scene layer (librt_host.so) against a Python restatement of src/parse_obj.cpp, src/geometry.cpp and the
scene / camera code of src/main.cpp."""
import math

import numpy as np
import pytest

from raytracer_amd import host, scenes, _abi


def py_parse_obj(path, rhs):
    """parse_obj.cpp:28-176: no dedup, quads -> (0 1 2)(0 2 3), face normals when the face carries no vn."""
    vs, vns, verts, idx = [], [], [], []
    for line in open(path):
        tok = line.split()
        if not tok:
            continue
        if tok[0] == "v":
            vs.append([np.float32(float(x)) for x in tok[1:4]])
        elif tok[0] == "vn":
            vns.append([np.float32(float(x)) for x in tok[1:4]])
        elif tok[0] == "f":
            corners = tok[1:5]
            vi = [int(c.split("/")[0]) - 1 for c in corners]
            ni = [int(c.split("/")[2]) - 1 for c in corners if c.count("/") == 2]
            base = len(verts)
            idx += [base, base + 1, base + 2] if len(corners) == 3 else [base, base + 1, base + 2, base, base + 2, base + 3]
            for k, v in enumerate(vi):
                pos = np.array(vs[v], dtype=np.float32)
                if ni:
                    nrm = np.array(vns[ni[k]], dtype=np.float32)
                else:
                    a = np.array(vs[vi[0]], dtype=np.float32)
                    n = np.cross(np.array(vs[vi[1]], dtype=np.float32) - a, np.array(vs[vi[2]], dtype=np.float32) - a)
                    nrm = (n / np.sqrt(np.dot(n, n))).astype(np.float32)
                if rhs:
                    pos = np.array([-pos[0], pos[2], pos[1]], dtype=np.float32)
                    nrm = np.array([-nrm[0], nrm[2], nrm[1]], dtype=np.float32)
                verts.append((pos, nrm))
    return verts, idx


@pytest.mark.parametrize("name,rhs", [("debug_cube.obj", False), ("debug_cube.obj", True), ("bunny.obj", True),
                                      ("cornell/floor.obj", True), ("cornell/largebox.obj", True), ("cornell/luminaire.obj", True)])
def test_parse_obj_matches_restatement(name, rhs):
    aabb = host.aabb_null()
    v, i = host.parse_obj_files(host.data_path(*name.split("/")), rhs, aabb)
    rv, ri = py_parse_obj(host.data_path(*name.split("/")), rhs)
    assert len(v) == len(rv) and list(i) == ri
    pos = np.array([p for p, _ in rv]); nrm = np.array([n for _, n in rv])
    assert np.array_equal(v["position"], pos)
    assert np.allclose(v["normal"], nrm, rtol=1e-6, atol=1e-7)
    assert np.array_equal(np.array(aabb[:3]), pos.min(axis=0)) and np.array_equal(np.array(aabb[3:]), pos.max(axis=0))


def test_expansion_counts_match_survey():
    # debug_cube 12 tris / 36 verts; bunny 948 tris / 2844 verts; cornell 36 tris in 7 geometries
    v, i = host.parse_obj_files(host.data_path("debug_cube.obj"), False)
    assert (len(v), len(i)) == (36, 36)
    v, i = host.parse_obj_files(host.data_path("bunny.obj"), True)
    assert (len(v), len(i)) == (2844, 948 * 3)
    s = scenes.cornell(64, 36)
    assert len(s.blas[0]) == 6 and s.triangle_count() == 36
    # files appended into one array rebase their indices (main.cpp:252-254 appends floor, back, ceiling)
    white = s.blas[0][0]
    assert len(white["vertices"]) == 12 and white["indices"].max() == 11


def test_missing_file_is_an_error():
    with pytest.raises(RuntimeError):
        host.parse_obj_files("/nonexistent.obj", True)


def test_normalizing_transform_and_widest():
    aabb = host.aabb_null()
    host.parse_obj_files(host.data_path("bunny.obj"), True, aabb)
    m = host.normalizing_transform(aabb)
    lo, hi = np.array(aabb[:3]), np.array(aabb[3:])
    assert abs(host.aabb_widest(aabb) - (hi - lo).max()) < 1e-9
    tl = lo @ m[:3, :3] + m[3, :3]; th = hi @ m[:3, :3] + m[3, :3]
    assert np.allclose(tl, -th, atol=1e-6) and abs((th - tl).max() - 1.0) < 1e-6   # widest = 1, centred (main.cpp:352-358)
    # bunny after RHS conversion is Z-up
    assert np.allclose(lo, (-0.058, -0.057, 0.034), atol=2e-3) and np.allclose(hi, (0.093, 0.058, 0.182), atol=2e-3)


def test_default_globals_and_camera():
    g = host.default_globals(16.0 / 9.0)
    assert (g.samples_per_pixel, g.bounces_per_sample, g.translucent_emission_bounces) == (1, 4, 1)   # main.cpp:413-415
    assert g.translucent_bssrdf_scale == 0.0 and g.translucent_bssrdf_fudge == 1.0                    # :418-419
    assert abs(g.translucent_refractive_index - 1.75) < 1e-7 and list(g.translucent_scattering) == [15.0] * 3
    assert np.allclose(list(g.translucent_absorption), 0.1) and list(g.light_color) == [50.0] * 3
    m = np.array(g.camera_to_world, dtype=np.float64).reshape(4, 4)
    assert np.allclose(m[:3, :3] @ m[:3, :3].T, np.eye(3), atol=1e-6)           # rigid
    assert abs(np.linalg.det(m[:3, :3]) - 1.0) < 1e-6                            # right-handed
    eye, target = m[3, :3], np.array([0, 0, -0.06])
    assert abs(np.linalg.norm(eye - target) - 2.5) < 1e-5                        # distance 2.5 (main.cpp:499)
    look = -m[2, :3]                                                             # camera looks down -Z
    assert np.allclose(look, (target - eye) / 2.5, atol=1e-6)
    assert abs(m[0, 2]) < 1e-6 and m[1, 2] > 0                                   # X right is horizontal, Y up has +Z
    assert abs(g.camera_focal_length - 1 / math.tan(math.radians(15))) < 1e-5    # fov_y 30 deg
    assert abs(math.degrees(math.asin((eye - target)[2] / 2.5)) - 9.0) < 1e-4    # elevation 9 deg


def test_frame_rng_chain_is_wang_hash(oracle_mod):
    r = 0x12345678
    for _ in range(5):
        assert host.next_frame_rng(r) == oracle_mod.hash1(r)
        r = host.next_frame_rng(r)


def test_image_writers(tmp_path):
    rgba = (np.arange(4 * 3 * 4) % 256).astype(np.uint8).reshape(3, 4, 4)
    p = str(tmp_path / "a.ppm")
    assert host.write_ppm(p, rgba) == 0
    data = open(p, "rb").read()
    assert data.startswith(b"P6\n4 3\n255\n") and data[11:] == rgba[:, :, :3].tobytes()
    f = np.random.RandomState(0).rand(3, 4, 4).astype(np.float32)
    p = str(tmp_path / "a.pfm")
    assert host.write_pfm(p, f, 0.5) == 0
    data = open(p, "rb").read()
    hdr = b"PF\n4 3\n-1.0\n"
    assert data.startswith(hdr)
    body = np.frombuffer(data[len(hdr):], dtype="<f4").reshape(3, 4, 3)
    assert np.allclose(body, (f[::-1, :, :3] * 0.5))


def test_scene_configs_have_the_documented_sizes():
    assert scenes.debug_cube().triangle_count() == 12 + 4
    assert scenes.bunny().triangle_count() == 948 + 4
    s5 = scenes.bunny_field(64, 36, grid=8, count=50)
    assert len(s5.instances) == 51 and s5.triangle_count() == 50 * 948 + 4
    m = np.array(s5.instances[3][0])
    sc = np.linalg.norm(m[:3, :3], axis=1)
    assert np.allclose(sc, sc[0], rtol=1e-5)      # uniform scale (required by bluenoise.cpp:122-127)


def test_png_writer_round_trips_through_zlib(tmp_path):
    """rt_write_png (main.cpp:750's stbi_write_png): signature, chunk CRCs, IHDR, and the IDAT stream inflates to the
    filter-0 scanlines of the image"""
    import struct, zlib
    from raytracer_amd import host
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, (37, 301, 4)).astype(np.uint8)     # 45 KB raw: one stored block; and a larger one below
    for image in (img, np.tile(img, (5, 1, 1))):
        path = str(tmp_path / "a.png")
        assert host.write_png(path, image) == 0
        data = open(path, "rb").read()
        assert data[:8] == b"\x89PNG\r\n\x1a\n"
        pos, chunks = 8, []
        while pos < len(data):
            n, kind = struct.unpack(">I4s", data[pos:pos + 8])
            body = data[pos + 8:pos + 8 + n]
            assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == (zlib.crc32(kind + body) & 0xffffffff)
            chunks.append((kind, body)); pos += 12 + n
        assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        h, w = image.shape[:2]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 6, 0, 0, 0)
        raw = zlib.decompress(chunks[1][1])
        rows = np.frombuffer(raw, dtype=np.uint8).reshape(h, 1 + 4 * w)
        assert (rows[:, 0] == 0).all() and np.array_equal(rows[:, 1:].reshape(h, w, 4), image)
