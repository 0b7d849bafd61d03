"""The headless frame loop raytracer_amd/rt_render (host/rt_render.cpp: main.cpp:427-792 without the window) run as a
subprocess: scene assembly through rt_scene_create, per-frame reseed (main.cpp:443-454), and the regenerate / reset
sequencing of main.cpp:632-658 driven by scripted parameter changes.  Its accumulator (.pfm) is compared with the CPU
oracle driven through the same sequence from Python."""
import os
import subprocess

import numpy as np
import pytest

from raytracer_amd import scenes, host
import parity_util as pu

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = [int(x) for x in f.readline().split()]
        assert float(f.readline()) < 0   # little endian
        data = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return data[::-1]                     # PFM rows run bottom to top


def _run_cli(tmp_path, args):
    prefix = str(tmp_path / "img")
    cmd = [os.path.join(ROOT, "raytracer_amd", "rt_render"), "--data", os.path.join(ROOT, "raytracer_amd", "data"), "--out", prefix,
           "--pfm", "raw"] + args
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    files = [f for f in os.listdir(str(tmp_path)) if f.endswith(".pfm")]
    assert len(files) == 1, files
    assert os.path.exists(os.path.join(str(tmp_path), files[0][:-4] + ".png"))
    return _read_pfm(os.path.join(str(tmp_path), files[0])), files[0], r.stdout


def _oracle_sequence(oracle_mod, scene, seed, frames, events):
    """main.cpp:427-669 in Python on the oracle: events = {frame: [(key, value)]} with the reference's reset rules"""
    from raytracer_amd import _abi
    cpu = oracle_mod.Oracle(brute_force=False, threads=8)
    radius, scene.sample_radius = scene.sample_radius, None
    scene.setup(cpu)
    g = scene.globals.copy()
    g.frame_rng = seed
    regen, reset_t, reset_a = radius or 0.0, False, True
    flags = scene.flags
    for f in range(frames):
        g.frame_rng = host.next_frame_rng(g.frame_rng)
        for key, value in events.get(f, []):
            if key == "radius": regen = value
            elif key == "spp": g.samples_per_pixel = int(value); reset_a = True
            elif key == "bounces": g.bounces_per_sample = int(value); reset_a = True
            elif key == "collect":
                flags = (flags | _abi.RT_FLAG_COLLECT_SAMPLES) if value else (flags & ~_abi.RT_FLAG_COLLECT_SAMPLES); reset_a = True
            elif key == "light": g.light_color[:] = [value] * 3; reset_t = True
            elif key == "reset-samples": reset_t = True
            elif key == "reset": reset_a = True
            else: raise KeyError(key)
        cur = cpu.get_globals()
        g.accumulator_count, g.translucent_accumulator_count = cur.accumulator_count, cur.translucent_accumulator_count
        if regen:
            cpu.generate_translucent_samples(regen, scene.rng_seed, scene.phase_seed); regen = 0.0; reset_t = True
        if reset_t:
            reset_a = True; g.translucent_accumulator_count = 0; reset_t = False
        if reset_a:
            g.accumulator_count = 0; reset_a = False
        cpu.set_flags(flags)
        cpu.set_globals(g)
        cpu.dispatch_rays()
    acc = cpu.read_accumulator()
    count = cpu.get_globals().accumulator_count
    cpu.close()
    return acc, count


def test_cli_c2_matches_the_oracle_bit_for_bit(tmp_path, oracle_mod):
    got, name, _ = _run_cli(tmp_path, ["--scene", "c2", "--width", "200", "--height", "120", "--frames", "6", "--seed", "77",
                                       "--at", "2:spp=2", "--at", "4:bounces=3"])
    want, count = _oracle_sequence(oracle_mod, scenes.bunny(200, 120), 77, 6, {2: [("spp", 2)], 4: [("bounces", 3)]})
    assert count == 2 and "-#2-200x120" in name          # frames 4 and 5 since the last reset
    assert np.array_equal(got.view(np.uint32), want[..., :3].view(np.uint32)) and want.any()


def test_cli_c4_regenerate_and_reset_sequencing(tmp_path, oracle_mod):
    """sample points regenerated at another radius in mid-run, the light changed (resets sample points and image), the
    sample collection switched off (resets the image only): main.cpp:544,590,595,632-658"""
    ev = {2: [("radius", 0.05)], 4: [("light", 30.0)], 6: [("collect", 0)]}
    args = ["--scene", "c4", "--width", "160", "--height", "96", "--frames", "8", "--spp", "2", "--seed", "99"]
    for f, lst in sorted(ev.items()):
        for k, v in lst:
            args += ["--at", "%d:%s=%g" % (f, k, v)]
    got, name, out = _run_cli(tmp_path, args)
    want, count = _oracle_sequence(oracle_mod, scenes.cornell_bunny(160, 96, spp=2), 99, 8, ev)
    assert count == 2 and "-r5.000e-02-#2-" in name and "2 regenerations" in out
    pu.compare_images(got, want[..., :3])
    assert want.any()


@pytest.mark.parametrize("scene_name", ["cornell", "c1", "c3", "c5"])
def test_cli_runs_every_scene(tmp_path, scene_name):
    args = ["--scene", scene_name, "--width", "96", "--height", "64", "--frames", "2"]
    if scene_name == "c5": args += ["--instances", "500"]
    if scene_name == "c3": args += ["--spp", "2"]
    got, _, out = _run_cli(tmp_path, args)
    assert got.any() and np.isfinite(got).mean() > 0.999 and "2 accumulated" in out   # (a NaN pixel is legitimate: raytracing.hlsl:64-65)
