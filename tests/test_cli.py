"""`path-tracer` CLI surface (src/config/mod.rs:20-52, src/main.rs:14-57): flags, env, exit code 2."""
import os
import subprocess

import pytest

from conftest import ROOT, SCENES

EXE = ROOT / "path-tracer_amd" / "path-tracer"


def run(*args, env=None):
    return subprocess.run([str(EXE), *args], capture_output=True, text=True, env=dict(os.environ, **(env or {})))


def test_cli_exists_and_prints_help():
    assert EXE.exists(), "run `make cli`"
    r = run("render", "--help")
    assert r.returncode == 0
    for flag in ("--output", "--quiet", "--viewer", "--debug-textures", "--profile", "OUTPUT", "PROFILE", "render.png"):
        assert flag in r.stdout
    assert run("--help").returncode == 0 and "convert" in run("--help").stdout


@pytest.mark.parametrize("args", [(), ("render",), ("render", "/no/such/scene.isf", "-q"), ("bogus",),
                                  ("render", "a.isf", "--nope"), ("convert", "a.glb", "out/"),
                                  ("render", str(SCENES / "cube" / "scene.isf"), "-q", "-p", "/no/such/profile.yml")])
def test_cli_errors_exit_with_code_2(args):
    r = run(*args)
    assert r.returncode == 2
    assert r.stderr.strip()


def test_cli_bad_profile_and_env(tmp_path):
    bad = tmp_path / "bad.yml"
    bad.write_text("tonemap: PURPLE\n")
    r = run("render", str(SCENES / "cube" / "scene.isf"), "-q", env={"PROFILE": str(bad)})
    assert r.returncode == 2 and "PURPLE" in r.stderr


@pytest.mark.gpu
def test_cli_viewer_flag_refreshes_the_output(tmp_path, pta, gpu_scene_cache):
    import numpy as np
    from PIL import Image
    prof = tmp_path / "p.yml"
    prof.write_text("resolution: {width: 64, height: 48}\nsamples: 40\nbounces: 2\n")
    out = tmp_path / "v.png"
    r = run("render", str(SCENES / "cube" / "scene.isf"), "-q", "-v", "-p", str(prof), "-o", str(out))
    assert r.returncode == 0, r.stderr
    rgb, _ = gpu_scene_cache("cube").render(pta.Profile.make(64, 48, 40, 2))
    assert np.array_equal(np.asarray(Image.open(out)).reshape(-1, 3), rgb)


@pytest.mark.gpu
def test_cli_debug_textures(tmp_path, pta, gpu_scene_cache):
    import numpy as np
    from PIL import Image
    prof = tmp_path / "p.yml"
    prof.write_text("resolution:\n  width: 80\n  height: 60\n")
    r = subprocess.run([str(EXE), "render", str(SCENES / "spheres" / "scene.isf"), "--debug-textures", "-p", str(prof)],
                       capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    planes = gpu_scene_cache("spheres").debug_render(80, 60)
    for name in pta.DEBUG_PLANES:
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"{name}.png")).reshape(-1, 3), planes[name]), name
    assert not (tmp_path / "render.png").exists()


@pytest.mark.gpu
def test_cli_renders_like_the_library(tmp_path, pta, gpu_scene_cache):
    import numpy as np
    from PIL import Image
    prof = tmp_path / "p.yml"
    prof.write_text("resolution:\n  width: 96\n  height: 64\nsamples: 4\nbounces: 2\n")
    out = tmp_path / "o.png"
    r = run("render", str(SCENES / "reflection" / "scene.isf"), "-q", "-p", str(prof), env={"OUTPUT": str(out)})
    assert r.returncode == 0, r.stderr
    rgb, _ = gpu_scene_cache("reflection").render(pta.Profile.make(96, 64, 4, 2))
    assert np.array_equal(np.asarray(Image.open(out)).reshape(-1, 3), rgb)


@pytest.mark.gpu
def test_cli_several_devices_equal_one(tmp_path, pta, gpu_scene_cache):
    """--devices A,B,..: one host thread per device, interleaved tiles, host-side assembly.  With the same
    device listed three times the three shards run on one GPU; the image must equal the single-device one."""
    import numpy as np
    from PIL import Image
    prof = tmp_path / "p.yml"
    prof.write_text("resolution:\n  width: 150\n  height: 70\nsamples: 5\nbounces: 3\n")
    out = tmp_path / "multi.png"
    r = run("render", str(SCENES / "alpha_transparency" / "scene.isf"), "-q", "-p", str(prof), "-o", str(out),
            "--devices", "0,0,0", "--stats")
    assert r.returncode == 0, r.stderr
    assert '"devices": 3' in r.stderr
    rgb, _ = gpu_scene_cache("alpha_transparency").render(pta.Profile.make(150, 70, 5, 3))
    assert np.array_equal(np.asarray(Image.open(out)).reshape(-1, 3), rgb)
    r = run("render", str(SCENES / "cube" / "scene.isf"), "-q", "-o", str(out), "--devices", "0,,1")
    assert r.returncode == 2 and "invalid value" in r.stderr


@pytest.mark.gpu
def test_cli_two_distinct_devices_gather_over_rccl(tmp_path, pta, gpu_scene_cache):
    """--devices 0,1 on a host with at least two GPUs: the library's own exchange step (pt_comm_create_all,
    ncclAllGather of the packed slices + scatter, one host thread per device) between REAL ranks.  Skipped on a
    one-GPU box; on the 8-GPU node this is the multi-rank RCCL path's first run on hardware."""
    import numpy as np
    import torch
    from PIL import Image
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("needs two GPUs")
    prof = tmp_path / "p.yml"
    prof.write_text("resolution:\n  width: 320\n  height: 200\nsamples: 8\nbounces: 3\n")
    rgb, _ = gpu_scene_cache("alpha_transparency").render(pta.Profile.make(320, 200, 8, 3))
    for devs in ("0,1", ",".join(str(k) for k in range(min(n, 8)))):
        out = tmp_path / f"multi_{devs.count(',') + 1}.png"
        r = run("render", str(SCENES / "alpha_transparency" / "scene.isf"), "-q", "-p", str(prof), "-o", str(out),
                "--devices", devs, "--stats")
        assert r.returncode == 0, r.stderr
        assert '"gather": "rccl"' in r.stderr
        assert np.array_equal(np.asarray(Image.open(out)).reshape(-1, 3), rgb), devs
