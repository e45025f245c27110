"""The measured options that stay in the library behind environment variables (read once per process) must render the
same bits as the default: every one of them only changes the order or the place in which paths are processed.

One child process per setting (the variables are process-static); a child renders three frames through the C ABI - a
golden scene with a directional light and translucency, the opaque generated scene (large enough for the drain hand-over
and several queue steps) and the translucent generated scene - and prints the SHA-1 of image + accumulation.
"""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import hashlib, sys
sys.path.insert(0, %(root)r)
import __graft_entry__ as e
pta = e.load_package()
h = hashlib.sha1()
sc = pta.HostScene.load_isf(%(scene)r)
g = pta.GpuScene(sc, device=0)
rgb, acc = g.render(pta.Profile.make(200, 150, 8, 4, "FILMIC"))
h.update(rgb.tobytes()); h.update(acc.tobytes())
for flags in (0, 1):
    sc = pta.HostScene.generate_ps5(30000, 0, flags)
    g = pta.GpuScene(sc, device=0)
    rgb, acc = g.render(pta.Profile.make(640, 360, 16, 5, "ACES"))
    h.update(rgb.tobytes()); h.update(acc.tobytes())
print("GRIDS", g.info().cam_grid_res, g.info().light_grids)
print("SHA", h.hexdigest())
"""

SETTINGS = [
    {},                                   # default
    {"PT_WF_DEFER": "0"},                 # no hand-over to k_wf_trace_wide
    {"PT_WF_DEFER": "2"},                 # ... as early as possible (many casts through the wide kernel)
    {"PT_CAM_CULL": "0"},                 # no camera-grid cull: every sample of an empty 8x8 block computes its ChaCha block and casts
    {"PT_WF_ALLWIDE": "1"},               # every cast of the bounces >= 1 through the cooperative kernel (an experiment's path)
    {"PT_GRAPH": "1"},                    # the frame's launches captured once, replayed with one hipGraphLaunch per frame
    {"PT_WF_SPLIT": "0"},                 # k_wf_trace_wide on the main stream, one shade pass after it
    {"PT_WF_SPLIT": "0", "PT_WF_DEFER": "2"},
    {"PT_WF_SORT": "1"},                  # octant bucketing of the survivors
    {"PT_WF_SORT": "2"},                  # hits shaded in material order
    {"PT_OG_FUSE_RNG": "0"},              # k_wf_rng stages the ChaCha words, bounce-0 kernel GRID 2
    {"PT_OG_INLINE_ALL": "1"},            # shadow casts inline at every bounce (GRID 1 at bounces >= 1)
    {"PT_WF_OVERLAP": "0"},               # everything on one stream
    {"PT_OG": "0"},                       # no origin grids at all: the KD-tree pipeline
    {"PT_WF_CHUNK": "1100000"},           # several queue chunks per frame
    {"PT_SHADE_BLOCKS_B0": "3", "PT_SHADE_BLOCKS": "1"},   # resident-only shade grids
    {"PT_WF_WALK": "3", "PT_WF_REFILL": "48"},              # odd traversal parameters
    {"PT_TILE_ORDER": "morton"},
    {"PT_WF_ENTRY": "1"},                 # casts of bounces >= 1 start at their primitive's home node (entry lists)
    {"PT_WF_ENTRY": "1", "PT_OG": "0"},   # ... on the KD-only pipeline
    {"PT_OG_HOST": "1"},                  # origin grids built by the host builder and uploaded (the round-2 path)
    {"PT_KD_PAD": "1"},                   # KD-tree over fattened primitives (the oracle filter's padding)
    {"PT_OG_BUDGET_GIB": "0.02"},         # grid memory budget of 20 MB: a coarser camera grid, the light on the KD-tree
    {"PT_OG_BUDGET_GIB": "0.001"},        # ... of 1 MB: no grid fits
]
EXPECT_GRIDS = {"0.02": "512 0", "0.001": "0 0"}


def run_child(extra):
    env = dict(os.environ)
    for k in list(env):
        if k.startswith(("PT_WF_", "PT_OG", "PT_SHADE_", "PT_TILE_", "PT_KD_", "PT_GRAPH", "PT_CAM_")):
            del env[k]
    env.update(extra)
    code = CHILD % {"root": str(ROOT), "scene": str(ROOT / "tests/golden/scenes/alpha_transparency/scene.isf")}
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (extra, out.stderr[-2000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("SHA ")]
    assert len(lines) == 1, (extra, out.stdout[-500:], out.stderr[-500:])
    grids = [ln for ln in out.stdout.splitlines() if ln.startswith("GRIDS ")][0][6:]
    if "PT_OG_BUDGET_GIB" in extra:
        assert grids == EXPECT_GRIDS[extra["PT_OG_BUDGET_GIB"]], (extra, grids)
    elif extra.get("PT_OG") != "0":
        assert grids == "1024 1", (extra, grids)
    return lines[0].split()[1]


@pytest.mark.gpu
def test_every_runtime_option_renders_the_same_bits():
    reference = run_child(SETTINGS[0])
    for extra in SETTINGS[1:]:
        assert run_child(extra) == reference, extra
