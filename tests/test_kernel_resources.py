"""Register / scratch budgets of the hot kernels, read from the code object inside the built libptgpu.so.

A change elsewhere in a kernel's template family can push a hot variant over its budget without any test failing
(round 2: the orthographic-grid branch cost the fused bounce-0 kernel 112 B of scratch per lane and 3.5 ms of a 40 ms
frame until it became a compile-time switch).  The budgets below are the measured state plus a little slack.
"""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
LLVM = Path("/opt/rocm/lib/llvm/bin")


def kernel_table(tmp_path):
    lib = ROOT / "path-tracer_amd" / "libptgpu.so"
    tools = [LLVM / "clang-offload-bundler", LLVM / "llvm-readelf", shutil.which("objcopy")]
    if not lib.exists() or not all(t and Path(t).exists() for t in tools):
        pytest.skip("libptgpu.so or the LLVM binutils are not here")
    fat, elf = tmp_path / "fat.bin", tmp_path / "gfx950.elf"
    subprocess.run([tools[2], "-O", "binary", "--only-section=.hip_fatbin", str(lib), str(fat)], check=True)
    subprocess.run([str(tools[0]), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--input={fat}", f"--output={elf}"], check=True)
    notes = subprocess.run([str(tools[1]), "--notes", str(elf)], check=True, capture_output=True, text=True).stdout
    table = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        table[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", block).group(1))
                       for k in ("private_segment_fixed_size", "vgpr_count", "sgpr_count", "group_segment_fixed_size")}
    return table


def find(table, fragment):
    hits = [v for k, v in table.items() if fragment in k]
    assert len(hits) == 1, (fragment, [k for k in table if fragment in k])
    return hits[0]


def test_hot_kernels_stay_within_their_register_and_scratch_budgets(tmp_path):
    t = kernel_table(tmp_path)
    # the fused bounce-0 kernel of opaque scenes without directional lights (config 3): 3 waves / SIMD, (almost) no scratch
    b0 = find(t, "k_wf_shadeILb0ELb0ELb1ELi3EE")
    assert b0["vgpr_count"] <= 170 and b0["private_segment_fixed_size"] <= 32, b0
    # ... and of translucent scenes (config 5)
    b0a = find(t, "k_wf_shadeILb1ELb0ELb1ELi3EE")
    # (80 B in round 2; + the hit's id kept for the scene-box test and the optional entry word: 88 B; + the escape-mask lookup
    # of round 4: 100 B)
    assert b0a["vgpr_count"] <= 170 and b0a["private_segment_fixed_size"] <= 104, b0a
    # shading of the later bounces
    sh = find(t, "k_wf_shadeILb0ELb0ELb0ELi0EE")
    assert sh["vgpr_count"] <= 128 and sh["private_segment_fixed_size"] <= 96, sh
    # persistent KD-tree casts: 5 waves / SIMD (<= 96 registers + the traversal stack in scratch)
    tr = find(t, "k_wf_traceILb0ELb0ELb0EE")
    assert tr["vgpr_count"] <= 96 and tr["private_segment_fixed_size"] <= 480, tr
    # shadow casts through the light grids: 8 waves / SIMD; the few bytes of scratch belong to the rarely taken branch of
    # kdtree-ray's box test (scene_slab): measured +-0 against a build without it (profiles/r03_experiments.txt item 1)
    og = find(t, "k_og_shadowILb0ELb0ELb0EE")
    assert og["vgpr_count"] <= 64 and og["private_segment_fixed_size"] <= 24, og
