#!/usr/bin/env python3
"""Registers / scratch / occupancy of every kernel of csrc/pt_gpu.hip as the compiler reports them
(hipcc -Rpass-analysis=kernel-resource-usage); extra arguments go to hipcc (e.g. -DWF_THREADS=320)."""
import re, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", f"-I{ROOT}/include",
       f"-I{ROOT}/path-tracer_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c",
       str(ROOT / "path-tracer_amd/csrc/pt_gpu.hip"), "-o", "/tmp/pt_gpu_resources.o"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        continue
    k, v = m.groups()
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(" ")[0]] = v
print(f"{'kernel':60s} {'VGPR':>5s} {'SGPR':>5s} {'scratch':>8s} {'waves':>6s} {'LDS':>7s}")
for r in rows:
    print(f"{r['name'][:60]:60s} {r.get('VGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('ScratchSize','?'):>8s} {r.get('Occupancy','?'):>6s} {r.get('LDS','?'):>7s}")
