#!/usr/bin/env python3
"""Print the per-stage HIP-event times of one render of the bench workload (diagnostics)."""
import argparse, ctypes as C, json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import __graft_entry__ as entry
ap = argparse.ArgumentParser()
ap.add_argument("--tris", type=int, default=500000); ap.add_argument("--spp", type=int, default=32)
ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--bounces", type=int, default=5); ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--counters", action="store_true")
ap.add_argument("--opt-flags", type=int, default=0, help="extra pt_opts.flags (4 = PT_FLAG_NO_GRIDS)")
a = ap.parse_args()
import torch
torch.cuda.init(); torch.zeros(1, device='cuda')
pta = entry.load_package()
scene = pta.HostScene.generate_ps5(a.tris, 0, a.flags)
g = pta.GpuScene(scene, 0)
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in g.info().as_dict().items()}))
prof = pta.Profile.make(a.width, a.height, a.spp, a.bounces)
rgb = torch.zeros(a.width * a.height * 3, dtype=torch.uint8, device="cuda")
acc = torch.zeros(a.width * a.height * 3, dtype=torch.float32, device="cuda")
for r in range(a.reps):
    g.render_device(prof, pta.Opts.make(flags=pta.PT_FLAG_TIMING | a.opt_flags), rgb.data_ptr(), acc.data_ptr(), 0)
    t = g.timing().as_dict()
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in t.items()}))
print("Msamples/s", a.width * a.height * a.spp / t["total_ms"] / 1e3)
if a.counters:
    g.render_device(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS | a.opt_flags), rgb.data_ptr(), acc.data_ptr(), 0)
    torch.cuda.synchronize()
    print(json.dumps(g.counters().as_dict()))
