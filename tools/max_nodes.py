import sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
sc = pta.HostScene.generate_ps5(500000, 0)
g = pta.GpuScene(sc, 0)
for bounces in (0, 1, 5):
    prof = pta.Profile.make(1920, 1080, 2, bounces, "FILMIC")
    g.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
    c = g.counters().as_dict()
    print(bounces, {k: c[k] for k in ("segments", "max_nodes_per_cast", "casts_over_1k_nodes", "trace_nodes")})
