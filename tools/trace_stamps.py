import sys
sys.path.insert(0,'/root/repo')
import torch; torch.zeros(1,device='cuda')
import __graft_entry__ as e
pta=e.load_package()
arg = sys.argv[1] if len(sys.argv)>1 else "500000"
sc = pta.HostScene.generate_ps5(int(arg),0) if arg.isdigit() else pta.HostScene.load_isf(arg)
g=pta.GpuScene(sc,0)
prof=pta.Profile.make(1920,1080,16,5)
g.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
c=g.counters().as_dict(); print({k:c[k] for k in ("samples","segments","trace_nodes","trace_tris")}, g.info().as_dict()["n_kd_nodes"])
