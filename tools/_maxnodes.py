import sys
sys.path.insert(0,'/root/repo')
import torch; torch.zeros(1,device='cuda')
import __graft_entry__ as e
pta=e.load_package()
sc=pta.HostScene.generate_ps5(500000,0); g=pta.GpuScene(sc,0)
prof=pta.Profile.make(1920,1080,4,5)
g.render(prof, pta.Opts.make(flags=pta.PT_FLAG_COUNTERS))
print(g.counters().as_dict())
