import sys; sys.path.insert(0,'.')
import torch; torch.cuda.init()
import __graft_entry__ as e
pta = e.load_package()
for nb in (1<<28, 1<<30, 2<<30, 8<<30):
    print(nb>>20, "MiB", round(pta.measure_copy_bandwidth(0, nb, 5),1), "GB/s")
a=torch.empty(2<<30,dtype=torch.uint8,device='cuda'); b=torch.empty_like(a)
s=torch.cuda.Event(enable_timing=True); t=torch.cuda.Event(enable_timing=True)
b.copy_(a); s.record(); b.copy_(a); t.record(); torch.cuda.synchronize(); print("torch copy", round(2*(2<<30)/s.elapsed_time(t)/1e6,1), "GB/s")
