"""pt_render (host buffers: rgb8 + f32 accumulation copied back over PCIe) against pt_render_device at config 3."""
import sys, time
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
import torch
torch.zeros(1, device='cuda')
import numpy as np
import __graft_entry__ as e
pta = e.load_package()
sc = pta.HostScene.generate_ps5(500000, 0)
g = pta.GpuScene(sc, 0)
prof = pta.Profile.make(1920, 1080, 128, 5, "FILMIC")
n = prof.width * prof.height
g.render(prof)
t0 = time.perf_counter()
for _ in range(3):
    g.render(prof)
host_ms = (time.perf_counter() - t0) / 3 * 1e3
rgb = torch.empty(n * 3, dtype=torch.uint8, device='cuda'); acc = torch.empty(n * 3, dtype=torch.float32, device='cuda')
g.render_device(prof, pta.Opts.make(), rgb.data_ptr(), acc.data_ptr(), 0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    g.render_device(prof, pta.Opts.make(), rgb.data_ptr(), acc.data_ptr(), 0)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / 3 * 1e3
s = n * prof.samples / 1e6
print(f"pt_render (host buffers, PCIe-inclusive): {host_ms:.2f} ms = {s / host_ms * 1e3:.0f} Msamples/s; "
      f"pt_render_device: {dev_ms:.2f} ms = {s / dev_ms * 1e3:.0f} Msamples/s")
