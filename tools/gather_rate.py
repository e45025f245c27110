"""Scattered-load ceiling of the GPU (pt_measure_gather_rate) for table sizes from L1-resident to HBM-resident."""
import sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for nbytes in (8, 16):
    for table in (16 << 10, 1 << 20, 32 << 20, 256 << 20, 4 << 30):
        g = pta.measure_gather_rate(0, table, nbytes, 1024)
        print(f"{nbytes:2d} B loads, table {table / 2**20:8.2f} MiB: {g:8.1f} G lane-loads/s = {g / n_cu / 2.1:.3f} per CU-cycle at 2.1 GHz"
              f" = {g * nbytes / 1e3:.2f} TB/s")
