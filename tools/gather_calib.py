"""One k_gather<8> and one k_gather<16> launch with known lane-load counts, for counter calibration under rocprofv3 --pmc."""
import sys
sys.path.insert(0, '.')
import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
for nbytes in (8, 16):
    pta.measure_gather_rate(0, 16 << 10, nbytes, 1024)
print("lane-loads per launch:", n_cu * 8 * 256 * 1024, "(4 launches per size)")
