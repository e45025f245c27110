"""Kernel timeline of ONE shard's frame (where the fixed per-frame cost of a 1/N shard sits):
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/shard_timeline.py run [--shards 8]
    python tools/shard_timeline.py show gpurun_out/tl [--frames 6]
(workload flags as tools/shard_times.py: --width --height --bounces --tonemap --scene-flags --tris --spp)
'run' renders shard 0 of N six times (plain options, no timing events); 'show' takes the last frame from the trace
and prints every launch with its start offset, duration and the idle gap before it on its stream."""
import argparse, csv, glob, sys
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
ap = argparse.ArgumentParser()
ap.add_argument('what', choices=['run', 'show'])
ap.add_argument('dir', nargs='?')
ap.add_argument('--shards', type=int, default=8)
ap.add_argument('--frames', type=int, default=6)
ap.add_argument('--spp', type=int, default=128)
ap.add_argument('--tris', type=int, default=500000)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
ap.add_argument('--bounces', type=int, default=5)
ap.add_argument('--tonemap', default='FILMIC')
ap.add_argument('--scene-flags', type=int, default=8)
ap.add_argument('--opt-flags', type=int, default=0)
a = ap.parse_args()
if a.what == 'run':
    import torch
    torch.zeros(1, device='cuda')
    import __graft_entry__ as e
    pta = e.load_package()
    g = pta.GpuScene(pta.HostScene.generate_ps5(a.tris, 0, a.scene_flags), 0)
    prof = pta.Profile.make(a.width, a.height, a.spp, a.bounces, a.tonemap)
    opts = pta.Opts.make(flags=a.opt_flags, shard_rank=0, shard_count=a.shards, tile_w=32, tile_h=32)
    npx = len(pta.local_pixel_map(prof, opts))
    rgb = torch.empty(npx * 3, dtype=torch.uint8, device='cuda')
    acc = torch.empty(npx * 3, dtype=torch.float32, device='cuda')
    for _ in range(a.frames):
        g.render_device(prof, opts, rgb.data_ptr(), acc.data_ptr(), 0)
        torch.cuda.synchronize()
    sys.exit(0)
rows = []
for f in glob.glob(a.dir + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:70], r.get('Stream_Id', r.get('Queue_Id', '?'))))
rows.sort()
# frames end with k_postprocess
ends = [i for i, r in enumerate(rows) if 'k_postprocess' in r[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
frame = rows[lo:hi]
t0 = frame[0][0]
last_end = {}
print(f"{'start us':>9} {'dur us':>8} {'gap us':>7}  q  kernel")
busy = 0
for s, e, name, q in frame:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {q}  {name}")
print(f"frame: {(frame[-1][1] - t0) / 1e3:.1f} us, {len(frame)} launches")
