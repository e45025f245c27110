"""Two sub-pipelines per GPU on CU-masked streams: does one pipeline's drain phase run under the
other's steady state?

One rank's share of a frame (shard r of N, default the whole frame) is cut once more into two disjoint tile sets - shards r and
r + N of 2N: the same tiles, pt_opts' stripe rule has the same stride for N and 2N - rendered by TWO scenes (own queues, own side
streams) at the same time:
    one        one scene, shard r of N                                          (what ships)
    two        two scenes on two plain streams, shards r and r + N of 2N        (round 2's attempt: every persistent grid fills the chip)
    two-half   ... their streams and the scenes' side streams CU-masked to the lower / the upper half of the CU numbers, grids for 128 CUs
    two-even   ... masked to the even / the odd CU numbers
    two-quarter-grid   two plain streams, persistent grids of half the slots each (PT_WF_BLOCKS_PER_CU), no masks
`--skew-ms x` delays the second pipeline of every frame by x ms (a spin on its stream) so that its phases fall under the other's.
Prints ms per frame (both pipelines complete) for each mode.

    python tools/cu_mask_pipelines.py [--shards 8] [--frames 12] [--skew-ms 0 2 4] [workload flags as tools/shard_times.py]
"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, '.')
import os
os.environ.setdefault('PT_ESCAPE_AFTER', '0')   # (a measurement: the escape masks from the scene's first frame, not its third)
ap = argparse.ArgumentParser()
ap.add_argument('--shards', type=int, default=8)
ap.add_argument('--rank', type=int, default=0)
ap.add_argument('--frames', type=int, default=12)
ap.add_argument('--skew-ms', type=float, nargs='*', default=[0.0])
ap.add_argument('--modes', nargs='*', default=['one', 'two', 'two-half', 'two-even'])
ap.add_argument('--spp', type=int, default=128)
ap.add_argument('--tris', type=int, default=500000)
ap.add_argument('--width', type=int, default=1920)
ap.add_argument('--height', type=int, default=1080)
ap.add_argument('--bounces', type=int, default=5)
ap.add_argument('--tonemap', default='FILMIC')
ap.add_argument('--scene-flags', type=int, default=8)
a = ap.parse_args()

import torch
torch.zeros(1, device='cuda')
import __graft_entry__ as e
pta = e.load_package()
lib = pta.gpu_lib()
host = pta.HostScene.generate_ps5(a.tris, 0, a.scene_flags)
prof = pta.Profile.make(a.width, a.height, a.spp, a.bounces, a.tonemap)
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
words = (n_cu + 31) // 32


def mask_of(pred):
    m = (C.c_uint32 * words)()
    for cu in range(n_cu):
        if pred(cu):
            m[cu // 32] |= 1 << (cu % 32)
    return m


def buffers(opts):
    npx = len(pta.local_pixel_map(prof, opts))
    return (torch.empty(npx * 3, dtype=torch.uint8, device='cuda'), torch.empty(npx * 3, dtype=torch.float32, device='cuda'))


def spin(stream_handle, ms):
    if ms <= 0:
        return
    with torch.cuda.stream(torch.cuda.ExternalStream(stream_handle)):
        torch.cuda._sleep(int(ms * 1e-3 * 2.4e9))      # (cycles of the ~2.4 GHz shader clock; the skew need not be exact)


def measure(mode, skew):
    N = a.shards
    if mode == 'one':
        pipes = [(pta.GpuScene(host, 0), pta.Opts.make(shard_rank=a.rank, shard_count=N, tile_w=32, tile_h=32), None)]
    else:
        masks = {'two': (None, None), 'two-quarter-grid': (None, None),
                 'two-half': (mask_of(lambda c: c < n_cu // 2), mask_of(lambda c: c >= n_cu // 2)),
                 'two-even': (mask_of(lambda c: c % 2 == 0), mask_of(lambda c: c % 2 == 1))}[mode]
        pipes = []
        for k in range(2):
            g = pta.GpuScene(host, 0)
            if masks[k] is not None:
                pta.check_gpu(lib.pt_scene_set_cu_mask(g.handle, masks[k], words))
            pipes.append((g, pta.Opts.make(shard_rank=a.rank + k * N, shard_count=2 * N, tile_w=32, tile_h=32), masks[k]))
    streams, bufs, owned = [], [], []
    for g, opts, mask in pipes:
        if mask is not None:
            h = C.c_void_p()
            pta.check_gpu(lib.pt_stream_create_cu_mask(0, mask, words, C.byref(h)))
            streams.append(h.value)
            owned.append(h.value)
        else:
            st = torch.cuda.Stream()
            owned.append(st)
            streams.append(st.cuda_stream)
        bufs.append(buffers(opts))

    def frame(k):
        for i, (g, opts, _) in enumerate(pipes):
            if i == 1:
                spin(streams[i], skew)
            g.render_device(prof, opts, bufs[i][0].data_ptr(), bufs[i][1].data_ptr(), streams[i])

    for k in range(3):                      # first frame, planned frame, one more
        frame(k)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.frames):
        frame(k)
        torch.cuda.synchronize()            # (a frame is complete when both pipelines are: what a rank hands to the gather)
    ms_sync = (time.perf_counter() - t0) / a.frames * 1e3
    t0 = time.perf_counter()
    for k in range(a.frames):               # frames back to back: the two pipelines drift apart by themselves
        frame(k)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.frames * 1e3
    for g, _, _ in pipes:
        g.close()
    for h in owned:
        if isinstance(h, int):
            pta.check_gpu(lib.pt_stream_destroy(h))
    return ms_sync, ms


if __name__ == '__main__':
    print(f"{a.width}x{a.height} x {a.spp} spp, {a.bounces} bounces, scene flags {a.scene_flags}; shard {a.rank} of {a.shards}; {n_cu} CUs", flush=True)
    for mode in a.modes:
        if mode == 'two-quarter-grid':
            os.environ['PT_WF_BLOCKS_PER_CU'] = os.environ.get('PT_HALF_GRID', '2')
        for skew in (a.skew_ms if mode != 'one' else [0.0]):
            ms_sync, ms = measure(mode, skew)
            print(f"{mode:18s} skew {skew:4.1f} ms: {ms_sync:8.3f} ms per frame one at a time, {ms:8.3f} ms back to back", flush=True)
