#!/usr/bin/env python3
"""Headline benchmark: Msamples/s of the path-tracing hot path on N MI355X.

Workload (BASELINE.json config 3, the one the metric is quoted on): the deterministic PS5
stand-in scene (500 k triangles, seed 0; the real PS5 ISF is not in the reference) in the
FRAMING OF THE REFERENCE'S OWN RENDER (readme/ps5_b5_s128.png: 38.8 % of its 8x8 pixel blocks are
sky; generator flag 8, host/scene_gen.cpp), 1920x1080, 128 spp, 5 bounces, Cook–Torrance, FILMIC.
The line also carries `config.legacy_framing`: the same measurement on the framing rounds 1-3
reported (52.5 % sky blocks), taken in the same run.  A "step" is one complete render of that frame:
every rank renders its interleaved 32x32 tiles (global pixel index in the seed formula, so
the image is bit-identical for any N), then — for N > 1 — one RCCL all-gather of the packed
u8 framebuffer slices and a scatter into the row-major image on every rank.  The scene, KD-tree,
origin grids and textures are resident in HBM before the timed region starts.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts the N ranks itself (one process per GPU,
torch.distributed.run, RCCL) before this process makes any GPU call, and exits with their status.

Rank 0 prints ONE JSON line (contract in the task statement) carrying `roofline` (dominant
kernel, HIP-event timed inside the timed region) and, at N=1, `cpu_baseline` (the CPU oracle —
a port, all host cores — on a bounded sample of the same workload).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def bytes_per_sample(c, spp):
    """SURVEY §8-d algorithmic bytes per path sample from exact work counters."""
    n = max(1, c["samples"])
    r_seg, r_sh = c["segments"] / n, c["shadow_rays"] / n
    v, t, h = c["nodes_visited"] / n, c["tris_tested"] / n, c["shaded_hits"] / n
    floor = r_seg * 160 + r_sh * 104 + 15.0 / spp
    ceiling = floor + 8 * v + 36 * t + 160 * h
    return floor, ceiling, dict(R_seg=r_seg, R_sh=r_sh, V=v, T=t, H=h)


def kernel_source_sha16():
    """Hash of the device sources: the PMC record of profiles/latest_traffic.json is only quoted for the code it was taken from."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted((ROOT / "path-tracer_amd" / "csrc").glob("*")):
        if f.suffix in (".h", ".hip"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def count_gpus():
    """GPUs of this node from the kernel driver's topology files (/sys/class/kfd): no GPU runtime is loaded, let alone
    initialised, by the parent that only starts the ranks.  A *_VISIBLE_DEVICES list narrows the count."""
    n = 0
    try:
        for node in sorted(Path("/sys/class/kfd/kfd/topology/nodes").iterdir()):
            props = dict(line.split(None, 1) for line in (node / "properties").read_text().splitlines() if " " in line)
            if int(props.get("simd_count", "0")) > 0:     # (CPU nodes have no SIMDs)
                n += 1
    except OSError:
        n = 0
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""])) if n else len([x for x in v.split(",") if x.strip() != ""])
    return n


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one per GPU) as children, before this
    process touches the GPU (it never does), and hand their exit status on."""
    have = count_gpus()
    if have < args.gpus:   # (no topology files in this container: ask the runtime - counting devices does not initialise them)
        import torch
        have = max(have, torch.cuda.device_count())
    if have < args.gpus and "PT_BENCH_DEVICE" not in os.environ:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--tris", type=int, default=500_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=128)
    ap.add_argument("--bounces", type=int, default=5)
    ap.add_argument("--tonemap", default="FILMIC")
    ap.add_argument("--scene-flags", type=int, default=8,
                    help="generator flags: 1 = translucent shells (config 5), 2 = textures, 4 = closed room, 8 = the framing of "
                         "the reference's PS5 render (default; 0 = the framing of rounds 1-3)")
    ap.add_argument("--no-legacy", action="store_true", help="skip the second measurement on the rounds 1-3 framing")
    ap.add_argument("--opt-flags", type=int, default=0, help="extra pt_opts.flags (4 = PT_FLAG_NO_GRIDS: KD-tree only)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="CPU baseline budget (0 = skip)")
    ap.add_argument("--no-counters", action="store_true")
    ap.add_argument("--save-png", default=None)
    ap.add_argument("--backend", default="nccl", help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 path with several ranks on ONE GPU: set PT_BENCH_DEVICE=0)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if "PT_BENCH_DEVICE" in os.environ:  # rehearsal only: several ranks on one GPU (gloo backend)
        local_rank = int(os.environ["PT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: {dist.get_world_size()} ranks came up, --gpus {args.gpus} asked for")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")  # where collective buffers live

    pta = entry.load_package()
    lib = pta.gpu_lib()

    # ---- scene (resident before the timed region)
    t0 = time.time()
    scene = pta.HostScene.generate_ps5(args.tris, 0, args.scene_flags)
    gscene = pta.GpuScene(scene, device=local_rank)
    info = gscene.info().as_dict()
    setup_s = time.time() - t0

    prof = pta.Profile.make(args.width, args.height, args.spp, args.bounces, args.tonemap)
    tile = 32
    opts = pta.Opts.make(flags=pta.PT_FLAG_TIMING | args.opt_flags, device=local_rank, shard_rank=rank,
                         shard_count=world, tile_w=tile, tile_h=tile)
    n_local = int(lib.pt_local_pixel_count(C.byref(prof), C.byref(opts)))
    npix = args.width * args.height
    slice_pixels = n_local
    if world > 1:
        t = torch.tensor([n_local], device=cdev, dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        slice_pixels = int(t.item())
    rgb_local = torch.zeros(slice_pixels * 3, dtype=torch.uint8, device=dev)
    acc_local = torch.zeros(max(1, n_local) * 3, dtype=torch.float32, device=dev)
    gathered = torch.zeros(world * slice_pixels * 3, dtype=torch.uint8, device=dev) if world > 1 else None
    image = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    stage_ms = {k: 0.0 for k in ("generate_ms", "trace_ms", "shade_ms", "shadow_ms", "accumulate_ms", "postprocess_ms",
                                 "integrate_ms", "total_ms", "bounce0_ms")}
    launches = {"launches": 0, "stage_launches": 0, "bounce0_launches": 0}

    # HIP events around every launch (PT_FLAG_TIMING: ~36 event records and one host synchronisation per frame): on every
    # timed step of the one-GPU run; on the FIRST timed step only when the frame is sharded - a rank's frame is a few
    # milliseconds there, and the other steps are enqueued without a host round trip, as a renderer would run them
    opts_plain = pta.Opts.make(flags=args.opt_flags, device=local_rank, shard_rank=rank, shard_count=world,
                               tile_w=tile, tile_h=tile)
    ev_steps = max(1, args.steps if world == 1 else 1)

    # the frame of this rank and - N > 1 - the exchange, timed with events on the launch stream (every timed step)
    ev_render = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                 for _ in range(args.steps)]
    step_no = [-1]

    def step(with_events=True):
        ks = step_no[0]   # timed step number, -1 during the warm-up
        if ks >= 0:
            ev_render[ks][0].record()
        gscene.render_device(prof, opts if with_events else opts_plain, rgb_local.data_ptr(), acc_local.data_ptr(), stream)
        if ks >= 0:
            ev_render[ks][1].record()
        if with_events:
            tm = gscene.timing().as_dict()
            for k in stage_ms:
                stage_ms[k] += tm[k]
            for k in launches:
                launches[k] += tm[k]
        if world > 1:
            if args.backend == "nccl":
                dist.all_gather_into_tensor(gathered, rgb_local)   # RCCL over xGMI, one exchange per frame
            else:
                parts = [torch.empty(slice_pixels * 3, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, rgb_local.cpu())
                gathered.copy_(torch.cat(parts))
            pta.check_gpu(lib.pt_assemble_tiles(C.byref(prof), world, tile, tile, slice_pixels, 3,
                                                gathered.data_ptr(), image.data_ptr(), stream))
        if ks >= 0:
            ev_render[ks][2].record()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # (the FIRST frame of a configuration runs in fixed-budget chunks and counts what every bounce produces; the frame after it
    # allocates the queues those counts ask for - pt_gpu.hip "frame plan".  Steady state is what the metric is quoted on: a
    # warm-up of at least four frames, each complete before the next is planned; --warmup 0 times the cold frames
    # (... and a scene builds its escape masks when it is about to render its third frame - a one-shot render is better off
    # without them -, after which the plan is made again: frames 1-2 without masks, 3 with masks counting, 4 planned)
    warm_frames = 0 if args.warmup == 0 else max(4, args.warmup)
    for _ in range(warm_frames):
        step()
        torch.cuda.synchronize()
    for k in stage_ms:
        stage_ms[k] = 0.0
    for k in launches:
        launches[k] = 0
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step_no[0] = k
        step(with_events=k < ev_steps)
    fence()
    elapsed = time.perf_counter() - t0
    q_info = gscene.info().as_dict()
    info.update({k: q_info[k] for k in ("escape_build_seconds", "escape_prims", "escape_clear_fraction", "device_bytes")})   # (built at frame 3)
    render_ms = sum(a.elapsed_time(b) for a, b, _ in ev_render) / args.steps     # this rank's frame (device time)
    gather_ms = sum(b.elapsed_time(c) for _, b, c in ev_render) / args.steps if world > 1 else 0.0
    per_rank = None
    if world > 1:
        mine = torch.tensor([render_ms, gather_ms], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[round(float(x[0]), 3), round(float(x[1]), 3)] for x in every]
    if world > 1:
        t = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_samples = npix * args.spp * args.steps
    value = total_samples / elapsed / 1e6

    # ---- roofline of the dominant kernel, this rank's shard.  Algorithmic bytes are SURVEY §8-d's per-unit terms for
    # the work the kernel does: 160 B per path segment (64 B ray + 16 B hit record, written and read), 104 B per shadow
    # ray, 8 B per KD node visited, 36 B per primitive tested, 160 B per shaded hit (96 B vertex attributes + 64 B
    # material record) - whatever the implementation keeps in registers or caches instead of moving it.
    roofline = None
    # the camera-grid cull of the timed frames (rank 0's tiles): blocks whose samples are the background without RNG or cast
    cull_blocks, cull_empty = gscene.cull_stats()
    counters = None
    if not args.no_counters:
        copts = pta.Opts.make(flags=pta.PT_FLAG_COUNTERS | args.opt_flags, device=local_rank, shard_rank=rank,
                              shard_count=world, tile_w=tile, tile_h=tile)
        gscene.render_device(prof, copts, rgb_local.data_ptr(), acc_local.data_ptr(), stream)
        torch.cuda.synchronize()
        counters = gscene.counters().as_dict()
        floor_b, ceil_b, per = bytes_per_sample(counters, args.spp)
        n_items = n_local * args.spp
        trace_launches = max(1, launches["launches"])
        kernels = {}
        masked = counters.get("masked_casts", 0)
        if launches["bounce0_launches"]:
            # The fused bounce-0 kernel (k_wf_shade<GRID >= 2>): ChaCha block, camera cast, shading, shadow casts.  Its wavefronts
            # do nothing at all for the samples of culled 8x8 blocks (k_cam_block_mask; the counters come from the instrumented
            # variant, which does not cull - but a culled block has no hits, no primitive tests and no shadow rays to count).
            per_frame = launches["bounce0_launches"] // ev_steps
            n_culled = cull_empty * 64 * args.spp if world == 1 else 0
            n_live = n_items - n_culled
            # SURVEY 8-d's terms for ALL samples (what the reference does for them) ...
            b0_alg_all = (n_items * 160 + counters["bounce0_shadow_rays"] * 104 + counters["bounce0_tris"] * 36
                          + counters["bounce0_hits"] * 160 + n_items * 12)
            # ... and what the fused kernel has to MOVE for the samples it touches: the ray, hit and shadow records stay in
            # registers, so what is left is the primitives tested (36 B + the 8-byte list entry that named them), the attribute and
            # material records of the shaded hits, two grid-cell words per cast, and the output - a 12-byte sample, or for a path
            # that goes on the 64-byte record + 16 B of RNG words (at most one per hit that no escape mask ended, at most the
            # casts the later bounces really made)
            survivors = max(0, min(counters["bounce0_hits"] - counters.get("bounce0_masked", 0),
                                   counters["segments"] - n_items - masked))
            b0_moved = (counters["bounce0_tris"] * 44 + counters["bounce0_hits"] * 160
                        + (n_live + counters["bounce0_shadow_rays"]) * 8 + (n_live - survivors) * 12 + survivors * 80)
            kernels["k_wf_shade<GRID> (bounce 0: ChaCha12 block + camera cast + shading + shadow casts)"] = dict(
                ms_per_frame=stage_ms["bounce0_ms"] / ev_steps, launches_per_frame=per_frame,
                bytes_per_frame=b0_moved, algorithmic_all_samples_per_frame=b0_alg_all,
                units_per_launch=n_live // max(1, per_frame), unit="path samples outside culled blocks",
                note="bytes = what the kernel moves for the samples it touches (records it keeps in registers not charged); "
                     "algorithmic_frac_all_samples prices SURVEY 8-d's per-sample terms for every sample, culled ones included")
        fused = bool(launches["bounce0_launches"])
        trace_segments = counters["segments"] - (n_items if fused else 0) - masked   # (casts an escape mask proved empty are not made)
        # closest-hit primitive tests of the KD casts (trace_tris also holds the camera casts of the fused kernel)
        trace_tris = counters["trace_tris"] - (counters["bounce0_cam_tris"] if fused else 0)
        trace_bytes = trace_segments * 80 + counters["trace_nodes"] * 8 + trace_tris * 36
        kernels["k_wf_trace (closest-hit KD-tree casts)"] = dict(
            ms_per_frame=stage_ms["integrate_ms"] / ev_steps, launches_per_frame=trace_launches // ev_steps,
            bytes_per_frame=trace_bytes, units_per_launch=trace_segments // max(1, trace_launches // ev_steps),
            unit="ray casts")
        # the shadow casts outside the fused kernel (k_og_shadow through the light grids, k_wf_shadow on the KD-tree):
        # 104 B per ray cast + its node visits and primitive tests
        sh_rays = counters["shadow_rays"] - counters["shadow_skipped"] - (counters["bounce0_shadow_rays"] if fused else 0)
        sh_tris = counters["tris_tested"] - counters["trace_tris"] - \
            ((counters["bounce0_tris"] - counters["bounce0_cam_tris"]) if fused else 0)
        sh_nodes = counters["nodes_visited"] - counters["trace_nodes"]
        # (one shadow stage per bounce iteration, i.e. per closest-hit launch or fused bounce-0 launch)
        sh_launches = max(1, (launches["launches"] + launches["bounce0_launches"]) // ev_steps)
        if sh_rays > 0 and stage_ms["shadow_ms"] > 0:
            sh_name = "k_og_shadow (shadow casts through the light grids)" if info["light_grids"] and \
                not (args.opt_flags & pta.PT_FLAG_NO_GRIDS) else "k_wf_shadow (any-hit KD-tree casts)"
            kernels[sh_name] = dict(
                ms_per_frame=stage_ms["shadow_ms"] / ev_steps, launches_per_frame=sh_launches,
                bytes_per_frame=sh_rays * 104 + sh_nodes * 8 + sh_tris * 36,
                units_per_launch=sh_rays // sh_launches, unit="shadow rays",
                # HIP events on the side stream: the launches share the chip with k_wf_trace of the next bounce, so this
                # is elapsed time beside another kernel, not the kernel alone (rocprofv3: profiles/*_kernel_stats.csv)
                note="elapsed on the side stream, overlapped with k_wf_trace")
        # HBM traffic and what bounds the kernel: from the PMC passes (separate rocprofv3 --pmc runs, tools/pmc_profile.sh) of the
        # same workload AND the same device code - profiles/latest_traffic.json carries the hash of csrc/ it was taken from; a
        # record of other code is not quoted (round-3 advisory)
        rec, rec_note = None, None
        tf = ROOT / "profiles" / "latest_traffic.json"
        if tf.exists():
            try:
                cand = json.loads(tf.read_text())
                if cand.get("workload") != [args.tris, args.width, args.height, args.spp, args.bounces, world] or args.opt_flags or \
                        cand.get("scene_flags", 0) != args.scene_flags:
                    rec_note = "profiles/latest_traffic.json is of another workload"
                elif cand.get("kernel_source_sha16") != kernel_source_sha16():
                    rec_note = (f"profiles/latest_traffic.json was taken from other device code "
                                f"({cand.get('kernel_source_sha16')} != {kernel_source_sha16()}): not quoted")
                else:
                    rec = cand
            except Exception as exc:
                rec_note = f"profiles/latest_traffic.json unreadable: {exc}"

        def pmc_of(kname):
            if rec is None:
                return None
            short = kname.split("<")[0].split(" ")[0]   # k_wf_shade / k_wf_trace / k_og_shadow
            per_kernel = rec.get("kernels", {})
            key = next((k for k in per_kernel if k.split("<")[0] == short and ("GRID" in k) == ("<GRID>" in kname)), None)
            return (key, per_kernel[key]) if key else None

        ranked = sorted(kernels.items(), key=lambda kv: -kv[1]["ms_per_frame"])
        name, dom = ranked[0]
        if len(ranked) > 1 and ranked[1][1]["ms_per_frame"] >= 0.98 * dom["ms_per_frame"]:
            # two kernels within 2 % by HIP events: the rocprofv3 total of the committed profile decides
            a_, b_ = pmc_of(ranked[0][0]), pmc_of(ranked[1][0])
            if a_ and b_ and b_[1].get("kernel_ms_profiled", 0) > a_[1].get("kernel_ms_profiled", 0):
                name, dom = ranked[1]
        avg_ms = dom["ms_per_frame"] / max(1, dom["launches_per_frame"])
        bytes_per_launch = dom["bytes_per_frame"] / max(1, dom["launches_per_frame"])
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # wall time of the kernel pipeline (shadow casts may overlap the next trace, so the stages do not add up)
        kernel_total = stage_ms["total_ms"] / ev_steps
        pipeline = ceil_b * n_items / (kernel_total * 1e-3) / 1e9
        traffic = valu_rate = traffic_src = lanes = None
        got = pmc_of(name)
        if got:
            # per launch like `achieved`: the record's bytes per FRAME over this run's launches per frame (the profiled run's
            # first frame also launches the bounces behind the last ray, which the planned frames timed here do not)
            traffic = (got[1]["hbm_bytes_per_frame"] / max(1, dom["launches_per_frame"]) if got[1].get("hbm_bytes_per_frame")
                       else got[1].get("hbm_bytes_per_launch"))
            valu_rate = got[1].get("valu_insts_per_cu_cycle")
            lanes = got[1].get("active_lanes_per_valu_inst")
            traffic_src = f"profiles/latest_traffic.json <- {rec.get('source')} ({got[0]}), device code {rec.get('kernel_source_sha16')}"
        elif rec_note:
            traffic_src = rec_note
        copy_gbs = pta.measure_copy_bandwidth(local_rank, 2 << 30, 5)   # achievable HBM rate on this box
        # What the kernel is bound by, from the PMC passes of the same workload and code (above): the contract's `bound` below
        # stays "hbm" (there is no contraction, so not "mfma"), this says which ceiling the counters show
        traffic_frac_now = traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if traffic else None
        if valu_rate is not None and valu_rate >= 0.8:
            bound_measured = {"bound": "valu_issue", "counter": "SQ_INSTS_VALU / (SQ_BUSY_CYCLES per CU)", "value": valu_rate,
                              "ceiling": 1.0, "frac": round(valu_rate, 3), "active_lanes_per_valu_inst": lanes,
                              "note": "a wave64 vector instruction occupies its SIMD for 4 cycles: 1.0 per CU-cycle is the issue ceiling"}
        elif traffic_frac_now is not None and traffic_frac_now >= 0.6:
            bound_measured = {"bound": "hbm", "counter": "2 x FETCH_SIZE + WRITE_SIZE", "value": round(traffic_frac_now * HBM_PEAK_GBS, 1),
                              "ceiling": HBM_PEAK_GBS, "frac": round(traffic_frac_now, 3)}
        elif valu_rate is not None:
            bound_measured = {"bound": "latency_of_dependent_scattered_fetches", "counter": "SQ_ACTIVE_INST_VALU lanes, SQ_WAIT_INST_ANY",
                              "value": lanes, "ceiling": 64, "frac": round((lanes or 0) / 64.0, 3), "valu_insts_per_cu_cycle": valu_rate,
                              "note": "lanes active per vector instruction: the wavefront waits for its slowest lane's fetch"}
        else:
            bound_measured = None
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "traffic_source": traffic_src,
                    "traffic_GBps": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                    "traffic_frac": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5) if traffic else None,
                    # what the kernel actually runs against (PMC): vector-instruction issue, 1.0 per CU-cycle = the ceiling
                    "valu_insts_per_cu_cycle": valu_rate,
                    "peak_measured_copy": round(copy_gbs, 1), "frac_of_measured_copy": round(achieved / copy_gbs, 5),
                    "bound_measured": bound_measured,
                    "avg_launch_ms": round(avg_ms, 4), "steps_timed_with_events": ev_steps, "launches_per_step": dom["launches_per_frame"],
                    "units_per_launch": dom["units_per_launch"], "unit_name": dom["unit"],
                    "algorithmic_bytes_per_launch": round(bytes_per_launch),
                    "algorithmic_bytes_per_unit": round(bytes_per_launch / max(1, dom["units_per_launch"]), 1),
                    "kernels": {k: dict({"ms_per_step": round(v["ms_per_frame"], 3), "launches_per_step": v["launches_per_frame"],
                                         "algorithmic_GBps": round(v["bytes_per_frame"] / max(1e-9, v["ms_per_frame"] * 1e-3) / 1e9, 1),
                                         "frac": round(v["bytes_per_frame"] / max(1e-9, v["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                                        **({"algorithmic_frac_all_samples": round(v["algorithmic_all_samples_per_frame"] /
                                                                                  max(1e-9, v["ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)}
                                           if "algorithmic_all_samples_per_frame" in v else {}),
                                        **({"note": v["note"]} if "note" in v else {}))
                                for k, v in kernels.items()},
                    "pipeline": {"algorithmic_bytes_per_sample": round(ceil_b, 1),
                                 "queue_floor_bytes_per_sample": round(floor_b, 1),
                                 "kernel_ms_per_step": round(kernel_total, 3),
                                 "achieved_GBps": round(pipeline, 1), "frac": round(pipeline / HBM_PEAK_GBS, 5),
                                 "queue_floor_frac": round(floor_b * n_items / (kernel_total * 1e-3) / 1e9
                                                           / HBM_PEAK_GBS, 5),
                                 "Grays_per_s": round((counters["segments"] + counters["shadow_rays"]
                                                       - counters["shadow_skipped"]) / (kernel_total * 1e-3) / 1e9, 3)},
                    "stage_ms_per_step": {k: round(v / ev_steps, 3) for k, v in stage_ms.items()},
                    "per_sample": {k: round(v, 3) for k, v in per.items()}}

    # ---- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N = 1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        oracle = entry.load_oracle()
        osc = oracle.OracleScene(scene.desc, oracle.PTO_BVH)
        rows_per_band = 2
        done, t_cpu, n_bands = 0, 0.0, 0
        # 2-row bands at low-discrepancy rows over the whole frame until the time budget is used
        while t_cpu < args.cpu_seconds and n_bands < args.height // rows_per_band:
            y0 = int(((n_bands * 0.6180339887498949) % 1.0) * (args.height - rows_per_band))
            begin, end = y0 * args.width, (y0 + rows_per_band) * args.width
            t1 = time.perf_counter()
            osc.render(prof, begin, end, 0)
            t_cpu += time.perf_counter() - t1
            done += (end - begin) * args.spp
            n_bands += 1
        cpu = {"value": round(done / t_cpu / 1e6, 4), "unit": "Msamples/s", "cores": oracle.max_threads(),
               "kind": "port",
               "sample": f"{n_bands} bands of {rows_per_band} rows at golden-ratio rows over the frame "
                         f"({done} of {npix * args.spp} samples, {t_cpu:.1f} s); oracle/pt_oracle.cpp (CPU restatement "
                         f"of the reference algorithm, AABB-tree candidate filter), OpenMP threads = cores; "
                         f"host reports {os.cpu_count()} logical CPUs"}

    if os.environ.get("PT_BENCH_CHECK") and world > 1:
        # rehearsal check: the assembled frame equals an unsharded render on this rank
        full = torch.zeros(npix * 3, dtype=torch.uint8, device=dev)
        gscene.render_device(prof, pta.Opts.make(device=local_rank), full.data_ptr(), None, stream)
        torch.cuda.synchronize()
        if not torch.equal(full, image):
            raise SystemExit(f"rank {rank}: assembled image differs from the unsharded render")
        print(f"rank {rank}: assembled image == unsharded render", file=sys.stderr)
    if args.save_png and rank == 0:
        img = (image if world > 1 else rgb_local[: npix * 3]).cpu().numpy()
        pta.check_host(pta.host_lib().pth_png_write_rgb8(os.fsencode(args.save_png), args.width, args.height,
                                                         img.ctypes.data))

    # ---- the same measurement on the framing rounds 1-3 reported (generator flags 0: 52.5 % of the pixel blocks are sky), so
    # that the two lines of numbers stay comparable across rounds
    legacy = None
    if rank == 0 and world == 1 and args.scene_flags == 8 and not args.no_legacy:
        gscene.close()
        scene0 = pta.HostScene.generate_ps5(args.tris, 0, 0)
        g0 = pta.GpuScene(scene0, device=local_rank)
        for _ in range(max(4, args.warmup)):
            g0.render_device(prof, opts_plain, rgb_local.data_ptr(), acc_local.data_ptr(), stream)
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            g0.render_device(prof, opts_plain, rgb_local.data_ptr(), acc_local.data_ptr(), stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        b0, e0 = g0.cull_stats()
        legacy = {"workload": f"PS5 stand-in {scene0.n_triangles} triangles (generator seed 0, flags 0: the framing of rounds 1-3)",
                  "value": round(npix * args.spp * args.steps / dt / 1e6, 3), "unit": "Msamples/s",
                  "ms_per_step": round(dt / args.steps * 1e3, 3), "background_block_fraction": round(e0 / b0, 4) if b0 else None}
        g0.close()

    dev_name = f"{local_rank}:{torch.cuda.get_device_name(local_rank)}"
    devices = [dev_name]
    if world > 1:
        names = [None] * world
        dist.all_gather_object(names, dev_name)
        devices = names
    if rank == 0:
        out = {
            "metric": "Msamples/sec (WxHxspp/s), PS5 stand-in scene, Cook-Torrance + FILMIC",
            "value": round(value, 3), "unit": "Msamples/s",
            "n_gpus": world if args.backend == "nccl" or world == 1 else len(set(devices)), "n_ranks": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"PS5 stand-in {scene.n_triangles} triangles (generator seed 0, flags "
                                   f"{args.scene_flags}" + (": framing of the reference's readme/ps5_b5_s128.png" if args.scene_flags & 8 else "")
                                   + f"), {args.width}x{args.height}, {args.spp} spp, "
                                   f"{args.bounces} bounces, COOK_TORRANCE, {args.tonemap}",
                       # 8x8 pixel blocks in which no camera ray can hit anything (the reference's PS5 image: 0.388 exactly black blocks)
                       "background_block_fraction": round(cull_empty / cull_blocks, 4) if cull_blocks else None,
                       "legacy_framing": legacy,
                       # proofs instead of casts: the escape masks (csrc/pt_escape.h) and the rays that leave the wavefront walker
                       "escape_masks": {"primitives_with_mask": info["escape_prims"], "clear_cell_fraction": round(info["escape_clear_fraction"], 4),
                                        "build_seconds": round(info["escape_build_seconds"], 3),
                                        "casts_proven_empty": (counters or {}).get("masked_casts"),
                                        "frac_of_casts_of_bounces_ge_1": round((counters["masked_casts"] / max(1, counters["segments"] - counters["samples"])), 4)
                                        if counters else None},
                       "exact_walker_casts": (counters or {}).get("exact_casts"),
                       # device memory of the path queues during the timed frames (sized by an earlier frame's counts when planned)
                       "queues": {"gib": round(q_info["queue_bytes"] / 2**30, 3), "chunk_items": q_info["queue_chunk_items"],
                                  "chunks_per_frame": -(-(args.width * args.height * args.spp) // max(1, q_info["queue_chunk_items"])) if world == 1 else None,
                                  "planned": bool(q_info["frame_planned"]), "warm_frames": warm_frames},
                       "parallelism": f"{world} x tile-sharded (32x32 interleaved)" + (
                           "" if world == 1 else ", RCCL all-gather of u8 framebuffer" if args.backend == "nccl"
                           else f", {args.backend} all-gather through host memory (REHEARSAL on {len(set(devices))} GPU(s), not an RCCL run)"),
                       "kd": {k: info[k] for k in ("n_prims", "n_kd_nodes", "n_kd_leaves", "n_leaf_refs", "kd_depth")},
                       "origin_grids": {k: info[k] for k in ("cam_grid_res", "light_grids", "grid_refs")},
                       "setup_seconds": round(setup_s, 2), "kd_build_seconds": round(info["kd_build_seconds"], 2),
                       "grid_build_seconds": round(info["grid_build_seconds"], 2),
                       "bounce0_cull": {"pixel_blocks_8x8": cull_blocks, "empty": cull_empty,
                                        "frac": round(cull_empty / cull_blocks, 4) if cull_blocks else None,
                                        "note": "blocks of rank 0 no camera ray can hit anything in (all cells of the camera grid under "
                                                "their pixels are empty): their samples are the background, without ChaCha block "
                                                "or cast; PT_CAM_CULL=0 renders them the long way, same bits"}},
            "roofline": roofline,
            "bound_measured": roofline["bound_measured"] if roofline else None,
            "cpu_baseline": cpu,
            "backend": ("rccl (torch.distributed nccl)" if args.backend == "nccl" else args.backend) if world > 1 else "none (one GPU)",
            "devices": devices,
            "per_rank_ms": None if per_rank is None else {
                "render": {"max": max(r[0] for r in per_rank), "min": min(r[0] for r in per_rank), "ranks": [r[0] for r in per_rank]},
                "gather_and_assemble": {"max": max(r[1] for r in per_rank), "min": min(r[1] for r in per_rank)}},
            "render_ms_rank0": round(render_ms, 3),
            # N > 1: the exchange step of the frame - all-gather of the u8 slices + scatter into the row-major image - per frame,
            # the slowest rank's (events on the launch stream); null on one GPU
            "gather_ms": None if per_rank is None else max(r[1] for r in per_rank),
        }
        if counters:
            out["counters"] = counters
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
