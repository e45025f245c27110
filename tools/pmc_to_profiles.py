#!/usr/bin/env python3
"""Fold the PMC passes of tools/pmc_profile.sh (full-size bench) into profiles/<tag>_pmc.json and refresh
profiles/latest_traffic.json (read by bench.py).

    python tools/pmc_to_profiles.py gpurun_out/pmc_r04_e r04_e 500000 1920 1080 128 5 1 --scene-flags=8 --frames=3
(--frames: frames rendered under the counters - bench.py renders at least two warm-up frames + the steps; per-frame figures are
the totals over that many frames, per-launch figures count the launches that found work)
"""
import csv, glob, json, sys
from collections import defaultdict
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import kernel_source_sha16   # (the record is only quoted by bench.py for the device code it was taken from)

no_latest = "--no-latest" in sys.argv   # (a workload other than the headline's: profiles/latest_traffic.json stays)
scene_flags, frames = 0, 1
for a in sys.argv:
    if a.startswith("--scene-flags="):
        scene_flags = int(a.split("=")[1])
    if a.startswith("--frames="):   # frames rendered under the counters (bench.py --warmup W --steps K: W + K, at least 2 + K)
        frames = int(a.split("=")[1])
argv = [a for a in sys.argv if a != "--no-latest" and not a.startswith("--scene-flags=") and not a.startswith("--frames=")]
src, tag = argv[1], argv[2]
workload = [int(v) for v in argv[3:9]]
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
fetch_of = defaultdict(dict)   # kernel -> dispatch -> FETCH_SIZE: which launches had work (a launch on an empty queue moves a few KB)
for path in sorted(glob.glob(f"{src}/pass*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if not any(s in k for s in ("k_wf_", "k_og_", "k_accumulate", "k_postprocess")):
            continue
        name = k.split("<")[0].split("(")[0].replace("void ", "")
        if name == "k_wf_shade":   # k_wf_shade<ALPHA, COUNT, PRIMARY, GRID>: the fused bounce-0 kernel is its own row
            targs = k.split("<", 1)[1].split(">", 1)[0].split(",")
            if len(targs) >= 4 and int(targs[3]) >= 2:
                name = "k_wf_shade<GRID=%d>" % int(targs[3])
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "FETCH_SIZE":
            fetch_of[name][r["Dispatch_Id"]] = fetch_of[name].get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
        seen.add((name, r["Dispatch_Id"]))
    for name, _ in seen:
        cnt[(name, path.split("/")[-3])] += 1
out = {}
for name, c in acc.items():
    n = max(v for (nm, _), v in cnt.items() if nm == name)
    fetch_kb, write_kb = c.get("FETCH_SIZE", 0), c.get("WRITE_SIZE", 0)
    top = max(fetch_of[name].values(), default=0.0)
    n_work = sum(1 for v in fetch_of[name].values() if v >= 1e-3 * top) if top > 0 else n
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    out[name] = {
        "dispatches_per_frame": n / frames, "frames_profiled": frames,
        # launches that found work (the first frame of a configuration also launches the bounces behind the last ray)
        "working_dispatches": n_work,
        # MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B request on gfx950 -> doubled; WRITE_SIZE exact
        "hbm_bytes_per_launch": (2 * fetch_kb + write_kb) * 1024 / max(1, n_work),
        "hbm_bytes_per_frame": (2 * fetch_kb + write_kb) * 1024 / frames,
        "FETCH_SIZE_KB_total": fetch_kb, "WRITE_SIZE_KB_total": write_kb,
        "active_lanes_per_valu_inst": round(c.get("SQ_THREAD_CYCLES_VALU", 0) / max(1, c.get("SQ_INSTS_VALU", 1)), 1),
        "wait_pct_of_wave_cycles": round(100 * c.get("SQ_WAIT_ANY", 0) / max(1, c.get("SQ_WAVE_CYCLES", 1)), 1),
        "l2_hit_pct": round(100 * c.get("TCC_HIT_sum", 0) / max(1, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)), 1),
        "l1_hit_pct": round(100 * (1 - c.get("TCP_TCC_READ_REQ_sum", 0) / max(1, c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1))), 1),
        "valu_insts": c.get("SQ_INSTS_VALU", 0) / frames, "salu_insts": c.get("SQ_INSTS_SALU", 0) / frames,
        "branch_insts": c.get("SQ_INSTS_BRANCH", 0) / frames, "vmem_rd_insts": c.get("SQ_INSTS_VMEM_RD", 0) / frames,
        "valu_insts_per_cu_cycle": round(c.get("SQ_INSTS_VALU", 0) / 256 / max(1, cyc), 3),
        "salu_insts_per_cu_cycle": round(c.get("SQ_INSTS_SALU", 0) / 256 / max(1, cyc), 3),
        "kernel_ms_profiled": round(cyc / 2.1e6 / frames, 2),
        # vector-L1 (TCP) accesses per CU and busy cycle: the traversal kernels sit at ~1.1 in every pass, whatever
        # their length - the rate at which scattered lane accesses get through the L1
        "tcp_accesses_per_cu_cycle": round(c.get("TCP_TOTAL_ACCESSES_sum", 0) / 256 / max(1, cyc), 3),
        "tcp_accesses_per_vmem_inst": round(c.get("TCP_TOTAL_ACCESSES_sum", 0) /
                                            max(1, c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)), 1),
    }
# the grid build of pt_scene_create (namespace ogb) runs once per scene in front of the timed region: its own section
setup = {k: out.pop(k) for k in list(out) if k.startswith("ogb::")}
json.dump({"workload": workload, "scene_flags": scene_flags, "kernels": out, "setup_kernels": setup}, open(f"profiles/{tag}_pmc.json", "w"), indent=1)
dominant = max(out, key=lambda k: out[k]["kernel_ms_profiled"])
t = out[dominant]
if not no_latest:
  json.dump({"workload": workload, "scene_flags": scene_flags, "kernel_source_sha16": kernel_source_sha16(), "kernel": dominant, "hbm_bytes_per_launch": round(t["hbm_bytes_per_launch"]),
           "tcp_accesses_per_cu_cycle": t["tcp_accesses_per_cu_cycle"],
           "valu_insts_per_cu_cycle": t["valu_insts_per_cu_cycle"],
           "active_lanes_per_valu_inst": t["active_lanes_per_valu_inst"],
           # every kernel's figures: bench.py reports the one with the largest time in ITS run, which may differ
           "kernels": {k: {"hbm_bytes_per_launch": round(v["hbm_bytes_per_launch"]), "hbm_bytes_per_frame": round(v["hbm_bytes_per_frame"]),
                           "valu_insts_per_cu_cycle": v["valu_insts_per_cu_cycle"],
                           "active_lanes_per_valu_inst": v["active_lanes_per_valu_inst"],
                           "kernel_ms_profiled": v["kernel_ms_profiled"]} for k, v in out.items()},
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_profile.sh); bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE) KiB per MI355X_MICROARCH.md (gfx950 FETCH_SIZE counts 64 B per 128 B "
                     "request); calibration in this pipeline: k_accumulate reads 12 B/sample -> FETCH_SIZE reads 0.48x",
           "source": f"profiles/{tag}_pmc.json"}, open("profiles/latest_traffic.json", "w"), indent=1)
for k, v in out.items():
    print(k, {a: v[a] for a in ("dispatches_per_frame", "hbm_bytes_per_frame", "active_lanes_per_valu_inst",
                                 "wait_pct_of_wave_cycles", "l1_hit_pct", "l2_hit_pct", "kernel_ms_profiled")})
for k, v in setup.items():
    print("(setup)", k, {a: v[a] for a in ("dispatches_per_frame", "hbm_bytes_per_frame", "kernel_ms_profiled")})
print("total HBM bytes per frame: %.1f GB" % (sum(v["hbm_bytes_per_frame"] for v in out.values()) / 1e9))
