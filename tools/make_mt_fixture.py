#!/usr/bin/env python3
"""Convert the reference's Möller–Trumbore YAML vectors into a compact fixture.

Input  (read-only reference data files):
    /root/reference/tests/moller_trumbore/hit_tests.yml   (3 012 cases)
    /root/reference/tests/moller_trumbore/miss_tests.yml  (3 012 cases)
Output: tests/golden/moller_trumbore.npz with float64 arrays
    hit_rays [N,6] (position, direction), hit_tris [N,9] (v0,v1,v2), hit_expect [N,3] (dist,u,v)
    miss_rays [M,6], miss_tris [M,9]
The values are kept in float64 exactly as the YAML holds them; the unit test
of the reference narrows them to f32 (src/scene/internal/triangle.rs:137-184).
"""
import sys
from pathlib import Path

import numpy as np
import yaml

SRC = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference/tests/moller_trumbore")
DST = Path(__file__).resolve().parent.parent / "tests" / "golden" / "moller_trumbore.npz"


def load(name):
    with open(SRC / name) as f:
        cases = yaml.load(f, Loader=getattr(yaml, "CSafeLoader", yaml.SafeLoader))
    rays = np.array([c["ray"]["position"] + c["ray"]["direction"] for c in cases], np.float64)
    tris = np.array([c["triangle"]["v0"] + c["triangle"]["v1"] + c["triangle"]["v2"] for c in cases], np.float64)
    hits = [c.get("hit") for c in cases]
    return rays, tris, hits


hr, ht, hh = load("hit_tests.yml")
mr, mt, mh = load("miss_tests.yml")
assert all(h is not None for h in hh) and all(h is None for h in mh)
he = np.array([[h["dist"], h["u"], h["v"]] for h in hh], np.float64)
np.savez_compressed(DST, hit_rays=hr, hit_tris=ht, hit_expect=he, miss_rays=mr, miss_tris=mt)
print(f"{DST}: {len(hr)} hit cases, {len(mr)} miss cases, min dist {he[:, 0].min():.3g}")
