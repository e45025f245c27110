"""Mutation fuzzing of the parsers that take untrusted bytes: ISF scene files, PNG / JPEG textures, profile YAML, glTF / GLB.
Every input must come back as a status code (PT_OK or an error with a message), never as a crash.  Meant to run
against the sanitized host library:  bash tools/asan_host.sh builds it; or standalone:

    python tools/fuzz_host.py [iterations] [seed]
"""
import ctypes as C, os, random, shutil, sys, tempfile, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import __graft_entry__ as e
pta = e.load_package()
L = pta.host_lib()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
golden = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"
scenes = sorted(golden.glob("scenes/*/scene.isf")) or sorted(golden.rglob("*.isf"))
pngs = sorted(golden.rglob("*.png"))
assert scenes and pngs, (len(scenes), len(pngs))
# JPEG corpus: baseline + progressive, grey + 4:4:4 / 4:2:0 colour, with restart markers (PIL is the ENcoder only)
import io
import numpy as np
from PIL import Image
_r = np.random.default_rng(7)
_img = np.clip(np.add.outer(np.arange(40) * 5, np.arange(56) * 3)[..., None] + _r.integers(0, 60, (40, 56, 3)), 0, 255).astype(np.uint8)
jpegs = []
for _mode, _kw in (("RGB", dict(subsampling=0)), ("RGB", dict(subsampling=2)), ("L", {}), ("RGB", dict(progressive=True)),
                   ("RGB", dict(subsampling=2, restart_marker_blocks=2))):
    _b = io.BytesIO()
    try:
        Image.fromarray(_img if _mode == "RGB" else _img[..., 0]).save(_b, "JPEG", quality=80, **_kw)
    except TypeError:
        continue
    jpegs.append(_b.getvalue())
# glTF corpus: the small scene of tests/test_convert.py as .gltf (base64 buffer) and as .glb
sys.path.insert(0, str(golden.parent))
import base64, json, struct
import test_convert
_doc, _blob, _ = test_convert.build_gltf(embed_png=True)
_doc["buffers"] = [{"byteLength": len(_blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(_blob).decode()}]
GLTF = json.dumps(_doc).encode()
_doc["buffers"] = [{"byteLength": len(_blob)}]
_js = json.dumps(_doc).encode(); _js += b" " * (-len(_js) % 4); _bin = _blob + b"\x00" * (-len(_blob) % 4)
GLB = (b"glTF" + struct.pack("<II", 2, 12 + 8 + len(_js) + 8 + len(_bin)) + struct.pack("<II", len(_js), 0x4E4F534A) + _js +
       struct.pack("<II", len(_bin), 0x004E4942) + _bin)


def mutate(data: bytes) -> bytes:
    b = bytearray(data)
    kind = rng.randrange(6)
    if kind == 0 and len(b) > 1:                       # truncate
        del b[rng.randrange(1, len(b)):]
    elif kind == 1:                                    # flip random bytes
        for _ in range(rng.randrange(1, 8)):
            b[rng.randrange(len(b))] = rng.randrange(256)
    elif kind == 2:                                    # delete a slice
        i = rng.randrange(len(b)); del b[i:i + rng.randrange(1, 64)]
    elif kind == 3:                                    # duplicate a slice
        i = rng.randrange(len(b)); b[i:i] = b[i:i + rng.randrange(1, 64)]
    elif kind == 4:                                    # insert structural characters / huge numbers
        i = rng.randrange(len(b)); b[i:i] = rng.choice([b"{", b"}", b"[", b"]", b",", b":", b'"', b"1e999", b"-", b"null", b"\\u12"])
    else:                                              # swap two regions
        i, j = rng.randrange(len(b)), rng.randrange(len(b)); b[i], b[j] = b[j], b[i]
    return bytes(b)


ok = err = 0
tmp = pathlib.Path(tempfile.mkdtemp(prefix="ptfuzz"))
try:
    for it in range(iters):
        which = it % 4
        if which == 0:     # ISF: mutated scene next to the original textures
            src = rng.choice(scenes)
            d = tmp / "scene"
            if d.exists(): shutil.rmtree(d)
            shutil.copytree(src.parent, d)
            (d / "scene.isf").write_bytes(mutate(src.read_bytes()))
            h = C.c_void_p()
            rc = L.pth_scene_load_isf(os.fsencode(str(d / "scene.isf")), C.byref(h))
            if rc == 0: L.pth_scene_free(h)
        elif which == 1:   # PNG
            data = mutate(rng.choice(jpegs) if it % 8 >= 4 else rng.choice(pngs).read_bytes())   # the decoder dispatches on the magic
            w, hh, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
            rc = L.pth_png_decode(data, len(data), rng.choice([1, 3]), C.byref(w), C.byref(hh), C.byref(px))
            if rc == 0: L.pth_free(px)
        elif which == 3:   # glTF / GLB -> converter
            name = "in.gltf" if it % 8 < 4 else "in.glb"
            (tmp / name).write_bytes(mutate(GLTF if name.endswith("gltf") else GLB))
            rc = L.pth_convert_gltf(os.fsencode(str(tmp / name)), os.fsencode(str(tmp / "conv")))
        else:              # profile YAML
            text = mutate(b"resolution:\n  width: 64\n  height: 48\nsamples: 4\nbounces: 2\nbrdf: COOK_TORRANCE\ntonemap: FILMIC\n")
            prof = pta.Profile()
            rc = L.pth_profile_parse(text.replace(b"\0", b" "), C.byref(prof))
        ok += rc == 0
        err += rc != 0
finally:
    shutil.rmtree(tmp, ignore_errors=True)
print(f"{iters} mutated inputs: {ok} accepted, {err} rejected with a status code, 0 crashes")
