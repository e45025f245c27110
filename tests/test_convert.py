"""`path-tracer convert` (src/scene/gltf.rs:146-265): glTF 2.0 -> ISF.  The reference has no test for its converter
and builds on the easy-gltf crate (not in the tree), so this is the converter checked against the glTF 2.0
specification and the mapping gltf.rs spells out (lights, material channels, texture files): a small scene written
here as .gltf (base64 buffers, external PNG) and as .glb must come out as the expected ISF, load, and render."""
import base64
import json
import struct
import subprocess
import zlib

import numpy as np
import pytest

from conftest import ROOT

CLI = ROOT / "path-tracer_amd" / "path-tracer"


def png_rgba(w, h, fn):
    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    raw = b"".join(b"\x00" + bytes(c for x in range(w) for c in fn(x, y)) for y in range(h))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 6, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


def jpeg_bytes(w, h, fn, **kw):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(np.array([[fn(x, y) for x in range(w)] for y in range(h)], np.uint8)).save(b, "JPEG", **kw)
    return b.getvalue()


def build_gltf(embed_png, jpeg=False):
    """One quad (indexed, u16) under a translated + scaled node, one non-indexed triangle under a rotated child,
    a perspective camera, a point / a spot / a directional light."""
    quad_pos = np.array([[-1, 0, -1], [1, 0, -1], [1, 0, 1], [-1, 0, 1]], np.float32)
    quad_nrm = np.array([[0, 2, 0]] * 4, np.float32)                      # not unit: the converter normalises
    quad_uv = np.array([[0, 0], [65535, 0], [65535, 65535], [0, 65535]], np.uint16)   # normalised u16
    quad_idx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    tri_pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    blob = b""
    views = []

    def view(data):
        nonlocal blob
        while len(blob) % 4:
            blob += b"\x00"
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)})
        blob += data
        return len(views) - 1
    v_pos, v_nrm, v_uv, v_idx, v_tri = (view(a.tobytes()) for a in (quad_pos, quad_nrm, quad_uv, quad_idx, tri_pos))
    base = png_rgba(4, 2, lambda x, y: (10 * x, 100 + y, 200, 50 * x + y))          # rgb -> albedo, a -> alpha
    mr = png_rgba(2, 2, lambda x, y: (9, 40 + x, 80 + y, 255))                      # g -> roughness, b -> metalness
    nrm = png_rgba(2, 2, lambda x, y: (128, 128, 255, 255))
    if jpeg:   # what real assets ship: JPEG base colour (no alpha: opaque), progressive JPEG metallic-roughness
        base = jpeg_bytes(16, 8, lambda x, y: (10 * x, 100 + 8 * y, 200 - 5 * x), quality=90, subsampling=2)
        mr = jpeg_bytes(8, 8, lambda x, y: (9, 40 + 20 * x, 80 + 10 * y), quality=95, progressive=True)
    images = []
    extra = {}
    for name, data in (("base.jpg" if jpeg else "base.png", base), ("mr.jpg" if jpeg else "mr.png", mr), ("n.png", nrm)):
        if embed_png:
            images.append({"bufferView": view(data), "mimeType": "image/jpeg" if data[:2] == b"\xff\xd8" else "image/png"})
        else:
            images.append({"uri": name})
            extra[name] = data
    doc = {
        "asset": {"version": "2.0"},
        "scenes": [{"nodes": [0, 2, 3, 4, 5]}],
        "nodes": [
            {"mesh": 0, "translation": [0, -1, 0], "scale": [2, 1, 2], "children": [1]},
            {"mesh": 1, "rotation": [0, 0, 0.7071068, 0.7071068], "translation": [0, 1, 0]},   # 90 deg about z
            {"camera": 0, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5, 1.0, 6.0, 1]},
            {"extensions": {"KHR_lights_punctual": {"light": 0}}, "translation": [1, 4, 2]},
            {"extensions": {"KHR_lights_punctual": {"light": 1}}, "translation": [-3, 5, 0]},
            {"extensions": {"KHR_lights_punctual": {"light": 2}}, "rotation": [-0.7071068, 0, 0, 0.7071068]},  # -z -> -y
        ],
        "meshes": [
            {"primitives": [{"attributes": {"POSITION": 0, "NORMAL": 1, "TEXCOORD_0": 2}, "indices": 3, "material": 0}]},
            {"primitives": [{"attributes": {"POSITION": 4}}]},
        ],
        "accessors": [
            {"bufferView": v_pos, "componentType": 5126, "count": 4, "type": "VEC3"},
            {"bufferView": v_nrm, "componentType": 5126, "count": 4, "type": "VEC3"},
            {"bufferView": v_uv, "componentType": 5123, "normalized": True, "count": 4, "type": "VEC2"},
            {"bufferView": v_idx, "componentType": 5123, "count": 6, "type": "SCALAR"},
            {"bufferView": v_tri, "componentType": 5126, "count": 3, "type": "VEC3"},
        ],
        "materials": [{"pbrMetallicRoughness": {"baseColorFactor": [0.5, 0.6, 0.7, 0.8], "metallicFactor": 0.25,
                                                "roughnessFactor": 0.75, "baseColorTexture": {"index": 0},
                                                "metallicRoughnessTexture": {"index": 1}},
                       "emissiveFactor": [0.1, 0.2, 0.3], "normalTexture": {"index": 2}}],
        "textures": [{"source": 0}, {"source": 1}, {"source": 2}],
        "images": images,
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.05, "zfar": 80.0, "aspectRatio": 1.5}}],
        "extensions": {"KHR_lights_punctual": {"lights": [
            {"type": "point", "color": [1, 0.5, 0.25], "intensity": 40},
            {"type": "spot", "intensity": 10, "spot": {}},
            {"type": "directional", "color": [0.2, 0.4, 0.6], "intensity": 3}]}},
        "bufferViews": views,
    }
    return doc, blob, extra


def check_isf(pta, out_dir):
    isf = json.loads((out_dir / "scene.isf").read_text())
    assert isf["background"] == [0.0, 0.0, 0.0]
    cam = isf["camera"]
    assert cam["fov"] == pytest.approx(0.7) and cam["znear"] == pytest.approx(0.05) and cam["zfar"] == pytest.approx(80.0)
    assert cam["transform"][3] == [0.5, 1.0, 6.0, 1.0]
    kinds = [(l["type"], l.get("size")) for l in isf["lights"]]
    assert kinds == [("Point", pytest.approx(0.1)), ("Point", pytest.approx(0.1)), ("Directional", None)]   # spot -> point
    assert isf["lights"][0]["position"] == [1.0, 4.0, 2.0] and isf["lights"][0]["color"] == pytest.approx([40, 20, 10])
    assert isf["lights"][1]["color"] == pytest.approx([10, 10, 10])
    assert isf["lights"][2]["direction"] == pytest.approx([0, -1, 0], abs=1e-6)
    assert isf["lights"][2]["color"] == pytest.approx([0.6, 1.2, 1.8])
    quad, tri = isf["models"]
    assert quad["type"] == "Mesh" and len(quad["triangles"]) == 2 and len(tri["triangles"]) == 1
    # world space: scale (2, 1, 2) then translate (0, -1, 0)
    p = np.array([[v["position"] for v in t] for t in quad["triangles"]])
    assert np.allclose(p[0], [[-2, -1, -2], [2, -1, -2], [2, -1, 2]]) and np.allclose(p[1][2], [-2, -1, 2])
    assert all(np.allclose(v["normal"], [0, 1, 0]) for t in quad["triangles"] for v in t)        # normalised
    assert quad["triangles"][0][1]["tex_coords"] == pytest.approx([1.0, 0.0])
    # child: rotate 90 deg about z, translate (0, 1, 0), then the parent's scale and translation
    q = np.array([v["position"] for v in tri["triangles"][0]])
    assert np.allclose(q, [[0, 0, 0], [0, 1, 0], [-2, 0, 0]], atol=1e-5)
    assert all(v["normal"] == [0.0, 0.0, 0.0] and v["tex_coords"] == [0.0, 0.0] for v in tri["triangles"][0])
    m = quad["material"]
    assert m["albedo"]["factor"] == pytest.approx([0.5, 0.6, 0.7]) and m["opacity"]["factor"] == pytest.approx(0.8)
    assert m["metalness"]["factor"] == pytest.approx(0.25) and m["roughness"]["factor"] == pytest.approx(0.75)
    assert m["emissive"]["factor"] == pytest.approx([0.1, 0.2, 0.3]) and m["ior"] == 1.0
    assert m["albedo"]["texture"] == "albedo_tex_0.png" and m["opacity"]["texture"] == "alpha_tex_0.png"
    assert {m["metalness"]["texture"], m["roughness"]["texture"]} == {"gray_tex_0.png", "gray_tex_1.png"}
    assert m["normal_texture"] == "vec_tex_0.png" and m["emissive"]["texture"] is None
    d = tri["material"]          # no glTF material: the specification's defaults
    assert d["albedo"]["factor"] == [1.0, 1.0, 1.0] and d["metalness"]["factor"] == 1.0 and d["albedo"]["texture"] is None
    # texture contents: rgb / alpha of the base colour image, blue / green of the metallic-roughness image
    lib = pta.host_lib()
    import ctypes as C

    def read(name, ch):
        w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        pta.check_host(lib.pth_png_read(str(out_dir / name).encode(), ch, C.byref(w), C.byref(h), C.byref(px)))
        a = np.ctypeslib.as_array(px, (h.value, w.value, ch)).copy()
        lib.pth_free(px)
        return a
    albedo, alpha = read("albedo_tex_0.png", 3), read("alpha_tex_0.png", 1)
    assert albedo.shape == (2, 4, 3) and tuple(albedo[1, 3]) == (30, 101, 200) and alpha[1, 3, 0] == 151
    metal, rough = read(m["metalness"]["texture"], 1), read(m["roughness"]["texture"], 1)
    assert metal[1, 0, 0] == 81 and rough[0, 1, 0] == 41
    # the result is a valid scene for the loader
    scene = pta.HostScene.load_isf(out_dir / "scene.isf")
    assert scene.n_triangles == 3 and scene.n_lights == 3
    return scene


def test_convert_gltf_with_external_files(pta, oracle, tmp_path):
    doc, blob, extra = build_gltf(embed_png=False)
    doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    src = tmp_path / "in"
    src.mkdir()
    (src / "scene.gltf").write_text(json.dumps(doc))
    for name, data in extra.items():
        (src / name).write_bytes(data)
    out = tmp_path / "out"
    pta.convert_gltf(src / "scene.gltf", out)
    scene = check_isf(pta, out)
    rgb, acc, st = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(pta.Profile.make(48, 32, 2, 2))
    assert st["numeric_errors"] == 0 and rgb.any()


def write_jpeg_gltf(tmp_path):
    doc, blob, extra = build_gltf(embed_png=False, jpeg=True)
    doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    src = tmp_path / "in"
    src.mkdir()
    (src / "scene.gltf").write_text(json.dumps(doc))
    for name, data in extra.items():
        (src / name).write_bytes(data)
    return src / "scene.gltf", extra


def test_convert_gltf_with_jpeg_textures(pta, oracle, tmp_path):
    """JPEG images (gltf.rs:27-45 goes through easy-gltf / image, which decode them): the textures written are the
    decoded pixels, PIL's decode being the yardstick; a JPEG has no alpha, so the alpha texture is all 255."""
    import io
    from PIL import Image
    gltf, extra = write_jpeg_gltf(tmp_path)
    out = tmp_path / "out"
    pta.convert_gltf(gltf, out)
    isf = json.loads((out / "scene.isf").read_text())
    m = isf["models"][0]["material"]
    albedo = np.asarray(Image.open(out / m["albedo"]["texture"]).convert("RGB"))
    assert np.array_equal(albedo, np.asarray(Image.open(io.BytesIO(extra["base.jpg"])).convert("RGB")))
    assert (np.asarray(Image.open(out / m["opacity"]["texture"]).convert("L")) == 255).all()
    mr = np.asarray(Image.open(io.BytesIO(extra["mr.jpg"])).convert("RGB"))
    assert np.array_equal(np.asarray(Image.open(out / m["metalness"]["texture"]).convert("L")), mr[..., 2])     # blue -> metalness
    assert np.array_equal(np.asarray(Image.open(out / m["roughness"]["texture"]).convert("L")), mr[..., 1])     # green -> roughness
    scene = pta.HostScene.load_isf(out / "scene.isf")
    rgb, _, st = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(pta.Profile.make(48, 32, 2, 2))
    assert st["numeric_errors"] == 0 and rgb.any()


@pytest.mark.gpu
@pytest.mark.parametrize("jpeg", [False, True])
def test_converted_scene_renders_like_the_oracle(pta, oracle, tmp_path, jpeg):
    """convert -> load -> render on the MI355X: image and f32 accumulation equal the oracle's bit for bit, on the
    grid path, the KD-only path and the megakernel; the G-buffer planes too."""
    if jpeg:
        gltf, _ = write_jpeg_gltf(tmp_path)
    else:
        doc, blob, extra = build_gltf(embed_png=False)
        doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
        (tmp_path / "in").mkdir()
        gltf = tmp_path / "in" / "scene.gltf"
        gltf.write_text(json.dumps(doc))
        for name, data in extra.items():
            (tmp_path / "in" / name).write_bytes(data)
    out = tmp_path / "out"
    pta.convert_gltf(gltf, out)
    scene = pta.HostScene.load_isf(out / "scene.isf")
    g = pta.GpuScene(scene)
    prof = pta.Profile.make(192, 128, 8, 4)
    o_rgb, o_acc, st = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(prof)
    assert st["numeric_errors"] == 0
    for flags in (0, pta.PT_FLAG_NO_GRIDS, pta.PT_FLAG_MEGAKERNEL):
        rgb, acc = g.render(prof, pta.Opts.make(flags=flags))
        assert np.array_equal(acc.view(np.uint32), o_acc.view(np.uint32)) and np.array_equal(rgb, o_rgb), flags
    assert len(np.unique(o_rgb.reshape(-1, 3), axis=0)) > 10
    got = g.debug_render(192, 128)
    ref = oracle.OracleScene(scene.desc, oracle.PTO_BVH).debug_render(192, 128)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k


def test_convert_glb_through_the_cli(pta, tmp_path):
    doc, blob, _ = build_gltf(embed_png=True)
    doc["buffers"] = [{"byteLength": len(blob)}]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    blob += b"\x00" * (-len(blob) % 4)
    glb = (b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)) + struct.pack("<II", len(js), 0x4E4F534A) + js +
           struct.pack("<II", len(blob), 0x004E4942) + blob)
    (tmp_path / "scene.glb").write_bytes(glb)
    out = tmp_path / "a" / "b" / "converted"      # create_dir_all (gltf.rs:152): the missing parents too
    r = subprocess.run([str(CLI), "convert", str(tmp_path / "scene.glb"), str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    check_isf(pta, out)


def test_convert_errors(pta, tmp_path):
    doc, blob, _ = build_gltf(embed_png=True)
    doc["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]

    def attempt(mutate, message):
        d = json.loads(json.dumps(doc))
        mutate(d)
        (tmp_path / "bad.gltf").write_text(json.dumps(d))
        r = subprocess.run([str(CLI), "convert", str(tmp_path / "bad.gltf"), str(tmp_path / "o")], capture_output=True, text=True)
        assert r.returncode == 2 and message in r.stderr, r.stderr   # every error -> message + exit code 2 (main.rs:14-22)
    attempt(lambda d: d["scenes"][0]["nodes"].remove(2), "No camera found")                     # gltf.rs:163-166
    attempt(lambda d: d.update(scenes=[]), "No scenes found in gltf file")                     # gltf.rs:159-161
    attempt(lambda d: d["cameras"][0].update(type="orthographic"), "Orthographic camera not supported")
    attempt(lambda d: d["meshes"][0]["primitives"][0].update(mode=5), "not a triangle list")
    attempt(lambda d: d["accessors"][3].update(count=600), "beyond its buffer")
    # size arithmetic that would wrap around in size_t: 2^60 elements of stride 16 from offset 4 sum to 0 (mod 2^64)
    def wrap(d):
        d["accessors"][0].update(count=2 ** 60, byteOffset=4)
        d["bufferViews"][d["accessors"][0]["bufferView"]].update(byteStride=16)
    attempt(wrap, "is not a valid size")
    def wrap53(d):   # the same below 2^53, where the field itself is a valid integer
        d["accessors"][0].update(count=2 ** 52, byteOffset=4)
        d["bufferViews"][d["accessors"][0]["bufferView"]].update(byteStride=4096)
    attempt(wrap53, "beyond its buffer")
    attempt(lambda d: d["accessors"][0].update(byteOffset=-8), "is not a valid size")
    attempt(lambda d: d["accessors"][0].update(count=2.5), "is not a valid size")
    attempt(lambda d: d["bufferViews"][d["accessors"][0]["bufferView"]].update(byteStride=2), "smaller than its")
    attempt(lambda d: d["bufferViews"][d["images"][0]["bufferView"]].update(byteOffset=2 ** 52, byteLength=2 ** 52), "image reaches beyond")
    (tmp_path / "file").write_text("x")
    r = subprocess.run([str(CLI), "convert", str(tmp_path / "bad.gltf"), str(tmp_path / "file")], capture_output=True, text=True)
    assert r.returncode == 2 and "is not a directory" in r.stderr                               # gltf.rs:153-155
    r = subprocess.run([str(CLI), "convert", str(tmp_path / "bad.gltf")], capture_output=True, text=True)
    assert r.returncode == 2 and "<OUTPUT>" in r.stderr
