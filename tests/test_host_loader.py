"""Host side that 'stays' around the hot path: ISF loader (serde defaults of src/scene/isf.rs),
PNG codec (image crate stand-in), profile.yml parser (src/config/profile.rs), scene generator."""
import ctypes as C
import json

import numpy as np
import pytest

from conftest import SCENES

TRI = [{"position": [0, 0, 0], "normal": [0, 0, 1], "tex_coords": [0, 0]},
       {"position": [1, 0, 0], "normal": [0, 0, 1], "tex_coords": [1, 0]},
       {"position": [0, 1, 0], "normal": [0, 0, 1], "tex_coords": [0, 1]}]
CAMERA = {"transform": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 5, 1]], "fov": 0.7, "zfar": 100.0, "znear": 0.1}


def write_scene(tmp_path, models, lights=(), background=(0.1, 0.2, 0.3), **extra):
    doc = {"models": models, "camera": CAMERA, "lights": list(lights), "background": list(background)}
    doc.update(extra)
    p = tmp_path / "scene.isf"
    p.write_text(json.dumps(doc))
    return p


def test_material_serde_defaults(pta, tmp_path):
    """isf.rs:83-138: missing emissive -> [0,0,0]; emissive without factor -> [1,1,1]; missing metalness -> 0,
    present without factor -> 1; missing opacity/roughness -> 1; ior default 1; albedo factor default [1,1,1]."""
    models = [
        {"type": "Mesh", "triangles": [TRI], "material": {"albedo": {}}},
        {"type": "Mesh", "triangles": [TRI, TRI], "material": {"albedo": {"factor": [0.5, 0.25, 0.125]}, "emissive": {},
                                                             "metalness": {}, "opacity": {"factor": 0.5},
                                                             "roughness": {"factor": 0.25}, "ior": 1.5,
                                                             "unknown_key": {"nested": [1, 2, {"x": None}]}}},
        {"material": {"albedo": {"factor": [1, 1, 1], "texture": None}, "normal_texture": None},
         "center": [1, 2, 3], "radius": 2.5, "type": "Sphere"},  # tag after the other keys
    ]
    s = pta.HostScene.load_isf(write_scene(tmp_path, models, ignored_top_level=[1, 2, 3]))
    d = s.desc.contents
    assert (d.n_models, d.n_materials, d.n_triangles, s.n_prims) == (3, 3, 3, 4)
    m0, m1, m2 = d.materials[0], d.materials[1], d.materials[2]
    assert list(m0.albedo) == [1, 1, 1] and list(m0.emissive) == [0, 0, 0]
    assert (m0.opacity, m0.metalness, m0.roughness, m0.ior) == (1, 0, 1, 1)
    assert list(m1.albedo) == [0.5, 0.25, 0.125] and list(m1.emissive) == [1, 1, 1]
    assert (m1.opacity, m1.metalness, m1.roughness, m1.ior) == (0.5, 1, 0.25, 1.5)
    assert all(t == -1 for t in (m0.tex_albedo, m0.tex_emissive, m0.tex_opacity, m0.tex_metalness, m0.tex_roughness, m0.tex_normal))
    assert d.models[1].tri_first == 1 and d.models[1].tri_count == 2
    assert d.models[2].kind == pta.PT_MODEL_SPHERE and d.models[2].radius == 2.5 and list(d.models[2].center) == [1, 2, 3]
    assert list(d.background) == pytest.approx([0.1, 0.2, 0.3])
    assert list(d.camera.transform)[12:15] == [0, 0, 5]


def test_f32_values_go_through_f64(pta, tmp_path):
    tri = json.loads(json.dumps(TRI))
    tri[0]["position"] = [0.24095196, 1e-45, 16777217.0]
    s = pta.HostScene.load_isf(write_scene(tmp_path, [{"type": "Mesh", "triangles": [tri], "material": {"albedo": {}}}]))
    got = np.ctypeslib.as_array(s.desc.contents.triangles, (24,))[:3]
    assert np.array_equal(got, np.array([0.24095196, 1e-45, 16777217.0], np.float64).astype(np.float32))


@pytest.mark.parametrize("mutate,needle", [
    (lambda d: d.pop("lights"), "lights"),
    (lambda d: d.pop("background"), "background"),
    (lambda d: d["models"][0].pop("material"), "material"),
    (lambda d: d["models"][0]["material"].pop("albedo"), "albedo"),
    (lambda d: d["models"][0].__setitem__("type", "Cube"), "Cube"),
    (lambda d: d["models"][0]["triangles"][0][0].pop("normal"), "normal"),
    (lambda d: d["camera"].pop("fov"), "fov"),
])
def test_loader_errors(pta, tmp_path, mutate, needle):
    doc = {"models": [{"type": "Mesh", "triangles": [json.loads(json.dumps(TRI))], "material": {"albedo": {}}}],
           "camera": dict(CAMERA), "lights": [], "background": [0, 0, 0]}
    mutate(doc)
    p = tmp_path / "bad.isf"
    p.write_text(json.dumps(doc))
    with pytest.raises(pta.PtError) as e:
        pta.HostScene.load_isf(p)
    assert needle in str(e.value)


def test_loader_syntax_and_io_errors(pta, tmp_path):
    p = tmp_path / "broken.isf"
    p.write_text('{"models": [')
    with pytest.raises(pta.PtError):
        pta.HostScene.load_isf(p)
    with pytest.raises(pta.PtError):
        pta.HostScene.load_isf(tmp_path / "missing.isf")
    q = write_scene(tmp_path, [{"type": "Mesh", "triangles": [TRI], "material": {"albedo": {"texture": "nope.png"}}}])
    with pytest.raises(pta.PtError) as e:
        pta.HostScene.load_isf(q)
    assert "Invalid path" in str(e.value)  # texture_bank.rs:26


def test_reference_scenes_load(pta, scene_cache):
    expect = {"cube": (1, 12), "reflection": (2, 1932), "head": (1, 2492), "spheres": (25, 0),
              "alpha_transparency": (8, 56), "white_furnace_indirect": (25, 0), "white_furnace_direct": (9, 108)}
    for name, (n_models, n_tris) in expect.items():
        d = scene_cache(name).desc.contents
        assert (d.n_models, d.n_triangles) == (n_models, n_tris), name
    head = scene_cache("head").desc.contents
    assert head.n_textures == 2 and head.n_lights == 2
    assert (head.textures[0].width, head.textures[0].height, head.textures[0].channels) == (1024, 1024, 3)
    assert head.textures[1].channels == 1


def test_png_decode_matches_pil(pta):
    from PIL import Image
    for rel, ch in (("head/albedo_tex_0.png", 3), ("head/alpha_tex_0.png", 1), ("alpha_transparency/albedo_tex_0.png", 3),
                    ("head/albedo_tex_0.png", 1)):
        w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        path = SCENES / rel
        pta.check_host(pta.host_lib().pth_png_read(str(path).encode(), ch, C.byref(w), C.byref(h), C.byref(px)))
        got = np.ctypeslib.as_array(px, (h.value, w.value, ch)).copy()
        pta.host_lib().pth_free(px)
        img = Image.open(path)
        if ch == 3:
            ref = np.asarray(img.convert("RGB"))
            assert np.array_equal(got, ref), rel
        elif img.mode in ("L", "P", "1"):
            assert np.array_equal(got[..., 0], np.asarray(img.convert("L"))), rel
        else:  # image 0.25 rgb -> luma: (2126 r + 7152 g + 722 b) / 10000
            rgb = np.asarray(img.convert("RGB")).astype(np.uint32)
            ref = ((2126 * rgb[..., 0] + 7152 * rgb[..., 1] + 722 * rgb[..., 2]) // 10000).astype(np.uint8)
            assert np.array_equal(got[..., 0], ref), rel


def test_png_write_read_round_trip(pta, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    path = tmp_path / "rt.png"
    pta.check_host(pta.host_lib().pth_png_write_rgb8(str(path).encode(), 53, 37, img.ctypes.data))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(path)), img)
    w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    pta.check_host(pta.host_lib().pth_png_read(str(path).encode(), 3, C.byref(w), C.byref(h), C.byref(px)))
    assert np.array_equal(np.ctypeslib.as_array(px, (37, 53, 3)), img)
    pta.host_lib().pth_free(px)


def test_png_other_formats_via_pil(pta, tmp_path):
    """Palette, grey+alpha, RGBA, 16-bit and sub-byte PNGs decode like `image` (unpinned by the reference)."""
    from PIL import Image
    rng = np.random.default_rng(1)
    rgb = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    cases = {"p.png": Image.fromarray(rgb).quantize(16), "la.png": Image.fromarray(rgb).convert("LA"),
             "rgba.png": Image.fromarray(rgb).convert("RGBA"), "one.png": Image.fromarray(rgb).convert("1")}
    for name, im in cases.items():
        im.save(tmp_path / name)
        w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        pta.check_host(pta.host_lib().pth_png_read(str(tmp_path / name).encode(), 3, C.byref(w), C.byref(h), C.byref(px)))
        got = np.ctypeslib.as_array(px, (9, 11, 3)).copy()
        pta.host_lib().pth_free(px)
        assert np.array_equal(got, np.asarray(im.convert("RGB"))), name


def test_png_trns_becomes_alpha(pta, tmp_path):
    """tRNS - per-entry alphas of a palette, or the one transparent grey / colour - comes out as the alpha channel of an
    RGBA decode, as the image crate's into_rgba8 delivers it (the glTF converter splits it into alpha_tex_N.png,
    gltf.rs:27-45).  PIL is the yardstick."""
    from PIL import Image
    rng = np.random.default_rng(2)
    pal = Image.fromarray(rng.integers(0, 256, (7, 5, 3), dtype=np.uint8)).quantize(8)
    pal.save(tmp_path / "pal.png", transparency=bytes([0, 60, 120, 255, 200, 255, 17, 99]))
    grey = Image.fromarray(rng.integers(0, 4, (7, 5), dtype=np.uint8) * 85)
    grey.save(tmp_path / "grey.png", transparency=85)
    rgb = Image.fromarray(rng.integers(0, 2, (7, 5, 3), dtype=np.uint8) * 255)
    rgb.save(tmp_path / "rgb.png", transparency=(255, 0, 255))
    for name in ("pal.png", "grey.png", "rgb.png"):
        w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        pta.check_host(pta.host_lib().pth_png_read(str(tmp_path / name).encode(), 4, C.byref(w), C.byref(h), C.byref(px)))
        got = np.ctypeslib.as_array(px, (7, 5, 4)).copy()
        pta.host_lib().pth_free(px)
        want = np.asarray(Image.open(tmp_path / name).convert("RGBA"))
        assert np.array_equal(got, want), name
        assert (got[..., 3] < 255).any() and (got[..., 3] > 0).any(), name


def test_jpeg_decode_matches_libjpeg(pta):
    """JPEG textures (`image::open` decodes them for the reference: ISF textures, texture_bank.rs:33,49, and the images
    of a glTF file, gltf.rs:27-45).  A JPEG decode is only specified up to the accuracy of the inverse DCT, so the
    yardstick is the decoder everybody ships: PIL's libjpeg-turbo with its defaults (slow-integer IDCT, fancy chroma
    upsampling) - bit for bit, baseline and progressive, 4:4:4 / 4:2:2 / 4:2:0 / greyscale, odd sizes, into rgb, rgba
    and luma."""
    import io
    from PIL import Image
    rng = np.random.default_rng(0)
    lib = pta.host_lib()

    def decode(data, want):
        w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        pta.check_host(lib.pth_png_decode(data, len(data), want, C.byref(w), C.byref(h), C.byref(px)))
        a = np.ctypeslib.as_array(px, (h.value, w.value, want)).copy()
        lib.pth_free(px)
        return a
    n = 0
    for (W, H) in ((64, 48), (37, 29), (8, 8), (1, 1), (17, 3), (130, 75)):
        y, x = np.mgrid[0:H, 0:W]
        img = np.stack([(x * 4 + y * 2) % 256, (y * 5) % 256, (x * y) % 256], -1)
        img = (img // 2 + rng.integers(0, 128, (H, W, 3))).astype(np.uint8)
        for mode in ("RGB", "L"):
            im = Image.fromarray(img).convert(mode)
            for sub in ((0, 1, 2) if mode == "RGB" else (0,)):
                for prog in (False, True):
                    for q in (30, 90):
                        b = io.BytesIO()
                        kw = dict(quality=q, progressive=prog)
                        if mode == "RGB":
                            kw["subsampling"] = sub
                        im.save(b, "JPEG", **kw)
                        data = b.getvalue()
                        ref = Image.open(io.BytesIO(data))
                        assert np.array_equal(decode(data, 3), np.asarray(ref.convert("RGB"))), (W, H, mode, sub, prog, q)
                        n += 1
    assert n == 96
    # rgba: opaque; luma of a greyscale file: the samples themselves; of a colour file: the integer weights of png_codec
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", quality=80)
    rgba = decode(b.getvalue(), 4)
    assert (rgba[..., 3] == 255).all() and np.array_equal(rgba[..., :3], decode(b.getvalue(), 3))
    rgb = decode(b.getvalue(), 3).astype(np.uint32)
    assert np.array_equal(decode(b.getvalue(), 1)[..., 0], (2126 * rgb[..., 0] + 7152 * rgb[..., 1] + 722 * rgb[..., 2]) // 10000)
    b = io.BytesIO()
    Image.fromarray(img).convert("L").save(b, "JPEG", quality=80)
    assert np.array_equal(decode(b.getvalue(), 1)[..., 0], np.asarray(Image.open(io.BytesIO(b.getvalue()))))


def test_jpeg_errors_are_errors(pta):
    """Truncated, damaged and unsupported files: an error code and a message, never a crash."""
    import io
    from PIL import Image
    lib = pta.host_lib()
    b = io.BytesIO()
    Image.fromarray(np.arange(64 * 64 * 3, dtype=np.uint8).reshape(64, 64, 3)).save(b, "JPEG", quality=85)
    good = b.getvalue()
    w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()

    def rc(data):
        r = lib.pth_png_decode(data, len(data), 3, C.byref(w), C.byref(h), C.byref(px))
        if r == 0:
            lib.pth_free(px)
        return r
    assert rc(good) == 0
    assert rc(good[:20]) != 0 and b"JPEG" in lib.pth_last_error()
    assert rc(good[:2] + b"\xff\xc9" + good[4:]) != 0                       # arithmetic coding (SOF9) in place of APP0
    sof = good.index(b"\xff\xc0")
    assert rc(good[:sof + 4] + b"\x0c" + good[sof + 5:]) != 0 and b"12-bit" in lib.pth_last_error()
    assert rc(good[:sof + 5] + b"\xff\xff\xff\xff" + good[sof + 9:]) != 0   # 65535 x 65535 pixels claimed by a 1 KB file
    # 16-bit quantiser tables with the largest entries: coefficient x 65535 reaches 2^31 - the DC-only shortcut of the inverse
    # DCT must not overflow an int (round-3 advisory); decodes to something, or fails cleanly
    dqt = good.index(b"\xff\xdb")
    seg_len = int.from_bytes(good[dqt + 2:dqt + 4], "big")
    n_tables = (seg_len - 2) // 65
    wide = b"".join(bytes([0x10 | t]) + b"\xff\xff" * 64 for t in range(n_tables))
    crafted = good[:dqt + 2] + (2 + len(wide)).to_bytes(2, "big") + wide + good[dqt + 2 + seg_len:]
    rc(crafted)
    # a DC predictor walked out of the 16-bit coefficient range is an error (it used to be a signed overflow): every Huffman
    # coded DC difference replaced by the largest one is the quickest way there - scan bytes set to 0xfe
    sos = good.index(b"\xff\xda")
    hdr = int.from_bytes(good[sos + 2:sos + 4], "big")
    rc(good[:sos + 2 + hdr] + b"\xfe" * (len(good) - sos - 2 - hdr - 2) + b"\xff\xd9")
    rng = np.random.default_rng(3)
    for _ in range(300):   # random damage decodes to something or fails cleanly
        d = bytearray(good)
        for k in rng.integers(2, len(d), 6):
            d[k] = rng.integers(0, 256)
        rc(bytes(d))


def test_isf_scene_with_a_jpeg_texture(pta, oracle, tmp_path):
    """An ISF material may name a .jpg (texture_bank.rs:33: image::open decodes whatever it finds)."""
    import json
    from PIL import Image
    img = (np.indices((16, 16)).sum(0) * 8 % 256).astype(np.uint8)
    Image.fromarray(np.stack([img, 255 - img, img // 2], -1)).save(tmp_path / "wood.jpg", quality=92)
    tri = [{"position": p, "normal": [0, 0, 1], "tex_coords": uv} for p, uv in (([-1, -1, 0], [0, 0]), ([1, -1, 0], [1, 0]), ([0, 1, 0], [0.5, 1]))]
    doc = {"models": [{"type": "Mesh", "triangles": [tri], "material": {"albedo": {"texture": "wood.jpg"}}}],
           "camera": {"transform": [[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 3, 1]], "fov": 0.8, "zfar": 100.0, "znear": 0.1},
           "lights": [{"type": "Point", "position": [0, 0, 2], "color": [30, 30, 30], "size": 0.1}], "background": [0.1, 0.1, 0.1]}
    (tmp_path / "scene.isf").write_text(json.dumps(doc))
    scene = pta.HostScene.load_isf(tmp_path / "scene.isf")
    d = scene.desc.contents
    assert d.n_textures == 1 and d.textures[0].width == 16 and d.textures[0].channels == 3
    texels = np.ctypeslib.as_array(d.texels, (d.n_texel_bytes,))[:16 * 16 * 3].reshape(16, 16, 3)
    assert np.array_equal(texels, np.asarray(Image.open(tmp_path / "wood.jpg").convert("RGB")))
    rgb, _, st = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(pta.Profile.make(32, 24, 2, 1))
    assert st["numeric_errors"] == 0 and len(np.unique(rgb.reshape(-1, 3), axis=0)) > 10


def test_texture_cache_and_sharing(pta, tmp_path):
    from PIL import Image
    Image.fromarray(np.full((4, 4, 3), 200, np.uint8)).save(tmp_path / "t.png")
    mat = {"albedo": {"texture": "t.png"}, "opacity": {"texture": "./t.png"}, "emissive": {"texture": "t.png"}}
    s = pta.HostScene.load_isf(write_scene(tmp_path, [{"type": "Mesh", "triangles": [TRI], "material": mat},
                                                      {"type": "Mesh", "triangles": [TRI], "material": mat}]))
    d = s.desc.contents
    assert d.n_textures == 2  # one rgb + one luma entry for the same canonical path (texture_bank.rs)
    assert d.materials[0].tex_albedo == d.materials[1].tex_albedo == d.materials[0].tex_emissive
    assert d.materials[0].tex_opacity == d.materials[1].tex_opacity != d.materials[0].tex_albedo
    tex = d.textures[d.materials[0].tex_opacity]
    assert tex.channels == 1 and d.texels[tex.offset] == 200


# ----------------------------------------------------------------------------- profile.yml
def test_profile_defaults(pta):
    p = pta.load_profile(None)
    assert (p.width, p.height, p.samples, p.bounces, p.brdf, p.tonemap) == (1920, 1080, 64, 4, 0, pta.PT_TONEMAP_FILMIC)
    p = pta.load_profile(text="")
    assert (p.width, p.height, p.samples, p.bounces) == (1920, 1080, 64, 4)


def test_profile_readme_example(pta):
    text = """resolution: # Resolution of the output image
  width: 1280
  height: 720
samples: 128 # Number of sample ray throw by pixel
bounces: 5 # Maximum number of bounces per sample
brdf: COOK_TORRANCE # Which brdf to use
tonemap: ACES # Which color tone map to use
"""
    p = pta.load_profile(text=text)
    assert (p.width, p.height, p.samples, p.bounces, p.tonemap) == (1280, 720, 128, 5, pta.PT_TONEMAP_ACES)


def test_profile_variants(pta, tmp_path):
    p = pta.load_profile(text="---\nresolution: {width: 800, height: 600}\ntonemap: 'REINHARD'\nunknown: 3\nother:\n  nested: 1\n")
    assert (p.width, p.height, p.tonemap, p.samples) == (800, 600, pta.PT_TONEMAP_REINHARD, 64)
    f = tmp_path / "p.yml"
    f.write_text("samples: 7\n")
    assert pta.load_profile(f).samples == 7
    for bad in ("resolution:\n  width: 5\n", "tonemap: BLUE\n", "samples: -3\n", "brdf: PHONG\n", "samples: many\n"):
        with pytest.raises(pta.PtError):
            pta.load_profile(text=bad)
    with pytest.raises(pta.PtError):
        pta.load_profile(tmp_path / "missing.yml")


# ----------------------------------------------------------------------------- generator
def test_generator_is_deterministic_and_round_trips(pta, tmp_path):
    a = pta.HostScene.generate_ps5(6000, 0)
    b = pta.HostScene.generate_ps5(6000, 0)
    c = pta.HostScene.generate_ps5(6000, 1)
    n = a.n_triangles
    assert 0.8 * 6000 <= n <= 1.2 * 6000
    ta = np.ctypeslib.as_array(a.desc.contents.triangles, (n * 24,))
    assert np.array_equal(ta, np.ctypeslib.as_array(b.desc.contents.triangles, (n * 24,)))
    assert not np.array_equal(ta, np.ctypeslib.as_array(c.desc.contents.triangles, (c.n_triangles * 24,))[: n * 24])
    a.save_isf(tmp_path / "gen")
    r = pta.HostScene.load_isf(tmp_path / "gen" / "scene.isf")
    assert r.n_triangles == n
    assert np.array_equal(ta.view(np.uint32), np.ctypeslib.as_array(r.desc.contents.triangles, (n * 24,)).view(np.uint32))
    for i in range(a.desc.contents.n_materials):
        ma, mr = a.desc.contents.materials[i], r.desc.contents.materials[i]
        assert bytes(ma)[:40] == bytes(mr)[:40]
    t = pta.HostScene.generate_ps5(3000, 0, flags=1)  # translucent shells + checker opacity texture
    assert t.desc.contents.n_textures == 1 and t.desc.contents.textures[0].channels == 1
    t.save_isf(tmp_path / "gen_alpha")
    rt = pta.HostScene.load_isf(tmp_path / "gen_alpha" / "scene.isf")
    nt = int(rt.desc.contents.n_texel_bytes)
    assert np.array_equal(np.ctypeslib.as_array(t.desc.contents.texels, (nt,)), np.ctypeslib.as_array(rt.desc.contents.texels, (nt,)))


def test_png_with_a_lying_header_is_rejected(pta, tmp_path):
    """A damaged IHDR must not drive the decoder's allocation (found by tools/fuzz_host.py): 60000 x 60000 pixels
    claimed over a few hundred bytes of image data is an error, not a 10 GB buffer."""
    import os, struct, zlib
    rgb = (np.arange(8 * 8 * 3) % 251).astype(np.uint8).reshape(8, 8, 3)
    path = tmp_path / "ok.png"
    lib = pta.host_lib()
    pta.check_host(lib.pth_png_write_rgb8(os.fsencode(str(path)), 8, 8, rgb.ctypes.data))
    data = bytearray(path.read_bytes())
    assert data[12:16] == b"IHDR"
    data[16:24] = struct.pack(">II", 60000, 60000)
    data[29:33] = struct.pack(">I", zlib.crc32(bytes(data[12:29])))   # keep the chunk CRC valid
    w, h, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    rc = lib.pth_png_decode(bytes(data), len(data), 3, C.byref(w), C.byref(h), C.byref(px))
    assert rc != 0 and b"cannot hold" in lib.pth_last_error()


def test_textured_scene_survives_the_isf_round_trip(pta, oracle, tmp_path):
    """Generator flag bit 1: normal map + emissive / metalness / roughness / albedo textures.  Written as ISF + PNG
    files and loaded back (isf.rs serde model, texture_bank.rs into_rgb8 / into_luma8), the scene must render to the
    same bits with the oracle - the loader and the oracle's texture paths (material.rs:132-214, hit.rs:55-82) agree."""
    import numpy as np
    scene = pta.HostScene.generate_ps5(6000, seed=0, flags=3)
    d = scene.desc.contents
    kinds = set()
    for i in range(d.n_materials):
        m = d.materials[i]
        kinds |= {k for k in ("tex_albedo", "tex_emissive", "tex_opacity", "tex_metalness", "tex_roughness", "tex_normal")
                  if getattr(m, k) >= 0}
    assert len(kinds) == 6
    scene.save_isf(tmp_path / "textured")
    loaded = pta.HostScene.load_isf(tmp_path / "textured" / "scene.isf")
    prof = pta.Profile.make(96, 54, 3, 4, "ACES")
    rgb_a, acc_a, st_a = oracle.OracleScene(scene.desc, oracle.PTO_BVH).render(prof)
    rgb_b, acc_b, st_b = oracle.OracleScene(loaded.desc, oracle.PTO_BVH).render(prof)
    assert st_a == st_b and st_a["numeric_errors"] == 0
    assert np.array_equal(acc_a.view(np.uint32), acc_b.view(np.uint32)) and np.array_equal(rgb_a, rgb_b)
    # the oracle's G-buffer shows the maps: bumps in the normals, stripes in the metalness, dots in the emissive plane
    planes = oracle.OracleScene(loaded.desc, oracle.PTO_BVH).debug_render(160, 90)
    assert len(np.unique(planes["normal"].reshape(-1, 3), axis=0)) > 200
    assert len(np.unique(planes["metalness"])) >= 3 and len(np.unique(planes["roughness"])) > 20
    assert planes["emissive"].any()
    # and the textures change the image
    plain = pta.HostScene.generate_ps5(6000, seed=0, flags=1)
    rgb_p, _, _ = oracle.OracleScene(plain.desc, oracle.PTO_BVH).render(prof)
    assert not np.array_equal(rgb_p, rgb_a)


def test_deeply_nested_json_is_a_parse_error_not_a_crash(pta, tmp_path):
    """An unknown key whose value is a million nested arrays: the reader's recursion is bounded (PT_ERR_PARSE = -3)."""
    deep = tmp_path / "deep.isf"
    deep.write_text('{"junk":' + "[" * 1000000 + "]" * 1000000 + ',"models":[],"lights":[],"background":[0,0,0]}')
    with pytest.raises(pta.PtError) as e:
        pta.HostScene.load_isf(deep)
    assert e.value.code == -3 and "nested too deeply" in str(e.value)
    ok = tmp_path / "ok.isf"   # 200 levels are fine
    cam = '{"transform":[[1,0,0,0],[0,1,0,0],[0,0,1,0],[0,0,4,1]],"fov":0.8,"zfar":100.0,"znear":0.1}'
    ok.write_text('{"junk":' + "[" * 200 + "]" * 200 + f',"models":[],"camera":{cam},"lights":[],"background":[0,0,0]}}')
    assert pta.HostScene.load_isf(ok).n_triangles == 0


def test_bad_palette_png_does_not_leak(pta):
    """A palette PNG whose pixels index past the palette: PT_ERR_PARSE, and the decoded-so-far buffer is released
    (repeated many times the process does not grow)."""
    import resource, struct, zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    w = h = 512
    raw = b"".join(b"\x00" + b"\x05" * w for _ in range(h))           # index 5 everywhere, palette has 2 entries
    png = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3, 0, 0, 0)) +
           chunk(b"PLTE", bytes([0, 0, 0, 255, 255, 255])) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    import ctypes as C
    lib = pta.host_lib()
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for _ in range(400):                                               # 400 x 768 KiB would be 300 MiB if leaked
        ww, hh, px = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
        rc = lib.pth_png_decode(png, len(png), 3, C.byref(ww), C.byref(hh), C.byref(px))
        assert rc == -3 and b"palette" in lib.pth_last_error()
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert after - before < 100 * 1024                                 # KiB
