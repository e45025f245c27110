// Deterministic synthetic stand-in for the PS5 scene + ISF writer.
//
// The reference's headline image (README.md:15, readme/ps5_b5_s128.png) was
// rendered from a glTF->ISF conversion that is NOT in the repository
// (.gitignore:30-31).  BASELINE configs 3-5 therefore use this generated
// scene: a tessellated ground plane, a glossy dark core, two
// tessellated curved white shells, two emissive strips, one point light,
// black background, camera fov 0.6911112 (as tests/scenes/spheres).  The
// triangle count is a parameter; vertex jitter comes from the same PCG32
// stream rand_core uses for seed expansion, so the scene is a pure function
// of (target_tris, seed, flags).
//
// flags bit 3 (8): the REFERENCE'S FRAMING.  readme/ps5_b5_s128.png is all that is
// known of the reference's config-3 workload: the console stands on a square ground
// seen corner-on from above, 40.5 % of the image's pixels and 38.8 % of its 8x8 pixel
// blocks are exactly black (sky), the ground fills the lower half, the object a
// quarter of the frame, a long shadow runs to the right.  The recipe below - a
// 12.94 x 12.94 ground instead of 24 x 24, the camera at azimuth 0.82 rad, 6.5 above
// the ground, 8.6 from the axis (0.54 of the ground's far corner behind the object),
// looking at (1, 1.87, 0), the light to the camera's left behind the object - was
// fitted (tools/framing_fit.py: random search over the six numbers against the CPU
// oracle's hit mask) to the image's fraction of empty blocks (0.387 vs 0.388) and to
// its profile of empty blocks over 16 bands of rows and of columns (rms deviation
// 0.03).  Without the flag: round 1-3's framing (camera in front of the object, low,
// 52.5 % of the blocks empty).
#include <cmath>
#include <cstring>
#include <fstream>
#include <functional>
#include <memory>
#include <sys/stat.h>

#include "host_common.hpp"

namespace pth {
namespace {

struct Pcg32 {
    uint64_t state;
    explicit Pcg32(uint64_t seed) : state(seed) {}
    uint32_t next() {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        return (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    float uniform() { return (float)(next() >> 8) * (1.0f / 16777216.0f); }
};

struct V3 {
    double x, y, z;
};
V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
V3 norm(V3 a) {
    double l = std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
    return l > 0 ? V3{a.x / l, a.y / l, a.z / l} : V3{0, 1, 0};
}

using Surface = std::function<V3(double s, double t)>;

// Tessellate a parametric patch into 2*ns*nt triangles with analytic
// (finite-difference) smooth normals and per-grid-vertex jitter along the normal.
void add_patch(pth_scene& sc, const Surface& f, uint32_t ns, uint32_t nt, bool wrap_s, Pcg32& rng,
               double jitter, double uv_scale, bool flip) {
    struct GV {
        float p[3], n[3], uv[2];
    };
    std::vector<GV> grid((size_t)(ns + 1) * (nt + 1));
    const double h = 1e-5;
    for (uint32_t j = 0; j <= nt; ++j)
        for (uint32_t i = 0; i <= ns; ++i) {
            double s = (double)i / ns, t = (double)j / nt;
            V3 p = f(s, t);
            V3 ds = f(s + h, t) - f(s - h, t);
            V3 dt = f(s, t + h) - f(s, t - h);
            V3 n = norm(cross(ds, dt));
            if (flip) n = {-n.x, -n.y, -n.z};
            GV& g = grid[(size_t)j * (ns + 1) + i];
            double a = jitter * (2.0 * rng.uniform() - 1.0);
            g.p[0] = (float)(p.x + a * n.x);
            g.p[1] = (float)(p.y + a * n.y);
            g.p[2] = (float)(p.z + a * n.z);
            g.n[0] = (float)n.x;
            g.n[1] = (float)n.y;
            g.n[2] = (float)n.z;
            g.uv[0] = (float)(s * uv_scale);
            g.uv[1] = (float)(t * uv_scale);
        }
    if (wrap_s)  // closed in s: share the seam vertices exactly
        for (uint32_t j = 0; j <= nt; ++j) {
            GV& a = grid[(size_t)j * (ns + 1)];
            GV& b = grid[(size_t)j * (ns + 1) + ns];
            memcpy(b.p, a.p, sizeof a.p);
            memcpy(b.n, a.n, sizeof a.n);
        }
    auto emit = [&](const GV& a, const GV& b, const GV& c) {
        for (const GV* g : {&a, &b, &c}) {
            sc.triangles.insert(sc.triangles.end(), g->p, g->p + 3);
            sc.triangles.insert(sc.triangles.end(), g->n, g->n + 3);
            sc.triangles.insert(sc.triangles.end(), g->uv, g->uv + 2);
        }
    };
    for (uint32_t j = 0; j < nt; ++j)
        for (uint32_t i = 0; i < ns; ++i) {
            const GV& g00 = grid[(size_t)j * (ns + 1) + i];
            const GV& g10 = grid[(size_t)j * (ns + 1) + i + 1];
            const GV& g01 = grid[(size_t)(j + 1) * (ns + 1) + i];
            const GV& g11 = grid[(size_t)(j + 1) * (ns + 1) + i + 1];
            if (flip) {  // reversed winding keeps e1 x e2 along the (negated) vertex normals
                emit(g00, g11, g10);
                emit(g00, g01, g11);
            } else {
                emit(g00, g10, g11);
                emit(g00, g11, g01);
            }
        }
}

pt_material plain(float r, float g, float b, float rough, float metal) {
    pt_material m{};
    m.albedo[0] = r;
    m.albedo[1] = g;
    m.albedo[2] = b;
    m.opacity = 1.f;
    m.metalness = metal;
    m.roughness = rough;
    m.ior = 1.f;
    m.tex_albedo = m.tex_emissive = m.tex_opacity = m.tex_metalness = m.tex_roughness = m.tex_normal = -1;
    return m;
}

void begin_mesh(pth_scene& sc, const pt_material& mat, uint64_t& first) {
    first = sc.triangles.size() / 24;
    sc.materials.push_back(mat);
}
void end_mesh(pth_scene& sc, uint64_t first) {
    pt_model m{};
    m.kind = PT_MODEL_MESH;
    m.material = (int32_t)sc.materials.size() - 1;
    m.tri_first = (uint32_t)first;
    m.tri_count = (uint32_t)(sc.triangles.size() / 24 - first);
    sc.models.push_back(m);
}

void generate(uint64_t target, uint64_t seed, uint32_t flags, pth_scene& sc) {
    if (target < 2048) target = 2048;
    if (target > 200000000ull) fail(PT_ERR_INVALID, "target_tris too large");
    Pcg32 rng(seed);
    const double jitter = 2e-4;
    const bool alpha = flags & 1u, textured = flags & 2u, ref_framing = flags & 8u;
    const double ground_half = ref_framing ? 6.47 : 12.0;   // half the ground's edge
    // procedural textures (flags bit 1): every texture kind of internal/material.rs:132-214 on the shells and the core
    auto add_texture = [&](uint32_t w, uint32_t h, uint32_t channels, const char* name,
                           const std::function<void(uint32_t, uint32_t, uint8_t*)>& texel) {
        pt_texture t{};
        t.offset = sc.texels.size();
        t.width = w;
        t.height = h;
        t.channels = channels;
        sc.texels.resize(sc.texels.size() + (size_t)w * h * channels);
        uint8_t* px = sc.texels.data() + t.offset;
        for (uint32_t y = 0; y < h; ++y)
            for (uint32_t x = 0; x < w; ++x) texel(x, y, px + ((size_t)y * w + x) * channels);
        sc.textures.push_back(t);
        sc.texture_paths.push_back(name);
        return (int32_t)sc.textures.size() - 1;
    };
    int32_t tex_normal = -1, tex_metal = -1, tex_rough = -1, tex_emissive = -1, tex_albedo = -1;
    if (textured) {
        tex_normal = add_texture(256, 256, 3, "generated:bumps_normal_256", [](uint32_t x, uint32_t y, uint8_t* p) {
            // a field of round bumps: tangent-space normal (dx, dy, 1) normalised, stored as n * 0.5 + 0.5
            double fx = (x % 32) / 32.0 - 0.5, fy = (y % 32) / 32.0 - 0.5, r2 = fx * fx + fy * fy;
            double dx = r2 < 0.2 ? -1.6 * fx : 0.0, dy = r2 < 0.2 ? -1.6 * fy : 0.0;
            double l = std::sqrt(dx * dx + dy * dy + 1.0);
            p[0] = (uint8_t)std::lround((dx / l * 0.5 + 0.5) * 255.0);
            p[1] = (uint8_t)std::lround((dy / l * 0.5 + 0.5) * 255.0);
            p[2] = (uint8_t)std::lround((1.0 / l * 0.5 + 0.5) * 255.0);
        });
        tex_metal = add_texture(64, 64, 1, "generated:stripes_metalness_64",
                                [](uint32_t x, uint32_t, uint8_t* p) { p[0] = (x / 8) % 2 ? 230 : 12; });
        tex_rough = add_texture(64, 32, 1, "generated:ramp_roughness_64x32",
                                [](uint32_t x, uint32_t y, uint8_t* p) { p[0] = (uint8_t)(40 + 3 * x + (y % 8 < 4 ? 0 : 20)); });
        tex_emissive = add_texture(128, 128, 3, "generated:dots_emissive_128", [](uint32_t x, uint32_t y, uint8_t* p) {
            bool dot = ((x % 16) - 8) * ((x % 16) - 8) + ((y % 16) - 8) * ((y % 16) - 8) < 10;
            p[0] = dot ? 255 : 0;
            p[1] = dot ? (uint8_t)(60 + x) : 0;
            p[2] = dot ? (uint8_t)(20 + y) : 4;
        });
        tex_albedo = add_texture(96, 96, 3, "generated:weave_albedo_96", [](uint32_t x, uint32_t y, uint8_t* p) {
            p[0] = (uint8_t)(150 + 100 * ((x / 12 + y / 12) % 2));
            p[1] = (uint8_t)(140 + (x * 7 + y * 3) % 100);
            p[2] = (uint8_t)(130 + (y * 5) % 120);
        });
    }

    // fixed parts scale gently with the budget
    uint32_t g_ground = (uint32_t)std::min<uint64_t>(256, std::max<uint64_t>(8, (uint64_t)std::sqrt(target / 128.0)));
    uint32_t n_strip = (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(16, target / 4096));
    uint64_t fixed = 2ull * g_ground * g_ground + 2ull * 2 * n_strip;
    uint64_t rest = target > fixed ? target - fixed : 1024;
    uint64_t core_tris = rest / 10, shell_tris = (rest - core_tris) / 2;
    auto dims = [](uint64_t tris, double aspect, uint32_t& ns, uint32_t& nt) {
        double cells = std::max(4.0, tris / 2.0);
        ns = (uint32_t)std::max(2.0, std::floor(std::sqrt(cells / aspect)));
        nt = (uint32_t)std::max(2.0, std::floor(cells / ns));
    };

    uint64_t first;
    // 1. ground
    begin_mesh(sc, plain(0.55f, 0.55f, 0.58f, 0.5f, 0.f), first);
    add_patch(sc, [=](double s, double t) { return V3{-ground_half + 2 * ground_half * s, 0.0, ground_half - 2 * ground_half * t}; },
              g_ground, g_ground, false, rng, 0.0, 24.0, false);
    end_mesh(sc, first);

    // 2. glossy dark core (closed ellipsoid-like body)
    uint32_t cs, ct;
    dims(core_tris, 1.0, cs, ct);
    {
        pt_material core = plain(0.03f, 0.03f, 0.035f, 0.25f, 0.f);
        if (textured) {   // glowing dots on the dark core: emissive texture x factor (material.rs:189-201)
            core.tex_emissive = tex_emissive;
            core.emissive[0] = 0.8f;
            core.emissive[1] = 0.6f;
            core.emissive[2] = 1.5f;
        }
        begin_mesh(sc, core, first);
    }
    add_patch(sc,
              [](double s, double t) {
                  double th = M_PI * (0.02 + 0.96 * t), ph = 2 * M_PI * s;
                  double r = std::sin(th);
                  return V3{0.34 * r * std::cos(ph), 1.95 - 1.85 * std::cos(th), 1.25 * r * std::sin(ph)};
              },
              cs, ct, true, rng, jitter, 4.0, true);
    end_mesh(sc, first);

    // 3. two white shells
    uint32_t ss, st;
    dims(shell_tris, 0.8, ss, st);
    int32_t opacity_tex = -1;
    if (alpha) {
        pt_texture t{};
        t.offset = sc.texels.size();
        t.width = t.height = 1024;
        t.channels = 1;
        sc.texels.resize(sc.texels.size() + 1024 * 1024);
        uint8_t* px = sc.texels.data() + t.offset;
        for (uint32_t y = 0; y < 1024; ++y)
            for (uint32_t x = 0; x < 1024; ++x) px[y * 1024 + x] = ((x >> 5) ^ (y >> 5)) & 1 ? 255 : 96;
        sc.textures.push_back(t);
        sc.texture_paths.push_back("generated:checker_opacity_1024");
        opacity_tex = (int32_t)sc.textures.size() - 1;
    }
    for (int side = -1; side <= 1; side += 2) {
        pt_material m = plain(0.9f, 0.9f, 0.92f, 0.35f, 0.f);
        if (alpha) {
            m.opacity = 0.5f;
            m.tex_opacity = opacity_tex;
        }
        if (textured) {   // normal map (hit.rs:64-71), metalness / roughness / albedo textures (material.rs:132-172)
            m.tex_normal = tex_normal;
            m.tex_metalness = tex_metal;
            m.metalness = 0.9f;
            m.tex_roughness = tex_rough;
            m.roughness = 0.8f;
            if (side > 0) m.tex_albedo = tex_albedo;
        }
        begin_mesh(sc, m, first);
        double sd = side;
        add_patch(sc,
                  [sd](double s, double t) {
                      double bulge = std::sin(M_PI * t) * (0.6 + 0.4 * std::cos(2 * M_PI * (s - 0.5)));
                      double x = sd * (0.50 + 0.22 * bulge + 0.35 * t * t);
                      double y = 0.10 + 3.9 * t;
                      double z = (1.7 - 0.25 * t) * (2 * s - 1) + 0.15 * std::sin(M_PI * t);
                      return V3{x, y, z};
                  },
                  ss, st, false, rng, jitter, 8.0, side > 0);
        end_mesh(sc, first);
    }

    // 4. emissive strips between shells and core
    {
        pt_material m = plain(0.f, 0.f, 0.f, 1.f, 0.f);
        m.emissive[0] = 0.3f;
        m.emissive[1] = 0.8f;
        m.emissive[2] = 3.0f;
        begin_mesh(sc, m, first);
        for (int side = -1; side <= 1; side += 2) {
            double sd = side;
            add_patch(sc,
                      [sd](double s, double t) {
                          return V3{sd * (0.42 + 0.04 * s), 0.3 + 3.4 * t, 1.32 + 0.02 * std::sin(6 * t)};
                      },
                      1, n_strip, false, rng, 0.0, 1.0, side < 0);
        }
        end_mesh(sc, first);
    }

    // 5. (flags bit 2) a closed room around everything: four walls and a ceiling on the edges of the ground
    // quad - no path leaves into the background, so the bounce loop (mod.rs:180) runs to its end
    if (flags & 4u) {
        const uint32_t g = std::max(8u, g_ground / 4u);
        begin_mesh(sc, plain(0.95f, 0.93f, 0.90f, 0.5f, 0.f), first);
        const double H = 10.0, E = ground_half;
        // (patch normals point INTO the room: cross(ds, dt) with the parametrisations below)
        add_patch(sc, [=](double u, double v) { return V3{-E + 2 * E * u, H * v, -E}; }, g, g, false, rng, 0.0, 6.0, false);  // back  (+z)
        add_patch(sc, [=](double u, double v) { return V3{E - 2 * E * u, H * v, E}; }, g, g, false, rng, 0.0, 6.0, false);   // front (-z)
        add_patch(sc, [=](double u, double v) { return V3{-E, H * v, E - 2 * E * u}; }, g, g, false, rng, 0.0, 6.0, false);  // left  (+x)
        add_patch(sc, [=](double u, double v) { return V3{E, H * v, -E + 2 * E * u}; }, g, g, false, rng, 0.0, 6.0, false);  // right (-x)
        add_patch(sc, [=](double u, double v) { return V3{-E + 2 * E * u, H, -E + 2 * E * v}; }, g, g, false, rng, 0.0, 6.0, false);  // ceiling (-y)
        end_mesh(sc, first);
    }

    // light, camera, background
    pt_light l{};
    l.kind = PT_LIGHT_POINT;
    l.vec[0] = ref_framing ? -5.5f : 3.5f;   // (reference framing: to the camera's left, behind the object - the shadow runs
    l.vec[1] = ref_framing ? 7.5f : 7.0f;    //  to the right and towards the camera)
    l.vec[2] = ref_framing ? 2.0f : 6.0f;
    l.color[0] = 3000.f;
    l.color[1] = 2900.f;
    l.color[2] = 2800.f;
    l.size = 0.1f;
    sc.lights.push_back(l);

    V3 P{0.6, 2.4, 9.0}, T{0.0, 1.9, 0.0};
    if (ref_framing) {
        const double az = 0.82, dist = 8.6;
        P = V3{dist * std::sin(az) + 1.0, 6.5, dist * std::cos(az)};
        T = V3{1.0, 1.87, 0.0};
    }
    V3 f = norm(T - P);
    V3 right = norm(cross(f, V3{0, 1, 0}));
    V3 up = cross(right, f);
    float* M = sc.desc.camera.transform;
    const V3 cols[4] = {right, up, V3{-f.x, -f.y, -f.z}, P};
    for (int k = 0; k < 4; ++k) {
        M[4 * k + 0] = (float)cols[k].x;
        M[4 * k + 1] = (float)cols[k].y;
        M[4 * k + 2] = (float)cols[k].z;
        M[4 * k + 3] = k == 3 ? 1.f : 0.f;
    }
    sc.desc.camera.fov = 0.6911112f;
    sc.desc.camera.zfar = 100.f;
    sc.desc.camera.znear = 0.1f;
    sc.desc.background[0] = sc.desc.background[1] = sc.desc.background[2] = 0.f;
    sc.finalize();
}

// ---------------------------------------------------------------- ISF writer
void write_f(std::ofstream& f, float v) {
    char buf[40];
    if (std::isfinite(v)) {
        snprintf(buf, sizeof buf, "%.9g", (double)v);
        // serde_json prints floats with a decimal point or exponent
        if (!strpbrk(buf, ".eEn")) strcat(buf, ".0");
    } else {
        snprintf(buf, sizeof buf, "null");
    }
    f << buf;
}
void write_arr(std::ofstream& f, const float* v, int n) {
    f << '[';
    for (int i = 0; i < n; ++i) {
        if (i) f << ',';
        write_f(f, v[i]);
    }
    f << ']';
}

void save_isf(const pth_scene& s, const std::string& dir) {
    mkdir(dir.c_str(), 0755);
    const pt_scene_desc& d = s.desc;
    // textures -> PNG (luma textures are written as grey RGB; into_luma8 maps them back exactly)
    std::vector<std::string> names(d.n_textures);
    for (uint32_t i = 0; i < d.n_textures; ++i) {
        const pt_texture& t = d.textures[i];
        // (the glTF converter names its textures like the reference's ReverseTextureBank, gltf.rs:47-76)
        const std::string& given = i < s.texture_paths.size() ? s.texture_paths[i] : std::string();
        const bool plain_png = given.size() > 4 && given.compare(given.size() - 4, 4, ".png") == 0 &&
                               given.find('/') == std::string::npos && given.find(':') == std::string::npos;
        names[i] = plain_png ? given : "tex_" + std::to_string(i) + ".png";
        std::vector<uint8_t> rgb((size_t)t.width * t.height * 3);
        const uint8_t* src = d.texels + t.offset;
        for (size_t p = 0; p < (size_t)t.width * t.height; ++p)
            for (int c = 0; c < 3; ++c) rgb[p * 3 + c] = t.channels == 3 ? src[p * 3 + c] : src[p];
        if (pth_png_write_rgb8((dir + "/" + names[i]).c_str(), t.width, t.height, rgb.data()) != PT_OK)
            fail(PT_ERR_IO, "%s", pth_last_error());
    }
    std::ofstream f(dir + "/scene.isf", std::ios::binary);
    if (!f) fail(PT_ERR_IO, "cannot write %s/scene.isf", dir.c_str());
    auto chan = [&](const char* name, const float* factor, int n, int32_t tex) {
        f << '"' << name << "\":{\"factor\":";
        if (n == 1) write_f(f, factor[0]);
        else write_arr(f, factor, n);
        f << ",\"texture\":";
        if (tex >= 0) f << '"' << names[tex] << '"';
        else f << "null";
        f << '}';
    };
    f << "{\"models\":[";
    for (uint32_t m = 0; m < d.n_models; ++m) {
        const pt_model& mo = d.models[m];
        const pt_material& ma = d.materials[mo.material];
        if (m) f << ',';
        if (mo.kind == PT_MODEL_MESH) {
            f << "{\"type\":\"Mesh\",\"triangles\":[";
            for (uint32_t t = 0; t < mo.tri_count; ++t) {
                const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                if (t) f << ',';
                f << '[';
                for (int k = 0; k < 3; ++k) {
                    if (k) f << ',';
                    f << "{\"position\":";
                    write_arr(f, v + 8 * k, 3);
                    f << ",\"normal\":";
                    write_arr(f, v + 8 * k + 3, 3);
                    f << ",\"tex_coords\":";
                    write_arr(f, v + 8 * k + 6, 2);
                    f << '}';
                }
                f << ']';
            }
            f << "],";
        } else {
            f << "{\"type\":\"Sphere\",\"radius\":";
            write_f(f, mo.radius);
            f << ",\"center\":";
            write_arr(f, mo.center, 3);
            f << ',';
        }
        f << "\"material\":{";
        chan("albedo", ma.albedo, 3, ma.tex_albedo);
        f << ',';
        chan("emissive", ma.emissive, 3, ma.tex_emissive);
        f << ',';
        chan("opacity", &ma.opacity, 1, ma.tex_opacity);
        f << ',';
        chan("metalness", &ma.metalness, 1, ma.tex_metalness);
        f << ',';
        chan("roughness", &ma.roughness, 1, ma.tex_roughness);
        f << ",\"ior\":";
        write_f(f, ma.ior);
        f << ",\"normal_texture\":";
        if (ma.tex_normal >= 0) f << '"' << names[ma.tex_normal] << '"';
        else f << "null";
        f << "}}";
    }
    f << "],\"camera\":{\"transform\":[";
    for (int k = 0; k < 4; ++k) {
        if (k) f << ',';
        write_arr(f, d.camera.transform + 4 * k, 4);
    }
    f << "],\"fov\":";
    write_f(f, d.camera.fov);
    f << ",\"zfar\":";
    write_f(f, d.camera.zfar);
    f << ",\"znear\":";
    write_f(f, d.camera.znear);
    f << "},\"lights\":[";
    for (uint32_t i = 0; i < d.n_lights; ++i) {
        const pt_light& l = d.lights[i];
        if (i) f << ',';
        if (l.kind == PT_LIGHT_POINT) {
            f << "{\"type\":\"Point\",\"position\":";
            write_arr(f, l.vec, 3);
            f << ",\"color\":";
            write_arr(f, l.color, 3);
            f << ",\"size\":";
            write_f(f, l.size);
            f << '}';
        } else {
            f << "{\"type\":\"Directional\",\"direction\":";
            write_arr(f, l.vec, 3);
            f << ",\"color\":";
            write_arr(f, l.color, 3);
            f << '}';
        }
    }
    f << "],\"background\":";
    write_arr(f, d.background, 3);
    f << "}\n";
    if (!f) fail(PT_ERR_IO, "write failed: %s/scene.isf", dir.c_str());
}

}  // namespace
}  // namespace pth

extern "C" {

int pth_scene_generate_ps5(uint64_t target_tris, uint64_t seed, uint32_t flags, pth_scene** out) {
    return pth::guarded([&] {
        if (!out) pth::fail(PT_ERR_INVALID, "pth_scene_generate_ps5: null output");
        auto s = std::make_unique<pth_scene>();
        pth::generate(target_tris, seed, flags, *s);
        *out = s.release();
    });
}

int pth_scene_save_isf(const pth_scene* s, const char* dir) {
    return pth::guarded([&] {
        if (!s || !dir) pth::fail(PT_ERR_INVALID, "pth_scene_save_isf: null argument");
        pth::save_isf(*s, dir);
    });
}

}  // extern "C"
