// ISF scene loader: JSON -> flat pt_scene_desc.
//
// Follows the serde model of the reference (src/scene/isf.rs:5-142) and the
// conversion into the runtime scene (src/scene/mod.rs:16-22,
// src/scene/internal/{mod.rs:35-51, model.rs:89-113, material.rs:103-113,
// texture_bank.rs:21-51}).  Written as a schema-driven streaming reader (no
// DOM) so that multi-hundred-megabyte scene files load in one pass.
//
// serde semantics kept:
//   * Scene: models, camera, lights, background all required (isf.rs:7-16)
//   * Model / Light internally tagged by "type"; the tag may come after the
//     other keys (isf.rs:31,58)
//   * Triangle = JSON array of 3 vertices (isf.rs:47); Vertex needs
//     position, normal, tex_coords (isf.rs:50-55)
//   * Material: albedo required, factor default [1,1,1] (isf.rs:101-106);
//     missing emissive -> Default => factor [0,0,0], present without factor
//     -> [1,1,1] (isf.rs:83-84,108-113); missing metalness -> 0, present
//     without factor -> 1 (isf.rs:89-90,124-129); missing opacity/roughness
//     -> 1 (isf.rs:86-87,92-93,115-138); ior default 1 (isf.rs:95-96);
//     normal_texture optional (isf.rs:98); unknown keys ignored
//   * f32 values are parsed as f64 and narrowed (serde_json)
//   * texture paths are relative to the .isf directory, canonicalised and
//     cached separately for rgb and luma use (texture_bank.rs:21-51)
#include <cerrno>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <charconv>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>

#include "host_common.hpp"

namespace pth {

static thread_local std::string g_last_error;
void set_last_error(const std::string& msg) { g_last_error = msg; }

namespace {

// ---------------------------------------------------------------- JSON
class JsonReader {
public:
    JsonReader(const char* begin, const char* end) : p_(begin), begin_(begin), end_(end) {}

    [[noreturn]] void error(const char* what) const {
        size_t line = 1, col = 1;
        for (const char* q = begin_; q < p_ && q < end_; ++q) {
            if (*q == '\n') {
                ++line;
                col = 1;
            } else {
                ++col;
            }
        }
        fail(PT_ERR_PARSE, "%s at line %zu column %zu", what, line, col);
    }

    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    char peek() {
        ws();
        if (p_ >= end_) error("EOF while parsing a value");
        return *p_;
    }
    void expect(char c) {
        if (peek() != c) {
            char msg[64];
            snprintf(msg, sizeof msg, "expected `%c`", c);
            error(msg);
        }
        ++p_;
    }
    bool consume(char c) {
        if (peek() == c) {
            ++p_;
            return true;
        }
        return false;
    }

    std::string string() {
        expect('"');
        std::string out;
        while (true) {
            if (p_ >= end_) error("EOF while parsing a string");
            char c = *p_++;
            if (c == '"') break;
            if (c == '\\') {
                if (p_ >= end_) error("EOF while parsing a string");
                char e = *p_++;
                switch (e) {
                    case '"': out += '"'; break;
                    case '\\': out += '\\'; break;
                    case '/': out += '/'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'n': out += '\n'; break;
                    case 'r': out += '\r'; break;
                    case 't': out += '\t'; break;
                    case 'u': {
                        unsigned cp = hex4();
                        if (cp >= 0xD800 && cp < 0xDC00 && p_ + 1 < end_ && p_[0] == '\\' &&
                            p_[1] == 'u') {
                            p_ += 2;
                            unsigned lo = hex4();
                            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                        }
                        utf8(out, cp);
                        break;
                    }
                    default: error("invalid escape");
                }
            } else {
                out += c;
            }
        }
        return out;
    }

    double number() {
        ws();
        const char* s = p_;
        if (p_ < end_ && *p_ == '-') ++p_;
        if (p_ >= end_ || !(*p_ >= '0' && *p_ <= '9')) error("expected a number");
        while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' ||
                             *p_ == 'E' || *p_ == '+' || *p_ == '-'))
            ++p_;
        // std::from_chars (correctly rounded like strtod, a third of its time: a 500 000-triangle ISF holds 12 M numbers - 0.5 of
        // the 0.8 s it took to load) for what it accepts; everything else - range errors, odd spellings - goes the old way
        {
            double fast;
            const std::from_chars_result r = std::from_chars(s, p_, fast);
            if (r.ec == std::errc() && r.ptr == p_) return fast;
        }
        // strtod needs a terminated buffer; numbers are short.
        char buf[80];
        size_t n = (size_t)(p_ - s);
        if (n >= sizeof buf) error("number too long");
        memcpy(buf, s, n);
        buf[n] = 0;
        char* endp = nullptr;
        double v = strtod(buf, &endp);
        if (endp != buf + n) error("invalid number");
        return v;
    }
    float f32() { return (float)number(); }

    bool null_() {
        if (peek() == 'n') {
            literal("null");
            return true;
        }
        return false;
    }

    // (recursive: an ISF of a few hundred thousand nested '[' must end in PT_ERR_PARSE, not in a stack overflow)
    void skip_value(int depth = 0) {
        if (depth > 256) error("value nested too deeply");
        char c = peek();
        if (c == '{') {
            ++p_;
            if (consume('}')) return;
            do {
                string();
                expect(':');
                skip_value(depth + 1);
            } while (consume(','));
            expect('}');
        } else if (c == '[') {
            ++p_;
            if (consume(']')) return;
            do {
                skip_value(depth + 1);
            } while (consume(','));
            expect(']');
        } else if (c == '"') {
            string();
        } else if (c == 't') {
            literal("true");
        } else if (c == 'f') {
            literal("false");
        } else if (c == 'n') {
            literal("null");
        } else {
            number();
        }
    }

    // Iterate an object: calls fn(key) for each member; fn must consume the value.
    template <class F>
    void object(F&& fn) {
        expect('{');
        if (consume('}')) return;
        do {
            std::string key = string();
            expect(':');
            fn(key);
        } while (consume(','));
        expect('}');
    }

    template <size_t N>
    void f32_array(float (&out)[N]) {
        expect('[');
        for (size_t i = 0; i < N; ++i) {
            if (i) expect(',');
            out[i] = f32();
        }
        expect(']');
    }

    void end_of_input() {
        ws();
        if (p_ != end_) error("trailing characters");
    }

private:
    void literal(const char* lit) {
        size_t n = strlen(lit);
        if ((size_t)(end_ - p_) < n || memcmp(p_, lit, n) != 0) error("expected value");
        p_ += n;
    }
    unsigned hex4() {
        if (end_ - p_ < 4) error("EOF in \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
            else error("invalid \\u escape");
        }
        return v;
    }
    static void utf8(std::string& out, unsigned cp) {
        if (cp < 0x80) out += (char)cp;
        else if (cp < 0x800) {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        } else if (cp < 0x10000) {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        } else {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }

    const char* p_;
    const char* begin_;
    const char* end_;
};

// ---------------------------------------------------------------- textures
struct TextureBank {
    std::string root;
    pth_scene* scene;
    std::map<std::string, int32_t> rgb, gray;

    int32_t get(const std::string& rel, uint32_t channels) {
        std::string joined = rel.size() && rel[0] == '/' ? rel : root + "/" + rel;
        char resolved[PATH_MAX];
        if (!realpath(joined.c_str(), resolved))
            fail(PT_ERR_IO, "Invalid path: %s: %s", joined.c_str(), strerror(errno));
        auto& cache = channels == 3 ? rgb : gray;
        auto it = cache.find(resolved);
        if (it != cache.end()) return it->second;
        uint32_t w = 0, h = 0;
        uint8_t* px = nullptr;
        if (pth_png_read(resolved, channels, &w, &h, &px) != PT_OK)
            fail(PT_ERR_IO, "texture %s: %s", resolved, pth_last_error());
        pt_texture t{};
        t.offset = scene->texels.size();
        t.width = w;
        t.height = h;
        t.channels = channels;
        scene->texels.insert(scene->texels.end(), px, px + (size_t)w * h * channels);
        // keep every texture 4-byte aligned in the blob
        while (scene->texels.size() & 3) scene->texels.push_back(0);
        pth_free(px);
        int32_t id = (int32_t)scene->textures.size();
        scene->textures.push_back(t);
        scene->texture_paths.push_back(resolved);
        cache.emplace(resolved, id);
        return id;
    }
};

// A {factor, texture} block (isf.rs:101-138).
template <size_t N>
struct Channel {
    bool present = false;
    bool has_factor = false;
    float factor[N];
    bool has_texture = false;
    std::string texture;
};

template <size_t N>
void parse_channel(JsonReader& r, Channel<N>& c) {
    c.present = true;
    r.object([&](const std::string& key) {
        if (key == "factor") {
            if constexpr (N == 1) {
                c.factor[0] = r.f32();
            } else {
                r.f32_array(c.factor);
            }
            c.has_factor = true;
        } else if (key == "texture") {
            if (!r.null_()) {
                c.texture = r.string();
                c.has_texture = true;
            }
        } else {
            r.skip_value();
        }
    });
}

int32_t parse_material(JsonReader& r, pth_scene& s, TextureBank& bank) {
    Channel<3> albedo, emissive;
    Channel<1> opacity, metalness, roughness;
    float ior = 1.0f;  // isf.rs:95-96
    bool has_normal = false;
    std::string normal_tex;
    r.object([&](const std::string& key) {
        if (key == "albedo") parse_channel(r, albedo);
        else if (key == "emissive") parse_channel(r, emissive);
        else if (key == "opacity") parse_channel(r, opacity);
        else if (key == "metalness") parse_channel(r, metalness);
        else if (key == "roughness") parse_channel(r, roughness);
        else if (key == "ior") ior = r.f32();
        else if (key == "normal_texture") {
            if (!r.null_()) {
                normal_tex = r.string();
                has_normal = true;
            }
        } else r.skip_value();
    });
    if (!albedo.present) r.error("missing field `albedo`");

    pt_material m{};
    for (int k = 0; k < 3; ++k) {
        m.albedo[k] = albedo.has_factor ? albedo.factor[k] : 1.0f;
        // Emissive::default() is all-zero (derive(Default)), a present block
        // without factor uses serde's `one` (isf.rs:108-113).
        m.emissive[k] = !emissive.present ? 0.0f : (emissive.has_factor ? emissive.factor[k] : 1.0f);
    }
    m.opacity = !opacity.present ? 1.0f : (opacity.has_factor ? opacity.factor[0] : 1.0f);
    // Metalness::default() is 0 (plain derive(Default), isf.rs:124-129).
    m.metalness = !metalness.present ? 0.0f : (metalness.has_factor ? metalness.factor[0] : 1.0f);
    m.roughness = !roughness.present ? 1.0f : (roughness.has_factor ? roughness.factor[0] : 1.0f);
    m.ior = ior;
    // Material::load order: albedo, emissive, opacity, metalness, roughness, normal.
    m.tex_albedo = albedo.has_texture ? bank.get(albedo.texture, 3) : -1;
    m.tex_emissive = emissive.has_texture ? bank.get(emissive.texture, 3) : -1;
    m.tex_opacity = opacity.has_texture ? bank.get(opacity.texture, 1) : -1;
    m.tex_metalness = metalness.has_texture ? bank.get(metalness.texture, 1) : -1;
    m.tex_roughness = roughness.has_texture ? bank.get(roughness.texture, 1) : -1;
    m.tex_normal = has_normal ? bank.get(normal_tex, 3) : -1;
    s.materials.push_back(m);
    return (int32_t)s.materials.size() - 1;
}

void parse_vertex(JsonReader& r, float* out /*8*/) {
    bool hp = false, hn = false, ht = false;
    r.object([&](const std::string& key) {
        if (key == "position") {
            float v[3];
            r.f32_array(v);
            memcpy(out, v, sizeof v);
            hp = true;
        } else if (key == "normal") {
            float v[3];
            r.f32_array(v);
            memcpy(out + 3, v, sizeof v);
            hn = true;
        } else if (key == "tex_coords") {
            float v[2];
            r.f32_array(v);
            memcpy(out + 6, v, sizeof v);
            ht = true;
        } else {
            r.skip_value();
        }
    });
    if (!hp) r.error("missing field `position`");
    if (!hn) r.error("missing field `normal`");
    if (!ht) r.error("missing field `tex_coords`");
}

// Appends triangles to s.triangles; returns the count.
uint64_t parse_triangles(JsonReader& r, pth_scene& s) {
    uint64_t n = 0;
    r.expect('[');
    if (r.consume(']')) return 0;
    do {
        r.expect('[');
        size_t base = s.triangles.size();
        s.triangles.resize(base + 24);
        for (int k = 0; k < 3; ++k) {
            if (k) r.expect(',');
            parse_vertex(r, &s.triangles[base + 8 * k]);
        }
        r.expect(']');
        ++n;
    } while (r.consume(','));
    r.expect(']');
    return n;
}

void parse_model(JsonReader& r, pth_scene& s, TextureBank& bank) {
    std::string type;
    bool has_radius = false, has_center = false, has_tris = false, has_mat = false;
    float radius = 0, center[3] = {0, 0, 0};
    uint64_t tri_first = s.triangles.size() / 24, tri_count = 0;
    int32_t material = -1;
    r.object([&](const std::string& key) {
        if (key == "type") type = r.string();
        else if (key == "radius") {
            radius = r.f32();
            has_radius = true;
        } else if (key == "center") {
            r.f32_array(center);
            has_center = true;
        } else if (key == "triangles") {
            if (has_tris) r.error("duplicate field `triangles`");
            tri_count = parse_triangles(r, s);
            has_tris = true;
        } else if (key == "material") {
            material = parse_material(r, s, bank);
            has_mat = true;
        } else r.skip_value();
    });
    pt_model m{};
    if (type == "Mesh") {
        if (!has_tris) r.error("missing field `triangles`");
        if (!has_mat) r.error("missing field `material`");
        m.kind = PT_MODEL_MESH;
        m.tri_first = (uint32_t)tri_first;
        m.tri_count = (uint32_t)tri_count;
    } else if (type == "Sphere") {
        if (!has_radius) r.error("missing field `radius`");
        if (!has_center) r.error("missing field `center`");
        if (!has_mat) r.error("missing field `material`");
        if (has_tris) s.triangles.resize(tri_first * 24);  // ignored unknown key
        m.kind = PT_MODEL_SPHERE;
        m.tri_first = (uint32_t)tri_first;
        m.tri_count = 0;
        memcpy(m.center, center, sizeof center);
        m.radius = radius;
    } else if (type.empty()) {
        r.error("missing field `type`");
    } else {
        fail(PT_ERR_PARSE, "unknown variant `%s`, expected `Sphere` or `Mesh`", type.c_str());
    }
    m.material = material;
    s.models.push_back(m);
}

void parse_light(JsonReader& r, pth_scene& s) {
    std::string type;
    bool hp = false, hd = false, hc = false, hs = false;
    float position[3] = {0, 0, 0}, direction[3] = {0, 0, 0}, color[3] = {0, 0, 0}, size = 0;
    r.object([&](const std::string& key) {
        if (key == "type") type = r.string();
        else if (key == "position") {
            r.f32_array(position);
            hp = true;
        } else if (key == "direction") {
            r.f32_array(direction);
            hd = true;
        } else if (key == "color") {
            r.f32_array(color);
            hc = true;
        } else if (key == "size") {
            size = r.f32();
            hs = true;
        } else r.skip_value();
    });
    pt_light l{};
    if (type == "Point") {
        if (!hp) r.error("missing field `position`");
        if (!hc) r.error("missing field `color`");
        if (!hs) r.error("missing field `size`");
        l.kind = PT_LIGHT_POINT;
        memcpy(l.vec, position, sizeof position);
    } else if (type == "Directional") {
        if (!hd) r.error("missing field `direction`");
        if (!hc) r.error("missing field `color`");
        l.kind = PT_LIGHT_DIRECTIONAL;
        memcpy(l.vec, direction, sizeof direction);
    } else if (type.empty()) {
        r.error("missing field `type`");
    } else {
        fail(PT_ERR_PARSE, "unknown variant `%s`, expected `Point` or `Directional`", type.c_str());
    }
    memcpy(l.color, color, sizeof color);
    l.size = size;
    s.lights.push_back(l);
}

void parse_camera(JsonReader& r, pt_camera& c) {
    bool ht = false, hf = false, hz = false, hn = false;
    r.object([&](const std::string& key) {
        if (key == "transform") {
            r.expect('[');
            for (int k = 0; k < 4; ++k) {
                if (k) r.expect(',');
                float col[4];
                r.f32_array(col);
                memcpy(&c.transform[4 * k], col, sizeof col);
            }
            r.expect(']');
            ht = true;
        } else if (key == "fov") {
            c.fov = r.f32();
            hf = true;
        } else if (key == "zfar") {
            c.zfar = r.f32();
            hz = true;
        } else if (key == "znear") {
            c.znear = r.f32();
            hn = true;
        } else r.skip_value();
    });
    if (!ht) r.error("missing field `transform`");
    if (!hf) r.error("missing field `fov`");
    if (!hz) r.error("missing field `zfar`");
    if (!hn) r.error("missing field `znear`");
}

}  // namespace

static void load_isf(const char* path, pth_scene& s) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(PT_ERR_IO, "%s: %s", path, strerror(errno));
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::string root = path;
    size_t slash = root.find_last_of('/');
    root = slash == std::string::npos ? "." : (slash == 0 ? "/" : root.substr(0, slash));

    TextureBank bank{root, &s, {}, {}};
    JsonReader r(text.data(), text.data() + text.size());
    bool hm = false, hc = false, hl = false, hb = false;
    r.object([&](const std::string& key) {
        if (key == "models") {
            r.expect('[');
            if (!r.consume(']')) {
                do parse_model(r, s, bank);
                while (r.consume(','));
                r.expect(']');
            }
            hm = true;
        } else if (key == "camera") {
            parse_camera(r, s.desc.camera);
            hc = true;
        } else if (key == "lights") {
            r.expect('[');
            if (!r.consume(']')) {
                do parse_light(r, s);
                while (r.consume(','));
                r.expect(']');
            }
            hl = true;
        } else if (key == "background") {
            r.f32_array(s.desc.background);
            hb = true;
        } else {
            r.skip_value();
        }
    });
    r.end_of_input();
    if (!hm) r.error("missing field `models`");
    if (!hc) r.error("missing field `camera`");
    if (!hl) r.error("missing field `lights`");
    if (!hb) r.error("missing field `background`");
    s.finalize();
}

}  // namespace pth

extern "C" {

int pth_scene_load_isf(const char* path, pth_scene** out) {
    return pth::guarded([&] {
        if (!path || !out) pth::fail(PT_ERR_INVALID, "pth_scene_load_isf: null argument");
        auto s = std::make_unique<pth_scene>();
        pth::load_isf(path, *s);
        *out = s.release();
    });
}

void pth_scene_free(pth_scene* s) { delete s; }

const pt_scene_desc* pth_scene_desc(const pth_scene* s) { return s ? &s->desc : nullptr; }

uint64_t pth_prim_count(const pt_scene_desc* d) {
    uint64_t n = 0;
    for (uint32_t m = 0; m < d->n_models; ++m)
        n += d->models[m].kind == PT_MODEL_MESH ? d->models[m].tri_count : 1;
    return n;
}

const char* pth_last_error(void) { return pth::g_last_error.c_str(); }

void pth_free(void* p) { free(p); }

}  // extern "C"
