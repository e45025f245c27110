// glTF 2.0 -> ISF converter: the `convert <INPUT> <OUTPUT>` subcommand.
//
// Reference: src/scene/gltf.rs:146-265 (convert_gltf_to_isf) on top of the easy-gltf crate, which is NOT part of
// the reference tree - what it does is restated here from the glTF 2.0 specification and easy-gltf's documented
// behaviour: scene 0 only; every mesh primitive of the node hierarchy becomes one ISF mesh with its vertices moved
// to world space (positions by the node's global matrix, normals by the same matrix with w = 0 and normalised);
// the first camera (perspective only, transform = its node's global matrix); KHR_lights_punctual lights.
// The mapping to ISF follows gltf.rs line by line:
//   * Light (gltf.rs:233-264): directional -> Directional{direction, color * intensity}; point AND spot ->
//     Point{position, color * intensity, size 0.1};
//   * Material (gltf.rs:78-129): albedo = base colour factor rgb + the base-colour texture's rgb
//     ("albedo_tex_N.png"); opacity = base colour factor alpha + the same texture's alpha channel as a grey image
//     ("alpha_tex_N.png"); metalness / roughness = factor + the blue / green channel of the metallic-roughness
//     texture as grey images ("gray_tex_N.png"); emissive = factor + texture ("vec_tex_N.png"); normal_texture
//     ("vec_tex_N.png"); ior 1.0;
//   * Camera (gltf.rs:200-215): fov = yfov, zfar, znear; orthographic cameras are an error (the reference panics);
//   * background [0, 0, 0] (Scene::default, gltf.rs:184-189); no camera is an error (the reference panics, :163-166).
// Supported input: .gltf (external or base64 buffers / images) and .glb; triangle-list primitives (other modes are
// an error, as easy-gltf's triangles() fails on them); PNG images (the only decoder in this build: JPEG textures are
// reported as unsupported).  No reference test covers the converter: parity unpinned.
#include <sys/stat.h>

#include <cerrno>
#include <cmath>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>

#include "host_common.hpp"

namespace pth {
namespace {

// ---------------------------------------------------------------- JSON DOM (glTF documents are small)
struct JVal {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;
    std::vector<JVal> arr;
    std::vector<std::pair<std::string, JVal>> obj;

    const JVal* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char* key) const { return get(key) != nullptr; }
    double number(const char* key, double dflt) const {
        const JVal* v = get(key);
        return v && v->kind == Num ? v->num : dflt;
    }
    int64_t index(const char* key) const {   // -1 when absent
        const JVal* v = get(key);
        return v && v->kind == Num && v->num >= 0 ? (int64_t)v->num : -1;
    }
    std::string string(const char* key) const {
        const JVal* v = get(key);
        return v && v->kind == Str ? v->str : std::string();
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
};

struct JParser {
    const char* p;
    const char* end;
    [[noreturn]] void error(const char* what) { fail(PT_ERR_PARSE, "glTF JSON: %s at byte %zu", what, (size_t)(p - begin)); }
    const char* begin;
    JParser(const char* s, size_t n) : p(s), end(s + n), begin(s) {}
    void ws() {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
    }
    JVal value(int depth = 0) {
        if (depth > 128) error("nested too deeply");
        ws();
        if (p >= end) error("unexpected end");
        JVal v;
        char c = *p;
        if (c == '{') {
            ++p;
            v.kind = JVal::Obj;
            ws();
            if (p < end && *p == '}') {
                ++p;
                return v;
            }
            while (true) {
                ws();
                if (p >= end || *p != '"') error("expected a key");
                std::string k = string();
                ws();
                if (p >= end || *p != ':') error("expected ':'");
                ++p;
                v.obj.emplace_back(std::move(k), value(depth + 1));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == '}') {
                    ++p;
                    return v;
                }
                error("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++p;
            v.kind = JVal::Arr;
            ws();
            if (p < end && *p == ']') {
                ++p;
                return v;
            }
            while (true) {
                v.arr.push_back(value(depth + 1));
                ws();
                if (p < end && *p == ',') {
                    ++p;
                    continue;
                }
                if (p < end && *p == ']') {
                    ++p;
                    return v;
                }
                error("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.kind = JVal::Str;
            v.str = string();
            return v;
        }
        auto lit = [&](const char* s) {
            size_t n = strlen(s);
            if ((size_t)(end - p) < n || memcmp(p, s, n) != 0) error("bad literal");
            p += n;
        };
        if (c == 't') {
            lit("true");
            v.kind = JVal::Bool;
            v.b = true;
            return v;
        }
        if (c == 'f') {
            lit("false");
            v.kind = JVal::Bool;
            return v;
        }
        if (c == 'n') {
            lit("null");
            return v;
        }
        const char* s = p;
        while (p < end && (strchr("+-.eE", *p) || (*p >= '0' && *p <= '9'))) ++p;
        if (p == s || p - s > 64) error("expected a value");
        std::string num(s, p);
        char* e = nullptr;
        v.num = strtod(num.c_str(), &e);
        if (e != num.c_str() + num.size()) error("bad number");
        v.kind = JVal::Num;
        return v;
    }
    std::string string() {
        ++p;   // opening quote
        std::string out;
        while (true) {
            if (p >= end) error("unterminated string");
            char c = *p++;
            if (c == '"') return out;
            if (c != '\\') {
                out += c;
                continue;
            }
            if (p >= end) error("unterminated escape");
            char e = *p++;
            switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': {
                    if (end - p < 4) error("bad \\u escape");
                    unsigned cp = (unsigned)strtoul(std::string(p, p + 4).c_str(), nullptr, 16);
                    p += 4;
                    if (cp < 0x80) out += (char)cp;   // (URIs and names: ASCII is all that matters here)
                    else out += '?';
                    break;
                }
                default: out += e;
            }
        }
    }
};

// ---------------------------------------------------------------- small linear algebra (column-major, f32)
struct M4 {
    float m[16];   // m[4 * col + row]
};
M4 identity() {
    M4 r{};
    r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
    return r;
}
M4 mul(const M4& a, const M4& b) {
    M4 r{};
    for (int c = 0; c < 4; ++c)
        for (int row = 0; row < 4; ++row) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s += a.m[4 * k + row] * b.m[4 * c + k];
            r.m[4 * c + row] = s;
        }
    return r;
}
M4 node_matrix(const JVal& node) {
    if (const JVal* mv = node.get("matrix")) {
        if (mv->size() != 16) fail(PT_ERR_PARSE, "glTF: node.matrix needs 16 numbers");
        M4 r;
        for (int i = 0; i < 16; ++i) r.m[i] = (float)mv->arr[i].num;
        return r;
    }
    float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
    auto rd = [&](const char* key, float* out, size_t n) {
        if (const JVal* v = node.get(key)) {
            if (v->size() != n) fail(PT_ERR_PARSE, "glTF: node.%s needs %zu numbers", key, n);
            for (size_t i = 0; i < n; ++i) out[i] = (float)v->arr[i].num;
        }
    };
    rd("translation", t, 3);
    rd("rotation", q, 4);
    rd("scale", s, 3);
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    M4 r = identity();   // T * R * S
    r.m[0] = (1 - 2 * (y * y + z * z)) * s[0];
    r.m[1] = (2 * (x * y + z * w)) * s[0];
    r.m[2] = (2 * (x * z - y * w)) * s[0];
    r.m[4] = (2 * (x * y - z * w)) * s[1];
    r.m[5] = (1 - 2 * (x * x + z * z)) * s[1];
    r.m[6] = (2 * (y * z + x * w)) * s[1];
    r.m[8] = (2 * (x * z + y * w)) * s[2];
    r.m[9] = (2 * (y * z - x * w)) * s[2];
    r.m[10] = (1 - 2 * (x * x + y * y)) * s[2];
    r.m[12] = t[0];
    r.m[13] = t[1];
    r.m[14] = t[2];
    return r;
}

// ---------------------------------------------------------------- files, buffers
std::vector<uint8_t> read_file(const std::string& path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) fail(PT_ERR_IO, "cannot open %s", path.c_str());
    std::vector<uint8_t> data((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return data;
}

std::vector<uint8_t> base64(const std::string& s, size_t from) {
    std::vector<uint8_t> out;
    uint32_t acc = 0;
    int bits = 0;
    for (size_t i = from; i < s.size(); ++i) {
        char c = s[i];
        int v;
        if (c >= 'A' && c <= 'Z') v = c - 'A';
        else if (c >= 'a' && c <= 'z') v = c - 'a' + 26;
        else if (c >= '0' && c <= '9') v = c - '0' + 52;
        else if (c == '+' || c == '-') v = 62;
        else if (c == '/' || c == '_') v = 63;
        else continue;   // padding, whitespace
        acc = (acc << 6) | (uint32_t)v;
        bits += 6;
        if (bits >= 8) {
            bits -= 8;
            out.push_back((uint8_t)(acc >> bits));
        }
    }
    return out;
}

std::vector<uint8_t> load_uri(const std::string& uri, const std::string& base_dir) {
    if (uri.rfind("data:", 0) == 0) {
        size_t comma = uri.find(',');
        if (comma == std::string::npos || uri.substr(0, comma).find(";base64") == std::string::npos)
            fail(PT_ERR_UNSUPPORTED, "glTF: only base64 data URIs are supported");
        return base64(uri, comma + 1);
    }
    std::string path = uri;   // (percent-encoding: only %20)
    for (size_t i; (i = path.find("%20")) != std::string::npos;) path.replace(i, 3, " ");
    return read_file(base_dir + path);
}

struct Document {
    JVal root;
    std::string base_dir;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<uint8_t> glb_bin;
};

const JVal& element(const JVal& root, const char* array, int64_t i) {
    const JVal* a = root.get(array);
    if (!a || i < 0 || (size_t)i >= a->size()) fail(PT_ERR_PARSE, "glTF: %s[%lld] does not exist", array, (long long)i);
    return a->arr[(size_t)i];
}

// A JSON number that must be a byte offset / length / count: finite, integral, in [0, 2^53) - the range in which a
// double holds integers exactly; anything else is a malformed file, not a value to cast (casting a negative or huge double
// to size_t is undefined behaviour, and sums of unchecked values wrap).
size_t size_field(const JVal& obj, const char* key, size_t dflt, const char* what) {
    if (!obj.has(key)) return dflt;
    const double d = obj.number(key, 0);
    if (!(d >= 0.0) || !(d < 9007199254740992.0) || d != std::floor(d))
        fail(PT_ERR_PARSE, "glTF: %s.%s = %g is not a valid size", what, key, d);
    return (size_t)d;
}
// off + len <= size, without overflow
bool range_fits(size_t off, size_t len, size_t size) { return off <= size && len <= size - off; }

// One accessor, converted to f32 components (normalised integers are scaled) or to u32 indices.
struct View {
    const uint8_t* data;
    size_t count, stride;
    int comp_type, n_comp;
    bool normalized;
};
View accessor_view(const Document& doc, int64_t index) {
    const JVal& acc = element(doc.root, "accessors", index);
    if (acc.has("sparse")) fail(PT_ERR_UNSUPPORTED, "glTF: sparse accessors are not supported");
    const std::string type = acc.string("type");
    View v;
    v.n_comp = type == "SCALAR" ? 1 : type == "VEC2" ? 2 : type == "VEC3" ? 3 : type == "VEC4" ? 4 : 0;
    if (!v.n_comp) fail(PT_ERR_UNSUPPORTED, "glTF: accessor type %s is not supported", type.c_str());
    v.comp_type = (int)acc.number("componentType", 0);
    v.count = size_field(acc, "count", 0, "accessor");
    v.normalized = acc.get("normalized") && acc.get("normalized")->b;
    size_t comp_size = v.comp_type == 5126 || v.comp_type == 5125 ? 4 : (v.comp_type == 5122 || v.comp_type == 5123) ? 2
                     : (v.comp_type == 5120 || v.comp_type == 5121) ? 1 : 0;
    if (!comp_size) fail(PT_ERR_PARSE, "glTF: bad accessor componentType %d", v.comp_type);
    const JVal& bv = element(doc.root, "bufferViews", acc.index("bufferView"));
    int64_t bi = bv.index("buffer");
    if (bi < 0 || (size_t)bi >= doc.buffers.size()) fail(PT_ERR_PARSE, "glTF: bufferView without a buffer");
    const std::vector<uint8_t>& buf = doc.buffers[(size_t)bi];
    const size_t view_off = size_field(bv, "byteOffset", 0, "bufferView"), acc_off = size_field(acc, "byteOffset", 0, "accessor");
    const size_t elem = comp_size * v.n_comp;
    v.stride = size_field(bv, "byteStride", 0, "bufferView");
    if (!v.stride) v.stride = elem;
    if (v.stride < elem) fail(PT_ERR_PARSE, "glTF: accessor %lld: byteStride %zu is smaller than its %zu-byte elements",
                              (long long)index, v.stride, elem);
    // view_off + acc_off + (count - 1) * stride + elem <= buf.size(), every step checked
    bool fits = range_fits(view_off, acc_off, buf.size());
    const size_t off = fits ? view_off + acc_off : 0;
    if (fits && v.count) fits = range_fits(off, elem, buf.size()) && (v.count - 1) <= (buf.size() - off - elem) / v.stride;
    if (!fits) fail(PT_ERR_PARSE, "glTF: accessor %lld reaches beyond its buffer", (long long)index);
    v.data = buf.data() + off;
    return v;
}
float component(const View& v, size_t i, int c) {
    const uint8_t* p = v.data + i * v.stride;
    switch (v.comp_type) {
        case 5126: {
            float f;
            memcpy(&f, p + 4 * c, 4);
            return f;
        }
        case 5121: return v.normalized ? p[c] / 255.f : (float)p[c];
        case 5123: {
            uint16_t u;
            memcpy(&u, p + 2 * c, 2);
            return v.normalized ? u / 65535.f : (float)u;
        }
        case 5120: return v.normalized ? std::max((int8_t)p[c] / 127.f, -1.f) : (float)(int8_t)p[c];
        case 5122: {
            int16_t s;
            memcpy(&s, p + 2 * c, 2);
            return v.normalized ? std::max(s / 32767.f, -1.f) : (float)s;
        }
        default: fail(PT_ERR_UNSUPPORTED, "glTF: component type %d is not a vertex attribute type", v.comp_type);
    }
}
uint32_t index_at(const View& v, size_t i) {
    const uint8_t* p = v.data + i * v.stride;
    switch (v.comp_type) {
        case 5121: return p[0];
        case 5123: {
            uint16_t u;
            memcpy(&u, p, 2);
            return u;
        }
        case 5125: {
            uint32_t u;
            memcpy(&u, p, 4);
            return u;
        }
        default: fail(PT_ERR_PARSE, "glTF: bad index component type %d", v.comp_type);
    }
}

// ---------------------------------------------------------------- textures (ReverseTextureBank, gltf.rs:18-76)
struct Converter {
    Document doc;
    pth_scene& sc;
    // decoded glTF images (RGBA8), by image index
    struct Image {
        uint32_t w = 0, h = 0;
        std::vector<uint8_t> rgba;
    };
    std::map<int64_t, Image> images;
    // output textures by (image index, kind): kind 0 albedo rgb, 1 alpha, 2 rgb ("vec"), 3 blue as grey, 4 green as grey
    std::map<std::pair<int64_t, int>, int32_t> textures;
    uint32_t n_vec = 0, n_gray = 0, n_alpha = 0, n_albedo = 0;

    Converter(pth_scene& s) : sc(s) {}

    const Image& image(int64_t texture_index) {
        const JVal& tex = element(doc.root, "textures", texture_index);
        int64_t src = tex.index("source");
        auto it = images.find(src);
        if (it != images.end()) return it->second;
        const JVal& img = element(doc.root, "images", src);
        std::vector<uint8_t> bytes;
        if (img.has("uri")) {
            bytes = load_uri(img.string("uri"), doc.base_dir);
        } else {
            const JVal& bv = element(doc.root, "bufferViews", img.index("bufferView"));
            int64_t bi = bv.index("buffer");
            if (bi < 0 || (size_t)bi >= doc.buffers.size()) fail(PT_ERR_PARSE, "glTF: image bufferView without a buffer");
            const size_t off = size_field(bv, "byteOffset", 0, "bufferView"), len = size_field(bv, "byteLength", 0, "bufferView");
            if (!range_fits(off, len, doc.buffers[(size_t)bi].size())) fail(PT_ERR_PARSE, "glTF: image reaches beyond its buffer");
            bytes.assign(doc.buffers[(size_t)bi].begin() + off, doc.buffers[(size_t)bi].begin() + off + len);
        }
        Image out;
        uint8_t* px = nullptr;
        if (pth_png_decode(bytes.data(), bytes.size(), 4, &out.w, &out.h, &px) != PT_OK)
            fail(PT_ERR_PARSE, "glTF: image %lld: %s", (long long)src, pth_last_error());
        out.rgba.assign(px, px + (size_t)out.w * out.h * 4);
        pth_free(px);
        return images[src] = std::move(out);
    }

    int32_t texture(int64_t texture_index, int kind) {
        if (texture_index < 0) return -1;
        const JVal& tex = element(doc.root, "textures", texture_index);
        auto key = std::make_pair(tex.index("source"), kind);
        auto it = textures.find(key);
        if (it != textures.end()) return it->second;
        const Image& im = image(texture_index);
        pt_texture t{};
        t.offset = sc.texels.size();
        t.width = im.w;
        t.height = im.h;
        t.channels = (kind == 0 || kind == 2) ? 3 : 1;
        const size_t n = (size_t)im.w * im.h;
        sc.texels.resize(sc.texels.size() + n * t.channels);
        uint8_t* dst = sc.texels.data() + t.offset;
        for (size_t i = 0; i < n; ++i) {
            const uint8_t* s = im.rgba.data() + 4 * i;
            if (t.channels == 3) {
                dst[3 * i] = s[0];
                dst[3 * i + 1] = s[1];
                dst[3 * i + 2] = s[2];
            } else {
                dst[i] = kind == 1 ? s[3] : kind == 3 ? s[2] : s[1];
            }
        }
        char name[64];
        if (kind == 0) snprintf(name, sizeof name, "albedo_tex_%u.png", n_albedo++);
        else if (kind == 1) snprintf(name, sizeof name, "alpha_tex_%u.png", n_alpha++);
        else if (kind == 2) snprintf(name, sizeof name, "vec_tex_%u.png", n_vec++);
        else snprintf(name, sizeof name, "gray_tex_%u.png", n_gray++);
        sc.textures.push_back(t);
        sc.texture_paths.push_back(name);
        return textures[key] = (int32_t)sc.textures.size() - 1;
    }

    // convert_material (gltf.rs:78-129)
    pt_material material(int64_t index) {
        pt_material m{};
        m.albedo[0] = m.albedo[1] = m.albedo[2] = 1.f;
        m.opacity = 1.f;
        m.metalness = 1.f;   // glTF defaults
        m.roughness = 1.f;
        m.ior = 1.f;
        m.tex_albedo = m.tex_emissive = m.tex_opacity = m.tex_metalness = m.tex_roughness = m.tex_normal = -1;
        if (index < 0) return m;
        const JVal& mat = element(doc.root, "materials", index);
        auto tex_index = [](const JVal* info) -> int64_t { return info ? info->index("index") : -1; };
        if (const JVal* pbr = mat.get("pbrMetallicRoughness")) {
            if (const JVal* f = pbr->get("baseColorFactor")) {
                if (f->size() != 4) fail(PT_ERR_PARSE, "glTF: baseColorFactor needs 4 numbers");
                for (int k = 0; k < 3; ++k) m.albedo[k] = (float)f->arr[k].num;
                m.opacity = (float)f->arr[3].num;
            }
            m.metalness = (float)pbr->number("metallicFactor", 1.0);
            m.roughness = (float)pbr->number("roughnessFactor", 1.0);
            int64_t base = tex_index(pbr->get("baseColorTexture"));
            m.tex_albedo = texture(base, 0);
            m.tex_opacity = texture(base, 1);
            int64_t mr = tex_index(pbr->get("metallicRoughnessTexture"));
            m.tex_metalness = texture(mr, 3);
            m.tex_roughness = texture(mr, 4);
        }
        if (const JVal* f = mat.get("emissiveFactor")) {
            if (f->size() != 3) fail(PT_ERR_PARSE, "glTF: emissiveFactor needs 3 numbers");
            for (int k = 0; k < 3; ++k) m.emissive[k] = (float)f->arr[k].num;
        }
        m.tex_emissive = texture(tex_index(mat.get("emissiveTexture")), 2);
        m.tex_normal = texture(tex_index(mat.get("normalTexture")), 2);
        return m;
    }

    bool have_camera = false;

    void mesh(const JVal& node, const M4& world) {
        const JVal& mesh = element(doc.root, "meshes", node.index("mesh"));
        const JVal* prims = mesh.get("primitives");
        for (size_t pi = 0; prims && pi < prims->size(); ++pi) {
            const JVal& prim = prims->arr[pi];
            int mode = (int)prim.number("mode", 4);
            if (mode != 4) fail(PT_ERR_UNSUPPORTED, "glTF: primitive mode %d is not a triangle list", mode);
            const JVal* attrs = prim.get("attributes");
            if (!attrs || attrs->index("POSITION") < 0) fail(PT_ERR_PARSE, "glTF: primitive without POSITION");
            View pos = accessor_view(doc, attrs->index("POSITION"));
            if (pos.n_comp != 3) fail(PT_ERR_PARSE, "glTF: POSITION must be VEC3");
            View nrm{}, uv{};
            const bool has_n = attrs->index("NORMAL") >= 0, has_uv = attrs->index("TEXCOORD_0") >= 0;
            if (has_n) nrm = accessor_view(doc, attrs->index("NORMAL"));
            if (has_uv) uv = accessor_view(doc, attrs->index("TEXCOORD_0"));
            if ((has_n && (nrm.n_comp != 3 || nrm.count < pos.count)) || (has_uv && (uv.n_comp != 2 || uv.count < pos.count)))
                fail(PT_ERR_PARSE, "glTF: NORMAL / TEXCOORD_0 do not match POSITION");
            auto vertex = [&](uint32_t i, float* out8) {
                if (i >= pos.count) fail(PT_ERR_PARSE, "glTF: vertex index %u out of range", i);
                const float p[3] = {component(pos, i, 0), component(pos, i, 1), component(pos, i, 2)};
                float w[4];
                for (int r = 0; r < 4; ++r) w[r] = world.m[r] * p[0] + world.m[4 + r] * p[1] + world.m[8 + r] * p[2] + world.m[12 + r];
                out8[0] = w[0] / w[3];
                out8[1] = w[1] / w[3];
                out8[2] = w[2] / w[3];
                out8[3] = out8[4] = out8[5] = 0.f;
                if (has_n) {
                    const float n[3] = {component(nrm, i, 0), component(nrm, i, 1), component(nrm, i, 2)};
                    float t[3];
                    for (int r = 0; r < 3; ++r) t[r] = world.m[r] * n[0] + world.m[4 + r] * n[1] + world.m[8 + r] * n[2];
                    const float inv = 1.0f / std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
                    for (int r = 0; r < 3; ++r) out8[3 + r] = t[r] * inv;
                }
                out8[6] = has_uv ? component(uv, i, 0) : 0.f;
                out8[7] = has_uv ? component(uv, i, 1) : 0.f;
            };
            const size_t first_tri = sc.triangles.size() / 24;
            auto emit = [&](uint32_t a, uint32_t b, uint32_t c) {
                float v[24];
                vertex(a, v);
                vertex(b, v + 8);
                vertex(c, v + 16);
                sc.triangles.insert(sc.triangles.end(), v, v + 24);
            };
            if (prim.index("indices") >= 0) {
                View idx = accessor_view(doc, prim.index("indices"));
                for (size_t i = 0; i + 2 < idx.count; i += 3) emit(index_at(idx, i), index_at(idx, i + 1), index_at(idx, i + 2));
            } else {
                for (size_t i = 0; i + 2 < pos.count; i += 3) emit((uint32_t)i, (uint32_t)i + 1, (uint32_t)i + 2);
            }
            sc.materials.push_back(material(prim.index("material")));
            pt_model m{};
            m.kind = PT_MODEL_MESH;
            m.material = (int32_t)sc.materials.size() - 1;
            m.tri_first = (uint32_t)first_tri;
            m.tri_count = (uint32_t)(sc.triangles.size() / 24 - first_tri);
            sc.models.push_back(m);
        }
    }

    void camera(const JVal& node, const M4& world) {
        if (have_camera) return;   // scenes[0].cameras[0] (gltf.rs:167)
        const JVal& cam = element(doc.root, "cameras", node.index("camera"));
        if (cam.string("type") != "perspective") fail(PT_ERR_UNSUPPORTED, "Orthographic camera not supported");
        const JVal* pers = cam.get("perspective");
        if (!pers) fail(PT_ERR_PARSE, "glTF: perspective camera without parameters");
        memcpy(sc.desc.camera.transform, world.m, sizeof world.m);
        sc.desc.camera.fov = (float)pers->number("yfov", 0.0);
        sc.desc.camera.znear = (float)pers->number("znear", 0.0);
        // (an infinite far plane has no JSON number: the largest f32 stands for it; the integrator does not use zfar)
        sc.desc.camera.zfar = (float)pers->number("zfar", 3.4028234663852886e38);
        have_camera = true;
    }

    void light(const JVal& node, const M4& world) {
        const JVal* ext = node.get("extensions");
        const JVal* kl = ext ? ext->get("KHR_lights_punctual") : nullptr;
        if (!kl) return;
        const JVal* root_ext = doc.root.get("extensions");
        const JVal* defs = root_ext ? root_ext->get("KHR_lights_punctual") : nullptr;
        const JVal* list = defs ? defs->get("lights") : nullptr;
        int64_t li = kl->index("light");
        if (!list || li < 0 || (size_t)li >= list->size()) fail(PT_ERR_PARSE, "glTF: light %lld does not exist", (long long)li);
        const JVal& def = list->arr[(size_t)li];
        float color[3] = {1, 1, 1};
        if (const JVal* c = def.get("color"))
            for (int k = 0; k < 3 && (size_t)k < c->size(); ++k) color[k] = (float)c->arr[k].num;
        const float intensity = (float)def.number("intensity", 1.0);
        pt_light l{};
        for (int k = 0; k < 3; ++k) l.color[k] = color[k] * intensity;   // (color * intensity).into() (gltf.rs:241,250,261)
        const std::string type = def.string("type");
        if (type == "directional") {
            // the light shines along the node's local -Z
            float f[3] = {world.m[8], world.m[9], world.m[10]};
            const float inv = 1.0f / std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
            l.kind = PT_LIGHT_DIRECTIONAL;
            for (int k = 0; k < 3; ++k) l.vec[k] = -1.f * (f[k] * inv);
            l.size = 0.f;
        } else {   // point, and spot -> point (gltf.rs:243-263)
            l.kind = PT_LIGHT_POINT;
            for (int k = 0; k < 3; ++k) l.vec[k] = world.m[12 + k];
            l.size = 0.1f;
        }
        sc.lights.push_back(l);
    }

    void node(int64_t index, const M4& parent, int depth) {
        if (depth > 256) fail(PT_ERR_PARSE, "glTF: node hierarchy too deep (a cycle?)");
        const JVal& n = element(doc.root, "nodes", index);
        const M4 world = mul(parent, node_matrix(n));
        if (n.index("mesh") >= 0) mesh(n, world);
        if (n.index("camera") >= 0) camera(n, world);
        light(n, world);
        if (const JVal* ch = n.get("children"))
            for (size_t i = 0; i < ch->size(); ++i) node((int64_t)ch->arr[i].num, world, depth + 1);
    }
};

void convert(const std::string& input, const std::string& out_dir) {
    struct stat st;
    if (stat(out_dir.c_str(), &st) == 0) {
        if (!S_ISDIR(st.st_mode)) fail(PT_ERR_IO, "'%s' is not a directory", out_dir.c_str());   // gltf.rs:153-155
    } else {   // fs::create_dir_all (gltf.rs:152): every missing component of the path
        for (size_t pos = 0; pos != std::string::npos;) {
            pos = out_dir.find('/', pos + 1);
            const std::string part = out_dir.substr(0, pos);
            if (part.empty()) continue;
            if (mkdir(part.c_str(), 0755) != 0 && errno != EEXIST)
                fail(PT_ERR_IO, "cannot create directory %s: %s", part.c_str(), strerror(errno));
        }
    }
    pth_scene sc;
    Converter cv(sc);
    Document& doc = cv.doc;
    size_t slash = input.find_last_of('/');
    doc.base_dir = slash == std::string::npos ? "" : input.substr(0, slash + 1);
    std::vector<uint8_t> file = read_file(input);
    std::string json;
    if (file.size() >= 12 && !memcmp(file.data(), "glTF", 4)) {   // GLB container: header, JSON chunk, BIN chunk
        auto le32 = [&](size_t o) { return (uint32_t)file[o] | (uint32_t)file[o + 1] << 8 | (uint32_t)file[o + 2] << 16 | (uint32_t)file[o + 3] << 24; };
        size_t pos = 12;
        while (pos + 8 <= file.size()) {
            uint32_t len = le32(pos), type = le32(pos + 4);
            if (pos + 8 + (size_t)len > file.size()) fail(PT_ERR_PARSE, "GLB: truncated chunk");
            if (type == 0x4E4F534A) json.assign((const char*)file.data() + pos + 8, len);
            else if (type == 0x004E4942 && doc.glb_bin.empty()) doc.glb_bin.assign(file.begin() + pos + 8, file.begin() + pos + 8 + len);
            pos += 8 + (size_t)len;
        }
        if (json.empty()) fail(PT_ERR_PARSE, "GLB: no JSON chunk");
    } else {
        json.assign((const char*)file.data(), file.size());
    }
    JParser parser(json.data(), json.size());
    doc.root = parser.value();
    if (doc.root.kind != JVal::Obj) fail(PT_ERR_PARSE, "glTF: the document is not a JSON object");
    if (const JVal* bufs = doc.root.get("buffers"))
        for (size_t i = 0; i < bufs->size(); ++i) {
            const JVal& b = bufs->arr[i];
            if (b.has("uri")) doc.buffers.push_back(load_uri(b.string("uri"), doc.base_dir));
            else doc.buffers.push_back(doc.glb_bin);   // the GLB's BIN chunk
        }
    const JVal* scenes = doc.root.get("scenes");
    if (!scenes || scenes->size() == 0) fail(PT_ERR_INVALID, "No scenes found in gltf file");   // gltf.rs:159-161
    const JVal* roots = scenes->arr[0].get("nodes");
    for (size_t i = 0; roots && i < roots->size(); ++i) cv.node((int64_t)roots->arr[i].num, identity(), 0);
    if (!cv.have_camera) fail(PT_ERR_INVALID, "No camera found");   // (the reference panics, gltf.rs:163-166)
    sc.desc.background[0] = sc.desc.background[1] = sc.desc.background[2] = 0.f;
    sc.finalize();
    if (pth_scene_save_isf(&sc, out_dir.c_str()) != PT_OK) fail(PT_ERR_IO, "%s", pth_last_error());
}

}  // namespace
}  // namespace pth

extern "C" int pth_convert_gltf(const char* input, const char* output_dir) {
    return pth::guarded([&] {
        if (!input || !output_dir) pth::fail(PT_ERR_INVALID, "pth_convert_gltf: null argument");
        pth::convert(input, output_dir);
    });
}
