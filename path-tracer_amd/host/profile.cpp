// profile.yml parser (YAML subset sufficient for the reference's Profile).
//
// Reference: src/config/profile.rs:10-40 (every key optional, serde defaults:
// bounces 4, samples 64, brdf COOK_TORRANCE, tonemap FILMIC; unknown keys
// ignored), src/config/resolution.rs:3-16 (default 1920x1080; when
// `resolution` is given both width and height are required),
// src/renderer/brdf/mod.rs:50-55, src/renderer/tonemap.rs:5-13,
// README.md:37-60.
//
// Accepted syntax: block mappings by indentation, one nested level
// (`resolution:`), the flow form `resolution: {width: 800, height: 600}`,
// `#` comments, single/double quoted scalars, a leading `---`.
#include <cctype>
#include <cerrno>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>

#include "host_common.hpp"

namespace pth {
namespace {

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && isspace((unsigned char)s[a])) ++a;
    while (b > a && isspace((unsigned char)s[b - 1])) --b;
    return s.substr(a, b - a);
}

std::string strip_comment(const std::string& line) {
    bool sq = false, dq = false;
    for (size_t i = 0; i < line.size(); ++i) {
        char c = line[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || isspace((unsigned char)line[i - 1])))
            return line.substr(0, i);
    }
    return line;
}

std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\'')))
        return s.substr(1, s.size() - 2);
    return s;
}

uint32_t parse_uint(const std::string& key, const std::string& raw) {
    std::string v = unquote(trim(raw));
    if (v.empty()) fail(PT_ERR_PARSE, "%s: invalid type: expected an unsigned integer", key.c_str());
    errno = 0;
    char* end = nullptr;
    if (v[0] == '-') fail(PT_ERR_PARSE, "%s: invalid value: integer `%s`, expected an unsigned integer", key.c_str(), v.c_str());
    unsigned long long x = strtoull(v.c_str(), &end, 0);
    if (errno || *end || x > 0xffffffffull)
        fail(PT_ERR_PARSE, "%s: invalid value `%s`, expected an unsigned integer", key.c_str(), v.c_str());
    return (uint32_t)x;
}

void set_key(pt_profile& p, const std::string& key, const std::string& value, bool& has_w, bool& has_h,
             bool in_resolution) {
    if (in_resolution) {
        if (key == "width") {
            p.width = parse_uint("resolution.width", value);
            has_w = true;
        } else if (key == "height") {
            p.height = parse_uint("resolution.height", value);
            has_h = true;
        }
        return;
    }
    if (key == "bounces") p.bounces = parse_uint(key, value);
    else if (key == "samples") p.samples = parse_uint(key, value);
    else if (key == "brdf") {
        std::string v = unquote(trim(value));
        if (v == "COOK_TORRANCE") p.brdf = PT_BRDF_COOK_TORRANCE;
        else fail(PT_ERR_PARSE, "brdf: unknown variant `%s`, expected `COOK_TORRANCE`", v.c_str());
    } else if (key == "tonemap") {
        std::string v = unquote(trim(value));
        if (v == "REINHARD") p.tonemap = PT_TONEMAP_REINHARD;
        else if (v == "FILMIC") p.tonemap = PT_TONEMAP_FILMIC;
        else if (v == "ACES") p.tonemap = PT_TONEMAP_ACES;
        else fail(PT_ERR_PARSE, "tonemap: unknown variant `%s`, expected one of `REINHARD`, `FILMIC`, `ACES`", v.c_str());
    }
    // unknown keys are ignored (no deny_unknown_fields)
}

void parse_flow_mapping(const std::string& body, pt_profile& p, bool& has_w, bool& has_h) {
    // body = "{k: v, k: v}"
    std::string inner = trim(body);
    if (inner.size() < 2 || inner.front() != '{' || inner.back() != '}')
        fail(PT_ERR_PARSE, "resolution: invalid type: expected struct Resolution");
    inner = inner.substr(1, inner.size() - 2);
    std::stringstream ss(inner);
    std::string item;
    while (std::getline(ss, item, ',')) {
        size_t c = item.find(':');
        if (c == std::string::npos) continue;
        set_key(p, unquote(trim(item.substr(0, c))), item.substr(c + 1), has_w, has_h, true);
    }
}

void parse(const std::string& text, pt_profile& p) {
    p.width = 1920;
    p.height = 1080;
    p.samples = 64;
    p.bounces = 4;
    p.brdf = PT_BRDF_COOK_TORRANCE;
    p.tonemap = PT_TONEMAP_FILMIC;

    bool in_res = false, res_given = false, has_w = false, has_h = false;
    size_t res_indent = 0;
    std::stringstream ss(text);
    std::string raw;
    while (std::getline(ss, raw)) {
        std::string line = strip_comment(raw);
        if (trim(line).empty()) continue;
        if (trim(line) == "---" || trim(line) == "...") continue;
        if (line.find('\t') != std::string::npos && line.find_first_not_of(" \t") > line.find('\t'))
            fail(PT_ERR_PARSE, "found character that cannot start any token (tab indentation)");
        size_t indent = line.find_first_not_of(' ');
        std::string body = trim(line);
        size_t colon = body.find(':');
        if (colon == std::string::npos) fail(PT_ERR_PARSE, "invalid type: expected `key: value` in `%s`", body.c_str());
        std::string key = unquote(trim(body.substr(0, colon)));
        std::string value = trim(body.substr(colon + 1));
        if (in_res && indent <= res_indent) in_res = false;
        if (in_res) {
            set_key(p, key, value, has_w, has_h, true);
            continue;
        }
        if (indent != 0) {
            // nested block of an unknown key: ignore
            continue;
        }
        if (key == "resolution") {
            res_given = true;
            if (value.empty()) {
                in_res = true;
                res_indent = indent;
            } else if (value == "~" || value == "null") {
                fail(PT_ERR_PARSE, "resolution: invalid type: unit value, expected struct Resolution");
            } else {
                parse_flow_mapping(value, p, has_w, has_h);
            }
        } else {
            set_key(p, key, value, has_w, has_h, false);
        }
    }
    if (res_given) {
        if (!has_w) fail(PT_ERR_PARSE, "resolution: missing field `width`");
        if (!has_h) fail(PT_ERR_PARSE, "resolution: missing field `height`");
    }
}

}  // namespace
}  // namespace pth

extern "C" {

int pth_profile_parse(const char* yaml_text, pt_profile* out) {
    return pth::guarded([&] {
        if (!out) pth::fail(PT_ERR_INVALID, "pth_profile_parse: null output");
        pth::parse(yaml_text ? yaml_text : "", *out);
    });
}

int pth_profile_load(const char* path, pt_profile* out) {
    return pth::guarded([&] {
        if (!out) pth::fail(PT_ERR_INVALID, "pth_profile_load: null output");
        if (!path) {
            pth::parse("", *out);
            return;
        }
        std::ifstream f(path, std::ios::binary);
        if (!f) pth::fail(PT_ERR_IO, "%s: %s", path, strerror(errno));
        std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        pth::parse(text, *out);
    });
}

}  // extern "C"
