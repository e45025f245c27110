// Shared helpers of the host library (error slot, owned scene storage).
#pragma once
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "pthost.h"

namespace pth {

// Error carrying a pt_status code; caught at the C boundary.
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

void set_last_error(const std::string& msg);

// Run `fn`, translate exceptions into a status code + thread-local message.
template <class F>
int guarded(F&& fn) {
    try {
        fn();
        return PT_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("out of memory");
        return PT_ERR_INVALID;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return PT_ERR_INVALID;
    }
}

}  // namespace pth

// Storage behind a pth_scene handle; `desc` points into the vectors.
struct pth_scene {
    std::vector<pt_model> models;
    std::vector<pt_material> materials;
    std::vector<pt_texture> textures;
    std::vector<std::string> texture_paths;  // canonical path or generator name
    std::vector<pt_light> lights;
    std::vector<float> triangles;
    std::vector<uint8_t> texels;
    pt_scene_desc desc{};

    void finalize() {
        desc.n_models = (uint32_t)models.size();
        desc.n_materials = (uint32_t)materials.size();
        desc.n_textures = (uint32_t)textures.size();
        desc.n_lights = (uint32_t)lights.size();
        desc.n_triangles = triangles.size() / 24;
        desc.n_texel_bytes = texels.size();
        desc.models = models.data();
        desc.materials = materials.data();
        desc.textures = textures.data();
        desc.lights = lights.data();
        desc.triangles = triangles.data();
        desc.texels = texels.data();
    }
};
