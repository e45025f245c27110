// slab_events — lists every ray_cast of a render in which kdtree-ray's slab test against a model's bounding box
// rejects a model that Triangle::intersect / the sphere test would have hit (study tool for DESIGN §6, test infrastructure).
// usage: slab_events scene.isf bounces [x y]...   (all pixels when no pixel is given)
#include "../../oracle/pt_oracle.cpp"
#include "pthost.h"
#include <cstdio>

namespace {
struct Log { uint32_t pixel, sample; int model; Ray ray; int bind_lo, bind_hi; float tmin, tmax; int nhits; };
thread_local std::vector<Log>* g_log = nullptr;
thread_local uint32_t g_pixel, g_sample;
}

int main(int argc, char** argv) {
    if (argc < 3) return 1;
    pth_scene* hs = nullptr;
    if (pth_scene_load_isf(argv[1], &hs)) { fprintf(stderr, "load: %s\n", pth_last_error()); return 1; }
    pto_scene* s = nullptr;
    pto_scene_create(pth_scene_desc(hs), PTO_BRUTE_FORCE | PTO_NO_SCENE_SLAB, &s);
    pt_profile p{};
    p.width = 800; p.height = 600; p.samples = 16; p.bounces = atoi(argv[2]); p.brdf = 0; p.tonemap = PT_TONEMAP_FILMIC;
    std::vector<uint64_t> pixels;
    for (int k = 3; k + 1 < argc; k += 2) pixels.push_back(atoi(argv[k]) + (uint64_t)atoi(argv[k + 1]) * p.width);
    uint64_t npix = (uint64_t)p.width * p.height;
    if (pixels.empty()) for (uint64_t i = 0; i < npix; ++i) pixels.push_back(i);
    std::vector<Log> all;
#pragma omp parallel
    {
        std::vector<Log> mine;
        Ctx c(*s, p);
        CastScratch sc;
#pragma omp for schedule(dynamic, 64)
        for (int64_t k = 0; k < (int64_t)pixels.size(); ++k) {
            uint64_t i = pixels[k];
            for (uint32_t cs = 1; cs <= p.samples; ++cs) {
                StdRng rng(cs + i * p.samples);
                Ray ray = primary_ray(*s, p, i, rng);
                // walk the path by hand so that every cast can be inspected: replicate render_pixel's casts by
                // intercepting ray_cast through a wrapper is not possible without touching the oracle, so only the
                // casts whose rays can be reconstructed here are examined: the camera ray and, through the hits, the
                // shadow rays of the first surface.
                ray_cast(*s, ray, sc, &c.st.numeric_errors);
                std::vector<int> seen;
                for (auto& h : sc.hits) {
                    if (std::find(seen.begin(), seen.end(), h.model) != seen.end()) continue;
                    seen.push_back(h.model);
                    if (!kdtree_ray_slab(s->model_box[h.model], ray)) {
                        const Box& b = s->model_box[h.model];
                        const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
                        float tmin = -INFINITY, tmax = INFINITY; int lo = -1, hi = -1;
                        for (int a = 0; a < 3; ++a) { float inv = 1.0f / d[a]; float t1 = (b.mn[a] - o[a]) * inv, t2 = (b.mx[a] - o[a]) * inv;
                            float l = fminf(t1, t2), u = fmaxf(t1, t2); if (l > tmin) { tmin = l; lo = a * 2 + (t1 < t2 ? 0 : 1); } if (u < tmax) { tmax = u; hi = a * 2 + (t1 < t2 ? 1 : 0); } }
                        mine.push_back({(uint32_t)i, cs, h.model, ray, lo, hi, tmin, tmax, (int)sc.hits.size()});
                    }
                }
            }
        }
#pragma omp critical
        all.insert(all.end(), mine.begin(), mine.end());
    }
    const char* fn[6] = {"x=min", "x=max", "y=min", "y=max", "z=min", "z=max"};
    for (auto& e : all) {
        const Box& b = s->model_box[e.model];
        printf("px (%u,%u) sample %u model %d: enters %s exits %s tmin %.9g tmax %.9g  box [%g %g %g]-[%g %g %g] hits on ray %d\n",
               e.pixel % p.width, e.pixel / p.width, e.sample, e.model, fn[e.bind_lo], fn[e.bind_hi], e.tmin, e.tmax,
               b.mn[0], b.mn[1], b.mn[2], b.mx[0], b.mx[1], b.mx[2], e.nhits);
    }
    printf("%zu camera-ray events\n", all.size());
    return 0;
}
