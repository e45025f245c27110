// wfd_probe — study tool for the white_furnace_direct golden (reference src/main.rs:149-165), DESIGN §6.
// TEST INFRASTRUCTURE (includes the oracle's source to reach its internals); never part of the product.
//
// Hypothesis family tested here: the reference's candidate filter (kdtree-ray's ray / AABB slab test) rejects a few
// primary rays that graze a cube whose faces coincide with the faces of its bounding boxes, so that a sample the
// brute-force cast counts as a hit is a miss (background) in the reference.  For each slab-test formulation the tool
// lists those samples, renders the image with them turned into misses and prints its SHA-1.
//
// build: g++ -std=c++17 -O2 -ffp-contract=off -fopenmp -Iinclude -Ioracle tools/wfd/wfd_probe.cpp \
//            -Lpath-tracer_amd -lpthost -Wl,-rpath,$PWD/path-tracer_amd -o build/wfd_probe
#include "../../oracle/pt_oracle.cpp"
#include "pthost.h"
#include <cstdio>
#include <map>

struct Sha1 {  // FIPS 180-1
    uint32_t h[5] = {0x67452301, 0xEFCDAB89, 0x98BADCFE, 0x10325476, 0xC3D2E1F0};
    static uint32_t rol(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
    void block(const uint8_t* p) {
        uint32_t w[80];
        for (int i = 0; i < 16; ++i) w[i] = (p[4 * i] << 24) | (p[4 * i + 1] << 16) | (p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 80; ++i) w[i] = rol(w[i - 3] ^ w[i - 8] ^ w[i - 14] ^ w[i - 16], 1);
        uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4];
        for (int i = 0; i < 80; ++i) {
            uint32_t f, k;
            if (i < 20) { f = (b & c) | (~b & d); k = 0x5A827999; }
            else if (i < 40) { f = b ^ c ^ d; k = 0x6ED9EBA1; }
            else if (i < 60) { f = (b & c) | (b & d) | (c & d); k = 0x8F1BBCDC; }
            else { f = b ^ c ^ d; k = 0xCA62C1D6; }
            uint32_t t = rol(a, 5) + f + e + k + w[i];
            e = d; d = c; c = rol(b, 30); b = a; a = t;
        }
        h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e;
    }
    static std::string hex(const uint8_t* data, size_t n) {
        Sha1 s;
        size_t i = 0;
        for (; i + 64 <= n; i += 64) s.block(data + i);
        uint8_t tail[128] = {0};
        size_t r = n - i;
        memcpy(tail, data + i, r);
        tail[r] = 0x80;
        size_t tl = (r + 9 <= 64) ? 64 : 128;
        uint64_t bits = (uint64_t)n * 8;
        for (int k = 0; k < 8; ++k) tail[tl - 1 - k] = (uint8_t)(bits >> (8 * k));
        s.block(tail);
        if (tl == 128) s.block(tail + 64);
        char out[41];
        for (int k = 0; k < 5; ++k) snprintf(out + 8 * k, 9, "%08x", s.h[k]);
        return out;
    }
};

static double GAP = 2e-6;
static const char* GOLD = "6838e727798bd33f2f796be3edaa893445087159";

// slab-test formulations
//  bit 0: strict ">" instead of ">="      bit 1: divide by d instead of multiplying by 1/d
//  bit 2: compare against tmin only (no max(tmin, 0))
static bool slab(int variant, const Box& b, const Ray& r) {
    const float o[3] = {r.origin.x, r.origin.y, r.origin.z}, d[3] = {r.direction.x, r.direction.y, r.direction.z};
    float tmin = -INFINITY, tmax = INFINITY;
    for (int a = 0; a < 3; ++a) {
        float t1, t2;
        if (variant & 2) { t1 = (b.mn[a] - o[a]) / d[a]; t2 = (b.mx[a] - o[a]) / d[a]; }
        else { float inv = 1.0f / d[a]; t1 = (b.mn[a] - o[a]) * inv; t2 = (b.mx[a] - o[a]) * inv; }
        tmin = fmaxf(tmin, fminf(t1, t2));
        tmax = fminf(tmax, fmaxf(t1, t2));
    }
    float lo = (variant & 4) ? tmin : fmaxf(tmin, 0.f);
    return (variant & 1) ? tmax > lo : tmax >= lo;
}

struct Event { uint32_t pixel, sample; int model; V3 color, alt; float gap; };

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "tests/golden/scenes/white_furnace_direct/scene.isf";
    if (argc > 3) GAP = atof(argv[3]);
    pth_scene* hs = nullptr;
    if (pth_scene_load_isf(path, &hs)) { fprintf(stderr, "load: %s\n", pth_last_error()); return 1; }
    pto_scene* s = nullptr;
    if (pto_scene_create(pth_scene_desc(hs), PTO_BRUTE_FORCE, &s)) { fprintf(stderr, "%s\n", pto_last_error()); return 1; }
    pt_profile p{};
    p.width = 800; p.height = 600; p.samples = 16; p.bounces = 0; p.brdf = 0; p.tonemap = PT_TONEMAP_FILMIC;
    size_t nm = s->models.size();
    std::vector<Box> mbox(nm);
    for (size_t m = 0; m < nm; ++m) {
        Box b; for (int a = 0; a < 3; ++a) { b.mn[a] = INFINITY; b.mx[a] = -INFINITY; }
        uint32_t first = s->prim_first[m], end = m + 1 < nm ? s->prim_first[m + 1] : (uint32_t)s->prim_model.size();
        for (uint32_t q = first; q < end; ++q) {  // exact bounds (Triangle::bound, triangle.rs:84-122), no padding
            const float* f = &s->triangles[(size_t)(s->models[m].tri_first + (q - first)) * 24];
            for (int k = 0; k < 3; ++k)
                for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], f[8 * k + a]); b.mx[a] = fmaxf(b.mx[a], f[8 * k + a]); } }
        mbox[m] = b;
    }
    uint64_t npix = (uint64_t)p.width * p.height;
    const int NV = 8;
    std::vector<V3> base(npix);
    std::vector<std::vector<Event>> ev(NV);
    std::vector<Event> near;
#pragma omp parallel
    {
        Ctx c(*s, p);
        std::vector<std::vector<Event>> lev(NV);
        std::vector<Event> lnear;
#pragma omp for schedule(dynamic, 64)
        for (int64_t i = 0; i < (int64_t)npix; ++i) {
            V3 pixel = v3(0, 0, 0);
            for (uint32_t cs = 1; cs <= p.samples; ++cs) {
                StdRng rng((uint64_t)cs + (uint64_t)i * p.samples);
                Ray ray = primary_ray(*s, p, (uint64_t)i, rng);
                StdRng rng2 = rng;
                V3 color = render_pixel(c, ray, rng2);
                pixel = pixel + color;
                ray_cast(*s, ray, c.primary, &c.st.numeric_errors);
                if (c.primary.hits.empty()) continue;
                int m = c.primary.hits[0].model;
                bool single = true;
                for (auto& h : c.primary.hits) single &= (h.model == m);
                V3 alt = v3(s->d.background[0], s->d.background[1], s->d.background[2]);
                if (!single) continue;  // another cube behind: not handled (does not happen in this scene)
                // gap of the slab test in f64
                {
                    double tmin = -1e300, tmax = 1e300;
                    const float o[3] = {ray.origin.x, ray.origin.y, ray.origin.z}, d[3] = {ray.direction.x, ray.direction.y, ray.direction.z};
                    for (int a = 0; a < 3; ++a) { double t1 = ((double)mbox[m].mn[a] - o[a]) / d[a], t2 = ((double)mbox[m].mx[a] - o[a]) / d[a];
                        tmin = std::max(tmin, std::min(t1, t2)); tmax = std::min(tmax, std::max(t1, t2)); }
                    double gap = (tmax - tmin) / tmin;
                    if (gap < GAP) lnear.push_back({(uint32_t)i, cs, m, color, alt, (float)gap});
                }
                for (int v = 0; v < NV; ++v)
                    if (!slab(v, mbox[m], ray)) lev[v].push_back({(uint32_t)i, cs, m, color, alt, 0.f});
            }
            base[i] = pixel;
        }
#pragma omp critical
        {
            for (int v = 0; v < NV; ++v) ev[v].insert(ev[v].end(), lev[v].begin(), lev[v].end());
            near.insert(near.end(), lnear.begin(), lnear.end());
        }
    }
    auto image_hash = [&](const std::vector<V3>& acc) {
        std::vector<uint8_t> img(npix * 3);
        for (uint64_t i = 0; i < npix; ++i) post_processing(p.tonemap, acc[i] / (float)p.samples, &img[3 * i]);
        return Sha1::hex(img.data(), img.size());
    };
    printf("baseline %s\n", image_hash(base).c_str());
    // re-accumulate a pixel with some samples replaced (accumulation order matters: redo the f32 sum)
    auto rerender = [&](const std::vector<Event>& evs) {
        std::vector<V3> acc = base;
        std::map<uint32_t, std::vector<const Event*>> by;
        for (auto& e : evs) by[e.pixel].push_back(&e);
        Ctx c(*s, p);
        for (auto& kv : by) {
            V3 pixel = v3(0, 0, 0);
            for (uint32_t cs = 1; cs <= p.samples; ++cs) {
                StdRng rng((uint64_t)cs + (uint64_t)kv.first * p.samples);
                Ray ray = primary_ray(*s, p, kv.first, rng);
                V3 color = render_pixel(c, ray, rng);
                for (auto* e : kv.second) if (e->sample == cs) color = e->alt;
                pixel = pixel + color;
            }
            acc[kv.first] = pixel;
        }
        return acc;
    };
    for (int v = 0; v < NV; ++v) {
        std::string h = image_hash(rerender(ev[v]));
        printf("variant %d (%s %s %s): %zu samples become misses -> %s%s\n", v, v & 1 ? ">" : ">=", v & 2 ? "div" : "mul-inv",
               v & 4 ? "tmin" : "max(tmin,0)", ev[v].size(), h.c_str(), h == GOLD ? "  *** MATCH ***" : "");
    }
    std::sort(near.begin(), near.end(), [](const Event& a, const Event& b) { return a.gap < b.gap; });
    printf("%zu samples hit a cube with a relative slab gap < %g:\n", near.size(), GAP);
    for (size_t k = 0; k < near.size() && k < 60; ++k)
        printf("  px %u (x %u y %u) sample %u cube %d gap %.3g color %.4f\n", near[k].pixel, near[k].pixel % p.width,
               near[k].pixel / p.width, near[k].sample, near[k].model, near[k].gap, near[k].color.x);
    // subset search over the nearest events
    size_t N = std::min<size_t>(near.size(), argc > 2 ? atoi(argv[2]) : 16);
    printf("subset search over the %zu nearest events\n", N);
    for (uint64_t mask = 1; mask < (1ull << N); ++mask) {
        std::vector<Event> sel;
        for (size_t k = 0; k < N; ++k) if (mask >> k & 1) sel.push_back(near[k]);
        std::string h = image_hash(rerender(sel));
        if (h == GOLD) { printf("*** MATCH with subset mask %llx\n", (unsigned long long)mask); return 0; }
    }
    printf("no subset matches\n");
    return 0;
}
