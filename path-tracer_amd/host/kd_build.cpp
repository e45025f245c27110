// SAH KD-tree builder over every primitive of the scene.
//
// Replaces the third-party `kdtree-ray` crate the reference builds its two
// tree levels with (src/scene/internal/mod.rs:42 — scene tree over model
// AABBs; src/scene/internal/model.rs:96 — per-mesh tree over triangle AABBs;
// Bounded impls: triangle.rs:84-122, model.rs:76-86).  The reference uses the
// tree only as a candidate filter in front of Triangle::intersect /
// Model::intersect (src/renderer/utils.rs:11-21), so any structure that
// never drops a primitive whose intersect() succeeds yields the same image
// (SURVEY §0.2).  This builder therefore uses ONE tree over all triangles
// and spheres, keyed by the global primitive id (model order, triangle order
// inside a model) that also defines the reference's tie order.
//
// Robustness rules that keep the "never drop a hit" property:
//   * "perfect splits": a primitive is assigned to a child by the bounds of (primitive ∩ node box) -
//     triangles are clipped to the node box (Sutherland–Hodgman in f64, rounded outward), spheres keep
//     their AABB; below = { clipped min < split }, above = { clipped max > split }, primitives lying IN
//     the plane go to the cheaper side.  A primitive that merely touches the plane from one side is NOT
//     duplicated (on tessellated meshes, where split planes sit on shared vertices, duplicating them
//     multiplies the leaf references ~10x), and a plane ON the node boundary may peel planar primitives
//     into a zero-thickness child (a ground plane on the scene-box face);
//   * split positions are (clipped) primitive bounds: exact f32 values;
//   * in exchange the device traversal visits BOTH children whenever the plane parameter lies within a
//     relative epsilon of the node's ray interval, accepts hits outside the current leaf interval, and
//     keeps walking while the next node starts before the best hit (plus slack), see
//     csrc/pt_integrator.h kd_traverse() and csrc/pt_wavefront.h trav_step().
//
// Cost model: surface-area heuristic with an exact sweep over the clipped bounds (after Wald & Havran /
// pbrt's KdTreeAccel), traversal cost 1, intersection cost 24, empty-space bonus 0.2, leaves of at most 4,
// depth cap 16 + 1.3 log2 n, the top 7 levels built in parallel; all tunable through PT_KD_ISECT_COST / PT_KD_MAX_LEAF / PT_KD_EMPTY_BONUS /
// PT_KD_MAX_DEPTH / PT_KD_CLIP for experiments (measured optima).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <memory>

#include "host_common.hpp"

namespace pth {
namespace {

struct Box {
    float mn[3], mx[3];
};

struct Edge {
    float t;
    uint32_t key;  // prim << 1 | is_end
};

struct Sub {
    std::vector<pth_kd_node> nodes;
    std::vector<uint32_t> refs;
    uint32_t depth = 0;
    uint64_t leaves = 0;
    double sa_interior = 0, sa_tests = 0;  // un-normalised surface-area sums
};

// Triangle vertices (doubles) for clipping; spheres keep their AABB.
struct PrimGeom {
    double v[3][3];
    bool is_tri;
};

// Bounds of (triangle ∩ box), Sutherland–Hodgman in double precision against a box grown by a
// relative epsilon (so a triangle that only touches the box is kept), rounded outward to f32 and
// clamped to `box` and to the triangle's own AABB.  Returns false when the triangle misses the box.
// `pad`: the primitive is FATTENED by pad[a] along axis a (see kd_build: the slop of the f32 Möller–Trumbore test): the
// clip planes move out by it and so do the resulting bounds (a box around (triangle + cube of half-edge pad) ∩ box).
static bool clip_triangle_bounds(const PrimGeom& g, const Box& tri_box, const Box& box, const float* pad, Box& out) {
    double poly[2][16][3];
    int n = 3, cur = 0;
    for (int i = 0; i < 3; ++i)
        for (int a = 0; a < 3; ++a) poly[0][i][a] = g.v[i][a];
    for (int a = 0; a < 3 && n > 0; ++a) {
        for (int side = 0; side < 2 && n > 0; ++side) {
            double scale = std::max(std::fabs((double)box.mn[a]), std::fabs((double)box.mx[a]));
            double eps = 1e-6 * scale + 1e-9 + (double)pad[a];
            double plane = side == 0 ? (double)box.mn[a] - eps : (double)box.mx[a] + eps;
            int m = 0;
            double(*in)[3] = poly[cur];
            double(*outp)[3] = poly[cur ^ 1];
            for (int i = 0; i < n; ++i) {
                const double* p = in[i];
                const double* q = in[(i + 1) % n];
                bool pin = side == 0 ? p[a] >= plane : p[a] <= plane;
                bool qin = side == 0 ? q[a] >= plane : q[a] <= plane;
                if (pin) {
                    for (int k = 0; k < 3; ++k) outp[m][k] = p[k];
                    ++m;
                }
                if (pin != qin) {
                    double t = (plane - p[a]) / (q[a] - p[a]);
                    for (int k = 0; k < 3; ++k) outp[m][k] = p[k] + t * (q[k] - p[k]);
                    outp[m][a] = plane;
                    ++m;
                }
            }
            n = m;
            cur ^= 1;
        }
    }
    if (n == 0) return false;
    for (int a = 0; a < 3; ++a) {
        double lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < n; ++i) {
            lo = std::min(lo, poly[cur][i][a]);
            hi = std::max(hi, poly[cur][i][a]);
        }
        lo -= (double)pad[a];
        hi += (double)pad[a];
        float flo = (float)lo, fhi = (float)hi;
        if ((double)flo > lo) flo = std::nextafterf(flo, -INFINITY);
        if ((double)fhi < hi) fhi = std::nextafterf(fhi, INFINITY);
        out.mn[a] = std::max(std::max(flo, box.mn[a]), tri_box.mn[a]);
        out.mx[a] = std::min(std::min(fhi, box.mx[a]), tri_box.mx[a]);
        if (out.mn[a] > out.mx[a]) {  // only by rounding at a touching contact: keep it as a point
            float mid = std::min(std::max(out.mn[a], box.mn[a]), box.mx[a]);
            out.mn[a] = out.mx[a] = mid;
        }
    }
    return true;
}

struct Builder {
    const std::vector<Box>& boxes;      // fattened (see kd_build)
    const std::vector<PrimGeom>& geom;
    const std::vector<float>& pads;     // 3 per primitive
    float isect_cost;
    float trav_cost = 1.0f;
    float empty_bonus = 0.5f;
    uint32_t max_leaf;
    int par_levels;
    size_t strict_below = 64;
    bool clip = true;

    static double area(const Box& b) {
        double dx = (double)b.mx[0] - b.mn[0], dy = (double)b.mx[1] - b.mn[1], dz = (double)b.mx[2] - b.mn[2];
        return 2.0 * (dx * dy + dx * dz + dy * dz);
    }

    void leaf(Sub& out, const Box& nb, const std::vector<uint32_t>& prims, uint32_t level) {
        out.sa_tests += area(nb) * (double)prims.size();
        pth_kd_node n;
        n.w0 = (uint32_t)out.refs.size();
        n.w1 = ((uint32_t)prims.size() << 2) | 3u;
        out.nodes.push_back(n);
        out.refs.insert(out.refs.end(), prims.begin(), prims.end());
        out.depth = std::max(out.depth, level);
        out.leaves++;
    }

    // cb[i] = bounds of primitive prims[i] clipped to the node box nb ("perfect splits")
    void build(Sub& out, const Box& nb, std::vector<uint32_t>&& prims, std::vector<Box>&& cb, int depth_left,
               int bad, uint32_t level) {
        const size_t n = prims.size();
        if (n <= max_leaf || depth_left == 0) {
            leaf(out, nb, prims, level);
            return;
        }
        // SAH sweep
        float d[3] = {nb.mx[0] - nb.mn[0], nb.mx[1] - nb.mn[1], nb.mx[2] - nb.mn[2]};
        float total_sa = 2.f * (d[0] * d[1] + d[0] * d[2] + d[1] * d[2]);
        float inv_total_sa = total_sa > 0 ? 1.f / total_sa : 0.f;
        float old_cost = isect_cost * (float)n;
        float best_cost = INFINITY;
        int best_axis = -1;
        struct AxisBest {
            float cost = INFINITY, split = 0.f;
            int axis = -1;
            bool planar_below = true;
        };
        // the sweep over one axis (the best plane of that axis; ties keep the first, as a loop over the axes would)
        auto sweep = [&](int axis, AxisBest& out, std::vector<Edge>& edges) {
            edges.resize(2 * n);
            float bc = INFINITY, bs = 0.f;
            int ba = -1;
            bool bp = true;
            auto run = [&]() {
            if (!(d[axis] > 0.f)) return;  // flat box: nothing to cut on this axis
                const float lo = nb.mn[axis], hi = nb.mx[axis];
                for (size_t i = 0; i < n; ++i) {
                    const Box& b = cb[i];  // already inside the node box
                    edges[2 * i] = {b.mn[axis], (uint32_t)i << 1};
                    edges[2 * i + 1] = {b.mx[axis], ((uint32_t)i << 1) | 1u};
                }
                std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) { return a.t < b.t; });
                // Candidate planes = distinct (clamped) AABB bounds.  For a plane at t:
                //   L = { min < t }, R = { max > t }, P = { min == max == t } (lying in the plane)
                // A primitive that only touches the plane from one side stays on that side; P goes
                // to whichever side is cheaper.  A plane ON the node boundary is allowed when it
                // peels planar primitives into a zero-thickness child (a ground plane at the
                // bottom of the scene box): rays that never reach the plane skip them entirely.
                size_t n_below = 0, n_above = n;
                int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
                for (size_t i = 0; i < 2 * n;) {
                    float t = edges[i].t;
                    size_t j = i, ends = 0, starts = 0, planar = 0;
                    for (; j < 2 * n && edges[j].t == t; ++j) {
                        if (edges[j].key & 1u) ++ends;
                        else {
                            ++starts;
                            if (cb[edges[j].key >> 1].mx[axis] == t) ++planar;
                        }
                    }
                    n_above -= ends;
                    const bool at_lo = t == lo, at_hi = t == hi;
                    if ((!at_lo && !at_hi) || planar > 0) {
                        float below_sa = 2.f * (d[a1] * d[a2] + (t - lo) * (d[a1] + d[a2]));
                        float above_sa = 2.f * (d[a1] * d[a2] + (hi - t) * (d[a1] + d[a2]));
                        float pb = below_sa * inv_total_sa, pa = above_sa * inv_total_sa;
                        for (int side = 0; side < 2; ++side) {
                            bool planar_below = side == 0;
                            if (at_lo && !planar_below) continue;  // would reproduce the parent
                            if (at_hi && planar_below) continue;
                            if (planar == 0 && !planar_below) continue;
                            size_t nbel = n_below + (planar_below ? planar : 0);
                            size_t nabv = n_above + (planar_below ? 0 : planar);
                            float eb = (nabv == 0 || nbel == 0) ? empty_bonus : 0.f;
                            float cost = trav_cost + isect_cost * (1.f - eb) * (pb * (float)nbel + pa * (float)nabv);
                            if (cost < bc) {
                                bc = cost;
                                ba = axis;
                                bs = t;
                                bp = planar_below;
                            }
                        }
                    }
                    n_below += starts;
                    i = j;
                }
            };
            run();
            out.cost = bc;
            out.split = bs;
            out.axis = ba;
            out.planar_below = bp;
        };
        AxisBest per_axis[3];
        if ((int)level < par_levels && n > 100000) {   // the top of a large tree: the three sweeps side by side
            std::vector<Edge> e0, e1, e2;
            auto f1 = std::async(std::launch::async, [&] { sweep(1, per_axis[1], e1); });
            auto f2 = std::async(std::launch::async, [&] { sweep(2, per_axis[2], e2); });
            sweep(0, per_axis[0], e0);
            f1.get();
            f2.get();
        } else {
            std::vector<Edge> edges;
            for (int axis = 0; axis < 3; ++axis) sweep(axis, per_axis[axis], edges);
        }
        float best_split = 0.f;
        bool best_planar_below = true;
        for (int axis = 0; axis < 3; ++axis)
            if (per_axis[axis].axis >= 0 && per_axis[axis].cost < best_cost) {
                best_cost = per_axis[axis].cost;
                best_axis = axis;
                best_split = per_axis[axis].split;
                best_planar_below = per_axis[axis].planar_below;
            }
        if (best_cost > old_cost) ++bad;
        // small nodes stop as soon as splitting no longer pays (Wald's automatic
        // termination); large ones tolerate a few bad refines like pbrt does
        if (best_axis == -1 || bad == 3 || (best_cost >= old_cost && n <= strict_below)) {
            leaf(out, nb, prims, level);
            return;
        }
        // classify (prims is in ascending id order, so both children stay sorted); every
        // primitive is re-clipped against the child box it goes to
        float split = best_split;
        Box bb = nb, ab = nb;
        bb.mx[best_axis] = split;
        ab.mn[best_axis] = split;
        std::vector<uint32_t> below, above;
        std::vector<Box> below_cb, above_cb;
        // (range [i0, i1) of the node's primitives into the given lists: in order, so the lists stay sorted by id)
        auto classify = [&](size_t i0, size_t i1, std::vector<uint32_t>& bl, std::vector<Box>& bl_cb, std::vector<uint32_t>& ab_,
                            std::vector<Box>& ab_cb) {
            for (size_t i = i0; i < i1; ++i) {
                uint32_t p = prims[i];
                float mn = cb[i].mn[best_axis], mx = cb[i].mx[best_axis];
                bool planar = mn == best_split && mx == best_split;
                bool to_below = mn < best_split || (planar && best_planar_below);
                bool to_above = mx > best_split || (planar && !best_planar_below);
                for (int side = 0; side < 2; ++side) {
                    if (!(side == 0 ? to_below : to_above)) continue;
                    const Box& child = side == 0 ? bb : ab;
                    Box c;
                    if (geom[p].is_tri && clip) {
                        if (!clip_triangle_bounds(geom[p], boxes[p], child, &pads[(size_t)p * 3], c)) continue;  // misses this child
                    } else {
                        for (int a = 0; a < 3; ++a) {
                            c.mn[a] = std::max(cb[i].mn[a], child.mn[a]);
                            c.mx[a] = std::min(cb[i].mx[a], child.mx[a]);
                        }
                    }
                    (side == 0 ? bl : ab_).push_back(p);
                    (side == 0 ? bl_cb : ab_cb).push_back(c);
                }
            }
        };
        if ((int)level < par_levels && n > 100000) {   // the top of a large tree: sixteen ranges side by side, joined in order
            constexpr size_t R = 16;
            std::vector<uint32_t> pb[R], pa[R];
            std::vector<Box> pbc[R], pac[R];
            std::vector<std::future<void>> fs;
            for (size_t r = 0; r < R; ++r)
                fs.push_back(std::async(std::launch::async, [&, r] { classify(n * r / R, n * (r + 1) / R, pb[r], pbc[r], pa[r], pac[r]); }));
            for (auto& f : fs) f.get();
            for (size_t r = 0; r < R; ++r) {
                below.insert(below.end(), pb[r].begin(), pb[r].end());
                below_cb.insert(below_cb.end(), pbc[r].begin(), pbc[r].end());
                above.insert(above.end(), pa[r].begin(), pa[r].end());
                above_cb.insert(above_cb.end(), pac[r].begin(), pac[r].end());
            }
        } else {
            below.reserve(n);
            above.reserve(n);
            below_cb.reserve(n);
            above_cb.reserve(n);
            classify(0, n, below, below_cb, above, above_cb);
        }
        std::vector<uint32_t>().swap(prims);
        std::vector<Box>().swap(cb);

        size_t me = out.nodes.size();
        out.nodes.push_back({0, 0});
        out.sa_interior += area(nb);
        uint32_t above_idx;
        if ((int)level < par_levels && n > 4000) {
            auto fut = std::async(std::launch::async, [&, this]() {
                auto sub = std::make_unique<Sub>();
                build(*sub, ab, std::move(above), std::move(above_cb), depth_left - 1, bad, level + 1);
                return sub;
            });
            build(out, bb, std::move(below), std::move(below_cb), depth_left - 1, bad, level + 1);
            std::unique_ptr<Sub> sub = fut.get();
            above_idx = (uint32_t)out.nodes.size();
            uint32_t ref_base = (uint32_t)out.refs.size();
            out.nodes.reserve(out.nodes.size() + sub->nodes.size());
            for (pth_kd_node nd : sub->nodes) {
                if ((nd.w1 & 3u) == 3u) nd.w0 += ref_base;
                else nd.w1 = (((nd.w1 >> 2) + above_idx) << 2) | (nd.w1 & 3u);
                out.nodes.push_back(nd);
            }
            out.refs.insert(out.refs.end(), sub->refs.begin(), sub->refs.end());
            out.depth = std::max(out.depth, sub->depth);
            out.leaves += sub->leaves;
            out.sa_interior += sub->sa_interior;
            out.sa_tests += sub->sa_tests;
        } else {
            build(out, bb, std::move(below), std::move(below_cb), depth_left - 1, bad, level + 1);
            above_idx = (uint32_t)out.nodes.size();
            build(out, ab, std::move(above), std::move(above_cb), depth_left - 1, bad, level + 1);
        }
        if (above_idx >= (1u << 30)) fail(PT_ERR_UNSUPPORTED, "KD-tree has too many nodes");
        pth_kd_node nd;
        memcpy(&nd.w0, &split, 4);
        nd.w1 = (above_idx << 2) | (uint32_t)best_axis;
        out.nodes[me] = nd;
    }
};

float env_float(const char* name, float dflt) {
    const char* v = getenv(name);
    return v && *v ? (float)atof(v) : dflt;
}

}  // namespace

void prim_boxes(const pt_scene_desc& d, std::vector<Box>& boxes) {
    boxes.clear();
    boxes.reserve(pth_prim_count(&d));
    for (uint32_t m = 0; m < d.n_models; ++m) {
        const pt_model& mo = d.models[m];
        if (mo.kind == PT_MODEL_MESH) {
            for (uint32_t t = 0; t < mo.tri_count; ++t) {
                const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                Box b;
                for (int a = 0; a < 3; ++a) {
                    // Triangle::bound (triangle.rs:84-122)
                    b.mn[a] = std::min(std::min(v[a], v[8 + a]), v[16 + a]);
                    b.mx[a] = std::max(std::max(v[a], v[8 + a]), v[16 + a]);
                }
                boxes.push_back(b);
            }
        } else {
            Box b;  // Model::bound for spheres (model.rs:80-84)
            for (int a = 0; a < 3; ++a) {
                b.mn[a] = mo.center[a] - mo.radius;
                b.mx[a] = mo.center[a] + mo.radius;
            }
            boxes.push_back(b);
        }
    }
}

static void kd_build(const pt_scene_desc& d, pth_kdtree& out) {
    auto t0 = std::chrono::steady_clock::now();
    memset(&out, 0, sizeof out);
    std::vector<Box> boxes;
    prim_boxes(d, boxes);
    size_t n = boxes.size();
    if (n >= (1u << 30)) fail(PT_ERR_UNSUPPORTED, "too many primitives for the KD-tree (%zu)", n);
    // Fattened primitives.  Triangle::intersect in f32 accepts rays that pass a triangle's edge on the OUTSIDE by a few
    // 1e-6 of the ray's length (u + v comes out just under 1): with exact primitive bounds the ray point at such a hit
    // can lie in a leaf that does not reference the triangle - e.g. beyond the split plane an axis-aligned edge sits in -
    // and the walk's slack along the ray (1e-5 relative) does not reach far enough when the ray runs nearly parallel
    // to that plane.  Found by the grid-vs-KD check on config 5 (4 M triangles, 8.5 G samples: one ray; the origin
    // grids, whose margins come from a rounding analysis of the test, had it right).  So every primitive is entered
    // into the tree with its bounds grown by the padding the oracle's candidate filter uses, 1e-4 |coordinate| + 1e-5
    // per axis (oracle/pt_oracle.cpp prim_box): both filters then see the same fattened primitives.  PT_KD_PAD scales
    // it.  MEASURED (profiles/r03_experiments.txt item 3): any non-zero padding ends the "a primitive that only touches
    // a split plane stays on one side" economy at shared mesh edges - +45 % nodes, +43 % leaf references, +10 % node
    // visits per ray - and costs 4 % of the frame on config 3, 9 % in the closed room.  The default is therefore 0 (exact
    // bounds); the walkers cover the case instead (csrc/pt_integrator.h, "slop model"): the wavefront walker's slack is the
    // ray's own, PT_SLACK_K x the largest |1 / d_axis| between PT_SLACK_MIN and PT_SLACK_MAX, and a ray that would need
    // more than PT_SLACK_MAX - a direction component below 8e-4 - leaves it for the fat-ray walker (k_wf_trace_exact).
    const float pad_scale = env_float("PT_KD_PAD", 0.f);
    std::vector<float> pads(n * 3);
    for (size_t i = 0; i < n; ++i)
        for (int a = 0; a < 3; ++a) {
            const float pad = pad_scale * (1e-4f * std::max(fabsf(boxes[i].mn[a]), fabsf(boxes[i].mx[a])) + 1e-5f);
            pads[i * 3 + a] = pad == pad ? pad : 0.f;
            boxes[i].mn[a] -= pads[i * 3 + a];
            boxes[i].mx[a] += pads[i * 3 + a];
        }
    Box root;
    for (int a = 0; a < 3; ++a) {
        root.mn[a] = INFINITY;
        root.mx[a] = -INFINITY;
    }
    for (const Box& b : boxes)
        for (int a = 0; a < 3; ++a) {
            if (!(b.mn[a] == b.mn[a]) || !(b.mx[a] == b.mx[a]))
                fail(PT_ERR_NUMERIC, "NaN vertex coordinate in scene");
            root.mn[a] = std::min(root.mn[a], b.mn[a]);
            root.mx[a] = std::max(root.mx[a], b.mx[a]);
        }
    if (n == 0)
        for (int a = 0; a < 3; ++a) root.mn[a] = root.mx[a] = 0.f;

    std::vector<PrimGeom> geom(n);
    {
        size_t prim = 0;
        for (uint32_t m = 0; m < d.n_models; ++m) {
            const pt_model& mo = d.models[m];
            if (mo.kind == PT_MODEL_MESH) {
                for (uint32_t t = 0; t < mo.tri_count; ++t, ++prim) {
                    const float* v = d.triangles + (size_t)(mo.tri_first + t) * 24;
                    geom[prim].is_tri = true;
                    for (int k = 0; k < 3; ++k)
                        for (int a = 0; a < 3; ++a) geom[prim].v[k][a] = v[8 * k + a];
                }
            } else {
                geom[prim++].is_tri = false;
            }
        }
    }
    Builder b{boxes, geom, pads, env_float("PT_KD_ISECT_COST", 24.f), 1.0f, 0.5f, 2, 4, 64, true};
    b.clip = env_float("PT_KD_CLIP", 1.f) != 0.f;
    b.max_leaf = (uint32_t)env_float("PT_KD_MAX_LEAF", 4.f);
    // top levels whose two children are built on threads of their own: 2^7 subtrees for the 256 logical CPUs of an MI355X
    // host (config 3: 1.84 s at 4 levels, 1.33 / 1.10 / 1.02 / 0.97 s at 5 / 6 / 7 / 8)
    b.par_levels = (int)env_float("PT_KD_PAR_LEVELS", 7.f);
    b.strict_below = (size_t)env_float("PT_KD_STRICT_BELOW", 64.f);
    // pbrt uses 0.5; on the GPU walk empty leaves are not free (a node fetch + a pop each, 85 % of all leaf
    // visits): 0.2 measured +14 % samples/s over 0.5 on the 500 k-triangle scene (0 / 0.1 / 0.3: +5 / +9 / +11 %)
    b.empty_bonus = env_float("PT_KD_EMPTY_BONUS", 0.2f);
    // pbrt's 8 + 1.3 log2(n) leaves dense regions with leaves of 100+ primitives on tessellated
    // meshes (measured: 1.52 -> 1.76 Gsamples/s going from 33 to 41 levels at 500 k triangles)
    int max_depth = n ? (int)std::lround(16 + 1.3 * std::log2((double)n)) : 0;
    if (max_depth > 62) max_depth = 62;  // device traversal stack bound
    max_depth = (int)env_float("PT_KD_MAX_DEPTH", (float)max_depth);

    Sub sub;
    std::vector<uint32_t> prims(n);
    for (size_t i = 0; i < n; ++i) prims[i] = (uint32_t)i;
    std::vector<Box> root_cb(boxes);
    b.build(sub, root, std::move(prims), std::move(root_cb), max_depth, 0, 0);

    out.n_nodes = sub.nodes.size();
    out.n_refs = sub.refs.size();
    out.n_leaves = sub.leaves;
    out.depth = sub.depth;
    memcpy(out.bounds_min, root.mn, sizeof root.mn);
    memcpy(out.bounds_max, root.mx, sizeof root.mx);
    out.nodes = (pth_kd_node*)malloc(std::max<size_t>(1, sub.nodes.size()) * sizeof(pth_kd_node));
    out.refs = (uint32_t*)malloc(std::max<size_t>(1, sub.refs.size()) * sizeof(uint32_t));
    if (!out.nodes || !out.refs) throw std::bad_alloc();
    memcpy(out.nodes, sub.nodes.data(), sub.nodes.size() * sizeof(pth_kd_node));
    memcpy(out.refs, sub.refs.data(), sub.refs.size() * sizeof(uint32_t));
    double root_area = Builder::area(root);
    out.expected_nodes = root_area > 0 ? sub.sa_interior / root_area : 0;
    out.expected_tests = root_area > 0 ? sub.sa_tests / root_area : 0;
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // namespace pth

extern "C" {

int pth_kd_build(const pt_scene_desc* desc, pth_kdtree* out) {
    return pth::guarded([&] {
        if (!desc || !out) pth::fail(PT_ERR_INVALID, "pth_kd_build: null argument");
        pth::kd_build(*desc, *out);
    });
}

void pth_kd_free(pth_kdtree* kd) {
    if (!kd) return;
    free(kd->nodes);
    free(kd->refs);
    kd->nodes = nullptr;
    kd->refs = nullptr;
}

}  // extern "C"
