// rtw_bvh.h — host BVH2 builder over world-space primitive bounds.
// Replaces the closed OptiX accel builds the reference calls per primitive and per scene
// (geometry/ioSphere.h:45-106, ioAARect.h:41-154, ioMovingSphere.h:46-72,90-218,
//  ioGeometryGroup.h:160-225): one AABB per primitive in object space, carried to world space by
// the instance transform (and, for moving spheres, swept over the motion keys).
// Volume primitives are kept out of the tree: they are tested first, in index order, because
// their intersection programs draw random numbers (DESIGN.md "candidate order").
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/rtw.h"

namespace rtwbvh {

struct Node {  // 32 B
    float mn[3];
    uint32_t left_first;  // inner: index of left child (right = left+1); leaf: first entry in prim_order
    float mx[3];
    uint32_t count;       // 0 = inner node, else number of primitives in the leaf
};

struct Box {
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
    float mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    void add(const float p[3]) {
        for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], p[i]); mx[i] = std::max(mx[i], p[i]); }
    }
    void add(const Box& b) {
        for (int i = 0; i < 3; i++) { mn[i] = std::min(mn[i], b.mn[i]); mx[i] = std::max(mx[i], b.mx[i]); }
    }
    float area() const {
        float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
        return 2.f * (dx * dy + dy * dz + dz * dx);
    }
};

inline bool is_volume(int type) { return type == RTW_PRIM_VOLUME_BOX || type == RTW_PRIM_VOLUME_SPHERE; }

// object-space bounds exactly as the reference's getBounds() functions state them
inline Box object_bounds(const rtw_prim& p) {
    Box b;
    const float e = 0.0001f;
    switch (p.type) {
    case RTW_PRIM_SPHERE:
    case RTW_PRIM_VOLUME_SPHERE: {
        float r = std::fabs(p.p[3]);
        float lo[3] = {p.p[0] - r, p.p[1] - r, p.p[2] - r}, hi[3] = {p.p[0] + r, p.p[1] + r, p.p[2] + r};
        b.add(lo); b.add(hi);
        break;
    }
    case RTW_PRIM_MOVING_SPHERE: {
        float r = std::fabs(p.p[3]);
        for (int k = 0; k < 2; k++) {
            const float* c = &p.p[4 * k];
            float lo[3] = {c[0] - r, c[1] - r, c[2] - r}, hi[3] = {c[0] + r, c[1] + r, c[2] + r};
            b.add(lo); b.add(hi);
        }
        break;
    }
    case RTW_PRIM_RECT_X: { float lo[3] = {p.p[4] - e, p.p[0], p.p[2]}, hi[3] = {p.p[4] + e, p.p[1], p.p[3]}; b.add(lo); b.add(hi); break; }
    case RTW_PRIM_RECT_Y: { float lo[3] = {p.p[0], p.p[4] - e, p.p[2]}, hi[3] = {p.p[1], p.p[4] + e, p.p[3]}; b.add(lo); b.add(hi); break; }
    case RTW_PRIM_RECT_Z: { float lo[3] = {p.p[0], p.p[2], p.p[4] - e}, hi[3] = {p.p[1], p.p[3], p.p[4] + e}; b.add(lo); b.add(hi); break; }
    default: {
        float lo[3] = {p.p[0] - e, p.p[1] - e, p.p[2] - e}, hi[3] = {p.p[3] + e, p.p[4] + e, p.p[5] + e};
        b.add(lo); b.add(hi);
        break;
    }
    }
    return b;
}

inline Box world_bounds(const rtw_prim& p, const rtw_xform& xf) {
    Box ob = object_bounds(p);
    if (p.type == RTW_PRIM_MOVING_SPHERE) {
        // matrix-motion keys translate(C0), translate(C1) act on the object-space box (ioMovingSphere.h:161-203)
        Box sw;
        for (int k = 0; k < 2; k++) {
            const float* c = &p.p[4 * k];
            float lo[3] = {ob.mn[0] + c[0], ob.mn[1] + c[1], ob.mn[2] + c[2]};
            float hi[3] = {ob.mx[0] + c[0], ob.mx[1] + c[1], ob.mx[2] + c[2]};
            sw.add(lo); sw.add(hi);
        }
        ob = sw;
    }
    Box wb;
    for (int c = 0; c < 8; c++) {
        float q[3] = {(c & 1) ? ob.mx[0] : ob.mn[0], (c & 2) ? ob.mx[1] : ob.mn[1], (c & 4) ? ob.mx[2] : ob.mn[2]};
        float w[3];
        for (int i = 0; i < 3; i++) w[i] = xf.m[4 * i] * q[0] + xf.m[4 * i + 1] * q[1] + xf.m[4 * i + 2] * q[2] + xf.m[4 * i + 3];
        wb.add(w);
    }
    // conservative padding: the tree only culls, exact hits are decided by the primitive tests
    for (int i = 0; i < 3; i++) {
        float pad = 1e-4f * std::max(1.0f, std::max(std::fabs(wb.mn[i]), std::fabs(wb.mx[i])));
        wb.mn[i] -= pad; wb.mx[i] += pad;
    }
    return wb;
}

// What the GPU walks: one 64-byte record per INNER node holding the bounds of both children, so a traversal
// step is one burst of four 16-byte loads and the child that is entered next needs no load of its own box.
// A child reference is idx | count << 30: count > 0 (1 or 2) is a leaf whose idx is its first entry in
// prim_order; count == 0 is an inner node and idx its wide-node index.
struct WideNode {
    float lmn[3]; uint32_t lref;
    float lmx[3]; uint32_t pad0;
    float rmn[3]; uint32_t rref;
    float rmx[3]; uint32_t pad1;
};

struct Bvh {
    std::vector<Node> nodes;
    std::vector<WideNode> wide;       // inner nodes only; wide[0] is the root
    std::vector<int32_t> prim_order;  // leaf entries -> primitive index
    int max_depth = 0;
};

namespace detail {
struct Item { Box b; float c[3]; int32_t prim; };

inline void build(std::vector<Item>& items, int lo, int hi, int node_idx, int depth, Bvh& out) {
    out.max_depth = std::max(out.max_depth, depth);
    Box nb, cb;
    for (int i = lo; i < hi; i++) { nb.add(items[i].b); cb.add(items[i].c); }
    for (int i = 0; i < 3; i++) { out.nodes[node_idx].mn[i] = nb.mn[i]; out.nodes[node_idx].mx[i] = nb.mx[i]; }
    int n = hi - lo;
    if (n <= 2) {
        out.nodes[node_idx].left_first = static_cast<uint32_t>(lo);
        out.nodes[node_idx].count = static_cast<uint32_t>(n);
        return;
    }
    // binned SAH (16 bins), best split over all three centroid axes
    int axis = 0;
    float ext = -1.f;
    for (int a = 0; a < 3; a++) { float e = cb.mx[a] - cb.mn[a]; if (e > ext) { ext = e; axis = a; } }
    int mid = (lo + hi) / 2;
    if (ext > 0.f) {
        const int NB = 16;
        float best = FLT_MAX; int best_split = -1, best_axis = axis; float best_k = 0.f;
        for (int ax = 0; ax < 3; ax++) {
            const float e = cb.mx[ax] - cb.mn[ax];
            if (!(e > 0.f)) continue;
            Box bb[NB]; int bc[NB] = {0};
            const float k = NB * (1.f - 1e-6f) / e;
            for (int i = lo; i < hi; i++) {
                int bi = std::min(NB - 1, std::max(0, static_cast<int>((items[i].c[ax] - cb.mn[ax]) * k)));
                bb[bi].add(items[i].b); bc[bi]++;
            }
            Box accl[NB]; int cntl[NB];
            Box run; int rc = 0;
            for (int i = 0; i < NB; i++) { if (bc[i]) run.add(bb[i]); rc += bc[i]; accl[i] = run; cntl[i] = rc; }
            Box runr; int rcr = 0;
            for (int i = NB - 1; i >= 1; i--) {
                if (bc[i]) runr.add(bb[i]);
                rcr += bc[i];
                if (cntl[i - 1] == 0 || rcr == 0) continue;
                float cost = accl[i - 1].area() * cntl[i - 1] + runr.area() * rcr;
                if (cost < best) { best = cost; best_split = i; best_axis = ax; best_k = k; }
            }
        }
        axis = best_axis;
        if (best_split > 0) {
            auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& it2) {
                int bi = std::min(NB - 1, std::max(0, static_cast<int>((it2.c[axis] - cb.mn[axis]) * best_k)));
                return bi < best_split;
            });
            mid = static_cast<int>(it - items.begin());
        }
        if (mid == lo || mid == hi) {
            mid = (lo + hi) / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi,
                             [&](const Item& x, const Item& y) { return x.c[axis] < y.c[axis]; });
        }
    }
    int left = static_cast<int>(out.nodes.size());
    out.nodes.push_back(Node{});
    out.nodes.push_back(Node{});
    out.nodes[node_idx].left_first = static_cast<uint32_t>(left);
    out.nodes[node_idx].count = 0;
    build(items, lo, mid, left, depth + 1, out);
    build(items, mid, hi, left + 1, depth + 1, out);
}
}  // namespace detail

inline Bvh build_bvh(const rtw_prim* prims, uint32_t n_prims, const rtw_xform* xforms) {
    Bvh out;
    std::vector<detail::Item> items;
    for (uint32_t i = 0; i < n_prims; i++) {
        if (is_volume(prims[i].type)) continue;
        detail::Item it;
        it.b = world_bounds(prims[i], xforms[prims[i].xform]);
        for (int a = 0; a < 3; a++) it.c[a] = 0.5f * (it.b.mn[a] + it.b.mx[a]);
        it.prim = static_cast<int32_t>(i);
        items.push_back(it);
    }
    out.nodes.push_back(Node{});
    if (items.empty()) {
        out.nodes[0].count = 0; out.nodes[0].left_first = 0;
        for (int i = 0; i < 3; i++) { out.nodes[0].mn[i] = 1.f; out.nodes[0].mx[i] = -1.f; }
        return out;
    }
    detail::build(items, 0, static_cast<int>(items.size()), 0, 1, out);
    out.prim_order.resize(items.size());
    for (size_t i = 0; i < items.size(); i++) out.prim_order[i] = items[i].prim;
    // inner nodes -> wide records, numbered breadth-first so that a prefix of the array is the top of the tree
    std::vector<int32_t> wid(out.nodes.size(), -1);
    int32_t nw = 0;
    if (out.nodes[0].count == 0) {
        std::vector<uint32_t> queue{0u};
        for (size_t q = 0; q < queue.size(); q++) {
            const Node& nd = out.nodes[queue[q]];
            wid[queue[q]] = nw++;
            for (uint32_t c = 0; c < 2; c++)
                if (out.nodes[nd.left_first + c].count == 0) queue.push_back(nd.left_first + c);
        }
    }
    if (nw == 0) {
        // The root is a leaf (one or two surfaces: a tree scene made of volumes, or RTW_BRUTE_MAX=0 on a tiny scene). The
        // device walk always starts at inner record 0, so the leaf becomes the left child of one inner record whose right
        // child is a box no ray reaches (a point at 3e38: its slab interval never meets [tmin, best_t]).
        WideNode w{};
        const Node& rt = out.nodes[0];
        for (int a = 0; a < 3; a++) { w.lmn[a] = rt.mn[a]; w.lmx[a] = rt.mx[a]; w.rmn[a] = 3.0e38f; w.rmx[a] = 3.0e38f; }
        w.lref = rt.left_first | (rt.count << 30);
        w.rref = w.lref;
        out.wide.push_back(w);
        return out;
    }
    out.wide.resize((size_t)nw);
    for (size_t i = 0; i < out.nodes.size(); i++) {
        const Node& nd = out.nodes[i];
        if (nd.count != 0) continue;
        WideNode& w = out.wide[(size_t)wid[i]];
        const Node& L = out.nodes[nd.left_first];
        const Node& R = out.nodes[nd.left_first + 1];
        for (int a = 0; a < 3; a++) { w.lmn[a] = L.mn[a]; w.lmx[a] = L.mx[a]; w.rmn[a] = R.mn[a]; w.rmx[a] = R.mx[a]; }
        w.lref = (L.count ? L.left_first : (uint32_t)wid[nd.left_first]) | (L.count << 30);
        w.rref = (R.count ? R.left_first : (uint32_t)wid[nd.left_first + 1]) | (R.count << 30);
        w.pad0 = w.pad1 = 0;
    }
    return out;
}

}  // namespace rtwbvh
