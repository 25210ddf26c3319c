// rtw_bvh.h — host tree builder over world-space primitive bounds: a binned-SAH BVH2, collapsed into the 4-wide tree with
// 8-bit child boxes that the GPU walks (Q4Node), and the leaf records in tree order (LeafRec).
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
#include <cstring>
#include <vector>

#include "../../include/rtw.h"

#ifndef RTW_LEAF_MAX
#define RTW_LEAF_MAX 1  // primitives per leaf: 1 or 2 (count bits 3 mark "nothing": kBvhDone); measured: 1 is 5-7 % faster on scenes 1, 2, 4
#endif

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

// What the GPU walks: a 4-wide tree. One 64-byte record per inner node holds the boxes of up to four children, quantised
// to 8 bits per plane on a grid of the node's own (origin p = the node's box minimum, one power-of-two step per axis, stored as a float), so a
// traversal step is one burst of four 16-byte loads for four box tests, the tree of a few thousand primitives fits the
// LDS, and a walk has half the dependent steps of the binary tree. Quantisation only ever grows a box (floor / ceil,
// checked in double), and boxes only cull: exact hits are decided by the primitive tests, so the image does not depend
// on the tree. A child reference is idx << 2 | count: count 1 or 2 = a leaf whose idx is its first LeafRec, count 0 = an
// inner node; kQ4Empty marks an unused slot.
constexpr uint32_t kQ4Empty = 0xffffffffu;
struct Q4Node {
    float p[3];
    float sx;           // grid steps: powers of two
    uint32_t lo[3];     // byte c of lo[a]: child c's lower plane on axis a, in grid steps from p[a]
    uint32_t hi[3];     // upper planes
    float sy, sz;
    uint32_t ref[4];
};
static_assert(sizeof(Q4Node) == 64, "Q4Node layout");

// A leaf entry in tree order: what the intersection programs of spheres and rectangles read (p[0..4] of rtw_prim), so a
// leaf visit is one 32-byte load instead of index -> primitive record. A moving sphere takes two consecutive slots:
// centre 0 and radius, t0 (p[4]) and t1 (aux) in the first, centre 1 in the second - its intersection program
// (geometry/movingSphere.cu:33-39 under the motion transform of ioMovingSphere.h:161-203) then needs no other load either.
// Primitives under an instance transform keep their slot(s) and fetch the matrix (and, for other kinds, the primitive).
struct LeafRec {
    float p[5];
    int32_t prim;
    uint32_t type_xform;  // rtw_prim_type | xform << 8
    uint32_t aux;         // moving sphere: the bits of t1
};
static_assert(sizeof(LeafRec) == 32, "LeafRec layout");

struct Bvh {
    std::vector<Node> nodes;
    std::vector<Q4Node> q4;           // the 4-wide tree, breadth-first; q4[0] is the root
    std::vector<LeafRec> leaves;      // leaf slots in tree order (+ one slot of padding at the end)
    std::vector<uint32_t> slot_of;    // leaf entry (index into prim_order) -> its first slot
    uint32_t n_slots = 0;             // slots in use
    std::vector<int32_t> prim_order;  // leaf entries -> primitive index
    int max_depth = 0;                // of the BVH2
    int stack_need = 0;               // most entries a walk of the 4-wide tree can have on its stack (+ 1 of slack)
    int max_exp = -100;               // largest grid step of a node is 2^max_exp
};

namespace detail {
struct Item { Box b; float c[3]; int32_t prim; };

inline void build(std::vector<Item>& items, int lo, int hi, int node_idx, int depth, Bvh& out) {
    out.max_depth = std::max(out.max_depth, depth);
    Box nb, cb;
    for (int i = lo; i < hi; i++) { nb.add(items[i].b); cb.add(items[i].c); }
    for (int i = 0; i < 3; i++) { out.nodes[node_idx].mn[i] = nb.mn[i]; out.nodes[node_idx].mx[i] = nb.mx[i]; }
    int n = hi - lo;
    if (n <= RTW_LEAF_MAX) {
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

inline float node_area(const Node& n) {
    float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

// BVH2 -> 4-wide: a node adopts its grandchildren, the child of the largest area first, until it has four children or
// only leaves. Nodes are numbered breadth-first so that a prefix of the array is the top of the tree.
inline void collapse(Bvh& out) {
    const std::vector<Node>& nodes = out.nodes;
    std::vector<std::vector<uint32_t>> kids;  // per 4-wide node: BVH2 node indices of its children
    std::vector<uint32_t> src;                // per 4-wide node: the BVH2 node it stands for
    std::vector<int32_t> q4_of(nodes.size(), -1);
    std::vector<uint32_t> queue;
    if (nodes[0].count == 0) queue.push_back(0u);
    for (size_t q = 0; q < queue.size(); q++) {
        const uint32_t ni = queue[q];
        q4_of[ni] = (int32_t)q;
        std::vector<uint32_t> ch{nodes[ni].left_first, nodes[ni].left_first + 1};
        while (ch.size() < 4) {
            int best = -1; float ba = -1.f;
            for (size_t k = 0; k < ch.size(); k++)
                if (nodes[ch[k]].count == 0 && node_area(nodes[ch[k]]) > ba) { ba = node_area(nodes[ch[k]]); best = (int)k; }
            if (best < 0) break;
            const uint32_t c = ch[(size_t)best];
            ch[(size_t)best] = nodes[c].left_first;
            ch.push_back(nodes[c].left_first + 1);
        }
        for (uint32_t c : ch) if (nodes[c].count == 0) queue.push_back(c);
        kids.push_back(ch); src.push_back(ni);
    }
    if (queue.empty()) {  // the root is a leaf (one or two surfaces): one 4-wide node with that single child
        kids.push_back(std::vector<uint32_t>{0u}); src.push_back(0u);
    }
    out.q4.assign(kids.size(), Q4Node{});
    std::vector<int> need(kids.size(), 0);
    for (size_t qi = kids.size(); qi-- > 0;) {
        const Node& nb = nodes[src[qi]];
        Q4Node& w = out.q4[qi];
        double step[3];
        for (int a = 0; a < 3; a++) {
            w.p[a] = nb.mn[a];
            const double ext = (double)nb.mx[a] - (double)nb.mn[a];
            int e = -100;  // 2^e * 255 >= extent
            if (ext > 0.0) { e = (int)std::ceil(std::log2(ext / 255.0)); if (e < -100) e = -100; }
            for (;;) {  // the children's planes must fit 0..255 steps, rounding outwards
                bool ok = true;
                const double s2 = std::ldexp(1.0, e);
                for (uint32_t c : kids[qi]) if (std::ceil(((double)nodes[c].mx[a] - (double)w.p[a]) / s2) > 255.0) ok = false;
                if (ok) break;
                e++;
            }
            step[a] = std::ldexp(1.0, e);
            out.max_exp = std::max(out.max_exp, e);
        }
        w.sx = (float)step[0]; w.sy = (float)step[1]; w.sz = (float)step[2];
        for (int a = 0; a < 3; a++) { w.lo[a] = 0xffffffffu; w.hi[a] = 0u; }  // unused slots: an inverted box
        int nd = 0;
        for (size_t k = 0; k < 4; k++) {
            if (k >= kids[qi].size()) { w.ref[k] = kQ4Empty; continue; }
            const Node& ch = nodes[kids[qi][k]];
            for (int a = 0; a < 3; a++) {
                double ql = std::floor(((double)ch.mn[a] - (double)w.p[a]) / step[a]);
                double qh = std::ceil(((double)ch.mx[a] - (double)w.p[a]) / step[a]);
                if (ql < 0.0) ql = 0.0;
                if (qh > 255.0) qh = 255.0;
                while (ql > 0.0 && (double)w.p[a] + ql * step[a] > (double)ch.mn[a]) ql -= 1.0;
                while (qh < 255.0 && (double)w.p[a] + qh * step[a] < (double)ch.mx[a]) qh += 1.0;
                w.lo[a] = (w.lo[a] & ~(0xffu << (8 * k))) | ((uint32_t)ql << (8 * k));
                w.hi[a] = (w.hi[a] & ~(0xffu << (8 * k))) | ((uint32_t)qh << (8 * k));
            }
            if (ch.count) w.ref[k] = (out.slot_of[ch.left_first] << 2) | ch.count;
            else { w.ref[k] = (uint32_t)q4_of[kids[qi][k]] << 2; nd = std::max(nd, need[(size_t)q4_of[kids[qi][k]]]); }
        }
        need[qi] = (int)kids[qi].size() - 1 + nd;
    }
    out.stack_need = need[0] + 1;
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
    for (size_t i = 0; i < items.size(); i++) {
        const rtw_prim& pr = prims[items[i].prim];
        LeafRec lr{};
        for (int k = 0; k < 5; k++) lr.p[k] = pr.p[k];
        lr.prim = items[i].prim;
        lr.type_xform = (uint32_t)pr.type | ((uint32_t)pr.xform << 8);
        out.slot_of.push_back((uint32_t)out.leaves.size());
        if (pr.type == RTW_PRIM_MOVING_SPHERE) {  // p[0..3] centre 0, radius; p[4..6] centre 1; p[7] t0, p[8] t1 (rtw.h)
            lr.p[4] = pr.p[7];
            memcpy(&lr.aux, &pr.p[8], sizeof(uint32_t));
            out.leaves.push_back(lr);
            LeafRec l2{};
            for (int k = 0; k < 3; k++) l2.p[k] = pr.p[4 + k];
            l2.prim = items[i].prim;
            l2.type_xform = lr.type_xform;
            out.leaves.push_back(l2);
        } else {
            out.leaves.push_back(lr);
        }
    }
    out.n_slots = (uint32_t)out.leaves.size();
    out.leaves.push_back(LeafRec{});  // a walk reads the slot after a record's first before it knows the kind
    detail::collapse(out);
    return out;
}

}  // namespace rtwbvh
