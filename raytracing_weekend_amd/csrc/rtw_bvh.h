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

#ifndef RTW_SAH_BINS
#define RTW_SAH_BINS -1  // -1: build with 16 bins, 64 bins and the full sweep, keep the cheapest 4-wide tree; 0: sweep; n: n bins (<= 64)
#endif
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

// The same node for the wave-coherent walk of the camera rays (rtw_device.h traverse_wave): every lane of a wave stands at the
// same node, so the record arrives through the scalar cache as instruction operands and can afford plain fp32 boxes - the
// children's own bounds, no grid, no byte unpacking: 6 fused multiply-adds per child and lane instead of 6 conversions + 6.
struct WNode {
    float box[4][6];   // child c: lo.x lo.y lo.z hi.x hi.y hi.z (unused slots: zeros, ref = kQ4Empty)
    uint32_t ref[4];
    uint32_t pad[4];
};
static_assert(sizeof(WNode) == 128, "WNode layout");

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
    std::vector<WNode> wq4;           // the same nodes with fp32 child boxes (camera rays: wave-coherent walk)
    std::vector<LeafRec> leaves;      // leaf slots in tree order (+ one slot of padding at the end)
    std::vector<uint32_t> slot_of;    // leaf entry (index into prim_order) -> its first slot
    uint32_t n_slots = 0;             // slots in use
    std::vector<int32_t> prim_order;  // leaf entries -> primitive index
    int max_depth = 0;                // of the BVH2
    int stack_need = 0;               // most entries a walk of the 4-wide tree can have on its stack (+ 1 of slack)
    int max_exp = -100;               // largest grid step of a node is 2^max_exp
    double cost = 0.0;                // steps per ray of the sample walks of walk_cost(): node visits + 1.5 x leaf tests
    int bins = 0;                     // how the BVH2 was split: SAH bins, 0 = full sweep
    int collapse_kind = 0;            // how it was collapsed (build_bvh_with)
};

namespace detail {
struct Item { Box b; float c[3]; int32_t prim; };

inline void build(std::vector<Item>& items, int lo, int hi, int node_idx, int depth, Bvh& out, int bins) {
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
    int mid = (lo + hi) / 2;
    if (bins > 0) {
    // binned SAH, best split over all three centroid axes
    int axis = 0;
    float ext = -1.f;
    for (int a = 0; a < 3; a++) { float e = cb.mx[a] - cb.mn[a]; if (e > ext) { ext = e; axis = a; } }
    if (ext > 0.f) {
        constexpr int kMaxBins = 64;
        const int NB = std::min(bins, kMaxBins);
        float best = FLT_MAX; int best_split = -1, best_axis = axis; float best_k = 0.f;
        for (int ax = 0; ax < 3; ax++) {
            const float e = cb.mx[ax] - cb.mn[ax];
            if (!(e > 0.f)) continue;
            Box bb[kMaxBins]; int bc[kMaxBins] = {0};
            const float k = NB * (1.f - 1e-6f) / e;
            for (int i = lo; i < hi; i++) {
                int bi = std::min(NB - 1, std::max(0, static_cast<int>((items[i].c[ax] - cb.mn[ax]) * k)));
                bb[bi].add(items[i].b); bc[bi]++;
            }
            Box accl[kMaxBins]; int cntl[kMaxBins];
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
    } else {
    // full-sweep SAH: per axis the primitives in centroid order, every split position priced by
    // area(left) * count(left) + area(right) * count(right); the cheapest of the three axes is taken
    // (a few thousand primitives: the exact sweep costs nothing next to a render)
    {
        float best = FLT_MAX; int best_axis = -1, best_pos = -1;
        std::vector<float> right_area((size_t)n);
        for (int ax = 0; ax < 3; ax++) {
            std::sort(items.begin() + lo, items.begin() + hi, [&](const Item& x, const Item& y) { return x.c[ax] < y.c[ax] || (x.c[ax] == y.c[ax] && x.prim < y.prim); });
            Box r;
            for (int i = n - 1; i >= 1; i--) { r.add(items[lo + i].b); right_area[(size_t)i] = r.area(); }
            Box l;
            for (int i = 1; i < n; i++) {
                l.add(items[lo + i - 1].b);
                const float cost = l.area() * (float)i + right_area[(size_t)i] * (float)(n - i);
                if (cost < best) { best = cost; best_axis = ax; best_pos = i; }
            }
        }
        if (best_axis >= 0) {
            if (best_axis != 2)  // (the items are in z order now)
                std::sort(items.begin() + lo, items.begin() + hi,
                          [&](const Item& x, const Item& y) { return x.c[best_axis] < y.c[best_axis] || (x.c[best_axis] == y.c[best_axis] && x.prim < y.prim); });
            mid = lo + best_pos;
        }
    }
    }
    int left = static_cast<int>(out.nodes.size());
    out.nodes.push_back(Node{});
    out.nodes.push_back(Node{});
    out.nodes[node_idx].left_first = static_cast<uint32_t>(left);
    out.nodes[node_idx].count = 0;
    build(items, lo, mid, left, depth + 1, out, bins);
    build(items, mid, hi, left + 1, depth + 1, out, bins);
}

inline float node_area(const Node& n) {
    float dx = n.mx[0] - n.mn[0], dy = n.mx[1] - n.mn[1], dz = n.mx[2] - n.mn[2];
    return 2.f * (dx * dy + dy * dz + dz * dx);
}

// BVH2 -> 4-wide: a node adopts its grandchildren, the child of the largest area first, until it has four children or
// only leaves. Nodes are numbered breadth-first so that a prefix of the array is the top of the tree.
// optimal = false: greedy, as above. optimal = true: the collapse that minimises the summed surface area of the 4-wide
// nodes (each area capped at area_cap), by dynamic programming over "this subtree may take i of its parent's slots":
// cost[n][i] = min(area(n) + best split of 4 slots between n's children,   -- n stays a node (one slot)
//                  min_j cost[left][j] + cost[right][i - j])               -- n dissolves into its parent
inline void collapse(Bvh& out, bool optimal = false, float area_cap = FLT_MAX) {
    const std::vector<Node>& nodes = out.nodes;
    std::vector<std::vector<uint32_t>> kids;  // per 4-wide node: BVH2 node indices of its children
    std::vector<uint32_t> src;                // per 4-wide node: the BVH2 node it stands for
    std::vector<int32_t> q4_of(nodes.size(), -1);
    std::vector<uint32_t> queue;
    if (nodes[0].count == 0) queue.push_back(0u);
    struct Dp { double c[5]; uint8_t split[5]; };  // split[i]: 0 = stays a node, j = left child takes j of the i slots
    std::vector<Dp> dp;
    if (optimal) {
        dp.assign(nodes.size(), Dp{});
        // children have larger indices than their parent (build() appends them), so a reverse scan is a post-order
        for (size_t n = nodes.size(); n-- > 0;) {
            if (nodes[n].count != 0) continue;  // a leaf: one slot, no node cost
            const uint32_t l = nodes[n].left_first, r = l + 1;
            auto cost_of = [&](uint32_t x, int i) { return nodes[x].count != 0 ? 0.0 : dp[x].c[i]; };
            double inner = 1e300;
            for (int j = 1; j <= 3; j++) inner = std::min(inner, cost_of(l, j) + cost_of(r, 4 - j));
            const double as_node = (double)std::min(node_area(nodes[n]), area_cap) + inner;
            dp[n].c[1] = as_node; dp[n].split[1] = 0;
            for (int i = 2; i <= 4; i++) {
                dp[n].c[i] = as_node; dp[n].split[i] = 0;
                for (int j = 1; j < i; j++) {
                    const double c = cost_of(l, j) + cost_of(r, i - j);
                    if (c < dp[n].c[i]) { dp[n].c[i] = c; dp[n].split[i] = (uint8_t)j; }
                }
            }
        }
    }
    // the children of wide node ni under the optimal collapse: its two BVH2 children given 4 slots
    auto forest = [&](auto&& self, uint32_t x, int i, std::vector<uint32_t>& ch) -> void {
        if (nodes[x].count != 0 || dp[x].split[i] == 0) { ch.push_back(x); return; }
        const int j = dp[x].split[i];
        self(self, nodes[x].left_first, j, ch);
        self(self, nodes[x].left_first + 1, i - j, ch);
    };
    for (size_t q = 0; q < queue.size(); q++) {
        const uint32_t ni = queue[q];
        q4_of[ni] = (int32_t)q;
        std::vector<uint32_t> ch;
        if (optimal) {
            const uint32_t l = nodes[ni].left_first, r = l + 1;
            auto cost_of = [&](uint32_t x, int i) { return nodes[x].count != 0 ? 0.0 : dp[x].c[i]; };
            int bj = 1; double bc = 1e300;
            for (int j = 1; j <= 3; j++) { const double c = cost_of(l, j) + cost_of(r, 4 - j); if (c < bc) { bc = c; bj = j; } }
            forest(forest, l, bj, ch);
            forest(forest, r, 4 - bj, ch);
        } else {
            ch = {nodes[ni].left_first, nodes[ni].left_first + 1};
            while (ch.size() < 4) {
                int best = -1; float ba = -1.f;
                for (size_t k = 0; k < ch.size(); k++)
                    if (nodes[ch[k]].count == 0 && node_area(nodes[ch[k]]) > ba) { ba = node_area(nodes[ch[k]]); best = (int)k; }
                if (best < 0) break;
                const uint32_t c = ch[(size_t)best];
                ch[(size_t)best] = nodes[c].left_first;
                ch.push_back(nodes[c].left_first + 1);
            }
        }
        for (uint32_t c : ch) if (nodes[c].count == 0) queue.push_back(c);
        kids.push_back(ch); src.push_back(ni);
    }
    if (queue.empty()) {  // the root is a leaf (one or two surfaces): one 4-wide node with that single child
        kids.push_back(std::vector<uint32_t>{0u}); src.push_back(0u);
    }
    out.q4.assign(kids.size(), Q4Node{});
    out.wq4.assign(kids.size(), WNode{});
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
            if (k >= kids[qi].size()) { w.ref[k] = kQ4Empty; out.wq4[qi].ref[k] = kQ4Empty; continue; }
            const Node& ch = nodes[kids[qi][k]];
            for (int a = 0; a < 3; a++) { out.wq4[qi].box[k][a] = ch.mn[a]; out.wq4[qi].box[k][3 + a] = ch.mx[a]; }
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
            out.wq4[qi].ref[k] = w.ref[k];
        }
        need[qi] = (int)kids[qi].size() - 1 + nd;
    }
    out.stack_need = need[0] + 1;
}

// What a walk of the 4-wide tree costs, measured rather than modelled: a few thousand sample rays - origins inside the
// boxes of randomly picked primitives (where bounce rays start), uniform directions, a fixed generator - walk the tree
// nearest child first like the GPU does, with a leaf's quantised box standing in for its primitive (entering it ends
// the search beyond). Cost = node visits + 1.5 x leaf tests per ray (a leaf step costs about 1.5 node steps on the GPU).
// The surface-area estimate is of no use here: one huge primitive (a ground sphere of radius 1000) makes the root box so
// large that every other node's share rounds to nothing.
inline double walk_cost(const Bvh& t, const std::vector<Item>& items, int n_rays = 4096) {
    if (t.q4.empty() || items.empty()) return 0.0;
    uint64_t st = 0x9e3779b97f4a7c15ull;
    auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (double)((st >> 11) & ((1ull << 53) - 1)) / (double)(1ull << 53); };
    double visits = 0.0, tests = 0.0;
    std::vector<uint32_t> stack;
    for (int r = 0; r < n_rays; r++) {
        const Item& it = items[(size_t)(rnd() * (double)items.size()) % items.size()];
        double o[3], d[3];
        for (int a = 0; a < 3; a++) o[a] = (double)it.b.mn[a] + rnd() * ((double)it.b.mx[a] - (double)it.b.mn[a]);
        const double z = 2.0 * rnd() - 1.0, ph = 6.283185307179586 * rnd(), rr = std::sqrt(std::max(0.0, 1.0 - z * z));
        d[0] = rr * std::cos(ph); d[1] = rr * std::sin(ph); d[2] = z;
        double best = 1e300;
        stack.clear();
        uint32_t cur = 0;
        for (;;) {
            visits += 1.0;
            const Q4Node& nd = t.q4[cur >> 2];
            const float step[3] = {nd.sx, nd.sy, nd.sz};
            double tn[4]; bool hit[4];
            for (int k = 0; k < 4; k++) {
                hit[k] = nd.ref[k] != kQ4Empty;
                double t0 = 1e-9, t1 = best;
                for (int a = 0; a < 3 && hit[k]; a++) {
                    const double lo = (double)nd.p[a] + (double)((nd.lo[a] >> (8 * k)) & 0xffu) * (double)step[a];
                    const double hi = (double)nd.p[a] + (double)((nd.hi[a] >> (8 * k)) & 0xffu) * (double)step[a];
                    if (lo > hi) { hit[k] = false; break; }
                    const double inv = 1.0 / (d[a] != 0.0 ? d[a] : 1e-300);
                    double ta = (lo - o[a]) * inv, tb = (hi - o[a]) * inv;
                    if (ta > tb) std::swap(ta, tb);
                    t0 = std::max(t0, ta); t1 = std::min(t1, tb);
                }
                hit[k] = hit[k] && t0 <= t1;
                tn[k] = t0;
            }
            // leaves first (their boxes shorten the ray), then the inner children far to near onto the stack
            int order[4] = {0, 1, 2, 3};
            std::sort(order, order + 4, [&](int x, int y) { return tn[x] < tn[y]; });
            for (int q = 0; q < 4; q++) {
                const int k = order[q];
                if (!hit[k] || (nd.ref[k] & 3u) == 0u || tn[k] > best) continue;
                tests += 1.0;
                best = std::min(best, std::max(tn[k], 1e-9));
            }
            for (int q = 3; q >= 0; q--) {
                const int k = order[q];
                if (hit[k] && (nd.ref[k] & 3u) == 0u && tn[k] <= best) stack.push_back(nd.ref[k]);
            }
            if (stack.empty()) break;
            cur = stack.back(); stack.pop_back();
        }
    }
    return (visits + 1.5 * tests) / (double)n_rays;
}
}  // namespace detail

// One tree, its BVH2 split by `bins` SAH bins (0 = full sweep).
inline Bvh build_bvh_with(const rtw_prim* prims, uint32_t n_prims, const rtw_xform* xforms, int bins, int collapse_kind = 0) {
    Bvh out;
    out.bins = bins;
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
    detail::build(items, 0, static_cast<int>(items.size()), 0, 1, out, bins);
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
    // collapse_kind 0: greedy; 1: optimal for the plain areas; 2: optimal with areas capped at 16 x the median primitive
    // box (a stand-in for "a ray that is already inside a huge box visits it whatever its size")
    float cap = FLT_MAX;
    if (collapse_kind == 2) {
        std::vector<float> ar;
        for (const detail::Item& it : items) ar.push_back(it.b.area());
        std::nth_element(ar.begin(), ar.begin() + ar.size() / 2, ar.end());
        cap = 16.0f * ar[ar.size() / 2];
    }
    out.collapse_kind = collapse_kind;
    detail::collapse(out, collapse_kind != 0, cap);
    out.cost = detail::walk_cost(out, items);
    return out;
}

// The scene's tree. Greedy SAH is not monotone in how finely the splits are searched (measured node visits per ray,
// 16 bins / 64 bins / sweep: scene 1 7.1 / 6.4 / 7.4, scene 4 7.8 / 8.1 / 6.8), so the three are built - a few
// thousand primitives, milliseconds -, each collapsed greedily and optimally, and the 4-wide tree that the sample walks
// of walk_cost() find cheapest is kept.
inline Bvh build_bvh(const rtw_prim* prims, uint32_t n_prims, const rtw_xform* xforms) {
    if (RTW_SAH_BINS >= 0) return build_bvh_with(prims, n_prims, xforms, RTW_SAH_BINS);
    Bvh best = build_bvh_with(prims, n_prims, xforms, 16, 0);
    for (int kind : {0, 1})
        for (int bins : {16, 64, 0}) {
            if (kind == 0 && bins == 16) continue;
            Bvh cand = build_bvh_with(prims, n_prims, xforms, bins, kind);
            if (!cand.q4.empty() && cand.cost < best.cost) best = std::move(cand);
        }
    return best;
}

}  // namespace rtwbvh
