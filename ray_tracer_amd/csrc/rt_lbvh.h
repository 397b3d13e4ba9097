// rt_lbvh.h -- per-element steps of the Morton-code LBVH build (Karras 2012 radix tree over
// unique 64-bit keys, then a bottom-up fit).  The reference has no acceleration structure beyond
// the per-mesh Box gate: Mesh::intersect scans every fragment (main.cpp:534-539).  The tree only
// decides WHICH triangles are tested; the triangle test itself and the tie-break order are those
// of the reference, so closest-hit results are identical to the linear scan.
//
// Each function handles one element (triangle / node) and is called by the HIP kernels in
// rt_device.hip, one thread per element.  Host-compilable for the CPU logic tests.
#pragma once

#include "rt_core.h"

namespace rtc {

// spreads the low 10 bits of v to every third bit
RT_HD uint32_t expand_bits10(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
RT_HD uint32_t morton30(float x, float y, float z) {  // x, y, z in [0, 1]
    float fx = fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    float fy = fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    float fz = fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return expand_bits10((uint32_t)fx) * 4u + expand_bits10((uint32_t)fy) * 2u + expand_bits10((uint32_t)fz);
}

// Sort key of triangle i of a mesh: Morton code of the centroid inside the mesh bounds in the high
// word, the triangle index in the low word (keys are unique, so the radix tree is well defined and
// the build is deterministic).  Non-finite coordinates sort last.
RT_HD uint64_t tri_sort_key(const float *p, const float *lo, const float *hi, uint32_t i) {
    float c[3];
    for (int a = 0; a < 3; a++) {
        float ctr = (p[a] + p[3 + a] + p[6 + a]) * (1.0f / 3.0f);
        float ext = hi[a] - lo[a];
        float u = ext > 0 ? (ctr - lo[a]) / ext : 0.5f;
        c[a] = (u == u) ? u : 1.0f;
    }
    return ((uint64_t)morton30(c[0], c[1], c[2]) << 32) | (uint64_t)i;
}

// Triangle record used by the intersection test: p1 p2 p3 and sn = cross(p2-p1, p3-p1)
// (main.cpp:377) as three 16-byte words.
RT_HD void make_tri_test(const float *p, f4 *out) {
    v3 p1 = ld3(p), p2 = ld3(p + 3), p3 = ld3(p + 6);
    v3 sn = cross(p2 - p1, p3 - p1);
    out[0].x = p1.x; out[0].y = p1.y; out[0].z = p1.z; out[0].w = p2.x;
    out[1].x = p2.y; out[1].y = p2.z; out[1].z = p3.x; out[1].w = p3.y;
    out[2].x = p3.z; out[2].y = sn.x; out[2].z = sn.y; out[2].w = sn.z;
}

RT_HD int clz64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __clzll((long long)x);
#else
    return x ? __builtin_clzll(x) : 64;
#endif
}
// length of the common prefix of keys i and j, -1 outside the array
RT_HD int lbvh_delta(const uint64_t *keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    return clz64(keys[i] ^ keys[j]);
}

// Internal node i (0 <= i < n-1) of the radix tree over n sorted keys: its two children, encoded
// like BvhNode::c0/c1 (>= 0 internal node, < 0 leaf ~position).
RT_HD void lbvh_node_children(const uint64_t *keys, int n, int i, int32_t &c0, int32_t &c1) {
    int d = (lbvh_delta(keys, n, i, i + 1) - lbvh_delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    int dmin = lbvh_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lbvh_delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (lbvh_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    int j = i + l * d;
    int dnode = lbvh_delta(keys, n, i, j);
    int s = 0, t = l;
    do {
        t = (t + 1) >> 1;
        if (lbvh_delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    int gamma = i + s * d + (d < 0 ? d : 0);
    int first = i < j ? i : j, last = i < j ? j : i;
    c0 = (first == gamma) ? ~gamma : gamma;
    c1 = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
}

// Box of leaf k (a triangle, or one piece of a split triangle: six floats lo, hi), grown by `pad` on
// every side so that the slab test never rejects a ray the exact triangle test accepts.
RT_HD void leaf_bounds(const float *leaf_box, int k, float pad, float *lo, float *hi) {
    for (int a = 0; a < 3; a++) { lo[a] = leaf_box[6 * k + a] - pad; hi[a] = leaf_box[6 * k + 3 + a] + pad; }
}
// unpadded box of the triangle (or piece) with vertices p[0..8]
RT_HD void piece_box(const float *p, float *box) {
    for (int a = 0; a < 3; a++) {
        box[a] = fminf(p[a], fminf(p[3 + a], p[6 + a]));
        box[3 + a] = fmaxf(p[a], fmaxf(p[3 + a], p[6 + a]));
    }
}

// ---- binned surface-area splits
// The tree over the Morton-sorted leaves can also be built top-down: a node's leaves are binned by the
// centre of their boxes (kSahBins bins on each axis), the split with the least surface-area cost
// area(left) * count(left) + area(right) * count(right) over all 3 * (kSahBins - 1) bin boundaries wins
// (ties: the first in axis-major order), and the leaves are partitioned in place, order preserved.
// Measured against the radix tree: a fifth fewer node visits (teapot and shells).  The id of an internal
// node is the position of the last leaf of its left part in the final leaf order: every gap between two
// neighbouring leaves belongs to exactly one node, so ids are unique and below n - 1.
// Nodes deeper than kSahDepthLimit, and nodes whose leaves all share one centre, split at the median, which
// bounds the height by kSahDepthLimit + log2(n).
constexpr int kSahBins = 16;
constexpr int kSahDepthLimit = 36;
RT_HD float sah_centre(const float *leaf_box, int q, int a) { return 0.5f * (leaf_box[6 * q + a] + leaf_box[6 * q + 3 + a]); }
RT_HD int sah_bin(float c, float lo, float ext) {  // ext > 0; a NaN centre goes to bin 0
    float u = (c - lo) / ext * (float)kSahBins;
    u = fminf(fmaxf(u, 0.0f), (float)(kSahBins - 1));
    return (int)u;
}
RT_HD float sah_area(const float *b) {
    float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx * dy + dy * dz + dz * dx;
}
RT_HD void sah_grow(float *b, const float *x) {
    for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], x[a]); b[3 + a] = fmaxf(b[3 + a], x[3 + a]); }
}
// Cost of the split after bin b of one axis (bins: kSahBins boxes of 6 floats, counts: leaves per bin);
// INFINITY when a side would be empty or the cost is not a number.  n_left: leaves on the left.
RT_HD float sah_split_cost(const float *bins, const int32_t *counts, int b, int32_t &n_left) {
    float l[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    float r[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int32_t nl = 0, nr = 0;
    for (int k = 0; k < kSahBins; k++) {
        if (counts[k] == 0) continue;
        if (k <= b) { sah_grow(l, bins + 6 * k); nl += counts[k]; }
        else { sah_grow(r, bins + 6 * k); nr += counts[k]; }
    }
    n_left = nl;
    if (nl == 0 || nr == 0) return INFINITY;
    float c = sah_area(l) * (float)nl + sah_area(r) * (float)nr + 0.0f;  // (+ 0: no negative zero, costs compare as bit patterns too)
    return (c == c) ? c : INFINITY;
}

// Internal node i from its children: the record with both children's boxes, the node's own bounds and height.
// Each child is a leaf or a node whose bounds and height are already in memory.
RT_HD void lbvh_make_node(int i, const int32_t *c0s, const int32_t *c1s, const float *leaf_box, float pad,
                          float *node_lo, float *node_hi, int32_t *height, BvhNode *nodes) {
    int32_t c0 = c0s[i], c1 = c1s[i];
    BvhNode nd;
    int h0 = 0, h1 = 0;
    if (c0 < 0) leaf_bounds(leaf_box, ~c0, pad, nd.lo0, nd.hi0);
    else { for (int a = 0; a < 3; a++) { nd.lo0[a] = node_lo[3 * c0 + a]; nd.hi0[a] = node_hi[3 * c0 + a]; } h0 = height[c0]; }
    if (c1 < 0) leaf_bounds(leaf_box, ~c1, pad, nd.lo1, nd.hi1);
    else { for (int a = 0; a < 3; a++) { nd.lo1[a] = node_lo[3 * c1 + a]; nd.hi1[a] = node_hi[3 * c1 + a]; } h1 = height[c1]; }
    nd.c0 = c0; nd.c1 = c1; nd.pad0 = 0; nd.pad1 = 0;
    for (int a = 0; a < 3; a++) {
        node_lo[3 * i + a] = fminf(nd.lo0[a], nd.lo1[a]);
        node_hi[3 * i + a] = fmaxf(nd.hi0[a], nd.hi1[a]);
    }
    height[i] = 1 + (h0 > h1 ? h0 : h1);
    nodes[i] = nd;
}

// One fitting pass over internal node i.  done_pass[i] is 0 until the node is completed and then
// holds the number of the pass that completed it.  A node may be completed in pass `pass` once each
// child is a leaf or was completed in an EARLIER pass (so its bounds are already in memory).
// Returns 0: was already done, 1: completed now, 2: still waiting.
RT_HD int lbvh_fit_node(int i, int pass, const int32_t *c0s, const int32_t *c1s, const float *leaf_box, float pad,
                        int32_t *done_pass, float *node_lo, float *node_hi, int32_t *height, BvhNode *nodes) {
    if (done_pass[i]) return 0;
    int32_t c0 = c0s[i], c1 = c1s[i];
    int d0 = c0 < 0 ? 0 : done_pass[c0], d1 = c1 < 0 ? 0 : done_pass[c1];
    bool r0 = c0 < 0 || (d0 != 0 && d0 < pass), r1 = c1 < 0 || (d1 != 0 && d1 < pass);
    if (!(r0 && r1)) return 2;
    lbvh_make_node(i, c0s, c1s, leaf_box, pad, node_lo, node_hi, height, nodes);
    done_pass[i] = pass;
    return 1;
}

// The variant of a plain node for ray-direction octant o: near faces in the lo slots, far faces in
// the hi slots (bit a of o set: direction component a negative, so the near face is the hi bound),
// and the child that lies in front along the octant's diagonal (by the projection of the box centres; a tie or
// a centre that is not a number keeps the order) in slot 0: the walk visits slot 0 first without comparing.
RT_HD BvhNode lbvh_octant_variant(const BvhNode &n, int o) {
    float k0 = 0, k1 = 0;
    for (int a = 0; a < 3; a++) {
        float c0 = n.lo0[a] + n.hi0[a], c1 = n.lo1[a] + n.hi1[a];
        k0 += ((o >> a) & 1) ? -c0 : c0;
        k1 += ((o >> a) & 1) ? -c1 : c1;
    }
    const bool swap = k1 < k0;
    BvhNode v = n;
    for (int a = 0; a < 3; a++) {
        const bool neg = ((o >> a) & 1) != 0;
        const float lo0 = neg ? n.hi0[a] : n.lo0[a], hi0 = neg ? n.lo0[a] : n.hi0[a];
        const float lo1 = neg ? n.hi1[a] : n.lo1[a], hi1 = neg ? n.lo1[a] : n.hi1[a];
        v.lo0[a] = swap ? lo1 : lo0; v.hi0[a] = swap ? hi1 : hi0;
        v.lo1[a] = swap ? lo0 : lo1; v.hi1[a] = swap ? hi0 : hi1;
    }
    v.c0 = swap ? n.c1 : n.c0;
    v.c1 = swap ? n.c0 : n.c1;
    return v;
}

}  // namespace rtc
