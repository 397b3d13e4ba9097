// rt_device.hip -- HIP kernels (gfx950) and the C ABI of librt_amd.so (include/rt_amd.h).
//
// Kernels:
//   k_sort_keys / k_bitonic_step / k_gather_tris     leaves of a mesh's tree in Morton order
//   k_sah_level (or k_children) / k_fit / k_octant_variants   topology, bounds, per-octant copies
//   k_render<FEATURES, COUNT>        one thread per pixel of a 16x16 tile, all taps, tile-major output
//   k_render_split<FEATURES, COUNT>  the same pixels with a pixel's taps spread over a block's waves
//   k_untile                   tile-major shards -> row-major frame
//   k_trace_rays / k_intersect_object   arbitrary-ray entry points used for parity checks
//
// There is no CPU path in this library: without a usable GPU every entry point fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_core.h"
#include "rt_lbvh.h"
#include "rt_prepare.h"

using namespace rtc;

namespace {

constexpr int kBlock = 256;
constexpr int kNumCounters = 14, kNumRayCounters = 4;  // the ray counters are kept by every kernel variant
// The counters are kept in kCounterStripes copies, one 128-byte line each; a wave adds to the copy
// its block and wave number select, and the host adds the copies up.  Hundreds of thousands of waves
// adding to one address serialise in the L2's atomic unit (measured: 10 % of a 4 ms frame).
constexpr int kCounterStripes = 4096, kCounterStride = 16;

// ------------------------------------------------------------------ traversal stack in LDS

// Resident waves per SIMD a render kernel variant is compiled for (register budget 512 / waves).
// The mesh-only variant fits 96 registers and the quadric-only variant 80 without spilling (a sixth wave did not pay); the
// others need more than 100.
constexpr int render_waves(int features, bool count) {
    // mesh-only scenes: 5 waves (96 vector registers, none spilled).  6 waves (80 registers, 19 values of the pixel loop in
    // scratch) are 2 % faster, the walk waits for its scalar loads and only other waves fill that time, but write 470 MB of
    // scratch per 4K frame to HBM against 25 MB of pixels; 4 waves are 13 % slower.
    return count ? 4 : features == 0 ? 8 : features == F_MESH ? 5 : features == (F_QUADRIC | F_SECONDARY) ? 5 : 4;
}
// (the kernels with two bodies have more to keep: at 6 waves the per-tile kernel is 13 % slower than at 5)
// (measured on the shells frame: 130 registers at 3 waves 40.7 ms, 121 at 4 waves 41.8 ms: these instantiations want registers
// more than they want waves)
constexpr int split_waves(int features, bool count) { return (features & F_SECONDARY) ? 4 : render_waves(features, count); }
constexpr int kVariantQueue = 4;     // k_render_queue
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kStackWords = 3 * kStackDepth;
// The packet traversal keeps ONE stack per wave: entries of (node, 64-bit lane mask), three LDS
// words each.  All lanes write / read the same words (a broadcast access); the values are made
// scalar again with readfirstlane.
struct WaveStack {
    uint32_t *base;  // this wave's kStackWords words of LDS
    float *cold;     // this thread's column of the block's [kColdSlots][kBlock] parking area
    __device__ __forceinline__ void cold_put(int slot, float v) { cold[slot * kBlock] = v; }
    __device__ __forceinline__ float cold_get(int slot) const { return cold[slot * kBlock]; }
    __device__ __forceinline__ void put(int slot, int32_t node, uint64_t mask) {
        if (slot < kStackDepth) {  // never false: rt_build_accel refuses a tree higher than the stack
            base[3 * slot] = (uint32_t)node; base[3 * slot + 1] = (uint32_t)mask; base[3 * slot + 2] = (uint32_t)(mask >> 32);
        }
    }
    __device__ __forceinline__ void get(int slot, int32_t &node, uint64_t &mask) {
        node = __builtin_amdgcn_readfirstlane((int32_t)base[3 * slot]);
        uint32_t lo = __builtin_amdgcn_readfirstlane(base[3 * slot + 1]), hi = __builtin_amdgcn_readfirstlane(base[3 * slot + 2]);
        mask = ((uint64_t)hi << 32) | lo;
    }
};
// extra: columns after the cold_slots(features) ones: kTapSlots for a pixel's tap state (PixelTaps), kSplitSlots for the
// kernels that spread taps over waves (the pixel's corner, then the round's tap results: see split_tap_lds)
#define RT_WAVE_STACK(name, features, extra)                             \
    __shared__ uint32_t name##_lds[kWavesPerBlock * kStackWords];        \
    __shared__ float name##_cold[(cold_slots(features) + (extra)) * kBlock]; \
    WaveStack name;                                                      \
    name.base = name##_lds + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kStackWords; /* (one address per wave: scalar arithmetic) */ \
    name.cold = name##_cold + threadIdx.x

__device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ void flush_work(const Work &wk, unsigned long long *counters, bool count) {
    unsigned long long vals[kNumCounters] = {wk.rays_primary, wk.rays_shadow, wk.rays_secondary, wk.shadow_q, wk.tri, wk.box,
                                             wk.gate, wk.gate_tri, wk.quadric, wk.plane, wk.shade, wk.tex, wk.node_steps, wk.node_lanes};
    int n = count ? kNumCounters : kNumRayCounters;
    unsigned long long *mine = counters + (size_t)((blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % kCounterStripes) * kCounterStride;
    for (int i = 0; i < n; i++) {
        unsigned long long s = wave_sum(vals[i]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&mine[i], s);
    }
}

// ------------------------------------------------------------------ tree build kernels

__global__ void k_sort_keys(const float *tri_pos, int n, int n_pow2, const float *lo, const float *hi,
                            unsigned long long *keys) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    float l[3] = {lo[0], lo[1], lo[2]}, h[3] = {hi[0], hi[1], hi[2]};
    keys[i] = i < n ? tri_sort_key(tri_pos + 9 * (size_t)i, l, h, (uint32_t)i) : ~0ull;
}

// one compare-exchange step (j, k) of a bitonic sorting network over n_pow2 keys
__global__ void k_bitonic_step(unsigned long long *keys, int n_pow2, int j, int k) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pow2) return;
    int partner = i ^ j;
    if (partner > i) {
        unsigned long long a = keys[i], b = keys[partner];
        bool up = (i & k) == 0;
        if ((a > b) == up) { keys[i] = b; keys[partner] = a; }
    }
}

// The steps of the same network that stay inside a chunk of kSortChunk keys, in LDS, by one block per chunk:
// every stage k = 2 .. k_last with all its steps when `whole`, otherwise the steps j < kSortChunk of stage k_last alone
// (the steps with j >= kSortChunk of that stage were made by k_bitonic_step).
constexpr int kSortChunk = 4096, kSortThreads = 1024;
__global__ __launch_bounds__(kSortThreads) void k_bitonic_chunk(unsigned long long *keys, int k_last, int whole) {
    __shared__ unsigned long long s_keys[kSortChunk];
    const int base = blockIdx.x * kSortChunk;
    for (int t = threadIdx.x; t < kSortChunk; t += kSortThreads) s_keys[t] = keys[base + t];
    __syncthreads();
    for (int k = whole ? 2 : k_last; k <= k_last; k <<= 1) {
        for (int j = (k >> 1) < kSortChunk ? (k >> 1) : (kSortChunk >> 1); j > 0; j >>= 1) {
            for (int t = threadIdx.x; t < kSortChunk / 2; t += kSortThreads) {
                int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // the lower index of the t-th pair of this step
                unsigned long long a = s_keys[i], b = s_keys[i | j];
                bool up = ((base + i) & k) == 0;
                if ((a > b) == up) { s_keys[i] = b; s_keys[i | j] = a; }
            }
            __syncthreads();
        }
    }
    for (int t = threadIdx.x; t < kSortChunk; t += kSortThreads) keys[base + t] = s_keys[t];
}

struct SahJob { int32_t first, last, slot, depth; };  // slot: 2 * parent id + side of the link to fill in, -1 for the root

// sorted position -> triangle record, tie-break key, original triangle and leaf box of the reference
// at that position (keys == nullptr: identity order).  With `iota`, also the start of the top-down build:
// the identity permutation, the root's job and the job counts of every level (one job at level 0).
__global__ void k_gather_tris(const unsigned long long *keys, int n, int ref_offset, const float *ref_pos, const int32_t *ref_tri,
                              const float *tri_pos, const int32_t *tri_frag, const int32_t *tri_sub, f4 *tri_test,
                              uint32_t *tri_key, int32_t *tri_orig, float *leaf_box, int32_t *iota, SahJob *first_job,
                              int32_t *level_jobs, int n_levels) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (iota && k <= n_levels) level_jobs[k] = k == 0 ? 1 : 0;
    if (iota && k == 0) *first_job = SahJob{0, n - 1, -1, 0};
    if (k >= n) return;
    if (iota) iota[k] = k;
    int local = keys ? (int)(uint32_t)(keys[k] & 0xffffffffull) : k;
    int orig = ref_tri[ref_offset + local];
    make_tri_test(tri_pos + 9 * (size_t)orig, tri_test + 3 * (size_t)(ref_offset + k));
    tri_key[ref_offset + k] = 2u * (uint32_t)tri_frag[orig] + (1u - (uint32_t)tri_sub[orig]);
    tri_orig[ref_offset + k] = orig;
    piece_box(ref_pos + 9 * (size_t)(ref_offset + local), leaf_box + 6 * (size_t)(ref_offset + k));
}

__global__ void k_children(const unsigned long long *keys, int n, int32_t *c0, int32_t *c1, int32_t *node_parent, int32_t *leaf_parent) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    int32_t a, b;
    lbvh_node_children(reinterpret_cast<const uint64_t *>(keys), n, i, a, b);
    c0[i] = a;
    c1[i] = b;
    if (i == 0) node_parent[0] = -1;  // the radix tree's root
    if (a < 0) leaf_parent[~a] = i; else node_parent[a] = i;
    if (b < 0) leaf_parent[~b] = i; else node_parent[b] = i;
}

// Bottom-up fit, a thread per leaf: each climbs from its leaf's parent; the first to arrive at a node stops, the
// second (both subtrees are complete then) makes the node and climbs on.  The counter's acquire-release order
// at device scope makes the first arrival's bounds visible to the second, whichever CU it runs on.  No thread
// waits for another.  arrived[] starts at zero.
__global__ void k_fit(int n, const int32_t *c0s, const int32_t *c1s, const int32_t *node_parent, const int32_t *leaf_parent,
                      const float *leaf_box, float pad, int32_t *arrived, float *node_lo, float *node_hi, int32_t *height,
                      BvhNode *nodes, int32_t *root_height) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    int cur = leaf_parent[t];
    for (int step = 0; cur >= 0 && step < n; step++) {  // (a climb is shorter than n; the bound only guards against a broken link)
        if (__hip_atomic_fetch_add(&arrived[cur], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
        lbvh_make_node(cur, c0s, c1s, leaf_box, pad, node_lo, node_hi, height, nodes);
        int up = node_parent[cur];
        if (up < 0) *root_height = height[cur];
        cur = up;
    }
}

// ---- top-down build with binned surface-area splits (rt_lbvh.h), one level of the tree per launch.
// A block takes one node (a job: a range of the leaf permutation): centre bounds by a strided loop, a wave
// reduction and integer atomics in LDS, bins in LDS (integer atomics on order-preserving keys, so the result does
// not depend on the order of arrival), one lane of wave 0 per candidate split, a 64-bit minimum over (cost,
// candidate) for the winner, and an order-preserving partition by ballots and a scan over the block's waves into
// the other permutation buffer.  The children's ranges become jobs of the next level; a child that is one leaf
// is linked at once.  The first levels have few, long jobs and are launched with large blocks, the later ones
// with one wave per job; the code is the same.
constexpr int kSahMaxWaves = 16;
__device__ __forceinline__ int32_t float_to_ordered(float f) { int32_t i = __float_as_int(f); return i ^ ((i >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ordered_to_float(int32_t i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

// level_jobs[0] jobs at this level (the grid covers the most a level can have); the next level's are counted in
// level_jobs[1].  A job of depth d reads the permutation idx[d & 1] and writes the other.  With `finish` a block does
// not hand the children on but keeps them on a stack of its own and builds the whole subtree of its job: the launch
// after the levels a balanced tree would need, for what an unbalanced one has left (short ranges).
constexpr int kSahStack = 64;  // pending siblings, one per level at most: kSahDepthLimit + log2(n) < 64
// a leaf's box (24 bytes at a multiple of 8) as three 8-byte loads: a gather costs the texture addresser one lane per clock
// whatever its width, and the first levels, one block each, are bound by exactly that
__device__ __forceinline__ void load_leaf_box(const float *leaf_box, int q, float *box) {
    const float2 *b2 = reinterpret_cast<const float2 *>(leaf_box) + 3 * (size_t)q;
    float2 a = b2[0], b = b2[1], c = b2[2];
    box[0] = a.x; box[1] = a.y; box[2] = b.x; box[3] = b.y; box[4] = c.x; box[5] = c.y;
}
__global__ __launch_bounds__(64 * kSahMaxWaves) void k_sah_level(const SahJob *jobs, int32_t *level_jobs, const float *leaf_box, int32_t *idx_even,
                                                                int32_t *idx_odd, int32_t *c0, int32_t *c1, int32_t *node_parent,
                                                                int32_t *leaf_parent, SahJob *next_jobs, int32_t *root, int finish) {
    __shared__ int32_t s_bins[3][kSahBins][6];
    __shared__ int32_t s_cnt[3][kSahBins];
    __shared__ int32_t s_centre[6];                 // ordered keys of the centre bounds
    __shared__ int32_t s_best, s_n_left, s_only[2];
    __shared__ int32_t s_wave[2][kSahMaxWaves];     // per wave: lefts and rights of the current chunk
    __shared__ SahJob s_stack[kSahStack];
    __shared__ int32_t s_sp;
    const int n_jobs = level_jobs[0];
    const int j = blockIdx.x;
    if (j >= n_jobs) return;  // the whole block
    const int tid = threadIdx.x, nt = blockDim.x, wave = tid >> 6, lane = tid & 63, n_waves = nt >> 6;
    SahJob job = jobs[j];
    if (tid == 0) s_sp = 0;
    for (;;) {
    const int32_t *idx_in = (job.depth & 1) ? idx_odd : idx_even;
    int32_t *idx_out = (job.depth & 1) ? idx_even : idx_odd;
    const int len = job.last - job.first + 1;
    // body(q, box) for the leaves of this thread.  Four leaves at a time while the range lasts: their index and box loads
    // are in flight together (a job of the first levels is one block, i.e. sixteen waves on one compute unit, walking tens of
    // thousands of leaves through dependent index -> box gathers: the time of those levels is load latency).
    auto for_each_leaf = [&](auto &&body) {
        int k = job.first + tid;
        for (; k + 3 * nt <= job.last; k += 4 * nt) {
            int q[4];
            float box[4][6];
            for (int u = 0; u < 4; u++) q[u] = idx_in[k + u * nt];
            for (int u = 0; u < 4; u++) load_leaf_box(leaf_box, q[u], box[u]);
            for (int u = 0; u < 4; u++) body(q[u], box[u]);
        }
        for (; k <= job.last; k += nt) {
            int q = idx_in[k];
            float box[6];
            load_leaf_box(leaf_box, q, box);
            body(q, box);
        }
    };
    // centre bounds
    if (tid < 6) s_centre[tid] = float_to_ordered(tid < 3 ? INFINITY : -INFINITY);
    __syncthreads();
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for_each_leaf([&](int, const float *box) {
        for (int a = 0; a < 3; a++) { float c = 0.5f * (box[a] + box[3 + a]); clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c); }
    });
    for (int off = 32; off >= 1; off >>= 1)
        for (int a = 0; a < 3; a++) { clo[a] = fminf(clo[a], __shfl_xor(clo[a], off)); chi[a] = fmaxf(chi[a], __shfl_xor(chi[a], off)); }
    if (lane == 0)
        for (int a = 0; a < 3; a++) { atomicMin(&s_centre[a], float_to_ordered(clo[a])); atomicMax(&s_centre[3 + a], float_to_ordered(chi[a])); }
    __syncthreads();
    for (int a = 0; a < 3; a++) { clo[a] = ordered_to_float(s_centre[a]); chi[a] = ordered_to_float(s_centre[3 + a]); }
    float ext[3];
    bool usable[3];
    for (int a = 0; a < 3; a++) { ext[a] = chi[a] - clo[a]; usable[a] = ext[a] > 0 && ext[a] < INFINITY; }
    int best = -1;      // winning candidate: axis * (kSahBins - 1) + bin, -1: split at the median
    int32_t n_left = 0;
    if (job.depth < kSahDepthLimit && (usable[0] || usable[1] || usable[2])) {  // (block-uniform)
        int32_t *bins = &s_bins[0][0][0], *cnt = &s_cnt[0][0];
        for (int i = tid; i < 3 * kSahBins * 6; i += nt) bins[i] = float_to_ordered((i % 6) < 3 ? INFINITY : -INFINITY);
        for (int i = tid; i < 3 * kSahBins; i += nt) cnt[i] = 0;
        if (tid == 0) s_best = -1;
        __syncthreads();
        // The leaves arrive in Morton order, so the 64 leaves of a wave mostly fall into ONE bin of an axis (always so in
        // the first levels, whose bins hold thousands of leaves): then the wave folds its boxes with lane shuffles and one
        // lane updates the bin -- 7 LDS atomics per wave instead of 7 x 64 on one address, which is what the first levels'
        // single blocks spent their time on.  Minimum, maximum and count do not depend on the order: the same bins.
        for_each_leaf([&](int, const float *box) {
            const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
            for (int a = 0; a < 3; a++) {
                if (!usable[a]) continue;
                int b = sah_bin(0.5f * (box[a] + box[3 + a]), clo[a], ext[a]);
                const int b_first = __builtin_amdgcn_readfirstlane(b);
                if (active == ~0ull && __builtin_amdgcn_ballot_w64(b != b_first) == 0) {  // (wave-uniform) a full wave, one bin
                    int32_t key[6];
                    for (int c = 0; c < 3; c++) {
                        float lo = box[c], hi = box[3 + c];
                        key[c] = lo == lo ? float_to_ordered(lo) : 0x7fffffff;       // (a NaN bound takes no part)
                        key[3 + c] = hi == hi ? float_to_ordered(hi) : (int32_t)0x80000000;
                    }
                    for (int off = 32; off >= 1; off >>= 1)
                        for (int c = 0; c < 3; c++) {
                            int32_t o_lo = __shfl_xor(key[c], off), o_hi = __shfl_xor(key[3 + c], off);
                            key[c] = o_lo < key[c] ? o_lo : key[c];
                            key[3 + c] = o_hi > key[3 + c] ? o_hi : key[3 + c];
                        }
                    if (lane == 0) {
                        for (int c = 0; c < 3; c++) { atomicMin(&s_bins[a][b_first][c], key[c]); atomicMax(&s_bins[a][b_first][3 + c], key[3 + c]); }
                        atomicAdd(&s_cnt[a][b_first], 64);
                    }
                    continue;
                }
                for (int c = 0; c < 3; c++) {
                    float lo = box[c], hi = box[3 + c];
                    if (lo == lo) atomicMin(&s_bins[a][b][c], float_to_ordered(lo));  // fminf / fmaxf skip a NaN
                    if (hi == hi) atomicMax(&s_bins[a][b][3 + c], float_to_ordered(hi));
                }
                atomicAdd(&s_cnt[a][b], 1);
            }
        });
        __syncthreads();
        for (int i = tid; i < 3 * kSahBins * 6; i += nt) bins[i] = __float_as_int(ordered_to_float(bins[i]));
        __syncthreads();
        if (wave == 0) {
            unsigned long long key = ~0ull;
            int32_t my_left = 0;
            if (lane < 3 * (kSahBins - 1)) {
                int a = lane / (kSahBins - 1), b = lane % (kSahBins - 1);
                if (usable[a]) {
                    float cost = sah_split_cost(reinterpret_cast<const float *>(&s_bins[a][0][0]), &s_cnt[a][0], b, my_left);
                    if (cost < INFINITY) key = ((unsigned long long)(uint32_t)__float_as_int(cost) << 32) | (unsigned long long)lane;
                }
            }
            unsigned long long m = key;
            for (int off = 32; off >= 1; off >>= 1) {
                unsigned long long o = __shfl_xor(m, off);
                m = o < m ? o : m;
            }
            if (m != ~0ull && lane == (int)(m & 0xffffffffull)) { s_best = lane; s_n_left = my_left; }
        }
        __syncthreads();
        best = s_best;
        n_left = best < 0 ? 0 : s_n_left;
    }
    const int32_t total_left = best < 0 ? (len + 1) / 2 : n_left;
    const int axis = best < 0 ? 0 : best / (kSahBins - 1), bin = best < 0 ? 0 : best % (kSahBins - 1);
    const float lo_a = axis == 0 ? clo[0] : axis == 1 ? clo[1] : clo[2], ext_a = axis == 0 ? ext[0] : axis == 1 ? ext[1] : ext[2];
    // order-preserving partition into the other buffer, a chunk of one element per thread at a time
    int lbase = job.first, rbase = job.first + total_left;
    auto place = [&](int k, int q, float centre) {  // the block's threads hold the leaves k0 .. k0 + nt - 1 of the range, in order
        bool act = k <= job.last;
        bool left = act && (best < 0 ? (k - job.first) < total_left : sah_bin(centre, lo_a, ext_a) <= bin);
        bool right = act && !left;
        unsigned long long ml = __builtin_amdgcn_ballot_w64(left), mr = __builtin_amdgcn_ballot_w64(right);
        if (lane == 0) { s_wave[0][wave] = __popcll(ml); s_wave[1][wave] = __popcll(mr); }
        __syncthreads();
        int lsum = 0, rsum = 0, lmine = 0, rmine = 0;
        for (int w = 0; w < n_waves; w++) {
            if (w == wave) { lmine = lsum; rmine = rsum; }
            lsum += s_wave[0][w]; rsum += s_wave[1][w];
        }
        unsigned long long below = (1ull << lane) - 1ull;
        if (left) idx_out[lbase + lmine + __popcll(ml & below)] = q;
        if (right) idx_out[rbase + rmine + __popcll(mr & below)] = q;
        if (left && total_left == 1) s_only[0] = q;          // the one leaf of a side (one writer)
        if (right && len - total_left == 1) s_only[1] = q;
        lbase += lsum; rbase += rsum;
        __syncthreads();
    };
    {   // (the next chunk's leaf and centre are fetched while this chunk is placed)
        int q = job.first + tid <= job.last ? idx_in[job.first + tid] : 0;
        float centre = sah_centre(leaf_box, q, axis);
        for (int k0 = job.first; k0 <= job.last; k0 += nt) {
            const int k_next = k0 + nt + tid;
            const int q_next = k_next <= job.last ? idx_in[k_next] : 0;
            const float centre_next = sah_centre(leaf_box, q_next, axis);
            place(k0 + tid, q, centre);
            q = q_next; centre = centre_next;
        }
    }
    if (tid == 0) {
        int32_t *next_count = level_jobs + 1;
        const int32_t g = job.first + total_left - 1;  // this node's id
        if (job.slot < 0) { *root = g; node_parent[g] = -1; }
        else {
            if (job.slot & 1) c1[job.slot >> 1] = g; else c0[job.slot >> 1] = g;
            node_parent[g] = job.slot >> 1;
        }
        SahJob kids[2] = {{job.first, g, 2 * g, job.depth + 1}, {g + 1, job.last, 2 * g + 1, job.depth + 1}};
        if (total_left == 1) { c0[g] = ~s_only[0]; leaf_parent[s_only[0]] = g; }
        if (len - total_left == 1) { c1[g] = ~s_only[1]; leaf_parent[s_only[1]] = g; }
        for (int side = 1; side >= 0; side--) {  // (the left child ends on top of the stack)
            if ((side == 0 ? total_left : len - total_left) == 1) continue;
            if (!finish) next_jobs[atomicAdd(next_count, 1)] = kids[side];
            else if (s_sp < kSahStack) s_stack[s_sp++] = kids[side];
            else level_jobs[2] = 1;  // cannot happen: reported as an unfinished build
        }
    }
    if (!finish) return;
    __syncthreads();
    if (s_sp == 0) return;  // (block-uniform)
    job = s_stack[s_sp - 1];
    __syncthreads();
    if (tid == 0) s_sp--;
    }
}

// ------------------------------------------------------------------ render kernels

// Which tile of the frame (row-major) is local tile `local` of a rank: the tiles are dealt in groups of world_size
// consecutive tiles, one to each rank, and the deal rotates by one from group to group.  Without the rotation a
// rank's tiles are the same columns of every row whenever the row length is a multiple of the world size, and
// the ranks' shares of a frame differ by what those columns show (measured on the teapot frame at 8 ranks:
// shares between 0.406 and 0.447 ms).
__host__ __device__ inline int shard_tile(int local, int rank, int world_size) {
    return local * world_size + (rank + local) % world_size;
}

struct RenderArgs {
    int32_t width, height, mode, grid_m, max_depth, accel, rank, world_size;
    int32_t tiles_x, tiles_total;
    Camera cam;
};

// Block b renders local tile b = global tile shard_tile(b, rank, world_size).  Each wave owns an 8x8 pixel
// quadrant of the 16x16 tile so that its 64 rays are spatially coherent.
// copies of the fitted tree for the ray-direction octants 1..7 (variant 0 is what k_fit wrote)
__global__ void k_octant_variants(BvhNode *nodes, int n) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * (kOctants - 1)) return;
    int o = 1 + idx / n, i = idx % n;
    nodes[(size_t)o * n + i] = lbvh_octant_variant(nodes[i], o);
}
// then the fitted tree itself becomes the variant of octant 0 (its children put in that octant's order), in place
__global__ void k_octant_variant0(BvhNode *nodes, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    nodes[i] = lbvh_octant_variant(nodes[i], 0);
}

// One 16x16 tile by one block: each wave an 8x8 quadrant, each lane a pixel with all its taps.
template <int F, bool COUNT, bool PARKED>
__device__ __forceinline__ void render_tile(SceneRef S, const RenderArgs &a, int local, WaveStack &stk, const float *grid_off,
                                            float *rgb_tiles, uint8_t *rgb8_tiles, Work &wk) {
    int tile = shard_tile(local, a.rank, a.world_size);
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int px = (wave & 1) * 8 + (lane & 7), py = (wave >> 1) * 8 + (lane >> 3);
    int i = (tile % a.tiles_x) * RT_TILE_W + px, j = (tile / a.tiles_x) * RT_TILE_H + py;
    bool inside = tile < a.tiles_total && i < a.width && j < a.height;
    v3 color = render_pixel<F, COUNT, PARKED>(S, a.cam, a.mode, a.grid_m, a.max_depth, a.accel, i, j, inside, grid_off, stk, wk);
    if (!inside) color = V3(0, 0, 0);
    size_t p = ((size_t)local * (RT_TILE_W * RT_TILE_H) + (size_t)py * RT_TILE_W + px) * 3;
    if (rgb_tiles) { rgb_tiles[p] = color.x; rgb_tiles[p + 1] = color.y; rgb_tiles[p + 2] = color.z; }
    if (rgb8_tiles) {
        v3 t = tone_map(color);
        rgb8_tiles[p] = inside ? quantise(t.x) : 0;
        rgb8_tiles[p + 1] = inside ? quantise(t.y) : 0;
        rgb8_tiles[p + 2] = inside ? quantise(t.z) : 0;
    }
}
// The kernel arguments arrive as wide scalar loads, and a register tuple that has to make room is saved and
// restored whole (eight lane reads for one field, fifteen times in the tap loop of the split kernel).  After
// this every field is a scalar register of its own.
__device__ __forceinline__ void unbundle(RenderArgs &a) {
    asm("" : "+s"(a.width), "+s"(a.height), "+s"(a.mode), "+s"(a.grid_m), "+s"(a.max_depth), "+s"(a.accel), "+s"(a.rank),
             "+s"(a.world_size), "+s"(a.tiles_x), "+s"(a.tiles_total));
}
__device__ __forceinline__ void fill_grid_table(float *s_grid_off, const RenderArgs &a) {
    if (threadIdx.x < kMaxGrid) s_grid_off[threadIdx.x] = (int)threadIdx.x < a.grid_m ? grid_offset(threadIdx.x, a.grid_m, a.cam.s) : 0.0f;
    __syncthreads();
}

template <int F, bool COUNT>
__global__ __launch_bounds__(kBlock, render_waves(F, COUNT)) void k_render(const DevScene *scene, RenderArgs a, float *rgb_tiles, uint8_t *rgb8_tiles,
                                                   unsigned long long *counters, uint32_t *tile_times) {
    // tile_times (may be null): the block leaves its tile's time there (100 MHz ticks), for the kernels that choose per
    // tile by the time a tile took in the frame before.  Only mesh scenes have such kernels: in the kernels of the other
    // scenes the pointer, the clock and the branch on them cost 10 % of a frame (registers of the tap loop spilled).
    // The pointer and the start time wait in LDS meanwhile.
    constexpr bool kTimes = (F & F_MESH) != 0 && !COUNT;
    __shared__ unsigned long long s_t0;
    __shared__ uint32_t *s_times;
    if constexpr (kTimes) {
        if (threadIdx.x == 0) {
            s_times = tile_times;
            s_t0 = tile_times ? wall_clock64() : 0ull;
        }
    }
    SceneRef S = scene_ref(scene);
    RT_WAVE_STACK(stk, F, kTapSlots);
    __shared__ float s_grid_off[kMaxGrid];
    fill_grid_table(s_grid_off, a);
    Work wk;
    work_zero(wk);
    render_tile<F, COUNT, true>(S, a, blockIdx.x, stk, s_grid_off, rgb_tiles, rgb8_tiles, wk);
    flush_work(wk, counters, COUNT);
    if constexpr (kTimes) {
        __syncthreads();
        if (threadIdx.x == 0 && s_times != nullptr) {
            unsigned long long dt = wall_clock64() - s_t0;
            s_times[blockIdx.x] = dt > 0xfffffffeull ? 0xfffffffeu : (uint32_t)dt;
        }
    }
}

// The same frame with the taps of a pixel spread over the waves of a block.  PH = 1: block b renders the
// 8x8 quadrant b & 3 of local tile b >> 2; in rounds of kRoundTaps taps, wave w traces its quarter of the
// round's taps for all 64 pixels and leaves each tap's radiance in LDS; wave 0 then adds them to the pixel
// sums in tap order, exactly as the one-lane-per-pixel kernel does.  PH = 2: a block has half a quadrant
// (8x4 pixels), and the two halves of a wave trace different taps of the same 32 pixels, so a round's eight
// taps are traced in one step.  This bounds the time of the most expensive block to a quarter (an eighth),
// which is what limits frames whose cost is concentrated in few pixels and launches of few tiles.
constexpr int kRoundTaps = 8;
// LDS columns of the tap-splitting kernels after the shading columns: two for the pixel's corner, then the results of a
// round's taps, kRoundTaps * 3 * 64 floats (one lane-pixel each; 6 columns of kBlock floats).  A block that renders a
// whole tile one pixel per lane instead (the per-tile kernel's light tiles) keeps that pixel's tap state (kTapSlots
// columns) in the same place.
constexpr int kSplitSlots = 2 + kRoundTaps * 3 * 64 / kBlock;
static_assert(kRoundTaps * 3 * 64 % kBlock == 0 && kSplitSlots >= kTapSlots, "tap results fill whole columns");
template <int F>
__device__ __forceinline__ float *split_tap_lds(WaveStack &stk) { return stk.cold - threadIdx.x + (cold_slots(F) + 2) * kBlock; }
template <int BASE>
struct TapSlice {  // a trace_loop source: taps [k, k_end) of one pixel; the pixel's corner in columns BASE, BASE + 1 of stk
    Camera cam;
    int mode, grid_m, k, k_end, base;
    WaveStack *stk;
    bool inside;            // the lane has a pixel of the frame and taps of its own in this round
    int col, row;           // GRID: k = row * grid_m + col
    const float *grid_off;  // GRID: the block's table of grid_offset(a, grid_m, cam.s)
    float wt;
    float *slots;  // this lane's column of the LDS array [kRoundTaps][3][pixels of the block]; nullptr: sum into `color` at once
    int stride;    // pixels of the block
    v3 color;      // the pixel's running sum when the lane traces all its taps itself (slots == nullptr)
    __device__ bool more() const { return k < k_end; }
    __device__ bool next(v3 &ro, v3 &rd) {
        if (k >= k_end) return false;
        ro = V3(0, 0, 0);
        const float dx = stk->cold_get(BASE), dy = stk->cold_get(BASE + 1);
        rd = tap_dir(cam, mode, grid_m, dx, dy, k, col, row, grid_off, wt);
        k++;
        return inside;
    }
    __device__ void done(v3 r) {
        if (slots == nullptr) {  // (wave-uniform)
            color = mode == RT_MODE_REF_1SPP ? r : color + r * wt;
            return;
        }
        float *p = slots + (k - 1 - base) * 3 * stride;
        p[0] = r.x; p[stride] = r.y; p[2 * stride] = r.z;
    }
    __device__ rt_hit_record *first_hit() { return nullptr; }
};

// One work item of k_render_queue by one block, in one body for both kinds (two inlined bodies cost registers).
// part < 0: the whole local tile, each wave a quadrant, each lane a pixel with all its taps, summed in registers as the
// taps finish.  part = 0..3: that quadrant alone, the taps of its pixels spread over the four waves in rounds of
// kRoundTaps and summed in tap order through LDS (tap_lds: kRoundTaps * 3 * 64 floats), as render_tile_part does.
template <int F>
__device__ __forceinline__ void render_item(SceneRef S, const RenderArgs &a, int local, int part, WaveStack &stk, const float *grid_off,
                                            float *rgb_tiles, uint8_t *rgb8_tiles, Work &wk) {
    float *tap_lds = split_tap_lds<F>(stk);
    const bool whole = part < 0;  // (block-uniform)
    const int tile = shard_tile(local, a.rank, a.world_size);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int quad = whole ? wave : part;
    const int px = (quad & 1) * 8 + (lane & 7), py = (quad >> 1) * 8 + (lane >> 3);
    const int i = (tile % a.tiles_x) * RT_TILE_W + px, j = (tile / a.tiles_x) * RT_TILE_H + py;
    const bool inside = tile < a.tiles_total && i < a.width && j < a.height;
    const int taps = taps_of(a.mode, a.grid_m);
    const int round_taps = whole ? taps : kRoundTaps;
    stk.cold_put(cold_slots(F), pixel_dx(a.cam, i)); stk.cold_put(cold_slots(F) + 1, pixel_dy(a.cam, j));
    v3 color = V3(0, 0, 0);
    for (int base = 0; base < taps; base += round_taps) {
        const int n = taps - base < round_taps ? taps - base : round_taps;
        const int chunk = whole ? n : (n + 3) / 4, slot = whole ? 0 : wave;
        TapSlice<cold_slots(F)> src;
        src.cam = a.cam; src.mode = a.mode; src.grid_m = a.grid_m; src.stk = &stk; src.base = base; src.wt = 1.0f;
        src.k = base + (slot * chunk < n ? slot * chunk : n);
        src.k_end = base + ((slot + 1) * chunk < n ? (slot + 1) * chunk : n);
        src.col = src.k % a.grid_m; src.row = src.k / a.grid_m; src.grid_off = grid_off;
        src.slots = whole ? nullptr : tap_lds + lane;
        src.stride = 64;
        src.color = color;
        src.inside = inside;
        trace_loop<F, false>(S, a.max_depth, a.accel, stk, wk, src);
        color = src.color;
        if (!whole) {
            __syncthreads();
            if (wave == 0 && inside) {
                for (int t = 0; t < n; t++) {
                    const float *p = tap_lds + t * 3 * 64 + lane;
                    v3 r = V3(p[0], p[64], p[128]);
                    color = a.mode == RT_MODE_REF_1SPP ? r : color + r * tap_weight(a.mode, base + t);
                }
            }
            __syncthreads();
        }
    }
    if (whole || wave == 0) {
        if (a.mode != RT_MODE_REF_1SPP) color = color / (a.mode == RT_MODE_REF_AA9 ? 16.f : (float)taps);
        size_t p = ((size_t)local * (RT_TILE_W * RT_TILE_H) + (size_t)py * RT_TILE_W + px) * 3;
        if (rgb_tiles) { rgb_tiles[p] = color.x; rgb_tiles[p + 1] = color.y; rgb_tiles[p + 2] = color.z; }
        if (rgb8_tiles) {
            v3 t = tone_map(color);
            rgb8_tiles[p] = inside ? quantise(t.x) : 0;
            rgb8_tiles[p + 1] = inside ? quantise(t.y) : 0;
            rgb8_tiles[p + 2] = inside ? quantise(t.z) : 0;
        }
    }
}

// Part `part` (of 4 * PH) of local tile `local` by one block; tap_lds: kRoundTaps * 3 * (64 / PH) floats of LDS.
template <int F, bool COUNT, int PH>
__device__ __forceinline__ void render_tile_part(SceneRef S, const RenderArgs &a, int local, int part, WaveStack &stk,
                                                 const float *grid_off, float *rgb_tiles, uint8_t *rgb8_tiles, Work &wk) {
    float *tap_lds = split_tap_lds<F>(stk);
    constexpr int PX = 64 / PH;  // pixels of a block
    const int quad = part / PH, sub = part % PH;
    const int tile = shard_tile(local, a.rank, a.world_size);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pl = lane % PX, phase = lane / PX;  // this lane's pixel of the block and which of the wave's PH taps it traces
    const int q = sub * PX + pl;                  // pixel of the quadrant, row-major
    const int px = (quad & 1) * 8 + (q & 7), py = (quad >> 1) * 8 + (q >> 3);
    const int i = (tile % a.tiles_x) * RT_TILE_W + px, j = (tile / a.tiles_x) * RT_TILE_H + py;
    bool inside = tile < a.tiles_total && i < a.width && j < a.height;
    const int taps = taps_of(a.mode, a.grid_m);
    const int slot = wave * PH + phase;  // one of the block's 4 * PH tap slots
    stk.cold_put(cold_slots(F), pixel_dx(a.cam, i)); stk.cold_put(cold_slots(F) + 1, pixel_dy(a.cam, j));
    v3 color = V3(0, 0, 0);
    for (int base = 0; base < taps; base += kRoundTaps) {
        int n = taps - base < kRoundTaps ? taps - base : kRoundTaps, chunk = (n + 4 * PH - 1) / (4 * PH);
        TapSlice<cold_slots(F)> src;
        src.cam = a.cam; src.mode = a.mode; src.grid_m = a.grid_m; src.stk = &stk; src.base = base; src.wt = 1.0f;
        src.k = base + (slot * chunk < n ? slot * chunk : n);
        src.k_end = base + ((slot + 1) * chunk < n ? (slot + 1) * chunk : n);
        src.col = src.k % a.grid_m; src.row = src.k / a.grid_m; src.grid_off = grid_off;  // (one division per round, not per tap)
        src.slots = tap_lds + pl;
        src.stride = PX;
        src.color = V3(0, 0, 0);
        src.inside = inside;
        trace_loop<F, COUNT>(S, a.max_depth, a.accel, stk, wk, src);
        __syncthreads();
        if (wave == 0 && phase == 0 && inside) {
            for (int t = 0; t < n; t++) {
                const float *p = tap_lds + t * 3 * PX + pl;
                v3 r = V3(p[0], p[PX], p[2 * PX]);
                color = a.mode == RT_MODE_REF_1SPP ? r : color + r * tap_weight(a.mode, base + t);
            }
        }
        __syncthreads();
    }
    if (wave == 0 && phase == 0) {
        if (a.mode != RT_MODE_REF_1SPP) color = color / (a.mode == RT_MODE_REF_AA9 ? 16.f : (float)taps);
        size_t p = ((size_t)local * (RT_TILE_W * RT_TILE_H) + (size_t)py * RT_TILE_W + px) * 3;
        if (rgb_tiles) { rgb_tiles[p] = color.x; rgb_tiles[p + 1] = color.y; rgb_tiles[p + 2] = color.z; }
        if (rgb8_tiles) {
            v3 t = tone_map(color);
            rgb8_tiles[p] = inside ? quantise(t.x) : 0;
            rgb8_tiles[p + 1] = inside ? quantise(t.y) : 0;
            rgb8_tiles[p + 2] = inside ? quantise(t.z) : 0;
        }
    }
}

template <int F, bool COUNT, int PH>
__global__ __launch_bounds__(kBlock, split_waves(F, COUNT)) void k_render_split(const DevScene *scene, RenderArgs a, float *rgb_tiles,
                                                                                 uint8_t *rgb8_tiles, unsigned long long *counters,
                                                                                 uint32_t * /*tile_times: k_render's argument list*/) {
    SceneRef S = scene_ref(scene);
    unbundle(a);
    RT_WAVE_STACK(stk, F, kSplitSlots);
    __shared__ float s_grid_off[kMaxGrid];
    fill_grid_table(s_grid_off, a);
    Work wk;
    work_zero(wk);
    render_tile_part<F, COUNT, PH>(S, a, blockIdx.x / (4 * PH), blockIdx.x % (4 * PH), stk, s_grid_off, rgb_tiles, rgb8_tiles, wk);
    flush_work(wk, counters, COUNT);
}

// Both ways in one launch, chosen per tile: four blocks per tile; a tile that took long in the previous frame
// (cost_in, in ticks of the 100 MHz clock, above `threshold`) is rendered by all four, a quadrant each with
// the taps spread over the waves; any other tile by one of its blocks alone, the other three leave at once.
// The heavy tiles then no longer decide the length of a short launch (one rank's share of a frame), and the
// light ones keep the cheaper kernel.  One block of every tile leaves the tile's time for the next frame in
// cost_out (another array than cost_in: the four blocks of a tile must read the same value).
struct TileCosts { const uint32_t *cost_in; uint32_t *cost_out; const uint32_t *threshold; };  // threshold: written by k_tile_threshold just before
template <int F>
__global__ __launch_bounds__(kBlock, split_waves(F, false)) void k_render_hybrid(const DevScene *scene, RenderArgs a, float *rgb_tiles,
                                                                                  uint8_t *rgb8_tiles, unsigned long long *counters,
                                                                                  TileCosts tc) {
    SceneRef S = scene_ref(scene);
    const int local = blockIdx.x >> 2, quad = blockIdx.x & 3;
    const bool heavy = tc.cost_in[local] > *tc.threshold;  // the same for the tile's four blocks
    // Which of the four blocks takes a light tile is scattered by a hash of the tile: blocks are handed to the
    // XCDs and their compute units in turn, and a regular pattern of working and leaving blocks keeps most of
    // them idle (measured: 12.7 ms with always the first block, 12.1 ms with a rotation, 3.3 ms scattered).
    const int worker = (int)(((uint32_t)local * 2654435761u) >> 13) & 3;
    if (!heavy && quad != worker) return;
    const unsigned long long t0 = wall_clock64();
    unbundle(a);
    RT_WAVE_STACK(stk, F, kSplitSlots);
    __shared__ float s_grid_off[kMaxGrid];
    fill_grid_table(s_grid_off, a);
    Work wk;
    work_zero(wk);
    if (heavy) render_tile_part<F, false, 1>(S, a, local, quad, stk, s_grid_off, rgb_tiles, rgb8_tiles, wk);
    else render_tile<F, false, true>(S, a, local, stk, s_grid_off, rgb_tiles, rgb8_tiles, wk);
    flush_work(wk, counters, false);
    __syncthreads();
    if (threadIdx.x == 0 && quad == worker) {
        unsigned long long dt = (wall_clock64() - t0) * (heavy ? 4ull : 1ull);  // a quadrant's time stands for a quarter of the tile's
        tc.cost_out[local] = dt > 0xffffffffull ? 0xffffffffu : (uint32_t)dt;
    }
}

// ---- persistent blocks fed from a work queue
//
// The launch has as many blocks as the GPU keeps resident; each block takes work items from a list until the
// list is empty (one atomic per item).  An item is a whole 16x16 tile (each wave a quadrant, each lane a pixel,
// all taps: the cheapest way to render a tile) or, for a tile that took long in the previous frame, one of its
// four quadrants with the pixel's taps spread over the block's waves (a quarter as long).  k_plan_tiles builds
// the list before every frame from the tiles' times in the previous one: the quadrants of the heavy tiles first,
// longest first, then the light tiles in frame order.  With the long items handed out first and the short ones
// last, the blocks finish together: the end of a launch no longer waits for the block that happened to get a
// heavy tile late, which is what limits a short launch (one GPU's share of a frame on a node of eight).
// Which tiles count as heavy changes only who renders a pixel, never its value.
//
// Queue memory (`QueueMem`, one per context, or the library's static array for the entry point that may not
// allocate): header, two arrays of tile times used in turn, the item list.
struct QueueHeader {
    uint32_t next;       // next item to hand out
    uint32_t n_items;
    uint32_t n_heavy;    // heavy tiles in this frame's list
    uint32_t threshold;  // ticks of the 100 MHz clock above which a tile counted as heavy
    uint32_t pad[60];
};
static_assert(sizeof(QueueHeader) == 256, "QueueHeader layout");
constexpr uint32_t kItemQuadrant = 0x80000000u;  // item = tile | quadrant << 28 | kItemQuadrant
constexpr int kPlanThreads = 1024, kCostBuckets = 32;

// The threshold of k_render_hybrid, one block: 1.5 times the mean of the tiles' known times in the previous frame
// (a tile of unknown time, 0xffffffff, is above any threshold and does not enter the mean).
__global__ __launch_bounds__(kPlanThreads) void k_tile_threshold(QueueHeader *q, const uint32_t *cost_in, int n_tiles, uint32_t forced_threshold,
                                                                 int use_forced) {
    __shared__ unsigned long long s_sum[kPlanThreads / 64], s_cnt[kPlanThreads / 64];
    const int t = threadIdx.x;
    unsigned long long sum = 0, cnt = 0;
    for (int i = t; i < n_tiles; i += kPlanThreads) { uint32_t c = cost_in[i]; if (c != 0xffffffffu) { sum += c; cnt++; } }
    sum = wave_sum(sum); cnt = wave_sum(cnt);
    if ((t & 63) == 0) { s_sum[t >> 6] = sum; s_cnt[t >> 6] = cnt; }
    __syncthreads();
    if (t == 0) {
        unsigned long long total = 0, n = 0;
        for (int w = 0; w < kPlanThreads / 64; w++) { total += s_sum[w]; n += s_cnt[w]; }
        unsigned long long th = n ? total * 3ull / (2ull * n) : 0ull;
        q->threshold = use_forced ? forced_threshold : (th > 0xfffffffeull ? 0xfffffffeu : (uint32_t)th);
    }
}

// The lanes of a wave with `on` add one each to s_bucket[b] (b: the lane's bucket), one atomic per distinct bucket among
// them; returns the lane's index within its bucket.  Called by whole waves.
__device__ __forceinline__ uint32_t bucket_take(uint32_t *s_bucket, bool on, int b) {
    const int lane = threadIdx.x & 63;
    uint32_t pos = 0;
    uint64_t todo = __builtin_amdgcn_ballot_w64(on);
    while (todo != 0) {  // (wave-uniform)
        const int leader = __builtin_ctzll(todo);
        const int lb = __builtin_amdgcn_readlane(b, leader);
        const bool mine = on && b == lb;
        const uint64_t same = __builtin_amdgcn_ballot_w64(mine);
        uint32_t first = 0;
        if (lane == leader) first = atomicAdd(&s_bucket[lb], (uint32_t)__builtin_popcountll(same));
        first = (uint32_t)__builtin_amdgcn_readlane((int)first, leader);
        if (mine) pos = first + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return pos;
}

// One block.  cost_in: last frame's time per tile (0xffffffff: unknown, counts as heavy); cost_out: zeroed here,
// filled by the render kernel.  slots: blocks of the render launch.
// Every pass takes the tiles in batches of kPlanBatch: a thread fetches its kPlanPerThread times of a batch with loads
// that are all in flight together and works on them from registers (one block alone has too few waves to hide a
// memory round trip per step: with a load in every step of every pass this kernel took 72 us for a 4K frame).
constexpr int kPlanPerThread = 32, kPlanBatch = kPlanPerThread * kPlanThreads, kPlanWaves = kPlanThreads / 64;
__device__ __forceinline__ void plan_fetch(const uint32_t *__restrict__ cost_in, int first, int n_tiles, uint32_t (&c)[kPlanPerThread]) {
#pragma unroll
    for (int k = 0; k < kPlanPerThread; k++) {
        const int i = first + k * kPlanThreads + (int)threadIdx.x;
        c[k] = i < n_tiles ? cost_in[i] : 0u;  // (0: neither heavy nor, as an index out of range, light)
    }
}
__global__ __launch_bounds__(kPlanThreads) void k_plan_tiles(QueueHeader *q, const uint32_t *__restrict__ cost_in, uint32_t *__restrict__ cost_out,
                                                             uint32_t *__restrict__ items, int n_tiles, int slots, uint32_t forced_threshold,
                                                             int use_forced) {
    __shared__ unsigned long long s_sum[kPlanWaves], s_known[kPlanWaves];
    __shared__ uint32_t s_light[kPlanPerThread * kPlanWaves];  // light tiles of a batch by (step, wave), then their running sum
    __shared__ uint32_t s_bucket[kCostBuckets], s_bucket_base[kCostBuckets];
    __shared__ uint32_t s_threshold, s_heavy, s_light_before;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    uint32_t c[kPlanPerThread];
    // The threshold: a sixteenth of the launch's expected length (sum of the known times / slots), so that whatever is
    // handed out last is short against the whole, but not under the mean tile: a quadrant item costs more per pixel than
    // a whole tile, and a list of nothing but quadrants is a quarter slower than one that splits the tiles above the
    // mean (measured on one rank's share of the 4K teapot frame at 1 / 2 / 4 / 8 ranks: 1.73 / 0.87 / 0.46 / 0.26 ms
    // with this rule, 1.76 / 0.90 / 0.57 / 0.31 with the sixteenth alone).  Tiles of unknown time do not enter the sums.
    unsigned long long sum = 0, known = 0;
    for (int first = 0; first < n_tiles; first += kPlanBatch) {
        plan_fetch(cost_in, first, n_tiles, c);
#pragma unroll
        for (int k = 0; k < kPlanPerThread; k++) {
            const bool is_known = first + k * kPlanThreads + t < n_tiles && c[k] != 0xffffffffu;
            sum += is_known ? c[k] : 0u; known += is_known ? 1u : 0u;
        }
    }
    sum = wave_sum(sum); known = wave_sum(known);
    if (lane == 0) { s_sum[wave] = sum; s_known[wave] = known; }
    if (t < kCostBuckets) s_bucket[t] = 0;
    __syncthreads();
    if (t == 0) {
        unsigned long long total = 0, n_known = 0;
        for (int w = 0; w < kPlanWaves; w++) { total += s_sum[w]; n_known += s_known[w]; }
        unsigned long long th = total / (unsigned long long)(slots > 0 ? slots : 1) / 16ull;
        const unsigned long long mean = total / (n_known > 0 ? n_known : 1ull);
        if (th < mean) th = mean;
        s_threshold = use_forced ? forced_threshold : (th > 0xfffffffeull ? 0xfffffffeu : (uint32_t)th);
        s_heavy = 0; s_light_before = 0;
    }
    __syncthreads();
    const uint32_t threshold = s_threshold;
    // heavy tiles by bucket of floor(log2(time)), longest first; within a bucket in order of arrival.  (The lanes of a
    // wave add to a bucket together, bucket_take: a frame's heavy tiles fall into two or three buckets, and one atomic
    // per lane on so few addresses is a queue of sixty-four.)
    for (int first = 0; first < n_tiles; first += kPlanBatch) {
        plan_fetch(cost_in, first, n_tiles, c);
#pragma unroll
        for (int k = 0; k < kPlanPerThread; k++) {
            if (first + k * kPlanThreads >= n_tiles) break;  // (block-uniform)
            bucket_take(s_bucket, c[k] > threshold, 31 - __clz(c[k] | 1u));
        }
    }
    __syncthreads();
    if (t == 0) {
        uint32_t base = 0;
        for (int b = kCostBuckets - 1; b >= 0; b--) { s_bucket_base[b] = base; base += s_bucket[b]; s_bucket[b] = 0; }
        s_heavy = base;
    }
    __syncthreads();
    const uint32_t n_heavy = s_heavy;
    // The heavy tiles' quadrants into their buckets; the light tiles behind all of them in frame order: step k of a batch
    // gives wave w tiles first + k * kPlanThreads + 64 w .. + 63, so the (step, wave) pairs in order are the frame order,
    // and a running sum over their counts of light tiles says where each pair's tiles go.
    for (int first = 0; first < n_tiles; first += kPlanBatch) {
        plan_fetch(cost_in, first, n_tiles, c);
#pragma unroll
        for (int k = 0; k < kPlanPerThread; k++) {
            const int i = first + k * kPlanThreads + t;
            const bool heavy = c[k] > threshold;
            uint64_t lights = 0;
            if (first + k * kPlanThreads < n_tiles) {  // (block-uniform)
                const int b = 31 - __clz(c[k] | 1u);
                const uint32_t pos = s_bucket_base[b] + bucket_take(s_bucket, heavy, b);
                if (heavy)
                    for (uint32_t quad = 0; quad < 4; quad++) items[4 * pos + quad] = (uint32_t)i | (quad << 28) | kItemQuadrant;
                lights = __builtin_amdgcn_ballot_w64(i < n_tiles && !heavy);
            }
            if (lane == 0) s_light[k * kPlanWaves + wave] = (uint32_t)__builtin_popcountll(lights);
        }
        __syncthreads();
        if (wave == 0) {  // the running sum, by one wave: a lane adds up its eight consecutive counts, the lanes' sums are scanned
            constexpr int kPer = kPlanPerThread * kPlanWaves / 64;
            uint32_t v[kPer], mine = 0;
#pragma unroll
            for (int j = 0; j < kPer; j++) { v[j] = s_light[lane * kPer + j]; mine += v[j]; }
            uint32_t incl = mine;
            for (int off = 1; off < 64; off <<= 1) { uint32_t o = __shfl_up(incl, off, 64); if (lane >= off) incl += o; }
            uint32_t run = s_light_before + incl - mine;
#pragma unroll
            for (int j = 0; j < kPer; j++) { s_light[lane * kPer + j] = run; run += v[j]; }
            if (lane == 63) s_light_before = run;  // (read by every lane above, before this store: one wave, in step)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPlanPerThread; k++) {
            const int i = first + k * kPlanThreads + t;
            const bool light = i < n_tiles && !(c[k] > threshold);
            const uint64_t m = __builtin_amdgcn_ballot_w64(light);
            if (light) items[4 * n_heavy + s_light[k * kPlanWaves + wave] + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
        }
        __syncthreads();  // (s_light is rewritten by the next batch)
    }
    for (int i = t; i < n_tiles; i += kPlanThreads) cost_out[i] = 0;
    if (t == 0) { q->next = 0; q->n_items = 3 * n_heavy + (uint32_t)n_tiles; q->n_heavy = n_heavy; q->threshold = threshold; }
}

// (Every pointer the kernel writes through is a __restrict__ argument of its own: the scene tables are read in every
// round of the item loop, after the stores of the rounds before, and only when the compiler knows that those stores
// cannot touch the tables does it keep fetching them through the scalar cache.)
struct QueueArgs { QueueHeader *q; const uint32_t *items; uint32_t *cost_out; };
template <int F>
__global__ __launch_bounds__(kBlock, split_waves(F, false)) void k_render_queue(const DevScene *scene, RenderArgs a, float *__restrict__ rgb_tiles,
                                                                                 uint8_t *__restrict__ rgb8_tiles,
                                                                                 unsigned long long *__restrict__ counters,
                                                                                 QueueHeader *__restrict__ q_hdr,
                                                                                 const uint32_t *__restrict__ q_items,
                                                                                 uint32_t *__restrict__ q_cost_out) {
    SceneRef S = scene_ref(scene);
    QueueArgs qa = {q_hdr, q_items, q_cost_out};
    unbundle(a);
    RT_WAVE_STACK(stk, F, kSplitSlots);
    __shared__ float s_grid_off[kMaxGrid];
    __shared__ uint32_t s_item;
    fill_grid_table(s_grid_off, a);
    Work wk;
    work_zero(wk);
    const uint32_t n_items = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)qa.q->n_items);
    for (;;) {
        if (threadIdx.x == 0) s_item = atomicAdd(&qa.q->next, 1u);
        __syncthreads();
        const uint32_t at = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)s_item);  // block-uniform, and known to be so
        __syncthreads();  // everybody has read s_item before thread 0 overwrites it in the next round
        if (at >= n_items) break;
        const uint32_t item = (uint32_t)__builtin_amdgcn_readfirstlane((int32_t)qa.items[at]);
        const int local = (int)(item & 0x0fffffffu);
        const unsigned long long t0 = wall_clock64();
        render_item<F>(S, a, local, (item & kItemQuadrant) ? (int)((item >> 28) & 3u) : -1, stk, s_grid_off, rgb_tiles, rgb8_tiles, wk);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long dt = wall_clock64() - t0;
            atomicAdd(&qa.cost_out[local], dt > 0x3fffffffull ? 0x3fffffffu : (uint32_t)dt);  // a heavy tile: the sum over its quadrants
        }
    }
    flush_work(wk, counters, false);
}

// shards[r][local_tile][16][16][3] -> frame[y][x][3]; first_rank/n_ranks select which shards exist.
// A row of a tile is 48 * sizeof(T) contiguous bytes on both sides: one thread moves 16 of them, so a wave
// reads 1 KiB of consecutive shard bytes and writes whole 48- or 192-byte runs of the frame.  `vec`: both
// buffers are 16-byte aligned and so is every frame row; tiles that stick out of the frame, and any
// other buffers, go byte by byte.
template <typename T>
constexpr int untile_chunks_per_tile() { return RT_TILE_W * 3 * (int)sizeof(T) / 16 * RT_TILE_H; }
template <typename T>
__global__ void k_untile(const T *shards, T *frame, int width, int height, int tiles_x, int tiles_total,
                         int tiles_per_rank, int world_size, int first_rank, int n_ranks, int vec) {
    constexpr int kPixelBytes = 3 * (int)sizeof(T), kChunks = RT_TILE_W * kPixelBytes / 16, kTileChunks = untile_chunks_per_tile<T>();
    static_assert(RT_TILE_W * kPixelBytes % 16 == 0, "a tile row is a whole number of 16-byte pieces");
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)n_ranks * tiles_per_rank * kTileChunks;
    if (idx >= total) return;
    int within = (int)(idx % kTileChunks);
    size_t tl = idx / kTileChunks;
    int local = (int)(tl % tiles_per_rank), r = (int)(tl / tiles_per_rank) + first_rank;
    int tile = shard_tile(local, r, world_size);
    if (tile >= tiles_total) return;
    int row = within / kChunks, part = within % kChunks;
    int x0 = (tile % tiles_x) * RT_TILE_W, y = (tile / tiles_x) * RT_TILE_H + row;
    if (y >= height) return;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(shards) + idx * 16;
    unsigned char *dst = reinterpret_cast<unsigned char *>(frame) + ((size_t)y * width + x0) * kPixelBytes + part * 16;
    if (vec && x0 + RT_TILE_W <= width) {
        *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(src);
        return;
    }
    for (int j = 0; j < 16; j++)
        if (x0 + (part * 16 + j) / kPixelBytes < width) dst[j] = src[j];
}
// grid size and `vec` for a launch over `tiles` tiles
template <typename T>
static void untile_launch_shape(size_t tiles, const T *shards, const T *frame, int width, dim3 &grid, int &vec) {
    size_t chunks = tiles * untile_chunks_per_tile<T>();
    grid = dim3((unsigned)((chunks + 255) / 256));
    vec = (reinterpret_cast<uintptr_t>(shards) % 16 == 0 && reinterpret_cast<uintptr_t>(frame) % 16 == 0 &&
           ((size_t)width * 3 * sizeof(T)) % 16 == 0) ? 1 : 0;
}

__global__ __launch_bounds__(kBlock) void k_trace_rays(const DevScene *scene, long long n, const float *o, const float *d,
                                                       int max_depth, int accel, rt_hit_record *hits, float *radiance,
                                                       unsigned long long *counters) {
    SceneRef S = scene_ref(scene);
    RT_WAVE_STACK(stk, F_ALL, 0);
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    Work wk;
    work_zero(wk);
    if (r < n) {
        rt_hit_record h;
        v3 c = trace<F_ALL, false>(S, ld3(o + 3 * r), ld3(d + 3 * r), max_depth, accel, stk, wk, hits ? &h : nullptr);
        if (hits) hits[r] = h;
        if (radiance) { radiance[3 * r] = c.x; radiance[3 * r + 1] = c.y; radiance[3 * r + 2] = c.z; }
    }
    flush_work(wk, counters, false);
}

__global__ __launch_bounds__(kBlock) void k_intersect_object(const DevScene *scene, int object, long long n, const float *o,
                                                             const float *d, int accel, rt_hit_record *hits) {
    SceneRef S = scene_ref(scene);
    RT_WAVE_STACK(stk, F_ALL, 0);
    long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    Work wk;
    work_zero(wk);
    rt_hit_record h;
    intersect_object<false>(S, object, ld3(o + 3 * r), ld3(d + 3 * r), accel, h, stk, wk);
    hits[r] = h;
}

}  // namespace

// ------------------------------------------------------------------ context

// What k_render_hybrid and k_render_queue remember between frames: how long every tile took (read one array,
// write the other), plus the queue's work list.
struct TileMem {
    QueueHeader *hdr = nullptr;
    uint32_t *cost[2] = {nullptr, nullptr};
    uint32_t *items = nullptr;  // nullptr: no room for the queue's list
    int cap_tiles = 0;          // tiles the arrays hold
    int turn = 0;               // cost[turn] holds the last frame's times
    int32_t key[10] = {0};      // the frame parameters and scene those times belong to
    bool valid = false;
};

constexpr int kBuildStreams = 4;
struct rt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;      // the stream in use
    hipStream_t own_stream = nullptr;  // created with the context
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool trace_timed = false;  // ev0 / ev1 were recorded around a render launch (not around a build, and not never)
    std::string err;
    bool uploaded = false, built = false;
    std::vector<void *> allocs;  // everything owned by the current scene
    DevScene S{};                 // the scene record as the host fills it
    DevScene *d_scene = nullptr;  // its copy in device memory: what the kernels read
    std::vector<DevMesh> meshes;
    std::vector<MeshBuild> mesh_build;
    int n_triangles = 0, n_meshes = 0, n_nodes = 0, n_textures = 0, n_materials = 0;
    std::vector<int32_t> mesh_ranges;  // (tri_offset, tri_count) of every uploaded mesh
    std::vector<const void *> tex_addr;  // device address of every texture's texel records (nullptr: procedural)
    // device arrays the build writes
    float *d_tri_pos = nullptr;
    float *d_ref_pos = nullptr;    // 9 floats per reference (a piece of a triangle), in cutting order
    int32_t *d_ref_tri = nullptr;  // the triangle each reference is a piece of
    float *d_leaf_box = nullptr;   // lo, hi of the reference at each sorted position
    int n_refs = 0;
    f4 *d_tri_test = nullptr;
    uint32_t *d_tri_key = nullptr;
    int32_t *d_tri_orig = nullptr;
    int32_t *d_tri_frag = nullptr, *d_tri_sub = nullptr;
    BvhNode *d_nodes = nullptr;
    DevMesh *d_meshes = nullptr;
    unsigned long long *d_counters = nullptr;
    float last_trace_ms = 0, last_build_ms = 0;
    int features = F_ALL;
    int bvh_max_depth = 0;
    int split_min_taps = 32;  // taps per pixel from which k_render_split is used for scenes with meshes
    bool split_forced = false;  // RT_SPLIT_MIN_TAPS (tests, tuning) sets the threshold for every scene
    int n_cus = 256;            // compute units of the device
    bool radix_tree = false;    // RT_BUILD=radix: the Morton radix tree instead of the surface-area splits
    int split_phases = 1;  // RT_SPLIT_PHASES: which split kernel RT_SPLIT_MIN_TAPS selects
    TileMem tiles;              // tile times (and the queue's item list) of this context's launches
    unsigned scene_serial = 0;  // counts scene uploads: tile times of another scene are not used (rt_scene_update keeps them: a pose changes little from one frame to the next, and they only decide who renders a tile)
    int queue_slots[16] = {0};  // resident blocks of k_render_queue<features> on this device, by feature set (0: not asked yet)
    int forced_variant = -1;          // RT_RENDER_VARIANT: this kernel for every frame (tests)
    long long forced_threshold = -1;  // RT_HYBRID_THRESHOLD: ticks of 10 ns above which a tile counts as heavy (tests)
    int last_variant = 0;
    hipStream_t build_stream[kBuildStreams] = {nullptr, nullptr, nullptr, nullptr};  // rt_build_accel: meshes side by side
    hipEvent_t build_done[kBuildStreams] = {nullptr, nullptr, nullptr, nullptr}, build_fork = nullptr;
};

#define RT_HIP(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                        \
            return RT_ERR_HIP;                                                                   \
        }                                                                                        \
    } while (0)

static int fail(rt_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    return code;
}

template <typename T>
static int dev_alloc(rt_ctx *ctx, T **out, size_t count, const void *src, bool scene_owned = true) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    void *p = nullptr;
    RT_HIP(hipMalloc(&p, bytes));
    if (scene_owned) ctx->allocs.push_back(p);
    if (src && count) RT_HIP(hipMemcpyAsync(p, src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    else RT_HIP(hipMemsetAsync(p, 0, bytes, ctx->stream));
    *out = static_cast<T *>(p);
    return RT_OK;
}

// ctx->S and, behind it, the records of the lights beyond its own table (queued on the context's stream; `lights` must
// outlive the copy: the callers synchronise)
static int upload_scene_record(rt_ctx *ctx, const std::vector<LightRec> &lights) {
    RT_HIP(hipMemcpyAsync(ctx->d_scene, &ctx->S, sizeof ctx->S, hipMemcpyHostToDevice, ctx->stream));
    if ((int)lights.size() > kEmbLights)
        RT_HIP(hipMemcpyAsync(reinterpret_cast<char *>(ctx->d_scene) + sizeof(DevScene), lights.data() + kEmbLights,
                              (lights.size() - kEmbLights) * sizeof(LightRec), hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

static void free_scene(rt_ctx *ctx) {
    for (void *p : ctx->allocs) (void)hipFree(p);
    ctx->allocs.clear();
    ctx->uploaded = ctx->built = false;
    ctx->meshes.clear();
    ctx->mesh_build.clear();
    ctx->S = DevScene{};
    ctx->d_scene = nullptr;
}

static int noalloc_setup(rt_ctx *ctx);  // (below, next to the static tile memory)

extern "C" int rt_context_create(int device, rt_ctx **out) {
    if (!out) return RT_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return RT_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return RT_ERR_INVALID;
    rt_ctx *ctx = new rt_ctx();
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&ctx->own_stream) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    if (hipDeviceGetAttribute(&ctx->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || ctx->n_cus <= 0) ctx->n_cus = 256;
    if (const char *e = std::getenv("RT_BUILD")) ctx->radix_tree = std::strcmp(e, "radix") == 0;
    if (const char *e = std::getenv("RT_SPLIT_PHASES")) ctx->split_phases = std::atoi(e) == 2 ? 2 : 1;
    if (const char *e = std::getenv("RT_RENDER_VARIANT")) ctx->forced_variant = std::max(0, std::min(kVariantQueue, std::atoi(e)));
    if (const char *e = std::getenv("RT_HYBRID_THRESHOLD")) ctx->forced_threshold = std::atoll(e);
    if (const char *e = std::getenv("RT_SPLIT_MIN_TAPS")) { ctx->split_min_taps = std::atoi(e); ctx->split_forced = true; }
    void *c = nullptr;
    if (hipMalloc(&c, (size_t)kCounterStripes * kCounterStride * sizeof(unsigned long long)) != hipSuccess ||
        hipMemset(c, 0, (size_t)kCounterStripes * kCounterStride * sizeof(unsigned long long)) != hipSuccess) {
        delete ctx;
        return RT_ERR_HIP;
    }
    ctx->d_counters = static_cast<unsigned long long *>(c);
    if (noalloc_setup(ctx) != RT_OK) {  // the address of this device's copy of the static tile memory
        (void)hipFree(c);
        delete ctx;
        return RT_ERR_HIP;
    }
    *out = ctx;
    return RT_OK;
}

extern "C" int rt_context_destroy(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scene(ctx);
    (void)hipFree(ctx->d_counters);
    (void)hipFree(ctx->tiles.hdr);
    (void)hipEventDestroy(ctx->ev0);
    (void)hipEventDestroy(ctx->ev1);
    for (int k = 0; k < kBuildStreams; k++) {
        if (ctx->build_done[k]) (void)hipEventDestroy(ctx->build_done[k]);
        if (ctx->build_stream[k]) (void)hipStreamDestroy(ctx->build_stream[k]);
    }
    if (ctx->build_fork) (void)hipEventDestroy(ctx->build_fork);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return RT_OK;
}

extern "C" const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void *rt_stream(rt_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int rt_set_stream(rt_ctx *ctx, void *hip_stream) {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return RT_OK;
}

extern "C" int rt_sync(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

extern "C" int rt_scene_upload(rt_ctx *ctx, const rt_scene_desc *sc) {
    if (!ctx || !sc) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    std::string bad = validate_scene(sc);
    if (!bad.empty()) return fail(ctx, RT_ERR_INVALID, bad);
    RT_HIP(hipStreamSynchronize(ctx->stream));
    free_scene(ctx);

    int rc;
    rt_object *d_objects; rt_sphere *d_spheres; rt_plane *d_planes; rt_cone *d_cones; rt_material *d_materials;
    float *d_nrm, *d_uv; int32_t *d_tmat;
    if ((rc = dev_alloc(ctx, &d_objects, sc->n_objects, sc->objects))) return rc;
    if ((rc = dev_alloc(ctx, &d_spheres, sc->n_spheres, sc->spheres))) return rc;
    if ((rc = dev_alloc(ctx, &d_planes, sc->n_planes, sc->planes))) return rc;
    if ((rc = dev_alloc(ctx, &d_cones, sc->n_cones, sc->cones))) return rc;
    if ((rc = dev_alloc(ctx, &d_materials, sc->n_materials, sc->materials))) return rc;
    size_t nt = sc->n_triangles;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_pos, 9 * nt, sc->tri_pos))) return rc;
    if ((rc = dev_alloc(ctx, &d_nrm, 9 * nt, sc->tri_nrm))) return rc;
    if ((rc = dev_alloc(ctx, &d_uv, 6 * nt, sc->tri_uv))) return rc;
    if ((rc = dev_alloc(ctx, &d_tmat, nt, sc->tri_material))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_frag, nt, sc->tri_frag))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_sub, nt, sc->tri_sub))) return rc;

    // textures (the interleaved texels are staged in host vectors that live until the stream has been synchronised below)
    std::vector<DevTexture> tex(sc->n_textures);
    std::vector<std::vector<u4>> staged_texels(sc->n_textures);
    for (int i = 0; i < sc->n_textures; i++) {
        const rt_texture &t = sc->textures[i];
        tex[i].kind = t.kind; tex[i].w = t.width; tex[i].h = t.height; tex[i].pad = 0;
        tex[i].texels = nullptr;
        for (int k = 0; k < 3; k++) tex[i].pad2[k] = nullptr;
        if (t.kind == RT_TEX_IMAGE) {
            std::vector<u4> &texels = staged_texels[i];
            interleave_texture(t, texels);
            u4 *d;
            if ((rc = dev_alloc(ctx, &d, texels.size(), texels.data()))) return rc;
            tex[i].texels = d;
        }
    }
    std::vector<float> lut;
    make_srgb_lut(lut);
    float *d_lut;
    if ((rc = dev_alloc(ctx, &d_lut, lut.size(), lut.data()))) return rc;
    DevTexture *d_tex;
    if ((rc = dev_alloc(ctx, &d_tex, tex.size(), tex.data()))) return rc;
    std::vector<int32_t> plane_obj, sphere_obj, cone_obj, mesh_obj;
    kind_to_object(sc, plane_obj, sphere_obj, cone_obj, mesh_obj);
    int32_t *d_plane_obj, *d_cone_obj;
    SphereCull *d_sphere_cull;
    std::vector<SphereCull> sphere_cull;
    sphere_culls(sc, sphere_obj, sphere_cull);
    if ((rc = dev_alloc(ctx, &d_plane_obj, plane_obj.size(), plane_obj.data()))) return rc;
    if ((rc = dev_alloc(ctx, &d_sphere_cull, sphere_cull.size(), sphere_cull.data()))) return rc;
    if ((rc = dev_alloc(ctx, &d_cone_obj, cone_obj.size(), cone_obj.data()))) return rc;
    std::vector<AxisPair> axis_pairs;
    std::vector<PlaneRec> gen_planes;
    const uint32_t planes_skip_lights = make_plane_recs(sc, plane_obj, axis_pairs, gen_planes);
    AxisPair *d_axis_overflow;
    PlaneRec *d_gen_planes;
    // sized for any split of the planes into the two classes, so that rt_scene_update can re-partition them
    if ((rc = dev_alloc(ctx, &d_axis_overflow, axis_pair_capacity(sc->n_planes), (const void *)nullptr))) return rc;
    if ((rc = dev_alloc(ctx, &d_gen_planes, gen_plane_capacity(sc->n_planes), (const void *)nullptr))) return rc;
    if (axis_pairs.size() > (size_t)kFixedAxisPairs)
        RT_HIP(hipMemcpyAsync(d_axis_overflow, axis_pairs.data() + kFixedAxisPairs, (axis_pairs.size() - kFixedAxisPairs) * sizeof(AxisPair),
                              hipMemcpyHostToDevice, ctx->stream));
    if (!gen_planes.empty()) RT_HIP(hipMemcpyAsync(d_gen_planes, gen_planes.data(), gen_planes.size() * sizeof(PlaneRec), hipMemcpyHostToDevice, ctx->stream));
    // the materials and objects as shading reads them
    ctx->tex_addr.assign(sc->n_textures, nullptr);
    for (int i = 0; i < sc->n_textures; i++) ctx->tex_addr[i] = tex[i].texels;
    std::vector<MatRec> mat_recs;
    std::vector<ObjRec> obj_recs;
    make_mat_recs(sc, ctx->tex_addr, mat_recs);
    make_obj_recs(sc, mat_recs, obj_recs);
    MatRec *d_mat_recs;
    ObjRec *d_obj_recs;
    if ((rc = dev_alloc(ctx, &d_mat_recs, mat_recs.size(), mat_recs.data()))) return rc;
    if ((rc = dev_alloc(ctx, &d_obj_recs, obj_recs.size(), obj_recs.data()))) return rc;

    // meshes: bounds, padding, gate triangles, node ranges
    ctx->n_meshes = sc->n_meshes;
    ctx->n_triangles = sc->n_triangles;
    std::vector<f4> gate;
    std::vector<float> ref_pos;
    std::vector<int32_t> ref_tri;
    int node_total = prepare_meshes(sc, ctx->meshes, ctx->mesh_build, gate, true, ref_pos, ref_tri);
    for (int i = 0; i < sc->n_meshes; i++) ctx->meshes[i].object = mesh_obj[i];
    size_t nr = ref_tri.size();
    ctx->n_refs = (int)nr;
    if ((rc = dev_alloc(ctx, &ctx->d_ref_pos, 9 * nr, ref_pos.data()))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_ref_tri, nr, ref_tri.data()))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_leaf_box, 6 * nr, (const void *)nullptr))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_test, 3 * nr, (const void *)nullptr))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_key, nr, (const void *)nullptr))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_tri_orig, nr, (const void *)nullptr))) return rc;
    ctx->n_nodes = node_total;
    f4 *d_gate;
    if ((rc = dev_alloc(ctx, &d_gate, gate.size(), gate.data()))) return rc;
    std::vector<LightCull> cull;
    uint32_t cullable = make_light_cull(sc, ctx->meshes, planes_skip_lights, cull);
    std::vector<LightRec> light_recs;
    make_light_recs(sc, cull, cullable, light_recs);
    if ((rc = dev_alloc(ctx, &ctx->d_nodes, (size_t)node_total, (const void *)nullptr))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->d_meshes, ctx->meshes.size(), ctx->meshes.data()))) return rc;

    // identity triangle order: valid for RT_ACCEL_BRUTE until rt_build_accel sorts it
    for (int i = 0; i < sc->n_meshes; i++) {
        const DevMesh &m = ctx->meshes[i];
        if (m.tri_count <= 0) continue;
        hipLaunchKernelGGL(k_gather_tris, dim3((m.tri_count + 255) / 256), dim3(256), 0, ctx->stream, nullptr, m.tri_count,
                           m.tri_offset, ctx->d_ref_pos, ctx->d_ref_tri, ctx->d_tri_pos, ctx->d_tri_frag, ctx->d_tri_sub,
                           ctx->d_tri_test, ctx->d_tri_key, ctx->d_tri_orig, ctx->d_leaf_box, nullptr, nullptr, nullptr, 0);
    }
    RT_HIP(hipGetLastError());

    DevScene &S = ctx->S;
    S.n_objects = sc->n_objects; S.n_lights = sc->n_lights;
    S.n_planes = sc->n_planes; S.n_spheres = sc->n_spheres; S.n_cones = sc->n_cones; S.n_meshes = sc->n_meshes;
    S.n_axis_overflow = (int32_t)axis_pairs.size() - kFixedAxisPairs; S.n_gen_planes = (int32_t)gen_planes.size();
    S.axis_overflow = d_axis_overflow; S.gen_planes = d_gen_planes; S.planes_skip_lights = planes_skip_lights;
    S.obj_recs = d_obj_recs; S.mat_recs = d_mat_recs;
    S.plane_obj = d_plane_obj; S.sphere_cull = d_sphere_cull; S.cone_obj = d_cone_obj;
    S.objects = d_objects; S.spheres = d_spheres; S.planes = d_planes; S.cones = d_cones; S.meshes = ctx->d_meshes;
    S.tri_test = ctx->d_tri_test; S.tri_key = ctx->d_tri_key; S.tri_orig = ctx->d_tri_orig;
    S.tri_nrm = d_nrm; S.tri_uv = d_uv; S.tri_material = d_tmat; S.tri_frag = ctx->d_tri_frag; S.tri_sub = ctx->d_tri_sub;
    S.gate_tri = d_gate; S.unused_lights = nullptr; S.cullable_lights = cullable; S.unused_gate_order = 0; S.nodes = ctx->d_nodes; S.materials = d_materials; S.textures = d_tex; S.srgb_lut = d_lut;
    fill_embedded(S, axis_pairs, gen_planes, ctx->meshes, obj_recs, light_recs);
    {   // the scene record, then the records of the lights that do not fit its own table (light_rec_at)
        char *mem;
        if ((rc = dev_alloc(ctx, &mem, scene_record_bytes(sc->n_lights), (const void *)nullptr))) return rc;
        ctx->d_scene = reinterpret_cast<DevScene *>(mem);
        if ((rc = upload_scene_record(ctx, light_recs))) return rc;
    }
    RT_HIP(hipStreamSynchronize(ctx->stream));  // the caller's arrays may go away after return
    ctx->features = scene_features(sc);
    ctx->uploaded = true;
    ctx->scene_serial++;
    ctx->built = false;
    ctx->n_textures = sc->n_textures;
    ctx->n_materials = sc->n_materials;
    ctx->mesh_ranges.clear();
    for (int i = 0; i < sc->n_meshes; i++) { ctx->mesh_ranges.push_back(sc->meshes[i].tri_offset); ctx->mesh_ranges.push_back(sc->meshes[i].tri_count); }
    return RT_OK;
}

template <typename T>
static int dev_update(rt_ctx *ctx, const T *dst, const void *src, size_t count) {
    if (count) RT_HIP(hipMemcpyAsync(const_cast<T *>(dst), src, count * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    return RT_OK;
}

extern "C" int rt_scene_update(rt_ctx *ctx, const rt_scene_desc *sc) {
    if (!ctx || !sc) return RT_ERR_INVALID;
    if (!ctx->uploaded) return fail(ctx, RT_ERR_STATE, "rt_scene_update before rt_scene_upload");
    RT_HIP(hipSetDevice(ctx->device));
    std::string bad = validate_scene(sc);
    if (!bad.empty()) return fail(ctx, RT_ERR_INVALID, bad);
    DevScene &S = ctx->S;
    bool same = sc->n_objects == S.n_objects && sc->n_lights == S.n_lights && sc->n_planes == S.n_planes &&
                sc->n_spheres == S.n_spheres && sc->n_cones == S.n_cones && sc->n_meshes == S.n_meshes &&
                sc->n_triangles == ctx->n_triangles && sc->n_textures == ctx->n_textures;
    for (int i = 0; same && i < sc->n_meshes; i++)
        same = sc->meshes[i].tri_offset == ctx->mesh_ranges[2 * i] && sc->meshes[i].tri_count == ctx->mesh_ranges[2 * i + 1];
    if (!same) return fail(ctx, RT_ERR_INVALID, "rt_scene_update: counts or mesh triangle ranges differ from the uploaded scene");
    RT_HIP(hipStreamSynchronize(ctx->stream));  // no frame in flight may see half of the new tables

    int rc;
    if ((rc = dev_update(ctx, S.objects, sc->objects, sc->n_objects))) return rc;
    if ((rc = dev_update(ctx, S.spheres, sc->spheres, sc->n_spheres))) return rc;
    if ((rc = dev_update(ctx, S.planes, sc->planes, sc->n_planes))) return rc;
    if ((rc = dev_update(ctx, S.cones, sc->cones, sc->n_cones))) return rc;
    const bool grown_materials = sc->n_materials > ctx->n_materials;
    if (grown_materials) {  // more distinct materials than the table holds: a larger table
        rt_material *d_materials;
        if ((rc = dev_alloc(ctx, &d_materials, sc->n_materials, (const void *)nullptr))) return rc;
        S.materials = d_materials;
        ctx->n_materials = sc->n_materials;
    }
    if ((rc = dev_update(ctx, S.materials, sc->materials, sc->n_materials))) return rc;
    if ((rc = dev_update(ctx, S.tri_material, sc->tri_material, (size_t)sc->n_triangles))) return rc;  // indices into the new table
    std::vector<int32_t> plane_obj, sphere_obj, cone_obj, mesh_obj;
    kind_to_object(sc, plane_obj, sphere_obj, cone_obj, mesh_obj);
    if ((rc = dev_update(ctx, S.plane_obj, plane_obj.data(), plane_obj.size()))) return rc;
    std::vector<SphereCull> sphere_cull;
    sphere_culls(sc, sphere_obj, sphere_cull);
    if ((rc = dev_update(ctx, S.sphere_cull, sphere_cull.data(), sphere_cull.size()))) return rc;
    if ((rc = dev_update(ctx, S.cone_obj, cone_obj.data(), cone_obj.size()))) return rc;
    std::vector<AxisPair> axis_pairs;
    std::vector<PlaneRec> gen_planes;
    S.planes_skip_lights = make_plane_recs(sc, plane_obj, axis_pairs, gen_planes);
    if ((rc = dev_update(ctx, S.axis_overflow, axis_pairs.data() + kFixedAxisPairs, axis_pairs.size() - kFixedAxisPairs))) return rc;
    if ((rc = dev_update(ctx, S.gen_planes, gen_planes.data(), gen_planes.size()))) return rc;
    S.n_axis_overflow = (int32_t)axis_pairs.size() - kFixedAxisPairs; S.n_gen_planes = (int32_t)gen_planes.size();
    std::vector<MatRec> mat_recs;
    std::vector<ObjRec> obj_recs;
    make_mat_recs(sc, ctx->tex_addr, mat_recs);
    make_obj_recs(sc, mat_recs, obj_recs);
    if (grown_materials) {
        MatRec *d_mat_recs;
        if ((rc = dev_alloc(ctx, &d_mat_recs, mat_recs.size(), (const void *)nullptr))) return rc;
        S.mat_recs = d_mat_recs;
    }
    if ((rc = dev_update(ctx, S.mat_recs, mat_recs.data(), mat_recs.size()))) return rc;
    if ((rc = dev_update(ctx, S.obj_recs, obj_recs.data(), obj_recs.size()))) return rc;
    // mesh records: new maps, gate boxes and world bounds; triangle and node ranges are unchanged
    std::vector<DevMesh> meshes;
    std::vector<MeshBuild> build = ctx->mesh_build;  // the reference ranges stay
    std::vector<f4> gate;
    std::vector<float> no_pos;
    std::vector<int32_t> no_tri;
    prepare_meshes(sc, meshes, build, gate, false, no_pos, no_tri);
    for (int i = 0; i < sc->n_meshes; i++) {
        meshes[i].object = mesh_obj[i];
        if (meshes[i].node_offset != ctx->meshes[i].node_offset) return fail(ctx, RT_ERR_STATE, "rt_scene_update: node layout changed");
    }
    ctx->meshes = meshes;
    if ((rc = dev_update(ctx, S.gate_tri, gate.data(), gate.size()))) return rc;
    std::vector<LightCull> cull;
    S.cullable_lights = make_light_cull(sc, ctx->meshes, S.planes_skip_lights, cull);
    std::vector<LightRec> light_recs;
    make_light_recs(sc, cull, S.cullable_lights, light_recs);
    if ((rc = dev_update(ctx, S.meshes, ctx->meshes.data(), ctx->meshes.size()))) return rc;
    fill_embedded(S, axis_pairs, gen_planes, ctx->meshes, obj_recs, light_recs);
    ctx->features = scene_features(sc);
    S.unused_gate_order = 0;
    if ((rc = upload_scene_record(ctx, light_recs))) return rc;  // (the number of lights is the uploaded scene's)
    RT_HIP(hipStreamSynchronize(ctx->stream));  // the caller's arrays may go away after return
    return RT_OK;
}

extern "C" int rt_build_accel(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->uploaded) return fail(ctx, RT_ERR_STATE, "rt_build_accel before rt_scene_upload");
    if (ctx->built) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    // The meshes are built side by side: each on one of a few streams, with a slice of every work array to itself.
    // A build is a chain of some twenty-five small dependent launches (its first tree levels are single blocks), so
    // one mesh leaves most of the GPU idle and several of them cost little more than the largest alone.
    const size_t n_meshes = ctx->meshes.size();
    int max_n = 0;
    size_t sum_n = 0, sum_pow2 = 0;
    std::vector<size_t> at_n(n_meshes + 1, 0), at_pow2(n_meshes + 1, 0);  // where each mesh's slices start
    for (size_t i = 0; i < n_meshes; i++) {
        const int n = std::max(ctx->meshes[i].tri_count, 0);
        int n_pow2 = kSortChunk;  // the sort works on whole chunks
        while (n_pow2 < n) n_pow2 <<= 1;
        max_n = std::max(max_n, n);
        at_n[i] = sum_n; at_pow2[i] = sum_pow2;
        sum_n += (size_t)n; sum_pow2 += (size_t)n_pow2;
    }
    int max_levels = 2;         // height of the top-down tree: kSahDepthLimit + log2(n) at most (rt_lbvh.h)
    for (int v = 1; v < max_n; v <<= 1) max_levels++;
    max_levels += kSahDepthLimit;
    unsigned long long *d_keys = nullptr;
    int32_t *d_c0 = nullptr, *d_c1 = nullptr, *d_done = nullptr, *d_height = nullptr, *d_root = nullptr, *d_root_id = nullptr;
    float *d_lo = nullptr, *d_hi = nullptr, *d_bounds = nullptr;
    int32_t *d_idx[2] = {nullptr, nullptr}, *d_level_jobs = nullptr, *d_node_parent = nullptr, *d_leaf_parent = nullptr;
    SahJob *d_jobs[2] = {nullptr, nullptr};
    int rc;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *p : tmp) (void)hipFree(p); };
#define TMP_ALLOC(ptr, count)                                                    \
    if ((rc = dev_alloc(ctx, &ptr, (count), (const void *)nullptr, false))) { cleanup(); return rc; } \
    tmp.push_back(ptr)
    TMP_ALLOC(d_keys, sum_pow2);
    TMP_ALLOC(d_c0, sum_n);
    TMP_ALLOC(d_c1, sum_n);
    TMP_ALLOC(d_done, sum_n);
    TMP_ALLOC(d_height, sum_n);
    TMP_ALLOC(d_node_parent, sum_n);
    TMP_ALLOC(d_leaf_parent, sum_n);
    TMP_ALLOC(d_root, n_meshes + 1);
    TMP_ALLOC(d_root_id, n_meshes + 1);
    TMP_ALLOC(d_lo, 3 * sum_n);
    TMP_ALLOC(d_hi, 3 * sum_n);
    TMP_ALLOC(d_bounds, 6 * n_meshes + 6);
    if (!ctx->radix_tree) {
        TMP_ALLOC(d_idx[0], sum_n);
        TMP_ALLOC(d_idx[1], sum_n);
        TMP_ALLOC(d_jobs[0], sum_n);
        TMP_ALLOC(d_jobs[1], sum_n);
        TMP_ALLOC(d_level_jobs, (size_t)(max_levels + 2) * (n_meshes + 1));  // a row per mesh: the last count of each is checked at the end
    }
#undef TMP_ALLOC
    std::vector<float> bounds(6 * n_meshes + 6, 0.0f);
    for (size_t i = 0; i < n_meshes; i++)
        for (int a = 0; a < 3; a++) { bounds[6 * i + a] = ctx->mesh_build[i].lo[a]; bounds[6 * i + 3 + a] = ctx->mesh_build[i].hi[a]; }
    hipError_t e = hipMemcpyAsync(d_bounds, bounds.data(), bounds.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_root_id, 0, (n_meshes + 1) * sizeof(int32_t), ctx->stream);  // the radix tree's root is node 0
    if (e == hipSuccess) e = hipMemsetAsync(d_root, 0, (n_meshes + 1) * sizeof(int32_t), ctx->stream);
    if (e == hipSuccess && d_level_jobs) e = hipMemsetAsync(d_level_jobs, 0, (size_t)(max_levels + 2) * (n_meshes + 1) * sizeof(int32_t), ctx->stream);
    if (e == hipSuccess && sum_n) e = hipMemsetAsync(d_done, 0, sum_n * sizeof(int32_t), ctx->stream);
    // the side streams (kept with the context) and the events that fork them off the context's stream and join them back
    int n_side = 0;
    for (size_t i = 0; i < n_meshes; i++) if (ctx->meshes[i].tri_count > 0) n_side++;
    n_side = std::min(n_side, kBuildStreams);
    if (n_side < 2) n_side = 0;  // a single mesh is built on the context's stream
    for (int k = 0; k < n_side && e == hipSuccess; k++) {
        if (!ctx->build_stream[k]) e = hipStreamCreateWithFlags(&ctx->build_stream[k], hipStreamNonBlocking);
        if (e == hipSuccess && !ctx->build_done[k]) e = hipEventCreateWithFlags(&ctx->build_done[k], hipEventDisableTiming);
    }
    if (e == hipSuccess && n_side && !ctx->build_fork) e = hipEventCreateWithFlags(&ctx->build_fork, hipEventDisableTiming);
    if (e != hipSuccess) { cleanup(); return fail(ctx, RT_ERR_HIP, hipGetErrorString(e)); }

    // Everything below is queued without waiting for the device: the number of jobs of a level stays there.
    std::vector<int> unfinished_at(n_meshes, -1);  // per mesh: the entry of its row of d_level_jobs that must still be zero at the end
    (void)hipEventRecord(ctx->ev0, ctx->stream);
    if (n_side) {
        (void)hipEventRecord(ctx->build_fork, ctx->stream);
        for (int k = 0; k < n_side; k++) (void)hipStreamWaitEvent(ctx->build_stream[k], ctx->build_fork, 0);
    }
    // The launches of a mesh are collected first and then issued in turn, one of every mesh at a time: issued mesh by
    // mesh, the second mesh's first kernel would wait on the host behind the forty launches of the first.
    int turn = 0;
    std::vector<std::vector<std::function<void()>>> ops(n_meshes);
    for (size_t i = 0; i < n_meshes; i++) {
        const DevMesh &m = ctx->meshes[i];
        int n = m.tri_count;
        if (n <= 0) continue;
        hipStream_t st = n_side ? ctx->build_stream[turn++ % n_side] : ctx->stream;
        std::vector<std::function<void()>> &q = ops[i];
        int n_pow2 = kSortChunk;
        while (n_pow2 < n) n_pow2 <<= 1;
        // this mesh's slices of the work arrays
        unsigned long long *keys = d_keys + at_pow2[i];
        int32_t *c0 = d_c0 + at_n[i], *c1 = d_c1 + at_n[i], *done = d_done + at_n[i], *height = d_height + at_n[i];
        int32_t *node_parent = d_node_parent + at_n[i], *leaf_parent = d_leaf_parent + at_n[i];
        float *lo = d_lo + 3 * at_n[i], *hi = d_hi + 3 * at_n[i];
        int32_t *idx0 = d_idx[0] ? d_idx[0] + at_n[i] : nullptr, *idx1 = d_idx[1] ? d_idx[1] + at_n[i] : nullptr;
        SahJob *jobs[2] = {d_jobs[0] ? d_jobs[0] + at_n[i] : nullptr, d_jobs[1] ? d_jobs[1] + at_n[i] : nullptr};
        dim3 blk(256), grd((n_pow2 + 255) / 256);
        q.push_back([=]() { hipLaunchKernelGGL(k_sort_keys, grd, blk, 0, st, ctx->d_ref_pos + 9 * (size_t)m.tri_offset, n, n_pow2,
                           d_bounds + 6 * i, d_bounds + 6 * i + 3, keys); });
        q.push_back([=]() { hipLaunchKernelGGL(k_bitonic_chunk, dim3(n_pow2 / kSortChunk), dim3(kSortThreads), 0, st, keys, kSortChunk, 1); });
        for (int k = 2 * kSortChunk; k <= n_pow2; k <<= 1) {
            for (int j = k >> 1; j >= kSortChunk; j >>= 1)
                q.push_back([=]() { hipLaunchKernelGGL(k_bitonic_step, grd, blk, 0, st, keys, n_pow2, j, k); });
            q.push_back([=]() { hipLaunchKernelGGL(k_bitonic_chunk, dim3(n_pow2 / kSortChunk), dim3(kSortThreads), 0, st, keys, k, 0); });
        }
        const bool top_down = !ctx->radix_tree && n >= 2;
        int32_t *level_jobs = top_down ? d_level_jobs + (size_t)(max_levels + 2) * i : nullptr;
        q.push_back([=]() { hipLaunchKernelGGL(k_gather_tris, dim3((n + 255) / 256), blk, 0, st, keys, n, m.tri_offset, ctx->d_ref_pos,
                           ctx->d_ref_tri, ctx->d_tri_pos, ctx->d_tri_frag, ctx->d_tri_sub, ctx->d_tri_test, ctx->d_tri_key,
                           ctx->d_tri_orig, ctx->d_leaf_box, top_down ? idx0 : nullptr, jobs[0], level_jobs, max_levels); });
        if (n >= 2) {
            if (ctx->radix_tree) {
                q.push_back([=]() { hipLaunchKernelGGL(k_children, dim3((n + 255) / 256), blk, 0, st, keys, n, c0, c1, node_parent, leaf_parent); });
            } else {
                // top-down, one launch per level for the levels a balanced tree has and two more, then one launch
                // that finishes what is left.  A level has at most min(2^level, n / 2) jobs: each covers two
                // leaves or more of its own.
                int levels = 2;
                for (int v = 1; v < n; v <<= 1) levels++;
                for (int level = 0; level <= levels; level++) {
                    const bool finish = level == levels;
                    long long most = level < 30 ? std::min<long long>(1ll << level, n / 2) : n / 2;
                    // a job of this level has n / 2^level leaves if the tree is balanced; no more than four per thread then
                    int waves = 1;
                    while (!finish && waves < kSahMaxWaves && (long long)waves * 256 * (1ll << std::min(level, 30)) < n) waves <<= 1;
                    q.push_back([=]() { hipLaunchKernelGGL(k_sah_level, dim3((unsigned)most), dim3(64 * waves), 0, st,
                                       jobs[level & 1], level_jobs + level, ctx->d_leaf_box + 6 * (size_t)m.tri_offset, idx0, idx1,
                                       c0, c1, node_parent, leaf_parent, jobs[(level & 1) ^ 1], d_root_id + i, finish ? 1 : 0); });
                }
                unfinished_at[i] = levels + 2;
            }
            q.push_back([=]() { hipLaunchKernelGGL(k_fit, dim3((n + 255) / 256), blk, 0, st, n, c0, c1, node_parent, leaf_parent,
                               ctx->d_leaf_box + 6 * (size_t)m.tri_offset, ctx->mesh_build[i].pad, done, lo, hi, height,
                               ctx->d_nodes + m.node_offset, d_root + i); });
            int nv = (n - 1) * (kOctants - 1);
            q.push_back([=]() { hipLaunchKernelGGL(k_octant_variants, dim3((nv + 255) / 256), blk, 0, st, ctx->d_nodes + m.node_offset, n - 1); });
            q.push_back([=]() { hipLaunchKernelGGL(k_octant_variant0, dim3((n - 1 + 255) / 256), blk, 0, st, ctx->d_nodes + m.node_offset, n - 1); });
        }
    }
    for (size_t step = 0;; step++) {
        bool any = false;
        for (size_t i = 0; i < n_meshes; i++)
            if (step < ops[i].size()) { ops[i][step](); any = true; }
        if (!any) break;
    }
    for (int k = 0; k < n_side; k++) {
        (void)hipEventRecord(ctx->build_done[k], ctx->build_stream[k]);
        (void)hipStreamWaitEvent(ctx->stream, ctx->build_done[k], 0);
    }
    (void)hipEventRecord(ctx->ev1, ctx->stream);
    ctx->trace_timed = false;
    if (e == hipSuccess) e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    std::vector<int32_t> roots(n_meshes + 1, 0), root_ids(n_meshes + 1, 0);
    if (e == hipSuccess) e = hipMemcpy(roots.data(), d_root, roots.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(root_ids.data(), d_root_id, root_ids.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
    bool unfinished = false;
    if (e == hipSuccess && d_level_jobs) {  // cannot happen: the height is bounded
        std::vector<int32_t> counts((size_t)(max_levels + 2) * (n_meshes + 1));
        e = hipMemcpy(counts.data(), d_level_jobs, counts.size() * sizeof(int32_t), hipMemcpyDeviceToHost);
        for (size_t i = 0; i < n_meshes; i++)
            unfinished = unfinished || (unfinished_at[i] >= 0 && counts[(size_t)(max_levels + 2) * i + unfinished_at[i]] != 0);
    }
    for (size_t i = 0; i < n_meshes; i++) { ctx->mesh_build[i].root = root_ids[i]; ctx->meshes[i].root = root_ids[i]; }
    if (e == hipSuccess)  // the roots
        e = hipMemcpy(ctx->d_meshes, ctx->meshes.data(), n_meshes * sizeof(DevMesh), hipMemcpyHostToDevice);
    for (size_t i = 0; i < n_meshes && i < (size_t)kEmbMeshes; i++) ctx->S.mesh_heads[i] = ctx->meshes[i];  // (the copies in the scene record)
    if (e == hipSuccess) e = hipMemcpy(ctx->d_scene, &ctx->S, sizeof ctx->S, hipMemcpyHostToDevice);
    cleanup();
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, std::string("tree build: ") + hipGetErrorString(e));
    if (unfinished) return fail(ctx, RT_ERR_LIMIT, "tree build: levels left over");
    (void)hipEventElapsedTime(&ctx->last_build_ms, ctx->ev0, ctx->ev1);
    ctx->bvh_max_depth = 0;
    for (size_t i = 0; i < ctx->meshes.size(); i++) {
        ctx->mesh_build[i].height = roots[i];
        ctx->bvh_max_depth = std::max(ctx->bvh_max_depth, roots[i]);
    }
    if (ctx->bvh_max_depth > kStackDepth)
        return fail(ctx, RT_ERR_LIMIT, "tree height " + std::to_string(ctx->bvh_max_depth) + " exceeds the traversal stack (" +
                                           std::to_string(kStackDepth) + ")");
    ctx->built = true;
    return RT_OK;
}

// ------------------------------------------------------------------ rendering

static int check_params(rt_ctx *ctx, const rt_render_params *p) {
    if (!p) return RT_ERR_INVALID;
    if (p->width <= 0 || p->height <= 0 || p->width > 65536 || p->height > 65536) return fail(ctx, RT_ERR_INVALID, "bad frame size");
    if (p->mode < RT_MODE_REF_1SPP || p->mode > RT_MODE_GRID) return fail(ctx, RT_ERR_INVALID, "bad sampling mode");
    if (p->mode == RT_MODE_GRID && (p->grid_m < 1 || p->grid_m > 64)) return fail(ctx, RT_ERR_INVALID, "grid_m out of range");
    if (p->max_depth < 0 || p->max_depth > RT_MAX_DEPTH) return fail(ctx, RT_ERR_INVALID, "max_depth out of range");
    if (p->accel != RT_ACCEL_LBVH && p->accel != RT_ACCEL_BRUTE) return fail(ctx, RT_ERR_INVALID, "bad accel");
    if (p->world_size < 1 || p->rank < 0 || p->rank >= p->world_size) return fail(ctx, RT_ERR_INVALID, "bad rank/world_size");
    if (p->kernel < RT_KERNEL_AUTO || p->kernel > RT_KERNEL_QUEUE) return fail(ctx, RT_ERR_INVALID, "bad kernel");
    if (!(p->fov > 0 && p->fov < 180)) return fail(ctx, RT_ERR_INVALID, "bad fov");
    return RT_OK;
}

extern "C" int rt_shard_tiles(const rt_render_params *p, int32_t *tiles_per_rank, int32_t *tiles_total) {
    if (!p || p->width <= 0 || p->height <= 0 || p->world_size < 1) return RT_ERR_INVALID;
    int tx = (p->width + RT_TILE_W - 1) / RT_TILE_W, ty = (p->height + RT_TILE_H - 1) / RT_TILE_H;
    int total = tx * ty;
    if (tiles_total) *tiles_total = total;
    if (tiles_per_rank) *tiles_per_rank = (total + p->world_size - 1) / p->world_size;
    return RT_OK;
}

static int ready_for(rt_ctx *ctx, int accel) {
    if (!ctx->uploaded) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (accel == RT_ACCEL_LBVH && !ctx->built && ctx->n_triangles > 0)
        return fail(ctx, RT_ERR_STATE, "RT_ACCEL_LBVH requested before rt_build_accel");
    return RT_OK;
}

// Memory of the kernels that remember how long every tile took (k_render_hybrid, k_render_queue), for `tiles`
// tiles in one piece: header, two time arrays used in turn and, for the queue, the item list (4 per tile at most).
static size_t tile_mem_bytes(int tiles, bool with_items) { return sizeof(QueueHeader) + (size_t)tiles * (with_items ? 6 : 2) * sizeof(uint32_t); }
static void tile_mem_carve(TileMem &tm, void *base, int tiles, bool with_items) {
    unsigned char *b = static_cast<unsigned char *>(base);
    tm.hdr = reinterpret_cast<QueueHeader *>(b);
    tm.cost[0] = reinterpret_cast<uint32_t *>(b + sizeof(QueueHeader));
    tm.cost[1] = tm.cost[0] + tiles;
    tm.items = with_items ? tm.cost[1] + tiles : nullptr;
    tm.cap_tiles = tiles;
}

// ---- the entry point that allocates nothing: its tile times live in a file-scope device array
//
// One array per GPU (every device that loads this library holds its own copy, at its own address, until the
// process ends).  Host code never uses the array's name as an address: rt_context_create asks the runtime
// for the address on the context's device (hipGetSymbolAddress) and everything below works with that pointer.
constexpr int kNoAllocMaxTiles = 65536;  // e.g. a 4096 x 4096 frame on one GPU, or 8192 x 8192 split over four
constexpr size_t kNoAllocBytes = sizeof(QueueHeader) + (size_t)kNoAllocMaxTiles * 2 * sizeof(uint32_t);
static __device__ __align__(256) unsigned char scratch[kNoAllocBytes];
constexpr int kMaxDevices = 64;
static TileMem g_noalloc[kMaxDevices];  // per device: the carved pointer into that device's array and what its times belong to
static bool g_noalloc_ready[kMaxDevices];
// All contexts of a GPU share that GPU's array, and the per-tile kernel reads one frame's tile times while it writes the
// next frame's: two launches in flight at once would read each other's half-written times and disagree, block by
// block, on which tiles are split.  Launches through this entry point are therefore ordered per device: each waits
// for the event the previous one recorded (created with the array; waiting and recording allocate nothing), and the
// host-side bookkeeping (which of the two time arrays is current, whose frame they describe) is done under a lock.
static hipEvent_t g_noalloc_done[kMaxDevices];
static std::mutex g_noalloc_mutex;

static int noalloc_setup(rt_ctx *ctx) {  // once per device, with that device current
    if (ctx->device < 0 || ctx->device >= kMaxDevices) return RT_OK;  // (such a device gets RT_ERR_LIMIT from the entry point)
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    if (g_noalloc_ready[ctx->device]) return RT_OK;
    void *ptr = nullptr;
    RT_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    RT_HIP(hipEventCreateWithFlags(&g_noalloc_done[ctx->device], hipEventDisableTiming));
    tile_mem_carve(g_noalloc[ctx->device], ptr, kNoAllocMaxTiles, false);
    g_noalloc_ready[ctx->device] = true;
    return RT_OK;
}

template <int F>
static int queue_slots_of(rt_ctx *ctx) {
    int &slots = ctx->queue_slots[F];
    if (slots == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_render_queue<F>, kBlock, 0) != hipSuccess || per_cu <= 0) per_cu = 4;
        slots = per_cu * ctx->n_cus;
    }
    return slots;
}

// Which kernel renders a launch.  The rule is fixed (no timing).  A block of k_render walks all taps of its pixels
// one after the other, and a tile on a mesh takes many times as long as the average tile, so the longest blocks
// decide when a launch ends: the whole frame on one GPU loses a tenth to them, a rank's share of a frame (a launch
// of a few rounds of blocks) half its time.  Scenes with meshes and four or more taps per pixel therefore go to the
// persistent k_render_queue, which hands out the quadrants of the tiles that took long in the previous frame first and
// the other tiles whole after them (the 4K teapot frame: 1.69 ms against k_render's 1.93; one GPU's share of it at
// 2 / 4 / 8 GPUs: 0.87 / 0.46 / 0.26 ms).  k_render_hybrid, which decides per tile without a list, is 2 to 20 % behind it
// at these sizes and stays selectable by name.  Everything else goes to k_render, which has the fewest instructions per
// ray; so does the first frame of a long launch, whose tiles' times it records (render_shard).
// rt_render_params.kernel or RT_RENDER_VARIANT name another kernel.
static int pick_variant(const rt_ctx *ctx, const rt_render_params *p, int local_tiles) {
    if (p->count_work) return 0;  // the counting instantiation is k_render's
    if (ctx->forced_variant >= 0) return ctx->forced_variant;
    if (p->kernel > 0) return p->kernel - 1;
    const int taps = taps_of(p->mode, p->grid_m);
    if (ctx->split_forced) return taps >= ctx->split_min_taps ? ctx->split_phases : 0;
    if (!(ctx->features & F_MESH) || taps < 4) return 0;
    (void)local_tiles;
    return kVariantQueue;  // (a long launch whose tiles' times are not known yet is rendered by k_render, which records them: render_shard)
}

// This rank's tiles of one frame into tile-major device buffers.  tm: where the tiles' times are kept (the
// context's own memory, grown on demand when may_allocate, or the static array).
// by_rule: the kernel was chosen by pick_variant's rule, not named by the caller.
static int render_shard(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_tiles, uint8_t *d_rgb8_tiles, TileMem &tm, bool may_allocate,
                        int variant, bool by_rule) {
    RenderArgs a;
    a.width = p->width; a.height = p->height; a.mode = p->mode; a.grid_m = p->grid_m; a.max_depth = p->max_depth;
    a.accel = p->accel; a.rank = p->rank; a.world_size = p->world_size;
    a.tiles_x = (p->width + RT_TILE_W - 1) / RT_TILE_W;
    int ty = (p->height + RT_TILE_H - 1) / RT_TILE_H;
    a.tiles_total = a.tiles_x * ty;
    a.cam = make_camera(p->width, p->height, p->fov);
    // one tile of every group of world_size tiles (shard_tile); in the last group it may lie past the frame: its block leaves at once
    const int local_tiles = (a.tiles_total + p->world_size - 1) / p->world_size;
    if (local_tiles <= 0) return RT_OK;
    const bool timed_tiles = variant == 3 || variant == kVariantQueue;
    uint32_t *record_times = nullptr;  // k_render leaves the tiles' times here (see below)
    if (timed_tiles) {
        if (local_tiles > tm.cap_tiles || (variant == kVariantQueue && !tm.items)) {
            if (!may_allocate)
                return fail(ctx, RT_ERR_LIMIT, "the launch has " + std::to_string(local_tiles) + " tiles; the entry point that allocates nothing holds " +
                                                   std::to_string(tm.cap_tiles));
            RT_HIP(hipStreamSynchronize(ctx->stream));
            (void)hipFree(tm.hdr);
            tm = TileMem();
            void *mem = nullptr;
            RT_HIP(hipMalloc(&mem, tile_mem_bytes(local_tiles, true)));
            tile_mem_carve(tm, mem, local_tiles, true);
        }
        int32_t fov_bits;
        std::memcpy(&fov_bits, &p->fov, sizeof fov_bits);
        const int32_t key[10] = {p->width, p->height, p->mode, p->grid_m, p->max_depth, p->rank, p->world_size, fov_bits, ctx->features,
                                 (int32_t)ctx->scene_serial};
        if (!tm.valid || std::memcmp(key, tm.key, sizeof key) != 0) {  // nothing known about these tiles
            std::memcpy(tm.key, key, sizeof key);
            tm.valid = true;
            // A long launch (a whole frame on one GPU) goes through k_render once, which is the fastest kernel for a frame
            // about which nothing is known, and leaves every tile's time for the frames that follow; a short one (a rank's
            // share) counts all tiles as heavy once, which is what bounds its longest block.
            if (by_rule && local_tiles >= 16 * 5 * ctx->n_cus) record_times = tm.cost[tm.turn ^ 1];
            else RT_HIP(hipMemsetAsync(tm.cost[tm.turn], 0xff, (size_t)local_tiles * sizeof(uint32_t), ctx->stream));
        }
    }
    if (record_times) variant = 0;
    RT_HIP(hipEventRecord(ctx->ev0, ctx->stream));
    ctx->last_variant = variant;
    const int f = ctx->features;
    const uint32_t forced_threshold = ctx->forced_threshold >= 0 ? (uint32_t)std::min<long long>(ctx->forced_threshold, 0xfffffffell) : 0u;
    const int use_forced = ctx->forced_threshold >= 0 ? 1 : 0;
    if (variant == kVariantQueue) {
        typedef void (*queue_fn)(const DevScene *, RenderArgs, float *, uint8_t *, unsigned long long *, QueueHeader *, const uint32_t *, uint32_t *);
        queue_fn kernel;
        int slots;
#define RT_QUEUE(FEATURES) (kernel = k_render_queue<FEATURES>, slots = queue_slots_of<FEATURES>(ctx))
        if ((f & ~F_MESH) == 0) RT_QUEUE(F_MESH);
#ifndef RT_LEAN_BUILD
        else if (f == 0) RT_QUEUE(0);
        else if ((f & ~(F_MESH | F_SECONDARY)) == 0) RT_QUEUE(F_MESH | F_SECONDARY);
        else if ((f & ~(F_QUADRIC | F_SECONDARY)) == 0) RT_QUEUE(F_QUADRIC | F_SECONDARY);
        else if ((f & ~(F_QUADRIC | F_TEXTURE | F_SECONDARY)) == 0) RT_QUEUE(F_QUADRIC | F_TEXTURE | F_SECONDARY);
        else if ((f & ~(F_MESH | F_TEXTURE | F_SECONDARY)) == 0) RT_QUEUE(F_MESH | F_TEXTURE | F_SECONDARY);
#endif
        else RT_QUEUE(F_ALL);
#undef RT_QUEUE
        const int blocks = std::min(slots, 4 * local_tiles);
        hipLaunchKernelGGL(k_plan_tiles, dim3(1), dim3(kPlanThreads), 0, ctx->stream, tm.hdr, tm.cost[tm.turn], tm.cost[tm.turn ^ 1], tm.items,
                           local_tiles, blocks, forced_threshold, use_forced);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBlock), 0, ctx->stream, ctx->d_scene, a, d_rgb_tiles, d_rgb8_tiles, ctx->d_counters, tm.hdr,
                           (const uint32_t *)tm.items, tm.cost[tm.turn ^ 1]);
        tm.turn ^= 1;
    } else if (variant == 3) {
        typedef void (*hybrid_fn)(const DevScene *, RenderArgs, float *, uint8_t *, unsigned long long *, TileCosts);
        hybrid_fn hk = k_render_hybrid<F_ALL>;
        if ((f & ~F_MESH) == 0) hk = k_render_hybrid<F_MESH>;
#ifndef RT_LEAN_BUILD
        else if ((f & ~(F_MESH | F_SECONDARY)) == 0) hk = k_render_hybrid<F_MESH | F_SECONDARY>;
        else if ((f & ~(F_MESH | F_TEXTURE | F_SECONDARY)) == 0) hk = k_render_hybrid<F_MESH | F_TEXTURE | F_SECONDARY>;
#endif
        // a tile is heavy when it took more than 1.5 times the average tile: the threshold is formed on the device
        // from last frame's times, so that no frame waits for the host to look at them
        hipLaunchKernelGGL(k_tile_threshold, dim3(1), dim3(kPlanThreads), 0, ctx->stream, tm.hdr, (const uint32_t *)tm.cost[tm.turn], local_tiles,
                           forced_threshold, use_forced);
        TileCosts tc = {tm.cost[tm.turn], tm.cost[tm.turn ^ 1], &tm.hdr->threshold};
        hipLaunchKernelGGL(hk, dim3(4 * local_tiles), dim3(kBlock), 0, ctx->stream, ctx->d_scene, a, d_rgb_tiles, d_rgb8_tiles, ctx->d_counters, tc);
        tm.turn ^= 1;
    } else {
        typedef void (*render_fn)(const DevScene *, RenderArgs, float *, uint8_t *, unsigned long long *, uint32_t *);
        render_fn kernel;
#define RT_PICK(FEATURES, COUNT)                                                              \
    (variant == 0 ? (render_fn)k_render<FEATURES, COUNT>                                      \
                  : variant == 1 ? (render_fn)k_render_split<FEATURES, COUNT, 1> : (render_fn)k_render_split<FEATURES, COUNT, 2>)
        if (p->count_work) kernel = (render_fn)k_render<F_ALL, true>;
        else if ((f & ~F_MESH) == 0) kernel = RT_PICK(F_MESH, false);
#ifndef RT_LEAN_BUILD  // (a developer's build with -DRT_LEAN_BUILD has the mesh-only and the all-features kernels alone: a quarter of the compile time)
        else if (f == 0) kernel = RT_PICK(0, false);
        else if ((f & ~(F_MESH | F_SECONDARY)) == 0) kernel = RT_PICK(F_MESH | F_SECONDARY, false);
        else if ((f & ~(F_QUADRIC | F_SECONDARY)) == 0) kernel = RT_PICK(F_QUADRIC | F_SECONDARY, false);
        else if ((f & ~(F_QUADRIC | F_TEXTURE | F_SECONDARY)) == 0) kernel = RT_PICK(F_QUADRIC | F_TEXTURE | F_SECONDARY, false);
        else if ((f & ~(F_MESH | F_TEXTURE | F_SECONDARY)) == 0) kernel = RT_PICK(F_MESH | F_TEXTURE | F_SECONDARY, false);
#endif
        else kernel = RT_PICK(F_ALL, false);
#undef RT_PICK
        const int blocks_per_tile = variant == 0 ? 1 : variant == 2 ? 8 : 4;
        hipLaunchKernelGGL(kernel, dim3(blocks_per_tile * local_tiles), dim3(kBlock), 0, ctx->stream, ctx->d_scene, a, d_rgb_tiles, d_rgb8_tiles,
                           ctx->d_counters, record_times);
        if (record_times) tm.turn ^= 1;
    }
    RT_HIP(hipGetLastError());
    RT_HIP(hipEventRecord(ctx->ev1, ctx->stream));
    ctx->trace_timed = true;
    return RT_OK;
}

extern "C" int rt_render_shard_device(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_tiles, uint8_t *d_rgb8_tiles) {
    if (!ctx) return RT_ERR_INVALID;
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if ((rc = ready_for(ctx, p->accel))) return rc;
    RT_HIP(hipSetDevice(ctx->device));
    int per_rank = 0, total = 0;
    rt_shard_tiles(p, &per_rank, &total);
    const bool by_rule = !p->count_work && ctx->forced_variant < 0 && p->kernel <= 0 && !ctx->split_forced;
    return render_shard(ctx, p, d_rgb_tiles, d_rgb8_tiles, ctx->tiles, true, pick_variant(ctx, p, per_rank), by_rule);
}

extern "C" int rt_render_shard_device_noalloc(rt_ctx *ctx, const rt_render_params *p, float *d_rgb_tiles, uint8_t *d_rgb8_tiles) {
    if (!ctx) return RT_ERR_INVALID;
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if ((rc = ready_for(ctx, p->accel))) return rc;
    if (p->count_work) return fail(ctx, RT_ERR_INVALID, "rt_render_shard_device_noalloc does not count work");
    if (ctx->device < 0 || ctx->device >= kMaxDevices || !g_noalloc_ready[ctx->device])
        return fail(ctx, RT_ERR_LIMIT, "no static tile memory for device " + std::to_string(ctx->device));
    RT_HIP(hipSetDevice(ctx->device));
    // always the kernel that keeps tile times (they are what lives in the static array); one launch after the other per device
    std::lock_guard<std::mutex> lock(g_noalloc_mutex);
    RT_HIP(hipStreamWaitEvent(ctx->stream, g_noalloc_done[ctx->device], 0));
    rc = render_shard(ctx, p, d_rgb_tiles, d_rgb8_tiles, g_noalloc[ctx->device], false, 3, false);
    RT_HIP(hipEventRecord(g_noalloc_done[ctx->device], ctx->stream));
    return rc;
}

static int untile_on(rt_ctx *ctx, hipStream_t stream, const rt_render_params *p, const float *d_rgb_shards, float *d_rgb_frame,
                     const uint8_t *d_rgb8_shards, uint8_t *d_rgb8_frame) {
    if (!ctx) return RT_ERR_INVALID;
    int rc = check_params(ctx, p);
    if (rc) return rc;
    RT_HIP(hipSetDevice(ctx->device));
    int per_rank, total;
    rt_shard_tiles(p, &per_rank, &total);
    int tiles_x = (p->width + RT_TILE_W - 1) / RT_TILE_W;
    size_t tiles = (size_t)p->world_size * per_rank;
    dim3 blk(256), grd;
    int vec;
    if (d_rgb_shards && d_rgb_frame) {
        untile_launch_shape(tiles, d_rgb_shards, d_rgb_frame, p->width, grd, vec);
        hipLaunchKernelGGL(k_untile<float>, grd, blk, 0, stream, d_rgb_shards, d_rgb_frame, p->width, p->height, tiles_x, total,
                           per_rank, p->world_size, 0, p->world_size, vec);
    }
    if (d_rgb8_shards && d_rgb8_frame) {
        untile_launch_shape(tiles, d_rgb8_shards, d_rgb8_frame, p->width, grd, vec);
        hipLaunchKernelGGL(k_untile<uint8_t>, grd, blk, 0, stream, d_rgb8_shards, d_rgb8_frame, p->width, p->height, tiles_x,
                           total, per_rank, p->world_size, 0, p->world_size, vec);
    }
    RT_HIP(hipGetLastError());
    return RT_OK;
}

extern "C" int rt_untile_device(rt_ctx *ctx, const rt_render_params *p, const float *d_rgb_shards, float *d_rgb_frame,
                                const uint8_t *d_rgb8_shards, uint8_t *d_rgb8_frame) {
    return untile_on(ctx, ctx ? ctx->stream : nullptr, p, d_rgb_shards, d_rgb_frame, d_rgb8_shards, d_rgb8_frame);
}

extern "C" int rt_untile_device_on(rt_ctx *ctx, void *hip_stream, const rt_render_params *p, const float *d_rgb_shards,
                                   float *d_rgb_frame, const uint8_t *d_rgb8_shards, uint8_t *d_rgb8_frame) {
    return untile_on(ctx, static_cast<hipStream_t>(hip_stream), p, d_rgb_shards, d_rgb_frame, d_rgb8_shards, d_rgb8_frame);
}

extern "C" int rt_render(rt_ctx *ctx, const rt_render_params *p, float *rgb_linear, uint8_t *rgb8) {
    if (!ctx) return RT_ERR_INVALID;
    int rc = check_params(ctx, p);
    if (rc) return rc;
    if ((rc = ready_for(ctx, p->accel))) return rc;
    RT_HIP(hipSetDevice(ctx->device));
    int per_rank, total;
    rt_shard_tiles(p, &per_rank, &total);
    size_t shard_elems = (size_t)per_rank * RT_TILE_W * RT_TILE_H * 3;
    size_t frame_elems = (size_t)p->width * p->height * 3;
    float *d_tiles = nullptr, *d_frame = nullptr;
    uint8_t *d_tiles8 = nullptr, *d_frame8 = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *q : tmp) (void)hipFree(q); };
    // seed: what the device buffer starts from.  A rank that shares the frame with others (world_size > 1) starts
    // from the caller's frame, so that the tiles of the other ranks come back as they went in.
    auto alloc = [&](void **ptr, size_t bytes, const void *seed) -> bool {
        if (hipMalloc(ptr, bytes) != hipSuccess) return false;
        tmp.push_back(*ptr);
        if (seed) return hipMemcpyAsync(*ptr, seed, bytes, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
        return hipMemsetAsync(*ptr, 0, bytes, ctx->stream) == hipSuccess;
    };
    const bool shared = p->world_size > 1;
    bool ok = true;
    if (rgb_linear) ok = ok && alloc((void **)&d_tiles, shard_elems * sizeof(float), nullptr) &&
                         alloc((void **)&d_frame, frame_elems * sizeof(float), shared ? rgb_linear : nullptr);
    if (rgb8) ok = ok && alloc((void **)&d_tiles8, shard_elems, nullptr) && alloc((void **)&d_frame8, frame_elems, shared ? rgb8 : nullptr);
    if (!ok) { cleanup(); return fail(ctx, RT_ERR_HIP, "rt_render: device allocation failed"); }
    rc = rt_render_shard_device(ctx, p, d_tiles, d_tiles8);
    if (rc == RT_OK) {
        int tiles_x = (p->width + RT_TILE_W - 1) / RT_TILE_W;
        dim3 blk(256), grd;
        int vec;
        if (d_tiles) {
            untile_launch_shape((size_t)per_rank, d_tiles, d_frame, p->width, grd, vec);
            hipLaunchKernelGGL(k_untile<float>, grd, blk, 0, ctx->stream, d_tiles, d_frame, p->width, p->height, tiles_x, total, per_rank,
                               p->world_size, p->rank, 1, vec);
        }
        if (d_tiles8) {
            untile_launch_shape((size_t)per_rank, d_tiles8, d_frame8, p->width, grd, vec);
            hipLaunchKernelGGL(k_untile<uint8_t>, grd, blk, 0, ctx->stream, d_tiles8, d_frame8, p->width, p->height, tiles_x, total,
                               per_rank, p->world_size, p->rank, 1, vec);
        }
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && rgb_linear) e = hipMemcpyAsync(rgb_linear, d_frame, frame_elems * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess && rgb8) e = hipMemcpyAsync(rgb8, d_frame8, frame_elems, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, std::string("rt_render: ") + hipGetErrorString(e));
        else if (ctx->trace_timed) (void)hipEventElapsedTime(&ctx->last_trace_ms, ctx->ev0, ctx->ev1);
    }
    cleanup();
    return rc;
}

extern "C" int rt_trace_rays(rt_ctx *ctx, int64_t n, const float *o, const float *d, int32_t max_depth, int32_t accel,
                             rt_hit_record *hits, float *radiance) {
    if (!ctx || n < 0 || (n > 0 && (!o || !d))) return RT_ERR_INVALID;
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH) return fail(ctx, RT_ERR_INVALID, "max_depth out of range");
    if (accel != RT_ACCEL_LBVH && accel != RT_ACCEL_BRUTE) return fail(ctx, RT_ERR_INVALID, "bad accel");
    int rc = ready_for(ctx, accel);
    if (rc) return rc;
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    float *d_o = nullptr, *d_d = nullptr, *d_rad = nullptr;
    rt_hit_record *d_hits = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *q : tmp) (void)hipFree(q); };
    hipError_t e = hipMalloc((void **)&d_o, 3 * n * sizeof(float));
    if (e == hipSuccess) { tmp.push_back(d_o); e = hipMalloc((void **)&d_d, 3 * n * sizeof(float)); }
    if (e == hipSuccess) { tmp.push_back(d_d); if (radiance) { e = hipMalloc((void **)&d_rad, 3 * n * sizeof(float)); if (e == hipSuccess) tmp.push_back(d_rad); } }
    if (e == hipSuccess && hits) { e = hipMalloc((void **)&d_hits, n * sizeof(rt_hit_record)); if (e == hipSuccess) tmp.push_back(d_hits); }
    if (e == hipSuccess) e = hipMemcpyAsync(d_o, o, 3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, d, 3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_trace_rays, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ctx->d_scene, (long long)n,
                           d_o, d_d, max_depth, accel, d_hits, d_rad, ctx->d_counters);
        e = hipGetLastError();
    }
    if (e == hipSuccess && radiance) e = hipMemcpyAsync(radiance, d_rad, 3 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && hits) e = hipMemcpyAsync(hits, d_hits, n * sizeof(rt_hit_record), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, std::string("rt_trace_rays: ") + hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_intersect_object(rt_ctx *ctx, int32_t object, int64_t n, const float *o, const float *d, int32_t accel,
                                   rt_hit_record *hits) {
    if (!ctx || n < 0 || (n > 0 && (!o || !d || !hits))) return RT_ERR_INVALID;
    if (accel != RT_ACCEL_LBVH && accel != RT_ACCEL_BRUTE) return fail(ctx, RT_ERR_INVALID, "bad accel");
    int rc = ready_for(ctx, accel);
    if (rc) return rc;
    if (object < 0 || object >= ctx->S.n_objects) return fail(ctx, RT_ERR_INVALID, "object index out of range");
    if (n == 0) return RT_OK;
    RT_HIP(hipSetDevice(ctx->device));
    float *d_o = nullptr, *d_d = nullptr;
    rt_hit_record *d_hits = nullptr;
    std::vector<void *> tmp;
    auto cleanup = [&]() { for (void *q : tmp) (void)hipFree(q); };
    hipError_t e = hipMalloc((void **)&d_o, 3 * n * sizeof(float));
    if (e == hipSuccess) { tmp.push_back(d_o); e = hipMalloc((void **)&d_d, 3 * n * sizeof(float)); }
    if (e == hipSuccess) { tmp.push_back(d_d); e = hipMalloc((void **)&d_hits, n * sizeof(rt_hit_record)); }
    if (e == hipSuccess) { tmp.push_back(d_hits); e = hipMemcpyAsync(d_o, o, 3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream); }
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, d, 3 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_intersect_object, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, ctx->stream, ctx->d_scene, object,
                           (long long)n, d_o, d_d, accel, d_hits);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(hits, d_hits, n * sizeof(rt_hit_record), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return fail(ctx, RT_ERR_HIP, std::string("rt_intersect_object: ") + hipGetErrorString(e));
    return RT_OK;
}

extern "C" int rt_get_counters(rt_ctx *ctx, rt_counters *out) {
    if (!ctx || !out) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> all((size_t)kCounterStripes * kCounterStride);
    RT_HIP(hipMemcpy(all.data(), ctx->d_counters, all.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    unsigned long long c[kNumCounters] = {};
    for (int st = 0; st < kCounterStripes; st++)
        for (int i = 0; i < kNumCounters; i++) c[i] += all[(size_t)st * kCounterStride + i];
    std::memset(out, 0, sizeof *out);
    out->rays_primary = c[0]; out->rays_shadow = c[1]; out->rays_secondary = c[2]; out->shadow_queries = c[3];
    out->tri_tests = c[4]; out->box_tests = c[5]; out->gate_tests = c[6]; out->gate_tri_tests = c[7]; out->quadric_tests = c[8];
    out->plane_tests = c[9]; out->shade_evals = c[10]; out->tex_fetches = c[11]; out->node_steps = c[12]; out->node_lane_visits = c[13];
    float ms = 0;
    if (ctx->trace_timed && hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) == hipSuccess) ctx->last_trace_ms = ms;
    out->last_trace_ms = ctx->last_trace_ms;
    out->last_build_ms = ctx->last_build_ms;
    out->bvh_max_depth = ctx->bvh_max_depth;
    out->render_variant = ctx->last_variant;
    return RT_OK;
}

extern "C" int rt_reset_counters(rt_ctx *ctx) {
    if (!ctx) return RT_ERR_INVALID;
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipMemsetAsync(ctx->d_counters, 0, (size_t)kCounterStripes * kCounterStride * sizeof(unsigned long long), ctx->stream));
    return RT_OK;
}

extern "C" int rt_debug_copy_bvh(rt_ctx *ctx, int32_t mesh, float *nodes, int32_t *n_nodes, int32_t *order, int32_t *n_tris) {
    if (!ctx) return RT_ERR_INVALID;
    if (!ctx->uploaded) return fail(ctx, RT_ERR_STATE, "no scene uploaded");
    if (mesh < 0 || mesh >= ctx->n_meshes) return fail(ctx, RT_ERR_INVALID, "mesh index out of range");
    RT_HIP(hipSetDevice(ctx->device));
    RT_HIP(hipStreamSynchronize(ctx->stream));
    const DevMesh &m = ctx->meshes[mesh];
    int nn = ctx->built ? std::max(m.tri_count - 1, 0) : 0;
    if (n_nodes) *n_nodes = nn;
    if (n_tris) *n_tris = m.tri_count;
    if (nodes && nn) RT_HIP(hipMemcpy(nodes, ctx->d_nodes + m.node_offset, (size_t)nn * sizeof(BvhNode), hipMemcpyDeviceToHost));
    if (order && m.tri_count) RT_HIP(hipMemcpy(order, ctx->d_tri_orig + m.tri_offset, (size_t)m.tri_count * sizeof(int32_t), hipMemcpyDeviceToHost));
    return RT_OK;
}
