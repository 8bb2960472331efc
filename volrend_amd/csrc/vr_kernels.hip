// vr_kernels.hip -- gfx950 (CDNA4) PlenOctree ray-march kernels.
//
// Replaces the reference's device path: device::render_kernel
// (src/cuda/volrend.cu:78-173) with trace_ray (include/volrend/cuda/rt_core.cuh:66-196),
// query_single_from_root (include/volrend/internal/n3tree_query.hpp:13-48) and
// maybe_precalc_basis (include/volrend/internal/lumisphere.hpp:9-87) -- written
// from the algorithm, not from the CUDA text: wave64 8x8 pixel tiles, a device
// re-layout built at upload (sigma packed into the node words, padded 16-byte
// aligned SH records, a top grid + bricks lookup structure), integer digit descent for N=2
// (bit-identical to the float descent, see query_n2), march / shade phases,
// deterministic expf, explicit FP contraction policy.
//
// Built with -ffp-contract=off; see vr_device_math.h.
#include "vr_device_math.h"
#include "vr_internal.h"

namespace vr {

namespace {

constexpr int kWave = 64;
// Register budget of the fused FAST flavours, waves per SIMD (each measured; EXPERIMENTS.md):
constexpr int kSh16Waves = 5;   // 96 VGPRs (6 needs <= 80 and spills: 0.44 ms per C1 frame)
constexpr int kSh25Waves = 4;   // 128 VGPRs (5 = 96 VGPRs + 64 B of scratch with fenced shade math: 30 % slower)
constexpr int kSh9Waves = 7;    // 72 VGPRs without scratch since the lane's ray id lives in LDS and the round
                                // counters in scalar registers (round 5; C3 -1...-3 % against 6)
constexpr int kSh16Rows = 64;   // SH16 items per shade round (56 fits 24 waves per CU into the LDS, but measured
                                // slower: 0.275 against 0.265 ms per C1 frame)
// Guard against rays that never end (upstream would spin forever): KParams.max_iter march rounds
// of a wave since its last retire / refill pass that retired a ray, default 2^22 (tuning key
// `max_iter`, for tests).

enum { BASIS_RGBA = -1, BASIS_1 = 1, BASIS_4 = 4, BASIS_9 = 9, BASIS_16 = 16, BASIS_25 = 25 };

// Kernel flavours.  FAST is the production path: N == 2 integer descent, SH/RGBA
// only, no instrumentation, zero scratch.  FULL adds the SG/ASG lobe code and
// the optional access counters (VrFrame.counters); GENERIC additionally swaps in
// the literal float descent for N != 2 (or trees deeper than 24 levels).
enum { MODE_FAST = 0, MODE_FULL = 1, MODE_GENERIC = 2 };

struct RayCounters {
    uint32_t samples = 0, child_reads = 0, hits = 0, early = 0, entered = 0;
};

// ---------------------------------------------------------------------------
// view-dependent basis, lumisphere.hpp:9-87 (double literals => FP64 products)
// ---------------------------------------------------------------------------
// LOBES=false compiles the SH branch only (the hot configuration keeps zero
// scratch); LOBES=true adds the SG / ASG lobes read from tree.extra.
// BD > 0: the basis size is known at compile time (the render kernel's refill: SH only).
template <int FMA, bool LOBES, int BD = 0>
__device__ __forceinline__ void precalc_basis(const KParams& p, const float* dir, float* out) {
    using P = Policy<FMA>;
    const int basis_dim = BD > 0 ? BD : p.basis_dim;
    // NB: every index into out[] is a compile-time constant (loops fully
    // unrolled, predicated on basis_dim) so the array stays in VGPRs.
    if (LOBES && p.format == VR_FORMAT_ASG) {  // lumisphere.hpp:14-29
#pragma unroll
        for (int i = 0; i < VR_MAX_BASIS; ++i) {
            if (i < basis_dim) {
                const float* ptr = p.extra + i * 11;
                const float S = dot3<FMA>(dir, ptr + 8);
                const float dot_x = dot3<FMA>(dir, ptr + 2);
                const float dot_y = dot3<FMA>(dir, ptr + 5);
                const float arg = P::msub(-ptr[0] * dot_x, dot_x, ptr[1] * dot_y * dot_y);
                out[i] = S * vr_expf(arg) / (float)basis_dim;
            }
        }
    } else if (LOBES && p.format == VR_FORMAT_SG) {  // lumisphere.hpp:30-37
#pragma unroll
        for (int i = 0; i < VR_MAX_BASIS; ++i) {
            if (i < basis_dim) {
                const float* ptr = p.extra + i * 4;
                out[i] = vr_expf(ptr[0] * (dot3<FMA>(dir, ptr + 1) - 1.f)) / (float)basis_dim;
            }
        }
    } else if (BD > 0 || p.format == VR_FORMAT_SH) {  // lumisphere.hpp:38-81
        out[0] = (float)0.28209479177387814;
        const float x = dir[0], y = dir[1], z = dir[2];
        const float xx = x * x, yy = y * y, zz = z * z;
        const float xy = x * y, yz = y * z, xz = x * z;
        if (basis_dim == 25) {
            out[16] = (float)(2.5033429417967046 * (double)xy * (double)(xx - yy));
            out[17] = (float)(-1.7701307697799304 * (double)yz * (double)P::msub(3.f, xx, yy));
            out[18] = (float)(0.9461746957575601 * (double)xy * (double)P::msub(7.f, zz, 1.f));
            out[19] = (float)(-0.6690465435572892 * (double)yz * (double)P::msub(7.f, zz, 3.f));
            out[20] = (float)(0.10578554691520431 *
                              (double)P::madd(zz, P::msub(35.f, zz, 30.f), 3.f));
            out[21] = (float)(-0.6690465435572892 * (double)xz * (double)P::msub(7.f, zz, 3.f));
            out[22] =
                (float)(0.47308734787878004 * (double)(xx - yy) * (double)P::msub(7.f, zz, 1.f));
            out[23] = (float)(-1.7701307697799304 * (double)xz * (double)P::nmadd(3.f, yy, xx));
            const float a = P::nmadd(3.f, yy, xx);
            const float b = P::msub(3.f, xx, yy);
            out[24] = (float)(0.6258357354491761 * (double)P::msub(xx, a, yy * b));
        }
        if (basis_dim == 25 || basis_dim == 16) {
            out[9] = (float)(-0.5900435899266435 * (double)y * (double)P::msub(3.f, xx, yy));
            out[10] = (float)(2.890611442640554 * (double)xy * (double)z);
            out[11] = (float)(-0.4570457994644658 * (double)y * (double)(P::msub(4.f, zz, xx) - yy));
            out[12] = (float)(0.3731763325901154 * (double)z *
                              (double)P::nmadd(3.f, yy, P::msub(2.f, zz, 3.f * xx)));
            out[13] = (float)(-0.4570457994644658 * (double)x * (double)(P::msub(4.f, zz, xx) - yy));
            out[14] = (float)(1.445305721320277 * (double)z * (double)(xx - yy));
            out[15] = (float)(-0.5900435899266435 * (double)x * (double)P::nmadd(3.f, yy, xx));
        }
        if (basis_dim == 25 || basis_dim == 16 || basis_dim == 9) {
            out[4] = (float)(1.0925484305920792 * (double)xy);
            out[5] = (float)(-1.0925484305920792 * (double)yz);
            out[6] = (float)(0.31539156525252005 *
                             (P::dmsub(2.0, (double)zz, (double)xx) - (double)yy));
            out[7] = (float)(-1.0925484305920792 * (double)xz);
            out[8] = (float)(0.5462742152960396 * (double)(xx - yy));
        }
        if (basis_dim == 25 || basis_dim == 16 || basis_dim == 9 || basis_dim == 4) {
            out[1] = (float)(-0.4886025119029199 * (double)y);
            out[2] = (float)(0.4886025119029199 * (double)z);
            out[3] = (float)(-0.4886025119029199 * (double)x);
        }
    }
}

// ---------------------------------------------------------------------------
// Device layout (built once at upload by the relayout kernels below; the
// tree.npz format and the reference's flat child_/data_ arrays are the INPUT):
//
//   Nodes are RENUMBERED depth-first (pre-order) at upload: a subtree is one contiguous
//   run of the arrays, so the rays of a screen tile -- which walk through one compact
//   region of space -- touch few cache lines / DRAM pages (the file's numbering is
//   whatever the exporter produced, typically breadth-first).
//   nodes[capacity*N3]   one 32-bit word per child slot
//        bit31 = 0 : internal -- ABSOLUTE index of the child node (> 0)
//        bit31 = 1 : leaf     -- low 16 bits = sigma as IEEE fp16
//     so the descent's last load already delivers sigma: an empty-space
//     sample never touches the (GB-sized) coefficient array.
//   leaves[capacity*N3*stride] the data_dim-1 colour coefficients of each slot,
//     fp16, zero padded to `stride` bytes (16-byte aligned; 128 B = one cache
//     line for SH16) so a record is read with a few aligned 16-byte loads.
//   Lookup structure (N == 2), built from nodes[] at upload.  A sample resolves its leaf with
//   ONE load when it stays inside the top cell of the previous sample, and without a loop for
//   trees of up to G0 + BL levels (lego-class trees: 9):
//   top[8^G0]  uint2     one entry per cell of the 2^G0-per-axis grid (default G0 = 6: 2 MB)
//        .x bit31 = 1 : the cell lies inside ONE leaf of depth d <= G0 (extent 2^-d):
//                       .x = leaf | d << 16 | sigma(fp16),  .y = leaf id (slot index)
//        .x bit31 = 0 : the cell is an internal node of level G0 with a brick:
//                       .x = brick index,  .y = that node's index
//   bricks[n_bricks * 8^BL]  u32   (default BL = 3: 512 entries = 2 KB per brick) entry per
//        cell of the 2^BL-per-axis subdivision of a top cell:
//        bit31 = 1 : inside one leaf of depth d = G0 + 1 + drel:
//                    leaf | drel << 29 | delta << 19 | slot << 16 | sigma(fp16), where the leaf
//                    is child `slot` of node root + delta (the brick root's descendants of the
//                    next BL - 1 levels are numbered right behind it: delta <= 8 + 64 + 512)
//        bit31 = 0 : an internal node of level G0 + BL: its index; the walk continues there
//                    with one child-word load per level.
//        Entry order inside a brick: x-major (index = x << 2 BL | y << BL | z: a 128-byte line is a
//        1 x 4 x 8 slab of an 8^3 brick), or -- KParams.brick_blocked, BL == 3 only, chosen per tree
//        at upload -- [x2 y2 z2 z1 | x1 x0 y1 y0 z0]: a line is a 4 x 4 x 2 block, which a ray
//        crosses 3.5 instead of 4.8 of per brick.  The blocked order costs six more vector
//        instructions per brick lookup, so it is for trees whose lookups are fabric traffic: C3
//        (286 MB of lookup structure) -15 % L2<->fabric bytes, -3 % frame time; C1 / C2 (92 MB)
//        +1 % / +4 % (profiles/r05_experiments.jsonl, r05d).
// ---------------------------------------------------------------------------
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr int kMaxBrickLevels = 4;   // delta field (10 bits): 8 + 64 + 512 nodes below a brick root

// Distinct-line meter of the instrumented flavours (SURVEY.md 8(d) "B_unique"): marks the 128-byte
// line(s) an access of `bytes` bytes at byte offset `off` of array `which` touches.
enum { TOUCH_LEAVES = 0, TOUCH_NODES = 1, TOUCH_TOP = 2, TOUCH_BRICKS = 3 };
__device__ __forceinline__ void touch(const KParams& p, int which, uint64_t off, uint32_t bytes) {
    uint32_t* bm = p.touch[which];
    if (!bm) return;
    const int sh = which == TOUCH_LEAVES ? kTouchLeafShift : 7;
    const uint64_t l0 = off >> sh, l1 = (off + bytes - 1) >> sh;
    atomicOr(&bm[l0 >> 5], 1u << (l0 & 31u));
    if (l1 != l0) atomicOr(&bm[l1 >> 5], 1u << (l1 & 31u));
}

// octree point query, n3tree_query.hpp:13-48 -- literal float descent (any N).
// xyz is rewritten to leaf-local coordinates; returns the leaf slot index.
template <int FMA, bool COUNT = false>
__device__ __forceinline__ int64_t query_generic(const KParams& p, float* xyz, float* cube_sz,
                                                 int* levels, uint32_t* word) {
    using P = Policy<FMA>;
    const float fN = (float)p.N;
    const float hi = 1.f - 1e-6f;
#pragma unroll
    for (int i = 0; i < 3; ++i) xyz[i] = vmax(vmin(xyz[i], hi), 0.f);
    int64_t node = 0;
    *cube_sz = fN;
    int64_t sub_ptr = 0;
    uint32_t w = kLeafBit;
    int l = 0;
    for (; l < 64; ++l) {
        float index = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xyz[i] *= fN;
            const float k = __builtin_floorf(xyz[i]);
            index = P::madd(index, fN, k);
            xyz[i] -= k;
        }
        sub_ptr = node * p.N3 + (int32_t)index;
        w = p.nodes[sub_ptr];
        if (COUNT) touch(p, TOUCH_NODES, (uint64_t)sub_ptr * 4u, 4u);
        if (w & kLeafBit) break;
        *cube_sz *= fN;
        node = (int64_t)w;
    }
    *levels = l + 1;
    *word = w;
    return sub_ptr;
}

// Per-lane traversal cache for the N == 2 lookup: the top cell of the previous sample and
// its entry.
struct Cursor {
    uint32_t cell = 0xFFFFFFFFu;  // top cell index (no sample yet: matches nothing)
    uint32_t e0 = 0, e1 = 0;      // top[cell]
};

// N == 2: the float recurrence {x*=2; k=floor(x); x-=k} is exact in binary32, so
// the level-l digit is bit (23-l) of floor(x * 2^24) and the leaf-local
// coordinate is fract(x * 2^d) for a leaf of depth d -- same leaf, same bits, no float
// chain, and the digits of several levels index a table at once.
// Valid while the deepest leaf has d <= 24 (checked at upload).
// Returns the leaf id; *depth = d (child words the reference reads = d), *word low 16 bits = sigma.
// BLK: entry order of the bricks -- 0 x-major, 1 blocked (both compile-time: the production flavours
// exist once per order, a launch-uniform branch in the march round costs C1 1.5 %), -1 = as
// KParams.brick_blocked says (the instrumented flavours).
template <bool COUNT = false, int BLK = -1>
__device__ __forceinline__ uint32_t query_n2(const KParams& p, float* xyz, int* depth,
                                             uint32_t* word, Cursor& cur) {
    // clamp to [0, 1 - 1e-6] (n3tree_query.hpp:17-19) as ONE v_med3_f32 per axis: identical to
    // max(min(x, hi), 0) for every non-NaN x (-0 -> +0 included).  Deviation, non-finite poses
    // only: the reference's max(min(NaN, hi), 0) is `hi` (fminf / fmaxf drop the NaN), the
    // median of (NaN, 0, hi) is 0 -- such a ray samples the other corner of the volume
    const float hi = 1.f - 1e-6f;
    xyz[0] = __builtin_amdgcn_fmed3f(xyz[0], 0.f, hi);
    xyz[1] = __builtin_amdgcn_fmed3f(xyz[1], 0.f, hi);
    xyz[2] = __builtin_amdgcn_fmed3f(xyz[2], 0.f, hi);
    const uint32_t ux = (uint32_t)(xyz[0] * 16777216.f);
    const uint32_t uy = (uint32_t)(xyz[1] * 16777216.f);
    const uint32_t uz = (uint32_t)(xyz[2] * 16777216.f);
    const uint32_t g0 = (uint32_t)p.top_levels, sh0 = 24u - g0;
    const uint32_t cell = ((((ux >> sh0) << g0) | (uy >> sh0)) << g0) | (uz >> sh0);
    if (cell != cur.cell) {
        // 32-bit byte offsets from a uniform base (top: <= 128 MB; bricks: < 4 GB, upload)
        const uint2 e = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(p.top) + (cell << 3));
        if (COUNT) touch(p, TOUCH_TOP, (uint64_t)cell * 8u, 8u);
        cur.cell = cell;
        cur.e0 = e.x;
        cur.e1 = e.y;
    }
    uint32_t w = cur.e0, id = cur.e1;
    int d;
    if (w & kLeafBit) {
        d = (int)__builtin_amdgcn_ubfe(w, 16u, 5u);
    } else {
        const uint32_t bl = (uint32_t)p.brick_levels, sh1 = sh0 - bl;
        uint32_t sub;
        if (BLK > 0 || (BLK < 0 && p.brick_blocked)) {  // (compile-time, or launch-uniform)
            // 8^3 bricks in entry order [x2 y2 z2 z1 | x1 x0 y1 y0 z0]: a 128-byte line holds a
            // 4 x 4 x 2 block of entries instead of a 1 x 4 x 8 slab -- a ray crosses 3.5 lines of
            // a brick instead of 4.8 (chosen per tree at upload: layout comment at the top)
            const uint32_t lo = (((__builtin_amdgcn_ubfe(ux, sh1, 2u) << 2) |
                                  __builtin_amdgcn_ubfe(uy, sh1, 2u)) << 1) |
                                __builtin_amdgcn_ubfe(uz, sh1, 1u);
            const uint32_t hi = (((__builtin_amdgcn_ubfe(ux, sh1 + 2u, 1u) << 1) |
                                  __builtin_amdgcn_ubfe(uy, sh1 + 2u, 1u)) << 2) |
                                __builtin_amdgcn_ubfe(uz, sh1 + 1u, 2u);
            sub = (hi << 5) | lo;
        } else {
            sub = (((__builtin_amdgcn_ubfe(ux, sh1, bl) << bl) | __builtin_amdgcn_ubfe(uy, sh1, bl)) << bl) |
                  __builtin_amdgcn_ubfe(uz, sh1, bl);
        }
        const uint32_t entry = (w << (3u * bl)) + sub;
        if (COUNT) touch(p, TOUCH_BRICKS, (uint64_t)entry * 4u, 4u);
        w = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p.bricks) + (entry << 2));
        if (w & kLeafBit) {
            d = (int)(g0 + 1u + __builtin_amdgcn_ubfe(w, 29u, 2u));
            id = (id << 3) + __builtin_amdgcn_ubfe(w, 16u, 13u);  // (root + delta) * 8 + slot
        } else {
            // deeper than the brick: one child word per level (32-bit byte offsets: the node
            // array is < 4 GB, checked at upload)
            const char* nodes_base = reinterpret_cast<const char*>(p.nodes);
            uint32_t node = w, slot;
            int l = (int)(g0 + bl);
            for (;; ++l) {
                const uint32_t sh = (uint32_t)(23 - l);
                slot = (__builtin_amdgcn_ubfe(ux, sh, 1u) << 2) |
                       (__builtin_amdgcn_ubfe(uy, sh, 1u) << 1) | __builtin_amdgcn_ubfe(uz, sh, 1u);
                w = *reinterpret_cast<const uint32_t*>(nodes_base + (node * 8u + slot) * 4u);
                if (COUNT) touch(p, TOUCH_NODES, (uint64_t)(node * 8u + slot) * 4u, 4u);
                if ((w & kLeafBit) || l >= 23) break;
                node = w;
            }
            d = l + 1;
            id = node * 8u + slot;
        }
    }
    *depth = d;
    *word = w;
    const float cs = u2f((uint32_t)(127 + d) << 23);  // 2^d
    xyz[0] = __builtin_amdgcn_fractf(xyz[0] * cs);
    xyz[1] = __builtin_amdgcn_fractf(xyz[1] * cs);
    xyz[2] = __builtin_amdgcn_fractf(xyz[2] * cs);
    return id;
}

// rt_core.cuh:37-49
template <int FMA>
__device__ __forceinline__ float dda_unit(const float* cen, const float* invdir) {
    using P = Policy<FMA>;
    float tmax = 1e4f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t1 = -cen[i] * invdir[i];
        const float t2 = FMA ? P::madd(-cen[i], invdir[i], invdir[i]) : (t1 + invdir[i]);
        tmax = vmin(tmax, vmax(t1, t2));
    }
    return tmax;
}

// ---------------------------------------------------------------------------
// One leaf record in registers.  NV = 16-byte vectors per record for the
// compile-time basis sizes; BASIS_1 / RGBA use narrower loads.
// ---------------------------------------------------------------------------
template <int BASIS>
struct RecTraits {
    static constexpr int kHalfs = BASIS > 1 ? 3 * BASIS : 3;
    static constexpr int kDwords = BASIS > 1 ? ((kHalfs * 2 + 15) / 16) * 4 : 2;
};

template <int BASIS>
struct Record {
    uint32_t w[RecTraits<BASIS>::kDwords];
    template <int E>  // the packed word that holds coefficient E
    __device__ __forceinline__ uint32_t word() const {
        return w[E >> 1];
    }
    // coefficient i (compile-time) as fp32
    __device__ __forceinline__ float at(int i) const {
        const uint32_t d = w[i >> 1];
        return h2f((uint16_t)((i & 1) ? (d >> 16) : (d & 0xFFFFu)));
    }
};

template <int BASIS>
__device__ __forceinline__ void load_record(const KParams& p, uint32_t leaf, Record<BASIS>& r) {
    const uint16_t* base = p.leaves + (uint64_t)leaf * (uint32_t)p.leaf_stride_h;
    if (BASIS == BASIS_RGBA) {
        const uint2 v = *reinterpret_cast<const uint2*>(base);  // stride >= 8 B
        r.w[0] = v.x;
        r.w[1] = v.y;
    } else if (BASIS == BASIS_1) {
        // runtime channel stride (basis_dim is not one of 4/9/16/25): only the
        // first coefficient of each channel is used, rt_core.cuh:131
        const uint32_t c0 = base[0], c1 = base[p.basis_dim], c2 = base[2 * p.basis_dim];
        r.w[0] = c0 | (c1 << 16);
        r.w[1] = c2;
    } else {
        const uint4* v = reinterpret_cast<const uint4*>(base);
#pragma unroll
        for (int j = 0; j < RecTraits<BASIS>::kDwords / 4; ++j) {
            const uint4 q = v[j];
            r.w[4 * j + 0] = q.x;
            r.w[4 * j + 1] = q.y;
            r.w[4 * j + 2] = q.z;
            r.w[4 * j + 3] = q.w;
        }
    }
}

// SH / SG colour of channel c: rt_core.cuh:125-165.  Group order 25 -> 16 -> 9
// -> 4, each group summed left to right, then added to tmp.  Coefficient e of the record is
// half (e & 1) of word e >> 1; products read it in place (mul_half / fma_half).
template <int E, typename SRC>
__device__ __forceinline__ float coef_mul(float b, const SRC& r) {
    return mul_half<E & 1>(b, r.template word<E>());
}
template <int FMA, int E, typename SRC>  // Policy<FMA>::madd(b, coefficient E, c)
__device__ __forceinline__ float coef_madd(float b, const SRC& r, float c) {
    if (FMA) return fma_half<E & 1>(b, r.template word<E>(), c);
    return mul_add_half<E & 1>(b, r.template word<E>(), c);
}
// g = b[LO]*v[LO] (+) b[LO+1]*v[LO+1] (+) ... (+) b[HI]*v[HI], coefficients at offset O
template <int FMA, int O, int LO, int HI>
struct DotGroup {
    template <int I, typename SRC>
    static __device__ __forceinline__ float step(const float* b, const SRC& r, float g) {
        if constexpr (I > HI) {
            return g;
        } else {
            return step<I + 1>(b, r, coef_madd<FMA, O + I>(b[I], r, g));
        }
    }
    template <typename SRC>
    static __device__ __forceinline__ float run(const float* b, const SRC& r) {
        const float first = coef_madd<FMA, O + LO>(b[LO], r, coef_mul<O + LO + 1>(b[LO + 1], r));
        return step<LO + 2>(b, r, first);
    }
};

// SRC = Record<BASIS> (whole record in registers) or GroupWin (words of a staged record)
template <int FMA, int BASIS, int C, typename SRC>
__device__ __forceinline__ float channel_dot(const float* basis_fn, const SRC& r) {
    static_assert(BASIS > 1, "SH / SG / ASG sizes only");
    constexpr int O = C * BASIS;
    float tmp = coef_mul<O>(basis_fn[0], r);
    if constexpr (BASIS == 25) tmp += DotGroup<FMA, O, 16, 24>::run(basis_fn, r);
    if constexpr (BASIS >= 16) tmp += DotGroup<FMA, O, 9, 15>::run(basis_fn, r);
    if constexpr (BASIS >= 9) tmp += DotGroup<FMA, O, 4, 8>::run(basis_fn, r);
    if constexpr (BASIS >= 4) tmp += DotGroup<FMA, O, 1, 3>::run(basis_fn, r);
    return tmp;
}

// The words of a staged record (LDS row) that hold coefficients LO..HI of channel C.
template <int BASIS, int C, int LO, int HI>
struct GroupWin {
    static constexpr int kW0 = (C * BASIS + LO) / 2, kW1 = (C * BASIS + HI) / 2;
    uint32_t w[kW1 - kW0 + 1];
    __device__ __forceinline__ void load(const char* row) {
        const uint32_t* r32 = reinterpret_cast<const uint32_t*>(row);
#pragma unroll
        for (int j = 0; j <= kW1 - kW0; ++j) w[j] = r32[kW0 + j];
    }
    template <int E>
    __device__ __forceinline__ uint32_t word() const {
        return w[(E >> 1) - kW0];
    }
};

// rt_core.cuh:131-160 for the three channels of one staged record, group by group: the group
// sums are independent subexpressions of `tmp`, so each group's basis values are fetched
// (get(i) = basis_fn[i] of the ray that owns the item) right before the three channels use
// them and are dead afterwards -- the same operations in the same association as
// channel_dot, with ~12 fewer live registers than gathering the whole basis up front.
template <int FMA, int BASIS, int LO, int HI, bool FENCE, typename GET>
__device__ __forceinline__ void add_group(const char* row, GET&& get, float* acc) {
    float b[VR_MAX_BASIS];
    // FENCE keeps the scheduler from hoisting the next group's fetches over this group's
    // arithmetic (lowest register use, but every group then waits for its own LDS round trip: for
    // flavours on a tighter register budget than they like -- none of the production ones)
    if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = LO; i <= HI; ++i) b[i] = get(i);
    {
        GroupWin<BASIS, 0, LO, HI> w;
        w.load(row);
        acc[0] += DotGroup<FMA, 0 * BASIS, LO, HI>::run(b, w);
    }
    {
        GroupWin<BASIS, 1, LO, HI> w;
        w.load(row);
        acc[1] += DotGroup<FMA, 1 * BASIS, LO, HI>::run(b, w);
    }
    {
        GroupWin<BASIS, 2, LO, HI> w;
        w.load(row);
        acc[2] += DotGroup<FMA, 2 * BASIS, LO, HI>::run(b, w);
    }
}

template <int FMA, int BASIS, bool FENCE = false, typename GET>
__device__ __forceinline__ void channel_sums(const char* row, GET&& get, float* acc) {
    static_assert(BASIS > 1, "SH / SG / ASG sizes only");
    {
        const float b0 = get(0);
        GroupWin<BASIS, 0, 0, 0> w0;
        GroupWin<BASIS, 1, 0, 0> w1;
        GroupWin<BASIS, 2, 0, 0> w2;
        w0.load(row);
        w1.load(row);
        w2.load(row);
        acc[0] = coef_mul<0 * BASIS>(b0, w0);
        acc[1] = coef_mul<1 * BASIS>(b0, w1);
        acc[2] = coef_mul<2 * BASIS>(b0, w2);
    }
    if constexpr (BASIS == 25) add_group<FMA, BASIS, 16, 24, FENCE>(row, get, acc);
    if constexpr (BASIS >= 16) add_group<FMA, BASIS, 9, 15, FENCE>(row, get, acc);
    if constexpr (BASIS >= 9) add_group<FMA, BASIS, 4, 8, FENCE>(row, get, acc);
    if constexpr (BASIS >= 4) add_group<FMA, BASIS, 1, 3, FENCE>(row, get, acc);
}

__device__ __forceinline__ uint32_t quant8(float v) {
    // float -> uint8 the way a host build of the reference converts
    // (truncate to int32, keep the low byte); volrend.cu:166
    const float s = v * 255.f;
    if (s != s) return 0u;
    if (s >= 2147483648.f || s < -2147483648.f) return 0u;
    return (uint32_t)(int32_t)s & 0xFFu;
}

typedef __attribute__((address_space(1))) uint32_t vr_gword_t;   // a dword of a frame buffer
typedef float vr_f4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) vr_f4_t vr_gfloat4_t;

__device__ __forceinline__ bool wave_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0ull; }
// The lane's id, recomputed where it is asked for (two instructions): for addresses that are needed
// once in a while -- a register that holds `lane * 4` across the march loop is one the hot path lacks.
__device__ __forceinline__ uint32_t lane_id_now() {
    uint32_t l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// ---------------------------------------------------------------------------
// render_kernel: render_kernel + trace_ray of the reference
// (volrend.cu:78-173, rt_core.cuh:66-196) as a PERSISTENT wave64 kernel.
//
//   * The launch covers one or more frames of the same size (a batch of poses).  Ray
//     generation is a kernel of its own (raygen_kernel below: one lane per pixel at full
//     occupancy, FP64-heavy); the rays that enter the volume sit compacted in the ray buffer,
//     ray id -> (frame, 8x8 pixel block, pixel) by locate().
//   * A fixed number of waves (one 64-thread workgroup each) stays resident.  A wave owns a
//     chunk of consecutive ray ids (one atomic add on a queue head per chunk); whenever
//     >= refill_min lanes are idle, the k-th idle lane loads ray chunk_next + k from the
//     buffer.  Terminated rays are replaced in place -- live rays never move between lanes.
//   * The colour of a sample never feeds back into the march (only the
//     attenuation does), so colour evaluation is decoupled from the ray that
//     produced it.  The wave alternates two phases, each with most lanes busy:
//       march : descent + sigma test + attenuation / light update / stop test;
//               samples with sigma > sigma_thresh append a (leaf, weight, owner)
//               item to a wave-level LDS ring (ballot + mbcnt compaction);
//       shade : as soon as 64 items wait, every lane takes ONE item -- whoever
//               owns it -- the SH records arrive by LDS-DMA, the owner's basis through
//               ds_bpermute; then each owner adds the results
//               of its own items, oldest first (the reference's order per ray).
//   * Finished rays composite over the background, quantise and store their
//     pixel -- retired and refilled in batches of >= refill_min lanes, one memory round
//     trip per batch.
// Per-ray arithmetic and its order are exactly the reference's.
// ---------------------------------------------------------------------------
struct Ray {
    float cen[3], dir[3], invdir[3];
    float t, tmax, delta_scale;
    float light;
    float out[4];
    bool active;      // lane holds an unfinished ray
    bool alive;       // still inside `while (t < tmax)`
    bool entered;     // passed the ray/box test of rt_core.cuh:88
    bool stopped;     // ended by stop_thresh (renormalised in finish_ray)
};

struct PixelRef {
    int32_t frame, x, y, k, lx, ly;
    bool in_image;
};

__device__ __forceinline__ PixelRef locate(const KParams& p, uint32_t id) {
    PixelRef r;
    // Ray-id order (scheduling only; consecutive ids are generated, queued and marched together):
    // the frames of the launch are taken in groups of p.frame_group consecutive poses; within a
    // group the 8x8 pixel block is the major index and the FRAME the minor one, so the same
    // block of neighbouring poses -- rays that walk through nearly the same leaves -- sits in
    // consecutive ids; the blocks of a tile are visited super-block by super-block
    // (p.super_block x p.super_block blocks, row-major inside), so that a wave's chunk of ids
    // covers a compact screen region instead of a thin strip.
    const uint32_t blk = id >> 6;
    const int32_t lane = (int32_t)(id & 63u);
    const uint32_t G = (uint32_t)p.frame_group, nwb = (uint32_t)p.n_wave_blocks;
    const uint32_t grp = blk / (nwb * G), rem = blk - grp * nwb * G;
    const uint32_t left = (uint32_t)p.n_frames - grp * G, gsz = left < G ? left : G;
    const int32_t wb = (int32_t)(rem / gsz);
    r.frame = (int32_t)(grp * G + (rem - (uint32_t)wb * gsz));
    // wave block -> local tile -> frame tile -> pixel
    r.k = wb / p.wblocks_per_tile;
    const int32_t sub = wb - r.k * p.wblocks_per_tile;
    const int32_t tile = r.k * p.world + p.rank;
    const int32_t ty = tile / p.tiles_x, tx = tile - ty * p.tiles_x;
    int32_t sx, sy;
    const int32_t nbx = p.wblocks_per_tile_x;
    if (p.super_block <= 1) {
        sy = sub / nbx;
        sx = sub - sy * nbx;
    } else {  // ragged edges: the last super-row / super-column is simply shorter
        const int32_t S = p.super_block, nby = p.wblocks_per_tile / nbx;
        const int32_t per_sr = S * nbx, full_sr = nby / S;
        int32_t sr = sub / per_sr, in_sr = sub - sr * per_sr, rows = S;
        if (sr >= full_sr) {
            sr = full_sr;
            in_sr = sub - full_sr * per_sr;
            rows = nby - full_sr * S;
        }
        const int32_t per_sc = rows * S, full_sc = nbx / S;
        int32_t sc = in_sr / per_sc, in_sc = in_sr - sc * per_sc, cols = S;
        if (sc >= full_sc) {
            sc = full_sc;
            in_sc = in_sr - full_sc * per_sc;
            cols = nbx - full_sc * S;
        }
        const int32_t iy = in_sc / cols;
        sx = sc * S + (in_sc - iy * cols);
        sy = sr * S + iy;
    }
    r.lx = sx * 8 + (lane & 7);
    r.ly = sy * 8 + (lane >> 3);
    r.x = tx * p.tile_w + r.lx;
    r.y = ty * p.tile_h + r.ly;
    r.in_image = r.x < p.width && r.y < p.height;
    return r;
}

__device__ __forceinline__ uint8_t* pixel_ptr(const KParams& p, const FrameDesc& fd,
                                              const PixelRef& r) {
    if (p.layout == VR_LAYOUT_COMPACT)
        return fd.rgba + ((int64_t)r.k * p.tile_w * p.tile_h + (int64_t)r.ly * p.tile_w + r.lx) * 4;
    return fd.rgba + (int64_t)r.y * p.pitch + (int64_t)r.x * 4;
}

// Ray generation + trace_ray prologue up to the ray/box test
// (volrend.cu:135-148, rt_core.cuh:74-92).  vdir = (rotated) view direction for the basis.
template <int FMA>
__device__ __forceinline__ void setup_ray(const KParams& p, const PixelRef& r, Ray& ray,
                                          float* vdir) {
    using P = Policy<FMA>;
    const FrameDesc& fd = p.frames[r.frame];
    ray.out[0] = ray.out[1] = ray.out[2] = ray.out[3] = 0.f;
    ray.light = 1.f;
    ray.alive = ray.entered = ray.stopped = false;
    ray.t = 0.f;
    if (p.N <= 0) return;  // enable_draw = tree.N > 0
    float dir[3], cen[3];
    // screen2worlddir, volrend.cu:22-32 (no +0.5 pixel centre offset)
    float xyz[3];
    xyz[0] = P::nmadd(0.5f, (float)p.width, (float)r.x) / p.fx;
    xyz[1] = -(P::nmadd(0.5f, (float)p.height, (float)r.y)) / p.fy;
    xyz[2] = -1.0f;
    float xf[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) xf[i] = fd.xf[i];
    mv3<FMA>(xf, xyz, dir);
    normalize3<FMA>(dir);
    cen[0] = xf[9];
    cen[1] = xf[10];
    cen[2] = xf[11];
    vdir[0] = dir[0];
    vdir[1] = dir[1];
    vdir[2] = dir[2];
    if (p.ndc_width > 0) {  // maybe_world2ndc, volrend.cu:34-54
        const float tt = -(1.f + cen[2]) / dir[2];
#pragma unroll
        for (int i = 0; i < 3; ++i) cen[i] = P::madd(tt, dir[i], cen[i]);
        dir[0] = -((2.f * p.ndc_focal) / p.ndc_width) * (dir[0] / dir[2] - cen[0] / cen[2]);
        dir[1] = -((2.f * p.ndc_focal) / p.ndc_height) * (dir[1] / dir[2] - cen[1] / cen[2]);
        dir[2] = -2.f / cen[2];
        cen[0] = -((2.f * p.ndc_focal) / p.ndc_width) * (cen[0] / cen[2]);
        cen[1] = -((2.f * p.ndc_focal) / p.ndc_height) * (cen[1] / cen[2]);
        cen[2] = 1.f + 2.f / cen[2];
        normalize3<FMA>(dir);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) cen[i] = P::madd(p.scale[i], cen[i], p.offset[i]);

    float tmax_bg = 1e9f;
    if (!p.offscreen && fd.depth) tmax_bg = fd.depth[(int64_t)r.y * p.width + r.x];

    if (p.rot_enabled) {  // rodrigues, volrend.cu:57-71 (uniform part done on host)
        float cr[3];
        cross3<FMA>(p.rot_k, vdir, cr);
        const float dot = dot3<FMA>(p.rot_k, vdir);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float a = P::madd(vdir[i], p.rot_cos, cr[i] * p.rot_sin);
            const double kd = (double)(p.rot_k[i] * dot);
            const double om = 1.0 - (double)p.rot_cos;
            vdir[i] = (float)P::dmadd(kd, om, (double)a);
        }
    }
    // _get_delta_scale, rt_core.cuh:52-63
    dir[0] *= p.scale[0];
    dir[1] *= p.scale[1];
    dir[2] *= p.scale[2];
    const float delta_scale = 1.f / norm3<FMA>(dir);
    dir[0] *= delta_scale;
    dir[1] *= delta_scale;
    dir[2] *= delta_scale;
    tmax_bg /= delta_scale;
    float invdir[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) invdir[i] = (float)(1.0 / ((double)dir[i] + 1e-9));
    // _dda_world, rt_core.cuh:18-34: the 1e-6 literals make this FP64
    float tmin = 0.0f, tmax = 1e4f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t1 = (float)((((double)p.bbox[i] + 1e-6) - (double)cen[i]) * (double)invdir[i]);
        const float t2 =
            (float)((((double)p.bbox[i + 3] - 1e-6) - (double)cen[i]) * (double)invdir[i]);
        tmin = vmax(tmin, vmin(t1, t2));
        tmax = vmin(tmax, vmax(t1, t2));
    }
    tmax = vmin(tmax, tmax_bg);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ray.cen[i] = cen[i];
        ray.dir[i] = dir[i];
        ray.invdir[i] = invdir[i];
    }
    ray.delta_scale = delta_scale;
    ray.tmax = tmax;
    if (tmax < 0 || tmin > tmax) {
        if (p.render_depth) ray.out[3] = 1.f;  // ray misses the box, rt_core.cuh:88-92
        return;
    }
    ray.entered = true;
    ray.t = tmin;
    ray.alive = tmin < tmax;
}

// End of trace_ray + the compositing tail of render_kernel (rt_core.cuh:176-194,
// volrend.cu:152-172): early-stop renormalisation / final alpha, optional debug
// outputs, composite, quantise, store.
// px = the pixel's RGBA8 in its frame buffer; xy (x | y << 16) and frame are only read by the
// optional outputs (accumulators / counters).
template <int FMA, bool COUNT>
__device__ __forceinline__ void finish_ray(const KParams& p, Ray& ray, const RayCounters& rc,
                                           uint8_t* px, uint32_t xy, int frame) {
    using P = Policy<FMA>;
    float* out = ray.out;
    // (COUNT <=> not the FAST flavour: render_depth launches never take FAST, launch_fp)
    float alpha = out[3];  // (0, or 1 for a depth-mode ray that misses the box: ray generation)
    if (ray.stopped) {  // rt_core.cuh:176-185, applied once every queued colour has landed
        if (COUNT && p.render_depth) out[0] = out[1] = out[2] = vmin(out[0] * 0.3f, 1.0f);
        const float scale = 1.f / (1.f - ray.light);
        out[0] *= scale;
        out[1] *= scale;
        out[2] *= scale;
        alpha = 1.f;
    } else if (ray.entered) {  // rt_core.cuh:189-194
        if (COUNT && p.render_depth) {
            out[0] = out[1] = out[2] = vmin(out[0] * 0.3f, 1.0f);
            alpha = 1.f;
        } else {
            alpha = 1.f - ray.light;
        }
    }
    if (COUNT && p.frames[frame].counters) {
        const FrameDesc& fd = p.frames[frame];
        // VrCounters: rays, rays_hit_box, samples, child_reads, hit_samples, alg_bytes,
        // early_stops.  alg_bytes per SURVEY.md 8(d):
        //   sum over samples (4*L + 2 + hit*2*(data_dim-1)) + 4 per pixel
        const unsigned long long bytes = 4ull * rc.child_reads + 2ull * rc.samples +
                                         2ull * (unsigned long long)(p.data_dim - 1) * rc.hits +
                                         4ull;
        if (p.N > 0) atomicAdd(&fd.counters[0], 1ull);
        atomicAdd(&fd.counters[1], ray.entered ? 1ull : 0ull);
        atomicAdd(&fd.counters[2], (unsigned long long)rc.samples);
        atomicAdd(&fd.counters[3], (unsigned long long)rc.child_reads);
        atomicAdd(&fd.counters[4], (unsigned long long)rc.hits);
        atomicAdd(&fd.counters[5], bytes);
        atomicAdd(&fd.counters[6], (unsigned long long)rc.early);
    }
    if (p.any_accum) {  // launch-uniform: the frame table is only consulted when some frame asks
        float* accum = p.frames[frame].accum;
        if (accum) {
            const int64_t pix = (int64_t)(xy >> 16) * p.width + (int64_t)(xy & 0xFFFFu);
            // (frame buffers are global memory: say so, a pointer read from a table is "flat" to the
            // compiler and would be accessed with flat_ instructions)
            ((vr_gfloat4_t*)accum)[pix] = (vr_f4_t){out[0], out[1], out[2], alpha};
        }
    }
    vr_gword_t* const gpx = (vr_gword_t*)px;
    // composite, volrend.cu:152-172
    const float nalpha = 1.f - alpha;
    if (p.offscreen) {
        out[0] = P::madd(p.background_brightness, nalpha, out[0]);
        out[1] = P::madd(p.background_brightness, nalpha, out[1]);
        out[2] = P::madd(p.background_brightness, nalpha, out[2]);
    } else {
        const uint32_t init = *gpx;
        out[0] = P::madd((float)(init & 0xFFu) / 255.f, nalpha, out[0]);
        out[1] = P::madd((float)((init >> 8) & 0xFFu) / 255.f, nalpha, out[1]);
        out[2] = P::madd((float)((init >> 16) & 0xFFu) / 255.f, nalpha, out[2]);
    }
    *gpx = quant8(out[0]) | (quant8(out[1]) << 8) | (quant8(out[2]) << 16) | 0xFF000000u;
}

// ---------------------------------------------------------------------------
// Ray buffer (global memory, written by raygen_kernel; blocked structure of arrays, see
// ray_slot), the words of a ray:
//   0-2 cen, 3-5 dir, 6-8 invdir, 9 t, 10 tmax, 11 delta_scale, 12 xy, 13-14 the 64-bit
//   address of the pixel's RGBA8 (so that retiring a ray needs no frame-table lookup),
//   15 frame, 16.. basis_fn[0..nb)
// Wave-private LDS of the march kernel (one wave per workgroup):
//   ring  : colour work items (leaf, weight, owner lane) in sample order
//   stage : the SH records of one shade round, DMA'd straight from HBM (global_load_lds)
//   res   : the three colour contributions of each item of the round (aliases the first
//           768 bytes of `stage`: every row has been consumed by then)
// The basis of a lane's ray lives in that lane's registers; the lane that shades one of its
// items reads it through the LDS crossbar (ds_bpermute).
// ---------------------------------------------------------------------------
constexpr int kRayWords = 16;
// Blocked structure of arrays: the rays are stored in blocks of 64, word k of the 64 rays of a
// block contiguous (256 bytes), the words of a block back to back.  So word k of ray r lives at
//   buf + ((r >> 6) * words_per_ray + k) * 64 + (r & 63)
// -- lanes that hold consecutive rays read / write consecutive dwords, and all the words of one
// ray hang off ONE per-lane address with compile-time offsets (k * 256 bytes: the immediate
// field of the load), so neither address arithmetic nor a base register per field is spent.
template <typename T>
__device__ __forceinline__ T* ray_slot(T* buf, int words_per_ray, uint32_t r) {
    return buf + ((size_t)(r >> 6) * (uint32_t)words_per_ray * 64u + (r & 63u));
}
__device__ __forceinline__ uint32_t ray_word(const uint32_t* slot, int k) { return slot[k * 64]; }
constexpr int kRing = 128;   // capacity; at most 127 items are ever outstanding
constexpr int kQueueStride = 16;  // words between queue heads (one 64-byte line each: head, count)
constexpr uint32_t kStealMin = 8192;  // rays a foreign queue must still hold to be worth a steal (or an eighth of its length)

// Ray queues.  The 8x8 pixel blocks of a launch (ray-id order: locate()) are cut into n_queues (1 or
// 8) contiguous runs -- screen regions of the batch -- at multiples of 16 blocks; queue x owns the ray
// slots of its blocks, [first_block(x) * 64, first_block(x + 1) * 64), and two words of one 64-byte
// line: head (rays handed out, render_kernel) and count (rays stored, raygen_kernel).  Ray generation
// compacts the rays that enter the volume to the front of their queue's region (one atomic on the
// queue's count word per workgroup: eight words share the load a single counter carried, which is
// what lets small launches generate their rays in workgroups of one or four waves, below).
__device__ __forceinline__ uint32_t queue_first_block(uint32_t n_groups16, uint32_t x, uint32_t sh) {
    return (uint32_t)(((uint64_t)n_groups16 * x) >> sh) << 4;
}

// A wave's next private range [lo, hi) of ray slots, or lo == hi when there is nothing left for it.
// A wave serves the queue of its XCD first (workgroup b runs on XCD b % 8 -- used for L2 affinity
// only, never for correctness) and steals from the others when that queue has run dry.  Chunk sizes
// shrink as a queue drains (guided self-scheduling) so the tail stays balanced.
//   * One lane walks the queues: one load per queue, and ONE returning atomic on the queue that is
//     picked.  A single word sustains ~90 accesses per microsecond chip-wide (one queue for the
//     whole chip: a one-frame launch takes 40 % longer, profiles/r03_steal_threshold.jsonl).
//   * Waves steal from a queue only while it holds a good part of its rays (an eighth, at most
//     kStealMin); the rest is left to the queue's own waves.  Stealing down to the last chunk
//     -- round 2 -- had every wave of the chip visit every queue when they ran dry, all at
//     about the same time, and scattered the last blocks of every screen region over all
//     XCDs: a one-frame launch marched at a third of its rate for 50 us
//     (profiles/r03_tail_profile.jsonl; without any stealing a 20-frame launch is 5 % slower).
//     A wave only reports "nothing left" after its OWN queue has run dry, so every queue is
//     drained by the waves it belongs to -- which a grid of fewer waves than queues does not
//     have for every queue: such a grid steals to the end.
__device__ __forceinline__ void grab_chunk(const KParams& p, int lane, uint32_t& lo, uint32_t& hi) {
    lo = hi = 0;
    if (lane == 0) {
        const uint32_t nq = (uint32_t)p.n_queues;  // 1 or 8
        const uint32_t sh = nq == 8u ? 3u : 0u;
        const uint32_t n16 = ((p.total_rays >> 6) + 15u) >> 4;
        const uint32_t mine = blockIdx.x & (nq - 1u);
        const uint32_t waves_per_q = (gridDim.x + nq - 1u) >> sh;
        for (uint32_t a = 0; a < nq; ++a) {
            const uint32_t x = (mine + a) & (nq - 1u);
            uint32_t* head = p.queue_head + x * kQueueStride;
            const uint32_t len = head[1];  // rays of this queue (written by raygen_kernel, constant here)
            const uint32_t seen = __hip_atomic_load(head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= len) continue;
            if (a != 0u && gridDim.x >= nq &&
                len - seen < ((len >> 3) < kStealMin ? (len >> 3) : kStealMin))
                continue;  // not worth a steal
            uint32_t size = (len - seen) / (2u * waves_per_q);
            size = size < 64u ? 64u : (size > (uint32_t)p.chunk_max ? (uint32_t)p.chunk_max : size);
            size &= ~63u;
            const uint32_t base = atomicAdd(head, size);
            if (base < len) {
                const uint32_t qlo = queue_first_block(n16, x, sh) << 6;
                lo = qlo + base;
                hi = qlo + (base + size < len ? base + size : len);
                break;
            }
        }
    }
}

// Record fetch of a shade round: a record of V 16-byte chunks is fetched by V adjacent lanes
// (one or two cache lines per group instead of one line per lane and chunk) with LDS-DMA loads:
// lane l of an instruction lands at base + 16*l, i.e. records sit in dense rows of V*16 bytes
// and nothing passes through registers.  All records of a round (SH25: of half a round) are in
// flight at once; one wait, then every lane reads the row of the item it shades.
template <int BASIS>
struct Stage {
    static constexpr bool kEnabled = BASIS > 1;
    static constexpr int kVec = kEnabled ? RecTraits<BASIS>::kDwords / 4 : 1;  // V: 2, 4, 6, 10
    static constexpr int kRow = kVec * 16;                            // bytes
    static constexpr int kPerInstr = kWave / kVec;                    // records per DMA instruction
    // Rows per pass: SH16 (96-byte rows) shades kSh16Rows = 64 items per round in one pass
    // (6 KB of rows), SH25 (160-byte rows) 64 items in two passes of 32, the narrower formats 64
    // items in one pass.
    static constexpr int kPass = !kEnabled ? kWave
                                 : (kRow * kWave <= 5504 ? kWave
                                    : (BASIS == BASIS_16 ? kSh16Rows : kWave / 2));  // rows per pass
    static constexpr int kPasses = (BASIS == BASIS_25) ? 2 : 1;
    static constexpr int kShade = kPass * kPasses;                    // items per shade round
    static constexpr int kInstr = (kPass + kPerInstr - 1) / kPerInstr;
    static constexpr int kBytes = (kEnabled && kPass * kRow > 768) ? kPass * kRow : 768;
};
typedef __attribute__((address_space(1))) const void* vr_gptr_t;
typedef __attribute__((address_space(3))) void* vr_lptr_t;
// Outstanding colour items per ray: four 8-bit ring positions packed in one register, the newest
// in the top byte (a push is ONE v_alignbit_b32), the oldest at bit `qsh` = 32 - 8 * count (a pop
// only moves qsh).  (Eight per ray, measured: -2 % on a lone 20-frame launch, nothing on a
// 64-frame one, for a second register and a 64-bit funnel shift per push.)
constexpr int kOwnerQ = 4;

// The record requests of one pass of a shade round (see Stage): lane l fetches 16-byte chunk
// l % V of record l / V of its instruction, straight into the stage rows.  NT = the non-temporal
// cache policy (an immediate of the instruction, hence a template parameter).
template <int BASIS, bool NT, int RING = kRing>
__device__ __forceinline__ void issue_records(const KParams& p, char* stage, const uint32_t* it_leaf,
                                              uint32_t ring_head, int lane, int n, int pass) {
    using ST = Stage<BASIS>;
#pragma unroll
    for (int k = 0; k < ST::kInstr; ++k) {
        const int rin = k * ST::kPerInstr + lane / ST::kVec;  // record within the pass
        const int item = pass * ST::kPass + rin;
        if (lane < ST::kPerInstr * ST::kVec && rin < ST::kPass && item < n) {
            const uint32_t leaf = it_leaf[(ring_head + (uint32_t)item) & (RING - 1)];
            const char* src = reinterpret_cast<const char*>(p.leaves) +
                              (uint64_t)leaf * (uint32_t)(p.leaf_stride_h * 2) + (lane % ST::kVec) * 16;
            // (the LDS address is formed in address space 3: a generic-pointer detour between two
            // casts does not fold when `stage` is not the first LDS object of the kernel)
            __builtin_amdgcn_global_load_lds(
                (vr_gptr_t)src,
                (vr_lptr_t)((__attribute__((address_space(3))) char*)stage + k * ST::kPerInstr * ST::kRow),
                16, 0, NT ? 2 /* nt */ : 0);
        }
    }
}

// Register budget of the fused FAST flavours (waves per SIMD), from their natural register use:
// SH16 96 VGPRs -> kSh16Waves = 5 (20 waves per CU; 6 needs <= 80 and spills), SH9 <= 72 ->
// kSh9Waves = 7, SH25 <= 128 -> 4 (it gathers its 25 basis values up front), the small
// records 8.  The instrumented / lobe / generic flavours keep their wider state in registers at
// 4 waves per SIMD (3 for SH25).  No render flavour uses scratch.
template <int BASIS, int MODE>
constexpr int min_waves_per_eu() {
    if (MODE != MODE_FAST) return BASIS == BASIS_25 ? 3 : 4;  // SH25 + counters needs > 128 VGPRs
    return BASIS == BASIS_25 ? kSh25Waves : BASIS == BASIS_16 ? kSh16Waves : BASIS == BASIS_9 ? kSh9Waves : 8;
}
// Waves one CU holds of a flavour: the register bound above or the LDS bound (512-byte granules).
template <int BASIS, int MODE>
constexpr int waves_per_cu() {
    const int lds = ((kRing * 9 + kWave * 4 + Stage<BASIS>::kBytes + 511) / 512) * 512;
    const int by_lds = 163840 / lds, by_reg = 4 * min_waves_per_eu<BASIS, MODE>();
    return by_lds < by_reg ? by_lds : by_reg;
}

template <int FMA, int BASIS, int MODE, bool BLK = false>
__global__ __launch_bounds__(kWave, (min_waves_per_eu<BASIS, MODE>())) void render_kernel(
    const KParams p) {
    using P = Policy<FMA>;
    constexpr bool N2 = MODE != MODE_GENERIC;
    constexpr bool LOBES = MODE != MODE_FAST;
    constexpr bool COUNT = MODE != MODE_FAST;
    constexpr int NB = BASIS > 1 ? BASIS : 1;
    constexpr bool HAS_BASIS = BASIS != BASIS_RGBA;
    using ST = Stage<BASIS>;
    __shared__ uint32_t it_leaf[kRing];
    __shared__ float it_w[kRing];
    __shared__ uint8_t it_own[kRing];
    __shared__ __attribute__((aligned(16))) char stage[ST::kBytes];
    float* const res = reinterpret_cast<float*>(stage);  // 3 x 64 floats, see above
    float mybasis[NB];  // basis_fn of this lane's ray (rt_core.cuh:96-103), read by shader lanes
#pragma unroll
    for (int i = 0; i < NB; ++i) mybasis[i] = 0.f;

    const int lane = threadIdx.x & (kWave - 1);
    Ray ray;
    ray.active = false;
    ray.alive = ray.entered = ray.stopped = false;
    // index of the lane's ray in the ray buffer: written when the lane takes the ray, read when it
    // retires it -- in LDS, not in a register that would sit idle through every march round
    __shared__ uint32_t ray_ids[kWave];
    ray.t = 0.f;
    ray.tmax = -1.f;
    ray.light = 1.f;
    ray.delta_scale = 1.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ray.cen[i] = 0.f;
        ray.dir[i] = 0.f;
        ray.invdir[i] = 1.f;
    }
    ray.out[0] = ray.out[1] = ray.out[2] = ray.out[3] = 0.f;
    RayCounters rc;
    Cursor cur;
    uint32_t qpos = 0;  // ring positions of this ray's outstanding items, see kOwnerQ
    uint32_t qsh = 32;  // 32 - 8 * (number of outstanding items)
    uint32_t rounds = 0, progress_round = 0;  // march rounds of this wave; the last one before a retire
    // wave-uniform scheduler state
    bool exhausted = false;  // the ray buffer has been handed out completely
    uint32_t chunk_next = 0, chunk_end = 0;  // this wave's private range of ray ids
    uint32_t ring_head = 0, ring_tail = 0;  // items [head, tail) are waiting for a shader lane
    const int wpr = kRayWords + p.ray_tail_words;  // words per ray in the ray buffer
    // scheduling statistics (instrumented flavours only): rounds and busy lanes per phase
    uint32_t st_march_r = 0, st_march_l = 0, st_shade_r = 0, st_shade_l = 0, st_distinct = 0,
             st_fin_r = 0, st_fin_l = 0, st_iter = 0;

    // Colour evaluation of up to 64 queued items, one per lane, whoever owns them;
    // afterwards every owner adds the contributions of its own items, oldest first
    // (= the reference's accumulation order, rt_core.cuh:161).
    auto shade_chunk = [&](int n) {
        __syncthreads();  // item pushes are visible
        if (COUNT) {
            st_shade_r++;
            st_shade_l += (uint32_t)n;
            // distinct leaves among the chunk's items (instrumentation only)
            const uint32_t myleaf =
                lane < n ? it_leaf[(ring_head + (uint32_t)lane) & (kRing - 1)] : 0xFFFFFFFFu;
            bool first = lane < n;
            for (int o = 0; o < kWave; ++o) {
                const uint32_t other = (uint32_t)__shfl((int)myleaf, o);
                if (o < lane && other == myleaf) first = false;
            }
            st_distinct += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(first));
            if (lane < n)  // the record this item reads (colour coefficients only: sigma rides in the node word)
                touch(p, TOUCH_LEAVES, (uint64_t)myleaf * (uint32_t)(p.leaf_stride_h * 2),
                      (uint32_t)(2 * (p.data_dim - 1)));
        }
        const bool have = lane < n;
        const uint32_t jmine = (ring_head + (uint32_t)lane) & (kRing - 1);
        const float weight = have ? it_w[jmine] : 0.f;
        // basis_fn[i] of the ray that owns my item, out of its lane's registers through the LDS
        // crossbar (every lane executes the permute: a bpermute only reads active lanes)
        const int own4 = (HAS_BASIS && have) ? (int)it_own[jmine] << 2 : lane << 2;
        auto basis_of = [&](int i) -> float {
            return u2f((uint32_t)__builtin_amdgcn_ds_bpermute(own4, (int)f2u(mybasis[i])));
        };
        // One-pass flavours fetch each group of basis values right where the (whole) wave uses
        // it; the two-pass flavour (SH25) computes with half the wave at a time, so it gathers
        // everything up front while every owner lane is still active.
        float bfull[ST::kPasses > 1 ? NB : 1];
        if constexpr (ST::kPasses > 1) {
#pragma unroll
            for (int i = 0; i < NB; ++i) bfull[i] = basis_of(i);
        }
        auto basis_get = [&](int i) -> float {
            if constexpr (ST::kPasses > 1) return bfull[i];
            else return basis_of(i);
        };
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if constexpr (ST::kEnabled) {
#pragma unroll
            for (int pass = 0; pass < ST::kPasses; ++pass) {
                if (pass * ST::kPass < n) {  // wave-uniform
                    // Cache policy of the record stream (launch-uniform, chosen at upload): by
                    // default the records allocate in L2 like any load -- neighbouring rays
                    // re-use a quarter of them; when the lookup structure is much larger than
                    // the L2s, the stream is marked non-temporal so that it stops evicting the
                    // top / brick lines every sample needs (C3: -11 % time; C1-class trees:
                    // +8 %, hence the switch).  The policy is an immediate of the instruction,
                    // so the issue loop exists twice.
                    if (p.records_nt)
                        issue_records<BASIS, true>(p, stage, it_leaf, ring_head, lane, n, pass);
                    else
                        issue_records<BASIS, false>(p, stage, it_leaf, ring_head, lane, n, pass);
                    __syncthreads();  // the DMAs have landed (vmcnt(0)) and are visible
                    // (one pass: ALL lanes run the arithmetic -- an owner lane without an item
                    // of its own must stay active for the permutes; only `have` lanes keep results)
                    if (ST::kPasses == 1 || lane / ST::kPass == pass) {
                        const char* row = stage + (lane % ST::kPass) * ST::kRow;
                        float acc[3];
                        channel_sums<FMA, BASIS>(row, basis_get, acc);
                        // rt_core.cuh:161: weight / (1 + expf(-tmp)) per channel
                        // (the sigmoids of channels 0 / 1 share packed mul / fma / add instructions)
                        const float2v e01 = vr_expf2((float2v){-acc[0], -acc[1]}) + splat2(1.f);
                        r0 = weight / e01.x;
                        r1 = weight / e01.y;
                        r2 = weight / (1.f + vr_expf(-acc[2]));
                    }
                    if (ST::kPasses > 1) __syncthreads();  // rows are free for the next pass
                }
            }
        } else {
          const float b0 = HAS_BASIS ? basis_of(0) : 0.f;  // (every lane: see above)
          if (have) {
            Record<BASIS> rec;
            load_record<BASIS>(p, it_leaf[jmine], rec);
            if (HAS_BASIS) {  // runtime basis size: first coefficient of each channel only
                r0 = weight / (1.f + vr_expf(-(b0 * rec.at(0))));
                r1 = weight / (1.f + vr_expf(-(b0 * rec.at(1))));
                r2 = weight / (1.f + vr_expf(-(b0 * rec.at(2))));
            } else {  // RGBA: out[c] = madd(colour, weight, out[c]) is formed by the owner
                r0 = rec.at(0);
                r1 = rec.at(1);
                r2 = rec.at(2);
            }
          }
        }
        __syncthreads();  // every row has been read: `res` may overwrite them
        if (have) {
            res[0 * kWave + lane] = r0;
            res[1 * kWave + lane] = r1;
            res[2 * kWave + lane] = r2;
        }
        __syncthreads();  // contributions are visible
        const uint32_t head8 = ring_head & 0xFFu;
#pragma unroll
        for (int d = 0; d < kOwnerQ; ++d) {
            const uint32_t pos = __builtin_amdgcn_ubfe(qpos, qsh, 8u);  // my oldest item
            const uint32_t idx = (pos - head8) & 0xFFu;               // its index within the round
            if (qsh < 32u && idx < (uint32_t)n) {
                if (HAS_BASIS) {
                    ray.out[0] += res[0 * kWave + idx];
                    ray.out[1] += res[1 * kWave + idx];
                    ray.out[2] += res[2 * kWave + idx];
                } else {
                    const float w = it_w[pos & (kRing - 1)];
                    ray.out[0] = P::madd(res[0 * kWave + idx], w, ray.out[0]);
                    ray.out[1] = P::madd(res[1 * kWave + idx], w, ray.out[1]);
                    ray.out[2] = P::madd(res[2 * kWave + idx], w, ray.out[2]);
                }
                qsh += 8u;
            }
        }
        ring_head += (uint32_t)n;
    };

    for (;;) {
        // ---- retire finished rays and hand their lanes new ones, in batches ----
        // A lane's ray is alive while t < tmax.  Nothing else says so: a ray that is cut short by
        // stop_thresh gets tmax = -1 (which finish_ray reads as "stopped"), a lane without a ray
        // has t = 0, tmax = -1.  (As loop-carried booleans the two cost the scalar unit -- shared
        // by the CU's four SIMDs -- about sixteen lane-mask copies and merges per march round.)
        const bool done = ray.active && !(ray.t < ray.tmax) && qsh == 32u;
        const unsigned long long m_done = __builtin_amdgcn_ballot_w64(done);
        const unsigned long long m_free = __builtin_amdgcn_ballot_w64(!ray.active);
        const unsigned long long m_busy =
            __builtin_amdgcn_ballot_w64(ray.active && (ray.t < ray.tmax || qsh < 32u));
        const int n_avail = __builtin_popcountll(m_done | m_free);
        if (COUNT) st_iter++;
        if (n_avail > 0 && (m_busy == 0ull || (!exhausted && n_avail >= p.refill_min))) {
            if (COUNT && m_done != 0ull) {
                st_fin_r++;
                st_fin_l += (uint32_t)__builtin_popcountll(m_done);
            }
            // The whole round costs ONE memory round trip: the pixel address of every finished
            // ray is requested here, the new rays right behind it, and the finished rays are
            // composited and stored once everything has landed (their colour state does not
            // overlap the registers the new rays load into).
            uint32_t px_lo = 0, px_hi = 0, fin_xy = 0, fin_frame = 0;
            ray.stopped = ray.tmax < 0.f;  // (read before a new ray's tmax lands in the register)
            if (done) {
                const uint32_t* rs = ray_slot(p.ray_buf, wpr, ray_ids[lane_id_now()]);
                px_lo = ray_word(rs, 13);
                px_hi = ray_word(rs, 14);
                if (COUNT || p.any_accum) {
                    fin_xy = ray_word(rs, 12);
                    fin_frame = ray_word(rs, 15);
                }
            }
            const bool vacant = done || !ray.active;
            bool take = false;
            // (a ray retires in this pass: the wave makes progress.  A finished ray that still WAITS for a
            // pass -- fewer than refill_min idle lanes -- does not count: were it to, a wave that holds one
            // finished and one endless ray after the queues ran dry would never trip the guard.  Both
            // counters are wave-uniform; saying so keeps them in scalar registers -- as vector values they
            // cost the SH16 flavour its last two)
            if (m_done != 0ull) progress_round = (uint32_t)__builtin_amdgcn_readfirstlane((int)rounds);
            // Idle lanes take consecutive rays from the buffer.  The wave owns a private
            // chunk [chunk_next, chunk_end) of ray ids and only goes to the global queue
            // head (ONE returning atomic -- a single word sustains ~90 of them per
            // microsecond chip-wide) when the chunk is used up; chunk sizes shrink as the
            // queue drains (guided self-scheduling) so the tail stays balanced.
            if (!exhausted && chunk_next >= chunk_end) {
                uint32_t lo, hi;
                grab_chunk(p, lane, lo, hi);
                lo = __builtin_amdgcn_readfirstlane(lo);
                hi = __builtin_amdgcn_readfirstlane(hi);
                if (hi == lo) {
                    exhausted = true;
                } else {
                    chunk_next = lo;
                    chunk_end = hi;
                }
            }
            if (!exhausted) {
                const unsigned long long idle = m_done | m_free;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi(
                    (uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t r = chunk_next + rank;
                const uint32_t c_end = chunk_end;
                const uint32_t left = chunk_end - chunk_next;
                chunk_next += (uint32_t)n_avail < left ? (uint32_t)n_avail : left;
                if (vacant && r < c_end) {
                    take = true;
                    const uint32_t* rs = ray_slot(p.ray_buf, wpr, r);
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        ray.cen[i] = u2f(ray_word(rs, 0 + i));
                        ray.dir[i] = u2f(ray_word(rs, 3 + i));
                        ray.invdir[i] = u2f(ray_word(rs, 6 + i));
                    }
                    ray.t = u2f(ray_word(rs, 9));
                    ray.tmax = u2f(ray_word(rs, 10));
                    ray.delta_scale = u2f(ray_word(rs, 11));
                    ray_ids[lane_id_now()] = r;
                    if (HAS_BASIS) {
                        if (BASIS > 1 && p.ray_vdir) {
                            // rt_core.cuh:96-103: the basis of the ray's view direction (SH:
                            // lumisphere.hpp:38-81), zeroed outside basis_minmax -- evaluated
                            // here, by the lane that takes the ray, from 3 words of the record
                            float vd[3];
#pragma unroll
                            for (int i = 0; i < 3; ++i) vd[i] = u2f(ray_word(rs, kRayWords + i));
                            precalc_basis<FMA, false, (BASIS > 1 ? BASIS : 1)>(p, vd, mybasis);
#pragma unroll
                            for (int i = 0; i < NB; ++i)
                                if (i < p.basis_min || i > p.basis_max) mybasis[i] = 0.f;
                        } else {
#pragma unroll
                            for (int i = 0; i < NB; ++i)
                                mybasis[i] = u2f(ray_word(rs, kRayWords + i));
                        }
                    }
                }
            }
            if (done)
                finish_ray<FMA, COUNT>(
                    p, ray, rc,
                    reinterpret_cast<uint8_t*>(((uint64_t)px_hi << 32) | (uint64_t)px_lo), fin_xy,
                    (int)fin_frame);
            if (vacant) {
                ray.active = ray.entered = take;
                if (!take) {  // (no ray: not alive)
                    ray.t = 0.f;
                    ray.tmax = -1.f;
                }
                ray.out[0] = ray.out[1] = ray.out[2] = ray.out[3] = 0.f;
                ray.light = 1.f;
                rc = RayCounters();
                cur = Cursor();
                qsh = 32u;
                qpos = 0;
            }
        }
        if (!wave_any(ray.active)) {
            if (exhausted) break;
            continue;
        }

        // ---- march: lanes with a live ray and room for another outstanding item ----
        // Guard against rays that never end (not in the reference, which would spin): when the
        // wave has marched p.max_iter rounds since its last retire / refill pass that retired a ray
        // (progress_round above: with the default of 2^22 rounds the difference to "without a
        // retired ray" is nil; a lowered max_iter trips earlier the larger refill_min is), whatever is still
        // marching is cut and reported (sticky status bit 0: the host layers fail loudly on it;
        // WHICH rays share a wave depends on the scheduling knobs, so the pixels of a launch that
        // tripped the guard are not tuning-independent -- they are wrong either way).  Wave-uniform
        // and checked once per pass through here (<= march_max rounds), so that a march round
        // carries nothing of it (five vector and four scalar instructions per round until round 3;
        // -2 % frame time, profiles/r04_*).
        if (rounds - progress_round >= (uint32_t)p.max_iter) {
            if (ray.t < ray.tmax) {
                ray.t = ray.tmax;
                if (p.status) atomicOr(p.status, 1u);
            }
            progress_round = rounds;
        }
        int m = 0;
        for (; m < p.march_max; ++m) {
            // (the wave's "anybody marching?" mask comes straight from the two compares: the
            // ballot of a combined boolean costs two more vector instructions)
            const unsigned long long m_go = __builtin_amdgcn_ballot_w64(ray.t < ray.tmax) &
                                            __builtin_amdgcn_ballot_w64(qsh != 0u);
            if (m_go == 0ull) break;
            const bool go = ray.t < ray.tmax && qsh != 0u;
            if (COUNT) {
                st_march_r++;
                st_march_l += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(go));
            }
            bool push = false;
            uint32_t leaf = 0;
            float weight = 0.f;
            if (go) {
                float pos[3];
                pos[0] = P::madd(ray.t, ray.dir[0], ray.cen[0]);
                pos[1] = P::madd(ray.t, ray.dir[1], ray.cen[1]);
                pos[2] = P::madd(ray.t, ray.dir[2], ray.cen[2]);
                float cube_sz = 0.f;
                int levels;
                uint32_t word;
                if (N2) {
                    leaf = query_n2<COUNT, (MODE == MODE_FAST ? (BLK ? 1 : 0) : -1)>(p, pos, &levels, &word, cur);
                } else {
                    leaf = (uint32_t)query_generic<FMA, COUNT>(p, pos, &cube_sz, &levels, &word);
                }
                if (COUNT) {
                    rc.samples++;
                    rc.child_reads += (uint32_t)levels;
                }
                // rt_core.cuh:116: dda / cube_sz
                const float dda = dda_unit<FMA>(pos, ray.invdir);
                // N2: cube_sz = 2^levels; x / 2^k == ldexp(x, -k), the same real number rounded once
                const float t_subcube =
                    N2 ? __builtin_amdgcn_ldexpf(dda, -levels) : dda / cube_sz;
                const float delta_t = t_subcube + p.step_size;
                const float sigma = h2f((uint16_t)(word & 0xFFFFu));
                bool stop = false;
                if (sigma > p.sigma_thresh) {
                    // rt_core.cuh:118-121,174: attenuation, weight and the light update are
                    // taken now; the colour of this sample -- which nothing else depends on --
                    // becomes a work item for the shade phase.
                    if (COUNT) rc.hits++;
                    const float att = vr_expf(-delta_t * ray.delta_scale * sigma);
                    weight = ray.light * (1.f - att);
                    if (COUNT && p.render_depth)  // (depth launches take the FULL flavour)
                        ray.out[0] = P::madd(weight, ray.t, ray.out[0]);
                    else
                        push = true;
                    ray.light *= att;
                    stop = ray.light < p.stop_thresh;
                }
                if (stop) {
                    ray.tmax = -1.f;  // stopped (and no longer alive)
                    if (COUNT) rc.early++;
                } else {
                    ray.t += delta_t;
                }
            }
            // append this step's items to the ring: k-th pushing lane -> tail + k
            const unsigned long long m_push = __builtin_amdgcn_ballot_w64(push);
            if (m_push != 0ull) {
                if (push) {
                    const uint32_t seq =
                        ring_tail + __builtin_amdgcn_mbcnt_hi(
                                        (uint32_t)(m_push >> 32),
                                        __builtin_amdgcn_mbcnt_lo((uint32_t)m_push, 0u));
                    const uint32_t j = seq & (kRing - 1);
                    it_leaf[j] = leaf;
                    it_w[j] = weight;
                    it_own[j] = (uint8_t)lane;
                    qpos = __builtin_amdgcn_alignbit(seq, qpos, 8u);  // (qpos >> 8) | seq << 24
                    qsh -= 8u;
                }
                ring_tail += (uint32_t)__builtin_popcountll(m_push);
                // (a loop: with rounds of fewer than 64 items -- kSh16Rows < 64 -- one round per
                // march step would let the ring overflow)
                while (ring_tail - ring_head >= (uint32_t)ST::kShade) shade_chunk(ST::kShade);
            }
            // Drain phase (the ray queues have run dry): a ray whose colour queue is full cannot
            // march until a shade round takes its items, and a round waits for 64 items or for the
            // moment NOBODY can march -- with few rays left in the wave the blocked ray waits for the
            // other rays to fill their queues too, i.e. the last rays of a launch take turns instead
            // of marching side by side (a one-frame launch ran 526 rounds in its longest-lived wave
            // for a longest ray of 230 samples; without colour work 279:
            // profiles/r05_tail_profile.jsonl).  So once the wave is down to drain_flush marching
            // lanes, a blocked ray gets a (partial) round at once.  (A partial round costs what a
            // full one costs: while the queues still feed the wave this would be a loss.)
            if (exhausted) {
                const unsigned long long m_alive = __builtin_amdgcn_ballot_w64(ray.t < ray.tmax);
                if (__builtin_popcountll(m_alive) <= p.drain_flush &&
                    (m_alive & __builtin_amdgcn_ballot_w64(qsh == 0u)) != 0ull && ring_tail != ring_head) {
                    const uint32_t waiting = ring_tail - ring_head;
                    shade_chunk(waiting < (uint32_t)ST::kShade ? (int)waiting : ST::kShade);
                }
            }
        }
        rounds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(rounds + (uint32_t)m));
        // nobody can march any more (queues full / rays ended): flush what is queued
        // (at most kShade - 1 + 64 items wait here: two rounds at most)
        while (ring_tail != ring_head && !wave_any(ray.t < ray.tmax && qsh > 0u)) {
            const uint32_t waiting = ring_tail - ring_head;
            shade_chunk(waiting < (uint32_t)ST::kShade ? (int)waiting : ST::kShade);
        }
    }
    if (COUNT && p.sched_stats && lane == 0) {
        atomicAdd(&p.sched_stats[0], (unsigned long long)st_march_r);
        atomicAdd(&p.sched_stats[1], (unsigned long long)st_march_l);
        atomicAdd(&p.sched_stats[2], (unsigned long long)st_shade_r);
        atomicAdd(&p.sched_stats[3], (unsigned long long)st_shade_l);
        atomicAdd(&p.sched_stats[4], (unsigned long long)st_distinct);
        atomicAdd(&p.sched_stats[5], (unsigned long long)st_fin_r);
        atomicAdd(&p.sched_stats[6], (unsigned long long)st_fin_l);
        atomicAdd(&p.sched_stats[7], (unsigned long long)st_iter);
    }
}


// ---------------------------------------------------------------------------
// raygen_kernel: one lane per pixel of every frame of the launch, at full occupancy.
// Ray generation, NDC warp, world->tree transform, view-direction rotation and the
// ray/box test (volrend.cu:135-148, rt_core.cuh:74-92) with their FP64 islands,
// plus the basis of the view direction (rt_core.cuh:96-103).  Rays that miss the
// volume are composited and stored right here; the others are appended to the
// ray buffer -- each wave compacts its survivors with a ballot / mbcnt prefix
// count and reserves their slots with ONE atomic.
// ---------------------------------------------------------------------------
// Waves (8x8 pixel blocks) per raygen workgroup: 16, 4 or 1 (launch_render picks; tuning key
// raygen_waves).  16 for batches: one atomic per 1024 pixels.  Launches of one or two frames -- the
// ones whose neighbour on another stream is still draining -- use 4: a workgroup of 16 waves needs
// four free wave slots AND 256 free vector registers on every SIMD of one CU at the same moment,
// which the previous launch's render kernel (5 waves x 96 registers per SIMD) does not offer until
// it is nearly done: the ray generation of launch k + 1 took 135 us instead of 16 beside the tail
// of launch k (kernel trace, profiles/r06_overlap_trace.jsonl), 51 with workgroups of 4 -- and the
// render kernel of launch k + 1 cannot start before it has ended.  Smaller workgroups mean more
// atomics: at 4 waves a 64-frame launch is 4 % slower, at 1 wave 35 % (profiles/r06_raygen_waves.jsonl).

template <int FMA, bool FULL, int GW>
__global__ __launch_bounds__(kWave* GW) void raygen_kernel(const KParams p) {
    __shared__ uint32_t wave_count[GW];
    __shared__ uint32_t wave_base[GW];
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const uint32_t id =
        (uint32_t)(((int64_t)blockIdx.x * GW + wave) * kWave + lane);
    bool valid = false;
    Ray nr;
    uint8_t* px = nullptr;
    uint32_t xy = 0;
    int frame = 0;
    float vdir[3] = {0.f, 0.f, 1.f};
    if (id < p.total_rays) {
        const PixelRef r = locate(p, id);
        if (r.in_image) {
            frame = r.frame;
            xy = (uint32_t)r.x | ((uint32_t)r.y << 16);
            px = pixel_ptr(p, p.frames[r.frame], r);
            setup_ray<FMA>(p, r, nr, vdir);
            if (nr.alive) {
                valid = true;
            } else {
                RayCounters z;  // a ray without a single sample
                finish_ray<FMA, FULL>(p, nr, z, px, xy, frame);
            }
        }
    }
    // Compaction: wave ballot + mbcnt prefix inside the wave, a scan over the workgroup's waves, and
    // ONE atomic per workgroup on the count word of the queue that owns the workgroup's blocks (a
    // single word only sustains ~90 returning atomics per microsecond chip-wide; workgroups never
    // straddle a queue boundary: those lie at multiples of 16 blocks).
    const unsigned long long m_valid = __builtin_amdgcn_ballot_w64(valid);
    const uint32_t nq = (uint32_t)p.n_queues, sh = nq == 8u ? 3u : 0u;
    const uint32_t n16 = ((p.total_rays >> 6) + 15u) >> 4;
    const uint32_t g16 = (uint32_t)(((int64_t)blockIdx.x * GW) >> 4);  // this workgroup's group of 16 blocks
    uint32_t qx = (uint32_t)(((uint64_t)g16 << sh) / n16);               // its queue: first guess, then exact
    while (qx + 1u < nq && (queue_first_block(n16, qx + 1u, sh) >> 4) <= g16) ++qx;
    while (qx > 0u && (queue_first_block(n16, qx, sh) >> 4) > g16) --qx;
    uint32_t* const q_count = p.queue_head + qx * kQueueStride + 1;
    const uint32_t q_base = queue_first_block(n16, qx, sh) << 6;
    uint32_t my_base;
    if constexpr (GW == 1) {
        const uint32_t n = (uint32_t)__builtin_popcountll(m_valid);
        uint32_t b = 0;
        if (lane == 0 && n) b = atomicAdd(q_count, n);
        my_base = q_base + (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
    } else {
        if (lane == 0) wave_count[wave] = (uint32_t)__builtin_popcountll(m_valid);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t sum = 0;
#pragma unroll
            for (int w = 0; w < GW; ++w) {
                wave_base[w] = sum;
                sum += wave_count[w];
            }
            const uint32_t base = q_base + (sum ? atomicAdd(q_count, sum) : 0u);
#pragma unroll
            for (int w = 0; w < GW; ++w) wave_base[w] += base;
        }
        __syncthreads();
        my_base = wave_base[wave];
    }
    if (!valid) return;
    const uint32_t slot =
        my_base + __builtin_amdgcn_mbcnt_hi((uint32_t)(m_valid >> 32),
                                            __builtin_amdgcn_mbcnt_lo((uint32_t)m_valid, 0u));
    uint32_t* rb = ray_slot(p.ray_buf_rw, kRayWords + p.ray_tail_words, slot);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        rb[(0 + i) * 64] = f2u(nr.cen[i]);
        rb[(3 + i) * 64] = f2u(nr.dir[i]);
        rb[(6 + i) * 64] = f2u(nr.invdir[i]);
    }
    rb[9 * 64] = f2u(nr.t);
    rb[10 * 64] = f2u(nr.tmax);
    rb[11 * 64] = f2u(nr.delta_scale);
    rb[12 * 64] = xy;
    rb[13 * 64] = (uint32_t)reinterpret_cast<uint64_t>(px);
    rb[14 * 64] = (uint32_t)(reinterpret_cast<uint64_t>(px) >> 32);
    rb[15 * 64] = (uint32_t)frame;
    if (p.ray_vdir) {
        // SH trees: the (rotated) view direction travels, its basis is evaluated when a lane takes
        // the ray (3 words instead of up to 25: the ray buffer is written and read once per ray)
        rb[(kRayWords + 0) * 64] = f2u(vdir[0]);
        rb[(kRayWords + 1) * 64] = f2u(vdir[1]);
        rb[(kRayWords + 2) * 64] = f2u(vdir[2]);
    } else if (p.basis_words > 0) {
        // rt_core.cuh:96-103: basis of the view direction, zeroed outside basis_minmax
        float b[VR_MAX_BASIS];
#pragma unroll
        for (int i = 0; i < VR_MAX_BASIS; ++i) b[i] = 0.f;
        precalc_basis<FMA, FULL>(p, vdir, b);
#pragma unroll
        for (int i = 0; i < VR_MAX_BASIS; ++i)
            if (i < p.basis_words)
                rb[(kRayWords + i) * 64] =
                    f2u((i < p.basis_min || i > p.basis_max) ? 0.f : b[i]);
    }
}

// Writes the per-launch frame table into device memory and resets the ray queue.
// (Stream-ordered replacement for a pinned-memory H2D copy + memset.)
__global__ void prepare_launch_kernel(FrameTable tbl, FrameDesc* frames, uint32_t* queue_head) {
    const int i = threadIdx.x;
    if (i < tbl.n) frames[tbl.first + i] = tbl.f[i];
    if (tbl.first == 0 && i < 8) {
        queue_head[i * kQueueStride] = 0u;      // rays handed out
        queue_head[i * kQueueStride + 1] = 0u;  // rays stored
    }
}

// Probe circle overlay, volrend.cu:100-134.  Pixels inside the circle skip the
// ray march (enable_draw=false) and end with alpha 1, so they are independent of
// the main kernel's result and simply overwrite it.  One thread per pixel of the
// (probe_disp_size+5)^2 corner square.
template <int FMA>
__global__ void probe_overlay_kernel(const KParams p) {
    using P = Policy<FMA>;
    const int side = p.probe_disp_size + 5;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= side * side) return;
    const int y = idx / side;
    const int x = p.width - side + (idx - y * side);
    if (x < 0 || x >= p.width || y >= p.height) return;
    // which tile / rank owns this pixel
    const int tx = x / p.tile_w, ty = y / p.tile_h;
    const int tile = ty * p.tiles_x + tx;
    if (tile % p.world != p.rank) return;
    float cen[3], dir[3], out[4] = {0.f, 0.f, 0.f, 0.f};
    const int xx = x - (p.width - p.probe_disp_size) + 5;
    const int yy = y - 5;
    cen[0] = -((float)xx / (0.5f * (float)p.probe_disp_size) - 1.f);
    cen[1] = ((float)yy / (0.5f * (float)p.probe_disp_size) - 1.f);
    const float c = P::madd(cen[0], cen[0], cen[1] * cen[1]);
    if (!(c <= 1.f)) return;
    if (p.basis_dim >= 0) {
        float basis_fn[VR_MAX_BASIS];
#pragma unroll
        for (int i = 0; i < VR_MAX_BASIS; ++i) basis_fn[i] = 0.f;
        cen[2] = -__builtin_sqrtf(1 - c);
        float xf[9];
        for (int i = 0; i < 9; ++i) xf[i] = p.frames[blockIdx.y].xf[i];
        mv3<FMA>(xf, cen, dir);
        precalc_basis<FMA, true>(p, dir, basis_fn);
        // upstream indexes past basis_dim with the default basis_minmax {0,24} (UB);
        // like the oracle, clamp to the coefficients that exist
        int hi = p.basis_max;
        if (hi > p.basis_dim - 1) hi = p.basis_dim - 1;
        const int lo = p.basis_min < 0 ? 0 : p.basis_min;
        for (int tt = 0; tt < 3; ++tt) {
            const int off = tt * p.basis_dim;
            float tmp = 0.f;
            for (int i = lo; i <= hi; ++i) tmp = P::madd(basis_fn[i], p.probe_coeffs[off + i], tmp);
            out[tt] = 1.f / (1.f + vr_expf(-tmp));
        }
    } else {
        for (int i = 0; i < 3; ++i) out[i] = p.probe_coeffs[i];
    }
    out[3] = 1.f;
    const int64_t pix = (int64_t)y * p.width + x;
    const FrameDesc& fd = p.frames[blockIdx.y];
    if (fd.accum) reinterpret_cast<float4*>(fd.accum)[pix] = make_float4(out[0], out[1], out[2], out[3]);
    // nalpha = 1 - out[3] = 0: the composite adds (+0 * anything) and leaves out[] as is
    uint8_t* px;
    if (p.layout == VR_LAYOUT_COMPACT) {
        const int k = tile / p.world;
        const int lx = x - tx * p.tile_w, ly = y - ty * p.tile_h;
        px = fd.rgba + ((int64_t)k * p.tile_w * p.tile_h + (int64_t)ly * p.tile_w + lx) * 4;
    } else {
        px = fd.rgba + (int64_t)y * p.pitch + (int64_t)x * 4;
    }
    *reinterpret_cast<uint32_t*>(px) =
        quant8(out[0]) | (quant8(out[1]) << 8) | (quant8(out[2]) << 16) | 0xFF000000u;
}

// retrieve_cursor_lumisphere_kernel, volrend.cu:175-191
__global__ void probe_kernel(const KParams p, float probe0, float probe1, float probe2,
                             float* out) {
    float cen[3] = {p.offset[0] + p.scale[0] * probe0, p.offset[1] + p.scale[1] * probe1,
                    p.offset[2] + p.scale[2] * probe2};
    float cube_sz;
    int levels;
    uint32_t word;
    const int64_t leaf = query_generic<0>(p, cen, &cube_sz, &levels, &word);
    const uint16_t* v = p.leaves + leaf * p.leaf_stride_h;
    for (int i = threadIdx.x; i < p.data_dim - 1; i += blockDim.x) out[i] = h2f(v[i]);
}

// De-interleave `world` gathered COMPACT buffers into frames.  blockIdx.z = frame of the
// batch; rank r's compact buffer of frame i starts at gathered + r*rank_stride + i*in_stride.
__global__ void assemble_kernel(uint8_t* frame, int64_t pitch, const uint8_t* gathered, int width,
                                int height, int tile_w, int tile_h, int tiles_x, int world,
                                int64_t out_stride, int64_t rank_stride, int64_t in_stride) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= width || y >= height) return;
    frame += (int64_t)blockIdx.z * out_stride;
    gathered += (int64_t)blockIdx.z * in_stride;
    const int tx = x / tile_w, ty = y / tile_h;
    const int tile = ty * tiles_x + tx;
    const int rank = tile % world;
    const int64_t k = tile / world;
    const int lx = x - tx * tile_w, ly = y - ty * tile_h;
    const int64_t src = (k * tile_w * tile_h + (int64_t)ly * tile_w + lx);
    *reinterpret_cast<uint32_t*>(frame + (int64_t)y * pitch + (int64_t)x * 4) =
        reinterpret_cast<const uint32_t*>(gathered + (int64_t)rank * rank_stride)[src];
}

// ---------------------------------------------------------------------------
// Upload-time re-layout (reference layout -> device layout, see above)
// ---------------------------------------------------------------------------
__global__ void relayout_nodes_kernel(const int32_t* child, const uint16_t* data,
                                      const int32_t* perm, uint32_t* nodes, int64_t n_slots,
                                      int N3, int data_dim) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_slots) return;
    const int64_t n = i / N3;
    const int s = (int)(i - n * N3);
    const int32_t skip = child[i];
    const int64_t target = n + skip;
    uint32_t w;
    // Links of nodes the root cannot reach are not covered by the host's topology check
    // (file capacity > used nodes): a link that leaves the array becomes a leaf word.
    if (skip == 0 || target <= 0 || target >= n_slots / N3) {
        w = kLeafBit | (uint32_t)data[i * data_dim + (data_dim - 1)];
    } else {
        w = (uint32_t)perm[target];
    }
    nodes[(int64_t)perm[n] * N3 + s] = w;
}

// one thread per 16-byte chunk of the padded record array
__global__ void relayout_leaves_kernel(const uint16_t* data, const int32_t* perm, uint16_t* leaves,
                                       int64_t n_slots, int N3, int data_dim, int stride_h) {
    const int chunks = stride_h / 8;
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t slot = gid / chunks;
    if (slot >= n_slots) return;
    const int c = (int)(gid - slot * chunks);
    const uint16_t* src = data + slot * data_dim;
    uint16_t h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int e = c * 8 + j;
        h[j] = e < data_dim - 1 ? src[e] : (uint16_t)0;
    }
    uint4 q;
    q.x = h[0] | ((uint32_t)h[1] << 16);
    q.y = h[2] | ((uint32_t)h[3] << 16);
    q.z = h[4] | ((uint32_t)h[5] << 16);
    q.w = h[6] | ((uint32_t)h[7] << 16);
    const int64_t n = slot / N3;
    const int64_t dst = (int64_t)perm[n] * N3 + (slot - n * N3);
    reinterpret_cast<uint4*>(leaves + dst * stride_h)[c] = q;
}

// Median-cut codebook decode on the device (reference host loop: src/n3tree.cpp:310-339):
//   data[slot, j + n_ret + c*n_basis] = quant_colors[j, quant_map[j, slot], c]   j < n_quant
//   data[slot, j + c*n_basis]         = data_retained[j, slot, c]                j < n_ret
//   data[slot, data_dim-1]            = sigma[slot]
// One work item per (slot, basis); the basis index is the fast one so that the three
// 2-byte stores of neighbouring lanes land in the same lines.  `data` is zeroed first.
__global__ void decode_quant_kernel(const uint16_t* __restrict__ colors,
                                    const uint16_t* __restrict__ map,
                                    const uint16_t* __restrict__ sigma,
                                    const uint16_t* __restrict__ retained,
                                    uint16_t* __restrict__ data, int64_t n_slots, int n_quant,
                                    int n_ret, int data_dim) {
    const int n_basis = n_quant + n_ret;
    const int64_t total = n_slots * n_basis;
    for (int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; w < total;
         w += (int64_t)gridDim.x * blockDim.x) {
        const int64_t slot = w / n_basis;
        const int j = (int)(w - slot * n_basis);
        const uint16_t* c;
        if (j < n_ret) {
            c = retained + ((int64_t)j * n_slots + slot) * 3;
        } else {
            const int q = j - n_ret;
            c = colors + ((int64_t)q * 65536 + map[(int64_t)q * n_slots + slot]) * 3;
        }
        uint16_t* o = data + slot * data_dim + j;
        o[0] = c[0];
        o[n_basis] = c[1];
        o[2 * n_basis] = c[2];
        if (j == 0) data[slot * data_dim + data_dim - 1] = sigma[slot];
    }
}

__global__ void popcount_kernel(const uint32_t* words, uint64_t n_words, unsigned long long* out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words;
         i += (uint64_t)gridDim.x * blockDim.x)
        acc += (unsigned long long)__builtin_popcount(words[i]);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

// Lookup structure, part 1: top[cell] for every cell of the 2^G0-per-axis grid (layout comment
// at the top of this file).  brick_root[] = indices of the internal nodes of level G0, ascending.
__global__ void build_top_kernel(const uint32_t* nodes, const int32_t* brick_root, int n_bricks,
                                 uint2* top, int G0, uint32_t* error_flag) {
    const uint32_t cell = blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (1u << (3 * G0))) return;
    const uint32_t mask = (1u << G0) - 1u;
    const uint32_t cx = (cell >> (2 * G0)) & mask, cy = (cell >> G0) & mask, cz = cell & mask;
    uint32_t node = 0;
    for (int l = 0; l < G0; ++l) {
        const int sh = G0 - 1 - l;
        const uint32_t slot = (((cx >> sh) & 1u) << 2) | (((cy >> sh) & 1u) << 1) | ((cz >> sh) & 1u);
        const uint32_t w = nodes[(uint64_t)node * 8u + slot];
        if (w & kLeafBit) {
            top[cell] = make_uint2(kLeafBit | ((uint32_t)(l + 1) << 16) | (w & 0xFFFFu),
                                   node * 8u + slot);
            return;
        }
        node = w;
    }
    // internal node of level G0: find its brick
    int lo = 0, hi = n_bricks - 1, found = -1;
    while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const uint32_t r = (uint32_t)brick_root[mid];
        if (r == node) {
            found = mid;
            break;
        }
        if (r < node) lo = mid + 1; else hi = mid - 1;
    }
    if (found < 0) {
        atomicOr(error_flag, 1u);  // host and device disagree on the level-G0 nodes
        found = 0;
    }
    top[cell] = make_uint2((uint32_t)found, node);
}

// Lookup structure, part 2: one thread per brick entry.
__global__ void build_bricks_kernel(const uint32_t* nodes, const int32_t* brick_root, int n_bricks,
                                    uint32_t* bricks, int BL, int blocked, uint32_t* error_flag) {
    const uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t per = 1u << (3 * BL);
    if (gid >= (uint64_t)n_bricks * per) return;
    const uint32_t b = (uint32_t)(gid >> (3 * BL)), e = (uint32_t)gid & (per - 1u);
    const uint32_t mask = (1u << BL) - 1u;
    uint32_t ex = (e >> (2 * BL)) & mask, ey = (e >> BL) & mask, ez = e & mask;
    if (blocked) {  // BL == 3: e = [x2 y2 z2 z1 | x1 x0 y1 y0 z0] (query_n2)
        ex = (((e >> 8) & 1u) << 2) | ((e >> 3) & 3u);
        ey = (((e >> 7) & 1u) << 2) | ((e >> 1) & 3u);
        ez = (((e >> 5) & 3u) << 1) | (e & 1u);
    }
    const uint32_t root = (uint32_t)brick_root[b];
    uint32_t node = root;
    for (int k = 0; k < BL; ++k) {
        const int sh = BL - 1 - k;
        const uint32_t slot = (((ex >> sh) & 1u) << 2) | (((ey >> sh) & 1u) << 1) | ((ez >> sh) & 1u);
        const uint32_t w = nodes[(uint64_t)node * 8u + slot];
        if (w & kLeafBit) {
            const uint32_t delta = node - root;
            if (delta > 1023u) atomicOr(error_flag, 2u);  // numbering contract broken
            bricks[gid] = kLeafBit | ((uint32_t)k << 29) | ((delta & 1023u) << 19) | (slot << 16) |
                          (w & 0xFFFFu);
            return;
        }
        node = w;
    }
    bricks[gid] = node;  // internal node of level G0 + BL
}

// grid = the persistent waves: as many as the chip holds of this flavour (or the tuning
// override), but no more than about one wave per 128 rays of a small launch
template <int FMA, int MODE>
hipError_t launch_basis(const KParams& p, int64_t want, int n_cus, int waves_override, hipStream_t s) {
    const dim3 block(kWave);
    int b;
    if (p.basis_dim < 0 || p.format == VR_FORMAT_RGBA) {
        b = BASIS_RGBA;
    } else {
        switch (p.basis_dim) {  // the reference's switch only knows 25/16/9/4 (rt_core.cuh:132-160)
            case 25: b = BASIS_25; break;
            case 16: b = BASIS_16; break;
            case 9: b = BASIS_9; break;
            case 4: b = BASIS_4; break;
            default: b = BASIS_1; break;
        }
    }
#define VR_LAUNCH(B)                                                                         \
    do {                                                                                     \
        const int64_t cap_ = (int64_t)n_cus * (waves_override > 0 ? waves_override           \
                                                                  : waves_per_cu<B, MODE>()); \
        const dim3 grid((unsigned)(want < cap_ ? want : cap_));                              \
        if (MODE == MODE_FAST && p.brick_blocked)                                            \
            hipLaunchKernelGGL((render_kernel<FMA, B, MODE, MODE == MODE_FAST>), grid, block, 0, s, p); \
        else                                                                                 \
            hipLaunchKernelGGL((render_kernel<FMA, B, MODE, false>), grid, block, 0, s, p);  \
    } while (0)
    switch (b) {
        case BASIS_RGBA: VR_LAUNCH(BASIS_RGBA); break;
        case BASIS_25: VR_LAUNCH(BASIS_25); break;
        case BASIS_16: VR_LAUNCH(BASIS_16); break;
        case BASIS_9: VR_LAUNCH(BASIS_9); break;
        case BASIS_4: VR_LAUNCH(BASIS_4); break;
        default: VR_LAUNCH(BASIS_1); break;
    }
#undef VR_LAUNCH
    return hipGetLastError();
}

template <int FMA>
hipError_t launch_fp(const KParams& p, int64_t want, int n_cus, int waves_override, hipStream_t s) {
    const bool n2 = (p.N == 2) && p.top_levels > 0;  // built at upload when the tree qualifies
    const bool lobes = p.format == VR_FORMAT_SG || p.format == VR_FORMAT_ASG;
    if (!n2) return launch_basis<FMA, MODE_GENERIC>(p, want, n_cus, waves_override, s);
    if (lobes || p.instrumented || p.render_depth)  // the depth visualisation lives outside FAST
        return launch_basis<FMA, MODE_FULL>(p, want, n_cus, waves_override, s);
    return launch_basis<FMA, MODE_FAST>(p, want, n_cus, waves_override, s);
}

}  // namespace

hipError_t launch_prepare(const KParams& p, const FrameTable& tbl, hipStream_t stream) {
    hipLaunchKernelGGL(prepare_launch_kernel, dim3(1), dim3(64), 0, stream, tbl,
                       const_cast<FrameDesc*>(p.frames), p.queue_head);
    return hipGetLastError();
}

hipError_t launch_render(const KParams& p, int fp_mode, int n_cus, int waves_override, int gen_waves,
                         hipStream_t stream) {
    if (p.n_wave_blocks <= 0 || p.n_frames <= 0) return hipSuccess;
    const int64_t total_blocks = p.n_wave_blocks * p.n_frames;
    {   // ray generation: gen_waves wave blocks (8x8 pixels each) per workgroup
        const bool full = p.instrumented || p.render_depth || p.format == VR_FORMAT_SG ||
                          p.format == VR_FORMAT_ASG;
#define VR_GEN(FMA_, FULL_, GW_)                                                                    \
    hipLaunchKernelGGL((raygen_kernel<FMA_, FULL_, GW_>),                                           \
                       dim3((unsigned)((total_blocks + GW_ - 1) / GW_)), dim3(kWave * GW_), 0, stream, p)
#define VR_GEN_GW(FMA_, FULL_)                                                                      \
    do {                                                                                            \
        if (gen_waves >= 16) VR_GEN(FMA_, FULL_, 16);                                               \
        else if (gen_waves >= 4) VR_GEN(FMA_, FULL_, 4);                                            \
        else VR_GEN(FMA_, FULL_, 1);                                                                \
    } while (0)
        if (fp_mode == VR_FP_FMA) {
            if (full) VR_GEN_GW(1, true); else VR_GEN_GW(1, false);
        } else {
            if (full) VR_GEN_GW(0, true); else VR_GEN_GW(0, false);
        }
#undef VR_GEN_GW
#undef VR_GEN
    }
    // persistent march grid: enough waves to fill the chip, but no more than one per
    // ~128 pixels so that small launches still rebalance through the ray queue
    int64_t want = total_blocks / 2;  // about one wave per 64 rays that enter the volume
    if (want < 256) want = 256;
    if (want > total_blocks) want = total_blocks;
    const hipError_t e = fp_mode == VR_FP_FMA
                             ? launch_fp<1>(p, want, n_cus, waves_override, stream)
                             : launch_fp<0>(p, want, n_cus, waves_override, stream);
    if (e != hipSuccess || !p.enable_probe || p.probe_disp_size <= 0) return e;
    const int side = p.probe_disp_size + 5;
    const dim3 pgrid((unsigned)((side * side + 255) / 256), (unsigned)p.n_frames);
    if (fp_mode == VR_FP_FMA)
        hipLaunchKernelGGL(probe_overlay_kernel<1>, pgrid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(probe_overlay_kernel<0>, pgrid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_assemble(uint8_t* frame, int64_t pitch, const uint8_t* gathered, int width,
                           int height, int tile_w, int tile_h, int world, int n_frames,
                           int64_t out_stride, int64_t rank_stride, int64_t in_stride,
                           hipStream_t stream) {
    const int tiles_x = (width + tile_w - 1) / tile_w;
    const dim3 block(64, 4);
    const dim3 grid((width + 63) / 64, (height + 3) / 4, n_frames);
    hipLaunchKernelGGL(assemble_kernel, grid, block, 0, stream, frame, pitch, gathered, width,
                       height, tile_w, tile_h, tiles_x, world, out_stride, rank_stride, in_stride);
    return hipGetLastError();
}

hipError_t launch_probe(const KParams& p, const float probe[3], float* out_dev,
                        hipStream_t stream) {
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, stream, p, probe[0], probe[1],
                       probe[2], out_dev);
    return hipGetLastError();
}

int leaf_stride_halfs(int data_dim) {
    const int bytes = 2 * (data_dim - 1);
    int stride = 16;
    while (stride < bytes && stride < 128) stride *= 2;  // 16, 32, 64, 128: never straddles a line
    if (stride < bytes) stride = (bytes + 31) / 32 * 32;
    return stride / 2;
}

hipError_t launch_relayout(const int32_t* child, const uint16_t* data, const int32_t* perm,
                           uint32_t* nodes, uint16_t* leaves, int64_t n_slots, int N3,
                           int data_dim, int stride_h, hipStream_t stream) {
    const int tpb = 256;
    hipLaunchKernelGGL(relayout_nodes_kernel, dim3((unsigned)((n_slots + tpb - 1) / tpb)), dim3(tpb),
                       0, stream, child, data, perm, nodes, n_slots, N3, data_dim);
    const int64_t n_chunks = n_slots * (stride_h / 8);
    hipLaunchKernelGGL(relayout_leaves_kernel, dim3((unsigned)((n_chunks + tpb - 1) / tpb)),
                       dim3(tpb), 0, stream, data, perm, leaves, n_slots, N3, data_dim, stride_h);
    return hipGetLastError();
}

hipError_t launch_build_lookup(const uint32_t* nodes, const int32_t* brick_root, int n_bricks,
                               uint2* top, uint32_t* bricks, int top_levels, int brick_levels,
                               int brick_blocked, uint32_t* error_flag, hipStream_t stream) {
    const uint32_t n_cells = 1u << (3 * top_levels);
    hipLaunchKernelGGL(build_top_kernel, dim3((n_cells + 255) / 256), dim3(256), 0, stream, nodes,
                       brick_root, n_bricks, top, top_levels, error_flag);
    if (n_bricks > 0 && brick_levels > 0) {
        const uint64_t n = (uint64_t)n_bricks << (3 * brick_levels);
        hipLaunchKernelGGL(build_bricks_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           stream, nodes, brick_root, n_bricks, bricks, brick_levels,
                           (brick_blocked && brick_levels == 3) ? 1 : 0, error_flag);
    }
    return hipGetLastError();
}

hipError_t launch_popcount(const uint32_t* words, uint64_t n_words, unsigned long long* out,
                           hipStream_t stream) {
    if (n_words == 0) return hipSuccess;
    uint64_t blocks = (n_words + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(popcount_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, words, n_words,
                       out);
    return hipGetLastError();
}

hipError_t launch_decode_quant(const uint16_t* colors, const uint16_t* map, const uint16_t* sigma,
                               const uint16_t* retained, uint16_t* data, int64_t n_slots,
                               int n_quant, int n_ret, int data_dim, hipStream_t stream) {
    hipError_t e = hipMemsetAsync(data, 0, (size_t)n_slots * data_dim * sizeof(uint16_t), stream);
    if (e != hipSuccess) return e;
    const int64_t total = n_slots * (int64_t)(n_quant + n_ret);
    const int tpb = 256;
    int64_t blocks = (total + tpb - 1) / tpb;
    if (blocks > (1 << 20)) blocks = 1 << 20;  // grid-stride beyond that
    hipLaunchKernelGGL(decode_quant_kernel, dim3((unsigned)blocks), dim3(tpb), 0, stream, colors,
                       map, sigma, retained, data, n_slots, n_quant, n_ret, data_dim);
    return hipGetLastError();
}

}  // namespace vr
