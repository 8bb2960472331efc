// vr_api.cpp -- the C ABI of libvolrend_hip.so (include/volrend_hip.h).
// Host side only: argument validation, device memory, launch set-up.
// Built with hipcc for gfx950, -ffp-contract=off (the host-side Rodrigues
// pre-computation below must round like the oracle).
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <atomic>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <cstring>
#include <new>
#include <vector>

#include "vr_internal.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// Scheduling / layout knobs.  They never change results.  Every tree carries its OWN copy
// (vr_tree_set_tuning), taken at upload from the process defaults below; the defaults come from
// the environment (VR_MARCH_MAX, VR_REFILL_MIN, VR_WAVES_PER_CU, ... read once) and
// vr_set_tuning, which only affects trees uploaded afterwards and is serialised by a mutex.
struct Tuning {
    int march_max = 12;
    int refill_min = 20;
    int drain_flush = 16;  // drain phase: partial round for a blocked ray when <= this many lanes march (0 = off;
                           // measured 4..64, profiles/r05_experiments.jsonl: one frame per launch -13 %, two / four -5 %)
    int waves_per_cu = 0;   // 0: what the kernel flavour fits (vr_kernels.hip waves_per_cu<>)
    int frame_group = 0;   // poses per ray-order group (0 = all poses of the launch, 1 = frame-major)
    int super_block = 1;   // 8x8 blocks per super-block edge in the ray order
    int records_nt = -1;   // record stream non-temporal: -1 = by lookup-structure size, 0 / 1 = forced
    int xcd_queues = 1;
    int chunk_max = 4096;
    int raygen_waves = 0;  // waves per ray-generation workgroup: 16 / 4 / 1; 0 = by launch size (vr_render_batch)
    int top_levels = 0;    // lookup structure built at upload (vr_kernels.hip); 0 = auto
    int brick_levels = 3;
    int brick_blocked = -1;  // 8^3 bricks in 4 x 4 x 2 line blocks: -1 = when the lookup structure exceeds 128 MB, 0 / 1 = forced
    int max_iter = 1 << 22;  // the sample guard (vr_kernels.hip); the one knob that is NOT scheduling-only:
                             // a launch that trips it reports through vr_tree_status (tests lower it)
};
std::mutex g_tuning_mutex;
Tuning& default_tuning_locked() {  // call with g_tuning_mutex held
    static Tuning tn = [] {
        Tuning x;
        if (const char* e = getenv("VR_MARCH_MAX")) x.march_max = atoi(e) < 1 ? 1 : atoi(e);
        if (const char* e = getenv("VR_REFILL_MIN")) x.refill_min = atoi(e) < 1 ? 1 : atoi(e);
        if (const char* e = getenv("VR_DRAIN_FLUSH")) x.drain_flush = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("VR_WAVES_PER_CU")) x.waves_per_cu = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("VR_FRAME_GROUP")) x.frame_group = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("VR_SUPER_BLOCK")) x.super_block = atoi(e) < 1 ? 1 : atoi(e);
        if (const char* e = getenv("VR_RECORDS_NT")) x.records_nt = atoi(e);
        if (const char* e = getenv("VR_XCD_QUEUES")) x.xcd_queues = atoi(e) != 0;
        if (const char* e = getenv("VR_CHUNK_MAX")) x.chunk_max = atoi(e) < 64 ? 64 : (atoi(e) & ~63);
        if (const char* e = getenv("VR_RAYGEN_WAVES")) x.raygen_waves = atoi(e) < 0 ? 0 : atoi(e);
        if (const char* e = getenv("VR_TOP_LEVELS")) x.top_levels = atoi(e);
        if (const char* e = getenv("VR_BRICK_LEVELS")) x.brick_levels = atoi(e);
        if (const char* e = getenv("VR_BRICK_BLOCKED")) x.brick_blocked = atoi(e);
        if (const char* e = getenv("VR_MAX_ITER")) x.max_iter = atoi(e) < 1 ? 1 : atoi(e);
        return x;
    }();
    return tn;
}
Tuning default_tuning() {
    std::lock_guard<std::mutex> g(g_tuning_mutex);
    return default_tuning_locked();
}
bool set_tuning_key(Tuning& tn, const char* key, int value) {
    if (!strcmp(key, "march_max")) tn.march_max = value < 1 ? 1 : value;
    else if (!strcmp(key, "refill_min")) tn.refill_min = value < 1 ? 1 : (value > 64 ? 64 : value);
    else if (!strcmp(key, "drain_flush")) tn.drain_flush = value < 0 ? 0 : (value > 64 ? 64 : value);
    else if (!strcmp(key, "waves_per_cu")) tn.waves_per_cu = value < 0 ? 0 : (value > 64 ? 64 : value);
    else if (!strcmp(key, "frame_group")) tn.frame_group = value < 0 ? 0 : value;
    else if (!strcmp(key, "super_block")) tn.super_block = value < 1 ? 1 : (value > 64 ? 64 : value);
    else if (!strcmp(key, "records_nt")) tn.records_nt = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "xcd_queues")) tn.xcd_queues = value != 0;
    else if (!strcmp(key, "chunk_max")) tn.chunk_max = value < 64 ? 64 : (value & ~63);
    else if (!strcmp(key, "raygen_waves")) tn.raygen_waves = value < 0 ? 0 : value;
    else if (!strcmp(key, "top_levels")) tn.top_levels = value;
    else if (!strcmp(key, "brick_levels")) tn.brick_levels = value;
    else if (!strcmp(key, "brick_blocked")) tn.brick_blocked = value < 0 ? -1 : (value != 0);
    else if (!strcmp(key, "max_iter")) tn.max_iter = value < 1 ? 1 : value;
    else return false;
    return true;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP,      \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,    \
                        __LINE__);                                                          \
    } while (0)

// A tree lives on ONE device; its calls run there whatever the calling thread's current device
// is (one host thread may drive the trees of several devices), and leave the thread's device
// as they found it.
class DeviceGuard {
   public:
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev_) == hipSuccess && prev_ != device) {
            switched_ = hipSetDevice(device) == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched_) (void)hipSetDevice(prev_);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;

   private:
    int prev_ = 0;
    bool switched_ = false;
};

}  // namespace

constexpr unsigned kLaunchSlots = 8;
constexpr unsigned kSlotWords = 160;  // [16 + 16*x] queue x (x < 8): rays handed out, rays stored

// Per-launch scratch that a kernel reads while it runs: frame table, queue heads, ray count,
// ray buffer, probe coefficients.  `done` is recorded on the launch's stream behind its last
// kernel and the next user of the slot makes ITS stream wait for it (device-side wait, the host
// never blocks), so any number of launches on any number of streams may be in flight -- beyond
// kLaunchSlots they simply serialise.  A launch prefers the slot its own stream used last (the
// stream orders the two launches anyway), then a slot whose last launch has finished, and only
// then the next slot of the ring: a render loop on one stream lives in ONE slot and one ray
// buffer however far the host runs ahead, two alternating streams in two.
struct LaunchSlot {
    hipEvent_t done = nullptr;
    bool used = false;          // `done` has been recorded at least once
    hipStream_t last_stream = nullptr;  // the stream of that launch
    bool growing = false;       // its ray buffer is being reallocated outside the launch mutex: skip it
    uint32_t* rays = nullptr;   // ray buffer, grown on demand (or up front by vr_reserve)
    size_t ray_bytes = 0;
};

struct VrTreeOpaque {
    int device = 0;
    uint32_t* nodes = nullptr;   // device layout (vr_kernels.hip)
    uint16_t* leaves = nullptr;
    uint2* top = nullptr;        // lookup structure (N == 2), see vr_kernels.hip
    uint32_t* bricks = nullptr;
    int top_levels = 0, brick_levels = 0, n_bricks = 0;
    int brick_blocked = 0;       // entry order of the bricks (vr_kernels.hip), fixed at upload
    int leaf_stride_h = 0;
    float* extra = nullptr;
    uint32_t* status = nullptr;
    unsigned long long* sched_stats = nullptr;  // 8 x u64, see vr_sched_stats
    uint32_t* touch[4] = {nullptr, nullptr, nullptr, nullptr};  // distinct-line bitmaps (vr_touch_enable)
    uint64_t array_bytes[4] = {0, 0, 0, 0};  // leaves, nodes, top, bricks
    unsigned long long* touch_out = nullptr;
    float* probe_buf = nullptr;  // kLaunchSlots x data_dim floats: the lumisphere at opt.probe
    vr::FrameDesc* slot_frames = nullptr;  // kLaunchSlots x kMaxBatch
    uint32_t* slot_heads = nullptr;        // kLaunchSlots x kSlotWords
    LaunchSlot slots[kLaunchSlots];
    unsigned launch_seq = 0;
    std::mutex launch_mutex;  // slot bookkeeping + enqueue order of one launch; guards `tn`
    Tuning tn;                // this tree's knobs (vr_tree_set_tuning)
    int n_cus = 256;
    VrTreeDesc desc{};  // pointers cleared; scalars kept
    int32_t max_depth = 0;
    uint64_t device_bytes = 0;
};

namespace {

// Walks the child links from the root: every link must land on a node that has
// not been reached before (a tree, not a DAG / cycle), inside [1, capacity).
// Returns the deepest leaf level or -1.  A malformed file would otherwise make
// the device descent loop forever.
// level[n] = depth of node n (root 0), 255 = not reachable from the root.
int validate_topology(const int32_t* child, int64_t cap, int N3, std::vector<uint8_t>& level,
                      char* why, size_t why_len) {
    if (cap <= 0) {
        snprintf(why, why_len, "capacity must be positive");
        return -1;
    }
    // Fast path: files written breadth- or depth-first link every child FORWARD (to a higher
    // index), and then one sweep in index order sees every parent before its children -- no
    // queue, sequential reads (a 2 M-node tree: ~20 ms instead of ~50).  The first backward link
    // abandons the sweep for the general walk below.
    {
        level.assign((size_t)cap, 255);
        level[0] = 0;
        int depth = 0;
        bool forward_only = true;
        for (int64_t n = 0; n < cap && forward_only; ++n) {
            const uint8_t ln = level[(size_t)n];
            if (ln == 255) continue;  // not reachable (so far: decided for good if all links go forward)
            const int32_t* c = child + n * N3;
            for (int s = 0; s < N3; ++s) {
                const int64_t skip = c[s];
                if (skip == 0) continue;
                const int64_t m = n + skip;
                if (m <= n) {
                    forward_only = false;
                    break;
                }
                if (m >= cap) {
                    snprintf(why, why_len, "node %lld slot %d links outside the tree (%lld)",
                             (long long)n, s, (long long)m);
                    return -1;
                }
                if (level[(size_t)m] != 255) {
                    snprintf(why, why_len, "node %lld is linked twice (cycle or DAG)", (long long)m);
                    return -1;
                }
                if (ln + 1 > 60) {
                    snprintf(why, why_len, "tree deeper than 60 levels");
                    return -1;
                }
                level[(size_t)m] = (uint8_t)(ln + 1);
                if (ln + 1 > depth) depth = ln + 1;
            }
        }
        if (forward_only) return depth;
    }
    std::vector<uint8_t> seen((size_t)cap, 0);
    level.assign((size_t)cap, 255);
    level[0] = 0;
    std::vector<int64_t> cur{0}, next;
    seen[0] = 1;
    int depth = 0;
    for (;;) {
        next.clear();
        for (int64_t n : cur) {
            const int32_t* c = child + n * N3;
            for (int s = 0; s < N3; ++s) {
                const int64_t skip = c[s];
                if (skip == 0) continue;
                const int64_t m = n + skip;
                if (m <= 0 || m >= cap) {
                    snprintf(why, why_len, "node %lld slot %d links outside the tree (%lld)",
                             (long long)n, s, (long long)m);
                    return -1;
                }
                if (seen[(size_t)m]) {
                    snprintf(why, why_len, "node %lld is linked twice (cycle or DAG)", (long long)m);
                    return -1;
                }
                seen[(size_t)m] = 1;
                level[(size_t)m] = (uint8_t)(depth + 1);
                next.push_back(m);
            }
        }
        if (next.empty()) break;
        if (++depth > 60) {
            snprintf(why, why_len, "tree deeper than 60 levels");
            return -1;
        }
        cur.swap(next);
    }
    return depth;
}

// New node numbering: pre-order depth-first from the root (children in slot order), so a
// subtree is one contiguous run of the arrays.  Exception for the lookup structure (N == 2,
// G0 > 0): behind an internal node of level G0 (a brick root) come first ALL its descendants of
// the next BL - 1 levels, breadth-first (<= 8 + 64 nodes: a brick entry names the parent of its
// leaf as root + delta), and only then the subtrees hanging below level G0 + BL - 1, each
// depth-first.  Unreachable nodes keep their relative order behind the reachable ones.
// brick_roots receives the new indices of the level-G0 internal nodes (ascending).
std::vector<int32_t> node_permutation(const int32_t* child, int64_t cap, int N3, int G0, int BL,
                                      const std::vector<uint8_t>& level,
                                      std::vector<int32_t>& brick_roots) {
    std::vector<int32_t> perm((size_t)cap, -1);
    brick_roots.clear();
    int32_t next = 0;
    std::vector<int64_t> stack{0}, ring, ring_next;
    while (!stack.empty()) {
        const int64_t n = stack.back();
        stack.pop_back();
        perm[(size_t)n] = next++;
        const int32_t* c = child + n * N3;
        if (G0 > 0 && level[(size_t)n] == G0) {
            brick_roots.push_back(perm[(size_t)n]);
            // levels G0+1 .. G0+BL-1 breadth-first right behind the root
            ring.assign(1, n);
            for (int k = 1; k < BL; ++k) {
                ring_next.clear();
                for (int64_t m : ring)
                    for (int s = 0; s < N3; ++s)
                        if (child[m * N3 + s] != 0) {
                            const int64_t ch = m + child[m * N3 + s];
                            perm[(size_t)ch] = next++;
                            ring_next.push_back(ch);
                        }
                ring.swap(ring_next);
            }
            // `ring` = the nodes of level G0+BL-1: their children start ordinary subtrees
            for (size_t i = ring.size(); i-- > 0;) {
                const int64_t m = ring[i];
                for (int s = N3 - 1; s >= 0; --s)
                    if (child[m * N3 + s] != 0) stack.push_back(m + child[m * N3 + s]);
            }
            continue;
        }
        for (int s = N3 - 1; s >= 0; --s)  // reversed: slot 0 is visited first
            if (c[s] != 0) stack.push_back(n + c[s]);
    }
    for (int64_t i = 0; i < cap; ++i)
        if (perm[(size_t)i] < 0) perm[(size_t)i] = next++;
    return perm;
}

// ---------------------------------------------------------------------------
// Host -> device copies of the tree arrays at link speed.  hipMemcpy from pageable memory
// stages through ONE thread's memcpy (~9 GB/s measured: 1.67 GB in 0.19 s); here up to
// kCopyWorkersMax threads each stream chunks through two pinned slots of their own: memcpy into
// slot (i & 1) while the DMA of the previous chunk drains slot (i & 1) ^ 1 (with the source pages
// mapped ahead of time -- prefault_host_range -- 8 threads keep the link busy: 36-45 GB/s
// measured; without, the memcpy is page-fault bound at ~20).  All the DMAs go to
// ONE stream per device (creating a stream costs milliseconds -- an HSA queue -- and the link is
// the shared resource anyway); that stream and the pinned slots (with their events) live in a
// process-wide cache, so only the first upload of a process pays for them.  Chunks are claimed
// dynamically across all segments of a call.  Anything small, or any failure to set the pipeline
// up, falls back to the plain blocking copy.  VR_UPLOAD_TIMING=1 prints the phases.
// ---------------------------------------------------------------------------
constexpr size_t kCopyChunk = 2u << 20;  // (pinned memory costs ~0.5 ms per MB to allocate: 8 workers x 2 slots = 32 MB)
constexpr int kCopyWorkersMax = 4;  // (with the pages mapped ahead, 4 memcpy threads fill the link; every slot is 2 MB of pinned memory to allocate)

struct CopySegment {
    void* dst;
    const void* src;
    size_t bytes;
};
struct PinnedSlot {
    void* mem = nullptr;
    hipEvent_t done = nullptr;  // the last DMA out of this slot
    bool used = false;
};
struct UploadCache {
    static constexpr int kDevices = 16;
    std::mutex mu;
    // per DEVICE: a slot's event belongs to the device that was current when it was created, and
    // recording it on another device's stream is an error
    std::vector<PinnedSlot> free_slots[kDevices];
    hipStream_t stream[kDevices] = {};  // per device, created on first use
    // (call with `device` current)
    bool take(PinnedSlot& out, int device) {
        if (device < 0 || device >= kDevices) return false;
        {
            std::lock_guard<std::mutex> g(mu);
            if (!free_slots[device].empty()) {
                out = free_slots[device].back();
                out.used = false;
                free_slots[device].pop_back();
                return true;
            }
        }
        PinnedSlot sl;
        if (hipHostMalloc(&sl.mem, kCopyChunk, hipHostMallocPortable) != hipSuccess ||
            hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (sl.mem) (void)hipHostFree(sl.mem);
            return false;
        }
        out = sl;
        return true;
    }
    void give(const PinnedSlot& sl, int device) {
        std::lock_guard<std::mutex> g(mu);
        free_slots[device].push_back(sl);
    }
    // the stream and 2 x kCopyWorkersMax slots up front (first upload of the process)
    void warm(int device) {
        (void)stream_of(device);
        std::vector<PinnedSlot> got;
        for (int i = 0; i < 2 * kCopyWorkersMax; ++i) {
            PinnedSlot sl;
            if (!take(sl, device)) break;
            got.push_back(sl);
        }
        for (const PinnedSlot& sl : got) give(sl, device);
    }
    hipStream_t stream_of(int device) {
        std::lock_guard<std::mutex> g(mu);
        if (device < 0 || device >= kDevices) return nullptr;
        if (!stream[device] &&
            hipStreamCreateWithFlags(&stream[device], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            stream[device] = nullptr;
        }
        return stream[device];
    }
};
UploadCache& upload_cache() {
    static UploadCache* c = new UploadCache();  // never destroyed: no HIP calls at exit
    return *c;
}

// Maps the pages of a host range into this process ahead of the staged copy (tree files are
// handed over as views of an mmap'ed npz: every 4 KB page of the 1.6 GB costs a minor fault the
// first time a copy worker reads it, and the copy is fault-bound).  Runs on a few threads while
// the HIP runtime starts up; best effort, no effect on results.
void prefault_host_range(const void* ptr, size_t bytes) {
    if (!ptr || bytes < (64u << 20)) return;
    const unsigned hw = std::thread::hardware_concurrency();
    const int n_thr = hw >= 32 ? 8 : (hw >= 8 ? 4 : 1);
    const uintptr_t page = 4096;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(ptr) + page - 1) & ~(page - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(ptr) + bytes) & ~(page - 1);
    if (hi <= lo) return;
    const uintptr_t per = ((hi - lo) / n_thr + page - 1) & ~(page - 1);
    auto work = [=](int i) {
        const uintptr_t a = lo + per * (uintptr_t)i, b = a + per < hi ? a + per : hi;
        if (a >= b) return;
        // (one read per page, not madvise(MADV_POPULATE_READ): the bulk call holds the process's
        // mmap lock for its whole range and the HIP runtime's own mappings -- start-up, every
        // allocation -- queue up behind it; single faults take the per-VMA lock only)
        volatile unsigned char sink = 0;
        for (uintptr_t q = a; q < b; q += page) sink = sink + *reinterpret_cast<const volatile unsigned char*>(q);
        (void)sink;
    };
    std::vector<std::thread> pool;
    try {
        for (int i = 1; i < n_thr; ++i) pool.emplace_back(work, i);
    } catch (...) {
    }
    work(0);
    for (auto& t : pool) t.join();
}

hipError_t staged_h2d_multi(const CopySegment* seg, int n_seg, int device) {
    const auto t0 = std::chrono::steady_clock::now();
    size_t total = 0, n_chunks = 0;
    std::vector<size_t> first_chunk((size_t)n_seg + 1, 0);
    for (int i = 0; i < n_seg; ++i) {
        first_chunk[(size_t)i] = n_chunks;
        n_chunks += (seg[i].bytes + kCopyChunk - 1) / kCopyChunk;
        total += seg[i].bytes;
    }
    first_chunk[(size_t)n_seg] = n_chunks;
    auto plain = [&]() {
        for (int i = 0; i < n_seg; ++i)
            if (seg[i].bytes) {
                const hipError_t e = hipMemcpy(seg[i].dst, seg[i].src, seg[i].bytes, hipMemcpyHostToDevice);
                if (e != hipSuccess) return e;
            }
        return hipSuccess;
    };
    const unsigned hw = std::thread::hardware_concurrency();
    int workers = hw >= 8 ? kCopyWorkersMax : (hw >= 4 ? 2 : 1);
    if ((size_t)workers > n_chunks) workers = (int)n_chunks;
    hipStream_t st = (total >= (32u << 20) && workers >= 2) ? upload_cache().stream_of(device) : nullptr;
    if (!st) return plain();
    std::atomic<int> failed{0};
    std::atomic<size_t> next{0};
    auto work = [&]() {
        PinnedSlot slot[2];
        bool ok = hipSetDevice(device) == hipSuccess && upload_cache().take(slot[0], device) &&
                  upload_cache().take(slot[1], device);
        // chunks are claimed dynamically (a worker that was scheduled late does not hold the others up)
        for (int k = 0; ok; k ^= 1) {
            const size_t c = next.fetch_add(1);
            if (c >= n_chunks) break;
            int si = 0;
            while (c >= first_chunk[(size_t)si + 1]) ++si;
            const size_t off = (c - first_chunk[(size_t)si]) * kCopyChunk;
            const size_t len = seg[si].bytes - off < kCopyChunk ? seg[si].bytes - off : kCopyChunk;
            if (slot[k].used) ok = hipEventSynchronize(slot[k].done) == hipSuccess;  // its last DMA is done
            if (!ok) break;
            memcpy(slot[k].mem, static_cast<const char*>(seg[si].src) + off, len);
            ok = hipMemcpyAsync(static_cast<char*>(seg[si].dst) + off, slot[k].mem, len,
                                hipMemcpyHostToDevice, st) == hipSuccess &&
                 hipEventRecord(slot[k].done, st) == hipSuccess;
            slot[k].used = ok;  // (only a RECORDED event may be waited for)
        }
        // A failed enqueue / record may have left a DMA out of a slot in flight with no event to
        // wait for: drain the stream before the slots go back to the cache.
        if (!ok) (void)hipStreamSynchronize(st);
        for (auto& sl : slot) {
            if (!sl.mem) continue;
            if (sl.used && hipEventSynchronize(sl.done) != hipSuccess) ok = false;  // before the slot is reused
            upload_cache().give(sl, device);
        }
        if (!ok) failed.store(1);
    };
    std::vector<std::thread> pool;
    try {
        for (int w = 1; w < workers; ++w) pool.emplace_back(work);
    } catch (...) {  // could not start (all) helpers: this thread copies what is left
    }
    work();
    for (auto& t : pool) t.join();
    if (failed.load()) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(st);
        return plain();  // plain copy of everything
    }
    if (getenv("VR_UPLOAD_TIMING")) {
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        fprintf(stderr, "[volrend_hip] staged H2D: %.1f MB in %d segments, %d workers, %.1f ms (%.1f GB/s)\n",
                total / 1e6, n_seg, workers, ms, total / ms / 1e6);
    }
    return hipSuccess;
}

hipError_t staged_h2d(void* dst, const void* src, size_t bytes, int device) {
    const CopySegment seg{dst, src, bytes};
    return staged_h2d_multi(&seg, 1, device);
}

// same rounding sequence as the oracle's norm3 (strict / fma)
float host_norm3(const float* d, int fma) {
    float s;
    if (fma) {
        s = std::fmaf(d[0], d[0], d[1] * d[1]);
        s = std::fmaf(d[2], d[2], s);
    } else {
        s = d[0] * d[0] + d[1] * d[1];
        s = d[2] * d[2] + s;
    }
    return std::sqrt(s);
}

// basis words kept per ray in the ray buffer: what the kernel flavour for this basis_dim reads
int basis_words_of(const VrTreeOpaque* t) {
    const int bd = t->desc.basis_dim;
    if (t->desc.format == VR_FORMAT_RGBA || bd < 0) return 0;
    return (bd == 4 || bd == 9 || bd == 16 || bd == 25) ? bd : 1;
}

// SH trees with a basis size the kernel knows: the ray record carries the view direction (3 words)
// and the lane that takes the ray evaluates the basis; everything else carries the basis values
bool ray_carries_vdir(const VrTreeOpaque* t) {
    const int bw = basis_words_of(t);
    return t->desc.format == VR_FORMAT_SH && bw > 3;
}
int ray_tail_words_of(const VrTreeOpaque* t) { return ray_carries_vdir(t) ? 3 : basis_words_of(t); }

size_t ray_buffer_bytes(uint32_t total_rays, int tail_words) {
    // the ray queues own whole groups of 16 blocks of 64 rays (vr_kernels.hip "Ray queues")
    const size_t slots = (((size_t)total_rays / 64 + 15) / 16) * 16 * 64;
    return slots * (16 + (size_t)tail_words) * sizeof(uint32_t);  // kRayWords + tail
}

void fill_tree_params(vr::KParams& k, const VrTreeOpaque* t) {
    k.nodes = t->nodes;
    k.leaves = t->leaves;
    k.top = t->top;
    k.bricks = t->bricks;
    k.top_levels = t->top_levels;
    k.brick_levels = t->brick_levels;
    k.brick_blocked = t->brick_blocked;
    k.extra = t->extra;
    for (int i = 0; i < 3; ++i) {
        k.offset[i] = t->desc.offset[i];
        k.scale[i] = t->desc.scale[i];
    }
    k.N = t->desc.N;
    k.N3 = t->desc.N * t->desc.N * t->desc.N;
    k.capacity = t->desc.capacity;
    k.data_dim = t->desc.data_dim;
    k.format = t->desc.format;
    k.basis_dim = t->desc.basis_dim;
    k.leaf_stride_h = t->leaf_stride_h;
    k.max_depth = t->max_depth;
    k.ndc_width = t->desc.ndc_width;
    k.ndc_height = t->desc.ndc_height;
    k.ndc_focal = t->desc.ndc_focal;
    k.status = t->status;
    k.sched_stats = t->sched_stats;
    for (int i = 0; i < 4; ++i) k.touch[i] = t->touch[i];
}

}  // namespace

extern "C" {

int vr_abi_version(void) { return VR_ABI_VERSION; }

const char* vr_last_error(void) { return g_err; }

int vr_device_count(int* count) {
    if (!count) return fail(VR_ERR_INVALID_ARGUMENT, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(VR_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return VR_OK;
}

int vr_set_device(int device) {
    if (device < 0) return VR_OK;
    HIP_TRY(hipSetDevice(device));
    return VR_OK;
}

int vr_device_name(int device, char* name, size_t name_len) {
    if (!name || name_len == 0) return fail(VR_ERR_INVALID_ARGUMENT, "name buffer is NULL");
    hipDeviceProp_t prop;
    if (device < 0) HIP_TRY(hipGetDevice(&device));
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    snprintf(name, name_len, "%s", prop.gcnArchName);
    return VR_OK;
}

void vr_default_tree_desc(VrTreeDesc* d) {
    if (!d) return;
    memset(d, 0, sizeof(*d));
    d->N = 2;
    d->format = VR_FORMAT_RGBA;
    d->basis_dim = -1;
    d->ndc_width = -1.f;
}

// Validates the codebook arrays of a quantised tree against the tree description.
static int check_quant(const VrTreeDesc* d, const VrQuantDesc* q) {
    if (q->n_quant < 0 || q->n_retained < 0 || q->n_quant + q->n_retained < 1)
        return fail(VR_ERR_INVALID_ARGUMENT, "quantised tree needs at least one basis function");
    if (3 * (q->n_quant + q->n_retained) + 1 > d->data_dim)
        return fail(VR_ERR_INVALID_ARGUMENT, "%d quantised + %d retained basis functions do not "
                    "fit data_dim=%d", q->n_quant, q->n_retained, d->data_dim);
    if (!q->sigma) return fail(VR_ERR_INVALID_ARGUMENT, "sigma is NULL");
    if (q->n_quant && (!q->quant_colors || !q->quant_map))
        return fail(VR_ERR_INVALID_ARGUMENT, "quant_colors/quant_map is NULL");
    if (q->n_retained && !q->data_retained)
        return fail(VR_ERR_INVALID_ARGUMENT, "data_retained is NULL");
    return VR_OK;
}

// Stages the codebook arrays on the device (unless they are there already) and decodes
// them into `d_data` (flat reference layout, n_slots * data_dim halfs, device memory).
static hipError_t decode_quant_on_device(const VrTreeDesc* d, const VrQuantDesc* q, size_t n_slots,
                                         uint16_t* d_data, int device) {
    const size_t sz_colors = (size_t)q->n_quant * 65536 * 3 * sizeof(uint16_t);
    const size_t sz_map = (size_t)q->n_quant * n_slots * sizeof(uint16_t);
    const size_t sz_sigma = n_slots * sizeof(uint16_t);
    const size_t sz_ret = (size_t)q->n_retained * n_slots * 3 * sizeof(uint16_t);
    const void* src[4] = {q->quant_colors, q->quant_map, q->sigma, q->data_retained};
    const size_t sz[4] = {sz_colors, sz_map, sz_sigma, sz_ret};
    void* tmp[4] = {nullptr, nullptr, nullptr, nullptr};
    const void* dev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipError_t e = hipSuccess;
    CopySegment segs[4];
    int n_seg = 0;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) {
        if (!sz[i]) continue;
        if (d->memory == 1) {
            dev[i] = src[i];
            continue;
        }
        e = hipMalloc(&tmp[i], sz[i]);
        segs[n_seg++] = CopySegment{tmp[i], src[i], sz[i]};
        dev[i] = tmp[i];
    }
    if (e == hipSuccess && n_seg) e = staged_h2d_multi(segs, n_seg, device);
    if (e == hipSuccess)
        e = vr::launch_decode_quant((const uint16_t*)dev[0], (const uint16_t*)dev[1],
                                    (const uint16_t*)dev[2], (const uint16_t*)dev[3], d_data,
                                    (int64_t)n_slots, q->n_quant, q->n_retained, d->data_dim,
                                    nullptr);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    for (int i = 0; i < 4; ++i)
        if (tmp[i]) (void)hipFree(tmp[i]);
    return e;
}

static int check_tree_desc(const VrTreeDesc* d, bool need_data) {
    if (!d->child || (need_data && !d->data))
        return fail(VR_ERR_INVALID_ARGUMENT, "child/data is NULL");
    if (d->N < 2 || d->N > 16) return fail(VR_ERR_INVALID_ARGUMENT, "N=%d out of range", d->N);
    if (d->capacity <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "capacity must be positive");
    if (d->format < VR_FORMAT_RGBA || d->format > VR_FORMAT_ASG)
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown data format %d", d->format);
    const int min_dim = d->format == VR_FORMAT_RGBA ? 4 : 3 * d->basis_dim + 1;
    if (d->format != VR_FORMAT_RGBA && (d->basis_dim < 1 || d->basis_dim > VR_MAX_BASIS))
        return fail(VR_ERR_INVALID_ARGUMENT, "basis_dim=%d out of range [1,%d]", d->basis_dim,
                    VR_MAX_BASIS);
    if (d->data_dim < min_dim)
        return fail(VR_ERR_INVALID_ARGUMENT, "data_dim=%d too small for the format (need %d)",
                    d->data_dim, min_dim);
    if (d->format == VR_FORMAT_SG && (!d->extra || d->extra_count < (uint64_t)d->basis_dim * 4))
        return fail(VR_ERR_INVALID_ARGUMENT, "SG needs basis_dim*4 extra floats");
    if (d->format == VR_FORMAT_ASG && (!d->extra || d->extra_count < (uint64_t)d->basis_dim * 11))
        return fail(VR_ERR_INVALID_ARGUMENT, "ASG needs basis_dim*11 extra floats");
    return VR_OK;
}

// Everything of a tree that is not tree data, on the current device (= t->device): status and
// tally words, the launch-slot ring (events, frame tables, queue heads, probe coefficients).
static hipError_t alloc_launch_scratch(VrTreeOpaque* t) {
    hipError_t e = hipMalloc((void**)&t->status, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(t->status, 0, sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&t->sched_stats, 8 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(t->sched_stats, 0, 8 * sizeof(unsigned long long));
    if (e == hipSuccess)
        e = hipMalloc((void**)&t->probe_buf,
                      sizeof(float) * (size_t)t->desc.data_dim * kLaunchSlots);
    for (unsigned i = 0; i < kLaunchSlots && e == hipSuccess; ++i)
        e = hipEventCreateWithFlags(&t->slots[i].done, hipEventDisableTiming);
    if (e == hipSuccess)
        e = hipMalloc((void**)&t->slot_frames, sizeof(vr::FrameDesc) * vr::kMaxBatch * kLaunchSlots);
    if (e == hipSuccess)
        e = hipMalloc((void**)&t->slot_heads, sizeof(uint32_t) * kSlotWords * kLaunchSlots);
    if (e == hipSuccess) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, t->device) ==
                hipSuccess && cus > 0)
            t->n_cus = cus;
    }
    return e;
}

static int upload_body(const VrTreeDesc* d, const VrQuantDesc* q, vr_tree_t* out) {
    if (!d || !out) return fail(VR_ERR_INVALID_ARGUMENT, "desc/out is NULL");
    *out = nullptr;
    if (int rc = check_tree_desc(d, q == nullptr)) return rc;
    if (q)
        if (int rc = check_quant(d, q)) return rc;

    const int N3 = d->N * d->N * d->N;
    const size_t n_slots = (size_t)d->capacity * N3;
    const size_t child_sz = n_slots * sizeof(int32_t);
    const size_t data_sz = n_slots * (size_t)d->data_dim * sizeof(uint16_t);

    const bool timing = getenv("VR_UPLOAD_TIMING") != nullptr;
    const auto t_start = std::chrono::steady_clock::now();
    auto since = [&]() {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
    };
    // topology check needs the child words on the host
    std::vector<int32_t> staged;
    const int32_t* host_child = d->child;
    if (d->memory == 1) {
        staged.resize(n_slots);
        HIP_TRY(hipMemcpy(staged.data(), d->child, child_sz, hipMemcpyDeviceToHost));
        host_child = staged.data();
    }
    // The big host-to-device copies (and the codebook decode of a quantised file) run on a
    // helper thread while this one walks the tree on the host (topology check, node numbering):
    // the walks only read the child array, and hide completely behind the copies.
    // (a malformed tree is reported as such even where no device exists: no HIP error before that)
    // The first HIP call of a process starts the runtime (~50 ms; 145-200 ms now and then, right
    // after another process released gigabytes of device memory -- the outlier of
    // tools/upload_bench.py).  It has to be made on THIS thread (the tree goes to the caller's
    // current device, and a new thread's current device is 0), so the topology check starts first,
    // on a thread of its own, and runs beside it.
    char why[256] = "";
    std::vector<uint8_t> level;
    int max_depth = -1;
    bool walk_threw = false;
    std::thread walker([&] {
        try {
            max_depth = validate_topology(host_child, d->capacity, N3, level, why, sizeof(why));
        } catch (...) {
            walk_threw = true;
        }
        if (timing) fprintf(stderr, "[volrend_hip] upload: topology checked at %.1f ms\n", since());
    });
    struct WalkerJoin {
        std::thread& th;
        ~WalkerJoin() {
            if (th.joinable()) th.join();
        }
    } walker_join{walker};
    int device = 0;
    const hipError_t e_dev = hipGetDevice(&device);
    if (timing) fprintf(stderr, "[volrend_hip] upload: HIP runtime up at %.1f ms\n", since());
    int32_t* d_child = nullptr;
    uint16_t* d_data = nullptr;
    hipError_t e_copy = e_dev;
    // (the copier only moves the small child array and pays the runtime's start-up before it
    // looks at `topo`: a malformed file is rejected after ~30 ms of host walk, not after a
    // multi-GB upload)
    std::atomic<int> topo{0};  // 0: the check is still running, 1: tree is sound, -1: bad tree
    // the file's pages are mapped (prefault_host_range) beside the runtime's start-up and the
    // topology check, ahead of the copy that reads them
    std::thread prefaulter([&] {
        if (d->memory != 1 && !q) prefault_host_range(d->data, data_sz);
    });
    struct PrefaultJoin {
        std::thread& th;
        ~PrefaultJoin() {
            if (th.joinable()) th.join();
        }
    } prefault_join{prefaulter};
    std::thread copier([&] {
        if (e_dev != hipSuccess) return;
        hipError_t e = hipSetDevice(device);
        // runtime start-up, the allocations and the copy pipeline's pinned slots + stream first:
        // they need the process's mmap lock exclusively, which a page-mapping pass would hold
        if (d->memory != 1 && e == hipSuccess) e = hipMalloc((void**)&d_child, child_sz);
        if ((q || d->memory != 1) && e == hipSuccess) e = hipMalloc((void**)&d_data, data_sz);
        if (d->memory != 1 && e == hipSuccess) upload_cache().warm(device);
        if (timing) fprintf(stderr, "[volrend_hip] upload: runtime + buffers ready at %.1f ms\n", since());
        while (topo.load(std::memory_order_acquire) == 0) std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (topo.load(std::memory_order_acquire) < 0) {
            e_copy = e;
            return;
        }
        if (q) {  // quantised file: only the codebook arrays cross PCIe, the decode runs on the device
            if (e == hipSuccess && d->memory != 1) e = staged_h2d(d_child, d->child, child_sz, device);
            if (e == hipSuccess) e = decode_quant_on_device(d, q, n_slots, d_data, device);
        } else if (d->memory != 1 && e == hipSuccess) {
            const CopySegment both[2] = {{d_child, d->child, child_sz}, {d_data, d->data, data_sz}};
            e = staged_h2d_multi(both, 2, device);
        }
        e_copy = e;
        if (timing) fprintf(stderr, "[volrend_hip] upload: copies done at %.1f ms\n", since());
    });
    struct Joiner {  // every exit below waits for the copies and drops the staging buffers
        std::thread& th;
        int32_t*& c;
        uint16_t*& dd;
        std::atomic<int>& topo;
        bool keep = false;
        ~Joiner() {
            int pending = 0;
            topo.compare_exchange_strong(pending, -1);  // (an early exit must not leave the copier waiting)
            if (th.joinable()) th.join();
            if (!keep) {
                if (c) (void)hipFree(c);
                if (dd) (void)hipFree(dd);
                c = nullptr;
                dd = nullptr;
            }
        }
    } joiner{copier, d_child, d_data, topo};

    walker.join();
    if (walk_threw) {  // (the copier must be released before the exception travels on)
        topo.store(-1, std::memory_order_release);
        throw std::bad_alloc();
    }
    topo.store(max_depth < 0 ? -1 : 1, std::memory_order_release);
    if (max_depth < 0) return fail(VR_ERR_BAD_TREE, "bad tree: %s", why);
    if (e_dev != hipSuccess)
        return fail(VR_ERR_HIP, "hipGetDevice failed: %s", hipGetErrorString(e_dev));
    // Lookup structure (N == 2 fast path): leaves must sit within 24 levels (exact integer
    // digits of a binary32 coordinate) and node*8+slot byte offsets must fit 32 bits.
    int G0 = 0, BL = 0;
    const Tuning tn = default_tuning();  // the new tree's own copy from here on
    if (vr_query_mode_for(d->N, max_depth, d->capacity) == VR_QUERY_LOOKUP) {
        // auto: top grid + brick reach the deepest leaf (depth max_depth + 1) without a child-word
        // walk where a top grid of <= 256^3 cells allows it -- 64^3 (2 MB) for lego-class trees of
        // 9 levels, 128^3 for 10 (measured: C1 0.269 ms at (6,3) against 0.301 at (5,3); C3 0.790
        // at (7,3) against 0.847 at (6,3))
        G0 = tn.top_levels > 0 ? tn.top_levels : (max_depth + 1 - 3 < 6 ? 6 : max_depth + 1 - 3);
        if (G0 > 8) G0 = 8;
        if (G0 > max_depth + 1) G0 = max_depth + 1;  // deepest leaf depth
        BL = tn.brick_levels < 1 ? 1 : (tn.brick_levels > 4 ? 4 : tn.brick_levels);
        if (BL > max_depth + 1 - G0) BL = max_depth + 1 - G0;  // 0: the top grid resolves every leaf
        // the kernel addresses brick entries with 32-bit byte offsets: keep the brick array < 4 GB
        uint64_t n_roots = 0;
        for (uint8_t l : level) n_roots += (l == G0);
        while (BL > 1 && ((n_roots << (3 * BL)) * sizeof(uint32_t)) >= (1ull << 32)) --BL;
    }

    VrTreeOpaque* t = new (std::nothrow) VrTreeOpaque();
    if (!t) return fail(VR_ERR_OUT_OF_MEMORY, "host allocation failed");
    t->desc = *d;
    t->desc.child = nullptr;
    t->desc.data = nullptr;
    t->desc.extra = nullptr;
    t->max_depth = max_depth;
    t->device = device;
    t->tn = tn;
    // new node numbering (host walk) while the copies are still in flight
    std::vector<int32_t> brick_roots;
    const std::vector<int32_t> perm =
        node_permutation(host_child, d->capacity, N3, G0, BL, level, brick_roots);
    // the reference arrays are staged on the device now (unless they already were there);
    // re-layout into nodes/leaves, build the lookup structure, drop the staging copies
    if (timing) fprintf(stderr, "[volrend_hip] upload: host walks done at %.1f ms\n", since());
    copier.join();
    hipError_t e = e_copy;
    const int32_t* src_child = d->memory != 1 ? d_child : d->child;
    const uint16_t* src_data = (q || d->memory != 1) ? d_data : d->data;
    t->leaf_stride_h = vr::leaf_stride_halfs(d->data_dim);
    const size_t leaves_sz = n_slots * (size_t)t->leaf_stride_h * sizeof(uint16_t);
    if (e == hipSuccess) e = hipMalloc((void**)&t->nodes, child_sz);
    if (e == hipSuccess) e = hipMalloc((void**)&t->leaves, leaves_sz);
    if (e == hipSuccess) e = alloc_launch_scratch(t);
    int32_t* d_perm = nullptr;
    if (e == hipSuccess) e = hipMalloc((void**)&d_perm, perm.size() * sizeof(int32_t));
    if (e == hipSuccess)
        e = hipMemcpy(d_perm, perm.data(), perm.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess)
        e = vr::launch_relayout(src_child, src_data, d_perm, t->nodes, t->leaves, (int64_t)n_slots,
                                N3, d->data_dim, t->leaf_stride_h, nullptr);
    t->device_bytes = child_sz + leaves_sz + sizeof(uint32_t);
    t->array_bytes[0] = leaves_sz;
    t->array_bytes[1] = child_sz;
    // lookup structure: top grid + bricks (vr_kernels.hip), built from the node words
    int32_t* d_roots = nullptr;
    if (e == hipSuccess && G0 > 0) {
        const size_t top_sz = ((size_t)1 << (3 * G0)) * sizeof(uint2);
        const int n_bricks = BL > 0 ? (int)brick_roots.size() : 0;
        const size_t brick_sz = ((size_t)n_bricks << (3 * BL)) * sizeof(uint32_t);
        e = hipMalloc((void**)&t->top, top_sz);
        if (e == hipSuccess && n_bricks) e = hipMalloc((void**)&t->bricks, brick_sz);
        if (e == hipSuccess && n_bricks) e = hipMalloc((void**)&d_roots, n_bricks * sizeof(int32_t));
        if (e == hipSuccess && n_bricks)
            e = hipMemcpy(d_roots, brick_roots.data(), n_bricks * sizeof(int32_t),
                          hipMemcpyHostToDevice);
        // entry order of the bricks: blocked where the lookups are fabric traffic (a lookup structure
        // far beyond the 32 MB of L2), x-major where they mostly hit (six instructions cheaper)
        const int blocked = (n_bricks && BL == 3)
                                ? (tn.brick_blocked >= 0 ? tn.brick_blocked
                                                         : (top_sz + brick_sz > (128ull << 20)))
                                : 0;
        if (e == hipSuccess)
            e = vr::launch_build_lookup(t->nodes, d_roots, n_bricks, t->top, t->bricks, G0, BL,
                                        blocked, t->status, nullptr);
        uint32_t flag = 0;
        if (e == hipSuccess) e = hipMemcpy(&flag, t->status, sizeof(flag), hipMemcpyDeviceToHost);
        if (e == hipSuccess && flag != 0) {
            if (d_roots) (void)hipFree(d_roots);
            if (d_perm) (void)hipFree(d_perm);
            vr_tree_free(t);
            return fail(VR_ERR_BAD_TREE, "lookup structure build failed (flag %u)", flag);
        }
        if (e == hipSuccess) {
            t->top_levels = G0;
            t->brick_levels = n_bricks ? BL : 0;
            t->brick_blocked = blocked;
            t->n_bricks = n_bricks;
            t->device_bytes += top_sz + brick_sz;
            t->array_bytes[2] = top_sz;
            t->array_bytes[3] = brick_sz;
        }
    }
    if (d_roots) (void)hipFree(d_roots);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (d_perm) (void)hipFree(d_perm);
    if (e == hipSuccess && d->extra && d->extra_count) {
        const size_t esz = (size_t)d->extra_count * sizeof(float);
        const hipMemcpyKind kind =
            d->memory == 1 ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
        e = hipMalloc((void**)&t->extra, esz);
        if (e == hipSuccess) e = hipMemcpy(t->extra, d->extra, esz, kind);
        t->device_bytes += esz;
    }
    if (e != hipSuccess) {
        vr_tree_free(t);
        return fail(e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP,
                    "tree upload failed: %s", hipGetErrorString(e));
    }
    if (timing) fprintf(stderr, "[volrend_hip] upload: device-ready at %.1f ms\n", since());
    *out = t;
    return VR_OK;
}

// The host side of an upload allocates (level / permutation vectors) and starts threads: nothing
// of that may leave through the C boundary as an exception.
static int upload_impl(const VrTreeDesc* d, const VrQuantDesc* q, vr_tree_t* out) {
    try {
        return upload_body(d, q, out);
    } catch (const std::bad_alloc&) {
        if (out) *out = nullptr;
        return fail(VR_ERR_OUT_OF_MEMORY, "tree upload: host allocation failed");
    } catch (const std::exception& e) {  // std::system_error of a thread that could not start
        if (out) *out = nullptr;
        return fail(VR_ERR_OUT_OF_MEMORY, "tree upload: %s", e.what());
    }
}

int vr_tree_upload(const VrTreeDesc* d, vr_tree_t* out) { return upload_impl(d, nullptr, out); }

int vr_tree_upload_quantized(const VrTreeDesc* d, const VrQuantDesc* q, vr_tree_t* out) {
    if (!q) return fail(VR_ERR_INVALID_ARGUMENT, "quant desc is NULL");
    return upload_impl(d, q, out);
}

int vr_decode_quantized(const VrTreeDesc* d, const VrQuantDesc* q, uint16_t* data_out) {
    if (!d || !q || !data_out) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (d->N < 2 || d->N > 16 || d->capacity <= 0 || d->data_dim < 1)
        return fail(VR_ERR_INVALID_ARGUMENT, "bad N / capacity / data_dim");
    if (int rc = check_quant(d, q)) return rc;
    const size_t n_slots = (size_t)d->capacity * d->N * d->N * d->N;
    const size_t data_sz = n_slots * (size_t)d->data_dim * sizeof(uint16_t);
    uint16_t* d_data = data_out;
    hipError_t e = hipSuccess;
    if (d->memory != 1) e = hipMalloc((void**)&d_data, data_sz);
    int device = 0;
    if (e == hipSuccess) e = hipGetDevice(&device);
    if (e == hipSuccess) e = decode_quant_on_device(d, q, n_slots, d_data, device);
    if (e == hipSuccess && d->memory != 1)
        e = hipMemcpy(data_out, d_data, data_sz, hipMemcpyDeviceToHost);
    if (d->memory != 1 && d_data) (void)hipFree(d_data);
    if (e != hipSuccess)
        return fail(e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP,
                    "quantised decode failed: %s", hipGetErrorString(e));
    return VR_OK;
}

int vr_tree_clone(vr_tree_t src, int device, vr_tree_t* out) {
    if (!src || !out) return fail(VR_ERR_INVALID_ARGUMENT, "tree/out is NULL");
    *out = nullptr;
    int n_dev = 0;
    HIP_TRY(hipGetDeviceCount(&n_dev));
    if (device < 0 || device >= n_dev)
        return fail(VR_ERR_INVALID_ARGUMENT, "device %d outside [0,%d)", device, n_dev);
    // the source may still be rendering on its own device; no launch may be enqueued on it (nor
    // its bitmaps / slots change) while its arrays are read
    std::lock_guard<std::mutex> src_lock(src->launch_mutex);
    {
        DeviceGuard g(src->device);
        HIP_TRY(hipDeviceSynchronize());
    }
    DeviceGuard guard(device);
    VrTreeOpaque* t = new (std::nothrow) VrTreeOpaque();
    if (!t) return fail(VR_ERR_OUT_OF_MEMORY, "host allocation failed");
    t->device = device;
    t->desc = src->desc;
    t->max_depth = src->max_depth;
    t->leaf_stride_h = src->leaf_stride_h;
    t->tn = src->tn;
    t->top_levels = src->top_levels;
    t->brick_levels = src->brick_levels;
    t->brick_blocked = src->brick_blocked;
    t->n_bricks = src->n_bricks;
    t->device_bytes = src->device_bytes;
    for (int i = 0; i < 4; ++i) t->array_bytes[i] = src->array_bytes[i];
    // the re-laid-out arrays travel device to device (over xGMI between two GPUs of a node):
    // no second pass over PCIe, no second re-layout
    // direct peer access (xGMI / PCIe P2P) when the two devices have it: hipMemcpyPeer then moves
    // the arrays device to device; without it the runtime stages them through host memory
    // (still correct, ~10x slower) -- a note goes to stderr
    bool p2p = true;
    if (src->device != device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device, src->device) != hipSuccess) can = 0;
        p2p = can != 0;
        if (p2p) {
            const hipError_t pe = hipDeviceEnablePeerAccess(src->device, 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) p2p = false;
        }
        (void)hipGetLastError();
    }
    void** dst[4] = {(void**)&t->leaves, (void**)&t->nodes, (void**)&t->top, (void**)&t->bricks};
    const void* from[4] = {src->leaves, src->nodes, src->top, src->bricks};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 4 && e == hipSuccess; ++i) {
        if (!from[i] || !t->array_bytes[i]) continue;
        e = hipMalloc(dst[i], t->array_bytes[i]);
        if (e == hipSuccess)
            e = hipMemcpyPeer(*dst[i], device, from[i], src->device, t->array_bytes[i]);
    }
    if (e == hipSuccess && src->extra && src->desc.extra_count) {
        const size_t esz = (size_t)src->desc.extra_count * sizeof(float);
        e = hipMalloc((void**)&t->extra, esz);
        if (e == hipSuccess) e = hipMemcpyPeer(t->extra, device, src->extra, src->device, esz);
    }
    if (e == hipSuccess) e = alloc_launch_scratch(t);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        vr_tree_free(t);
        return fail(e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP,
                    "tree clone from device %d to device %d failed: %s%s", src->device, device,
                    hipGetErrorString(e),
                    p2p ? "" : " (the devices have NO peer access: check `rocm-smi --showtopo`, "
                               "IOMMU / ACS settings and HSA_ENABLE_IPC_MODE_LEGACY=0)");
    }
    if (!p2p)  // (a note, not an error: vr_last_error() stays empty after a call that returned VR_OK)
        fprintf(stderr, "[volrend_hip] note: devices %d and %d have no peer access; the clone was "
                        "staged through host memory\n", src->device, device);
    *out = t;
    return VR_OK;
}

int vr_tree_free(vr_tree_t t) {
    if (!t) return VR_OK;
    DeviceGuard guard(t->device);
    if (t->nodes) (void)hipFree(t->nodes);
    if (t->leaves) (void)hipFree(t->leaves);
    if (t->top) (void)hipFree(t->top);
    if (t->bricks) (void)hipFree(t->bricks);
    if (t->extra) (void)hipFree(t->extra);
    if (t->status) (void)hipFree(t->status);
    if (t->sched_stats) (void)hipFree(t->sched_stats);
    for (int i = 0; i < 4; ++i)
        if (t->touch[i]) (void)hipFree(t->touch[i]);
    if (t->touch_out) (void)hipFree(t->touch_out);
    if (t->probe_buf) (void)hipFree(t->probe_buf);
    if (t->slot_frames) (void)hipFree(t->slot_frames);
    if (t->slot_heads) (void)hipFree(t->slot_heads);
    for (unsigned i = 0; i < kLaunchSlots; ++i) {
        if (t->slots[i].rays) (void)hipFree(t->slots[i].rays);
        if (t->slots[i].done) (void)hipEventDestroy(t->slots[i].done);
    }
    delete t;
    return VR_OK;
}

int vr_tree_info(vr_tree_t t, VrTreeInfo* info) {
    if (!t || !info) return fail(VR_ERR_INVALID_ARGUMENT, "tree/info is NULL");
    info->capacity = t->desc.capacity;
    info->N = t->desc.N;
    info->data_dim = t->desc.data_dim;
    info->format = t->desc.format;
    info->basis_dim = t->desc.basis_dim;
    info->max_depth = t->max_depth;
    info->device = t->device;
    info->device_bytes = t->device_bytes;
    info->leaf_stride = (uint64_t)t->leaf_stride_h * 2u;
    info->query_mode = t->top_levels > 0 ? VR_QUERY_LOOKUP : VR_QUERY_DESCENT;
    info->top_levels = t->top_levels;
    info->brick_levels = t->brick_levels;
    info->brick_blocked = t->brick_blocked;
    return VR_OK;
}

// The integer lookup needs exact digits of a binary32 coordinate (leaves within 24 levels: child
// words read <= 24) and 32-bit byte offsets into the node array (node * 8 + slot words < 2^30).
int vr_query_mode_for(int N, int max_depth, int64_t capacity) {
    return (N == 2 && max_depth <= 23 && capacity < (1ll << 27)) ? VR_QUERY_LOOKUP : VR_QUERY_DESCENT;
}

void vr_default_options(VrRenderOptions* o) {
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->step_size = 1e-4f;
    o->sigma_thresh = 1e-2f;
    o->stop_thresh = 1e-2f;
    o->background_brightness = 1.f;
    o->render_bbox[3] = o->render_bbox[4] = o->render_bbox[5] = 1.f;
    o->basis_minmax[0] = 0;
    o->basis_minmax[1] = VR_MAX_BASIS - 1;
    o->grid_max_depth = 4;
    o->probe[2] = 1.f;
    o->probe_disp_size = 100;
}

void vr_default_frame(VrFrame* f) {
    if (!f) return;
    memset(f, 0, sizeof(*f));
    f->offscreen = 1;
    f->layout = VR_LAYOUT_FRAME;
    f->world = 1;
    f->fp_mode = VR_FP_STRICT;
}

static int tile_geometry(int width, int height, int tile_w, int tile_h, int* tw, int* th,
                         int* tiles_x, int* tiles_y) {
    if (width <= 0 || height <= 0) return fail(VR_ERR_INVALID_ARGUMENT, "empty image");
    if (tile_w == 0 && tile_h == 0) {
        tile_w = (width + 7) & ~7;
        tile_h = (height + 7) & ~7;
    }
    if (tile_w <= 0 || tile_h <= 0 || (tile_w & 7) || (tile_h & 7))
        return fail(VR_ERR_INVALID_ARGUMENT, "tile size %dx%d must be positive multiples of 8",
                    tile_w, tile_h);
    *tw = tile_w;
    *th = tile_h;
    *tiles_x = (width + tile_w - 1) / tile_w;
    *tiles_y = (height + tile_h - 1) / tile_h;
    return VR_OK;
}

int64_t vr_compact_bytes(int width, int height, int tile_w, int tile_h, int world) {
    int tw, th, tx, ty;
    if (tile_geometry(width, height, tile_w, tile_h, &tw, &th, &tx, &ty) != VR_OK) return -1;
    if (world < 1) world = 1;
    const int64_t n_tiles = (int64_t)tx * ty;
    const int64_t per_rank = (n_tiles + world - 1) / world;
    return per_rank * tw * th * 4;
}

int vr_set_tuning(const char* key, int value) {
    if (!key) return fail(VR_ERR_INVALID_ARGUMENT, "key is NULL");
    std::lock_guard<std::mutex> g(g_tuning_mutex);
    if (!set_tuning_key(default_tuning_locked(), key, value))
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown tuning key '%s'", key);
    return VR_OK;
}

int vr_tree_set_tuning(vr_tree_t t, const char* key, int value) {
    if (!t || !key) return fail(VR_ERR_INVALID_ARGUMENT, "tree/key is NULL");
    if (!strcmp(key, "top_levels") || !strcmp(key, "brick_levels") || !strcmp(key, "brick_blocked"))
        return fail(VR_ERR_INVALID_ARGUMENT, "'%s' is fixed at upload (vr_set_tuning before it)", key);
    std::lock_guard<std::mutex> g(t->launch_mutex);
    if (!set_tuning_key(t->tn, key, value))
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown tuning key '%s'", key);
    return VR_OK;
}

int vr_sched_stats(vr_tree_t t, uint64_t out[8], int reset) {
    if (!t || !out) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(t->device);
    HIP_TRY(hipMemcpy(out, t->sched_stats, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(t->sched_stats, 0, 8 * sizeof(uint64_t)));
    return VR_OK;
}

static uint64_t touch_granule(int which) {  // bytes one bit of the bitmap stands for
    return which == 0 ? (1ull << vr::kTouchLeafShift) : 128ull;
}
static size_t touch_words(uint64_t array_bytes, int which) {  // one bit per 128-byte line
    const uint64_t g = touch_granule(which);
    return (size_t)(((array_bytes + g - 1) / g + 31) / 32);
}

int vr_touch_enable(vr_tree_t t, int enable) {
    if (!t) return fail(VR_ERR_INVALID_ARGUMENT, "tree is NULL");
    DeviceGuard guard(t->device);
    std::lock_guard<std::mutex> lock(t->launch_mutex);  // no launch is being enqueued meanwhile
    HIP_TRY(hipDeviceSynchronize());  // no launch may be using the bitmaps while they change
    for (int i = 0; i < 4; ++i) {
        if (t->touch[i]) {
            HIP_TRY(hipFree(t->touch[i]));
            t->touch[i] = nullptr;
        }
        const size_t words = touch_words(t->array_bytes[i], i);
        if (enable && words) {
            HIP_TRY(hipMalloc((void**)&t->touch[i], words * sizeof(uint32_t)));
            HIP_TRY(hipMemset(t->touch[i], 0, words * sizeof(uint32_t)));
        }
    }
    if (enable && !t->touch_out) HIP_TRY(hipMalloc((void**)&t->touch_out, 4 * sizeof(unsigned long long)));
    return VR_OK;
}

int vr_touch_count(vr_tree_t t, uint64_t out[4], int reset) {
    if (!t || !out) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (!t->touch_out) return fail(VR_ERR_INVALID_ARGUMENT, "vr_touch_enable(tree, 1) first");
    DeviceGuard guard(t->device);
    std::lock_guard<std::mutex> lock(t->launch_mutex);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemset(t->touch_out, 0, 4 * sizeof(unsigned long long)));
    for (int i = 0; i < 4; ++i) {
        if (!t->touch[i]) continue;
        const size_t words = touch_words(t->array_bytes[i], i);
        HIP_TRY(vr::launch_popcount(t->touch[i], words, t->touch_out + i, nullptr));
        if (reset) HIP_TRY(hipMemsetAsync(t->touch[i], 0, words * sizeof(uint32_t), nullptr));
    }
    HIP_TRY(hipMemcpy(out, t->touch_out, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return VR_OK;
}

int vr_touch_read(vr_tree_t t, int which, uint32_t* host_words, uint64_t n_words,
                  uint64_t* bitmap_words, uint64_t* granule_bytes) {
    if (!t || which < 0 || which > 3) return fail(VR_ERR_INVALID_ARGUMENT, "tree / array index");
    if (!t->touch_out) return fail(VR_ERR_INVALID_ARGUMENT, "vr_touch_enable(tree, 1) first");
    DeviceGuard guard(t->device);
    std::lock_guard<std::mutex> lock(t->launch_mutex);
    // (an array the tree does not have -- no bricks, say -- has an empty bitmap)
    const uint64_t words = t->touch[which] ? touch_words(t->array_bytes[which], which) : 0;
    if (bitmap_words) *bitmap_words = words;
    if (granule_bytes) *granule_bytes = touch_granule(which);
    const uint64_t n = n_words < words ? n_words : words;
    if (host_words && n) {
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(host_words, t->touch[which], n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    }
    return VR_OK;
}

int vr_render_batch(vr_tree_t t, int n_frames, const VrCamera* cams, const VrRenderOptions* opt,
                    const VrFrame* frames, void* stream) {
    if (!t || !cams || !opt || !frames) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    if (n_frames < 1 || n_frames > VR_MAX_BATCH)
        return fail(VR_ERR_INVALID_ARGUMENT, "n_frames=%d outside [1,%d]", n_frames, VR_MAX_BATCH);
    const VrFrame* f = &frames[0];
    const VrCamera* cam = &cams[0];
    if (f->fp_mode != VR_FP_STRICT && f->fp_mode != VR_FP_FMA)
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown fp_mode %d", f->fp_mode);
    if (f->layout != VR_LAYOUT_FRAME && f->layout != VR_LAYOUT_COMPACT)
        return fail(VR_ERR_INVALID_ARGUMENT, "unknown layout %d", f->layout);
    const int world = f->world < 1 ? 1 : f->world;
    if (f->rank < 0 || f->rank >= world)
        return fail(VR_ERR_INVALID_ARGUMENT, "rank %d outside world %d", f->rank, world);
    // pixel coordinates travel as 16+16 bits, pixel offsets as 32 bits
    if (cam->width < 1 || cam->height < 1 || cam->width > 65535 || cam->height > 65535)
        return fail(VR_ERR_INVALID_ARGUMENT, "image size %dx%d outside [1, 65535]", cam->width,
                    cam->height);
    {
        const int64_t pitch = f->pitch ? f->pitch : (int64_t)cam->width * 4;
        if (pitch < (int64_t)cam->width * 4 || pitch * cam->height >= (1ll << 32))
            return fail(VR_ERR_INVALID_ARGUMENT, "pitch %lld unusable for a %dx%d frame",
                        (long long)pitch, cam->width, cam->height);
    }
    if (!(cam->fx != 0.f) || !(cam->fy != 0.f))
        return fail(VR_ERR_INVALID_ARGUMENT, "focal length must be non-zero");

    bool instrumented = false, any_accum = false;
    for (int i = 0; i < n_frames; ++i) {
        const VrFrame& fi = frames[i];
        const VrCamera& ci = cams[i];
        if (!fi.rgba) return fail(VR_ERR_INVALID_ARGUMENT, "frame %d: rgba is NULL", i);
        // one launch shares everything but the pose and the buffers
        if (ci.width != cam->width || ci.height != cam->height || ci.fx != cam->fx ||
            ci.fy != cam->fy)
            return fail(VR_ERR_INVALID_ARGUMENT, "frame %d: intrinsics differ within the batch", i);
        if (fi.pitch != f->pitch || fi.offscreen != f->offscreen || fi.layout != f->layout ||
            fi.tile_w != f->tile_w || fi.tile_h != f->tile_h || fi.rank != f->rank ||
            fi.world != f->world || fi.fp_mode != f->fp_mode)
            return fail(VR_ERR_INVALID_ARGUMENT, "frame %d: layout/shard/fp_mode differ within the batch", i);
        instrumented = instrumented || fi.counters != nullptr;
        any_accum = any_accum || fi.accum != nullptr;
    }

    // the reference spins forever on step_size <= 0 (rt_core.cuh:108-175: t never advances past
    // a leaf face); the kernel's iteration cap would cut such rays short silently -- refuse.
    if (!(opt->step_size > 0.f))
        return fail(VR_ERR_INVALID_ARGUMENT, "step_size must be positive (got %g)",
                    (double)opt->step_size);

    DeviceGuard device_guard(t->device);
    vr::KParams k;
    memset(&k, 0, sizeof(k));
    fill_tree_params(k, t);
    k.width = cam->width;
    k.height = cam->height;
    k.fx = cam->fx;
    k.fy = cam->fy;
    k.step_size = opt->step_size;
    k.sigma_thresh = opt->sigma_thresh;
    k.stop_thresh = opt->stop_thresh;
    k.background_brightness = opt->background_brightness;
    memcpy(k.bbox, opt->render_bbox, sizeof(k.bbox));
    k.basis_min = opt->basis_minmax[0];
    k.basis_max = opt->basis_minmax[1];
    k.render_depth = opt->render_depth != 0;
    k.enable_probe = opt->enable_probe != 0;
    k.probe_disp_size = opt->probe_disp_size;

    // rodrigues (reference src/cuda/volrend.cu:57-71): angle/axis/cos/sin are
    // uniform over the frame -> once here, with the oracle's rounding sequence
    const float angle = host_norm3(opt->rot_dirs, f->fp_mode == VR_FP_FMA);
    if ((double)angle < 1e-6) {
        k.rot_enabled = 0;
    } else {
        k.rot_enabled = 1;
        for (int i = 0; i < 3; ++i) k.rot_k[i] = opt->rot_dirs[i] / angle;
        k.rot_cos = cosf(angle);
        k.rot_sin = sinf(angle);
    }

    int rc = tile_geometry(cam->width, cam->height, f->tile_w, f->tile_h, &k.tile_w, &k.tile_h,
                           &k.tiles_x, &k.tiles_y);
    if (rc != VR_OK) return rc;
    const int64_t n_tiles = (int64_t)k.tiles_x * k.tiles_y;
    k.rank = f->rank;
    k.world = world;
    k.n_local_tiles = (int32_t)((n_tiles - f->rank + world - 1) / world);
    k.wblocks_per_tile_x = k.tile_w / 8;
    k.wblocks_per_tile = (k.tile_w / 8) * (k.tile_h / 8);
    k.n_wave_blocks = (int64_t)k.n_local_tiles * k.wblocks_per_tile;
    const int64_t total = k.n_wave_blocks * 64 * n_frames;
    if (total >= (1ll << 30))  // ray-buffer fields are addressed with 32-bit byte offsets
        return fail(VR_ERR_INVALID_ARGUMENT, "batch of %lld rays exceeds the 2^30-ray queue",
                    (long long)total);
    k.total_rays = (uint32_t)total;
    k.n_frames = n_frames;
    k.pitch = f->pitch ? f->pitch : (int64_t)cam->width * 4;
    k.offscreen = f->offscreen != 0;
    k.layout = f->layout;
    k.instrumented = instrumented ? 1 : 0;
    k.any_accum = any_accum ? 1 : 0;
    hipStream_t hs = static_cast<hipStream_t>(stream);
    std::unique_lock<std::mutex> guard(t->launch_mutex);
    // vr_touch_enable / vr_touch_count (re)allocate these under the launch mutex: the launch must
    // carry what is current NOW, not what fill_tree_params saw before the lock
    auto refresh_instrumentation = [&]() {
        k.status = t->status;
        k.sched_stats = t->sched_stats;
        for (int i = 0; i < 4; ++i) k.touch[i] = t->touch[i];
    };
    refresh_instrumentation();
    const Tuning tn = t->tn;  // (a copy: the mutex is dropped once below, while a slot grows)
    // lookup structure (top + bricks) beyond 4x the aggregate L2 (8 x 4 MiB on MI355X): the record
    // stream would keep evicting it -- see the DMA loads in vr_kernels.hip
    k.records_nt = tn.records_nt >= 0 ? tn.records_nt
                                      : (t->array_bytes[2] + t->array_bytes[3] > (128ull << 20));
    k.march_max = tn.march_max;
    k.refill_min = tn.refill_min;
    k.drain_flush = tn.drain_flush;
    k.max_iter = tn.max_iter;
    k.frame_group = tn.frame_group < 1 || tn.frame_group > n_frames ? n_frames : tn.frame_group;
    k.super_block = tn.super_block;
    // launch slot: per-launch scratch in device memory (ring, see LaunchSlot)
    const size_t need = ray_buffer_bytes(k.total_rays, ray_tail_words_of(t));
    unsigned slot = kLaunchSlots;
    for (int want_fit = 1; want_fit >= 0 && slot == kLaunchSlots; --want_fit) {
        for (int pass = 0; pass < 2 && slot == kLaunchSlots; ++pass)
            for (unsigned i = 0; i < kLaunchSlots; ++i) {
                const LaunchSlot& c = t->slots[i];
                if (c.growing || (want_fit && c.ray_bytes < need)) continue;
                const bool ok = pass == 0 ? (c.used && c.last_stream == hs)
                                          : (!c.used || hipEventQuery(c.done) == hipSuccess);
                if (ok) {
                    slot = i;
                    break;
                }
            }
    }
    (void)hipGetLastError();  // hipEventQuery's hipErrorNotReady is an answer, not an error
    if (slot == kLaunchSlots) {  // all busy elsewhere: queue up behind one (not one that is growing)
        for (unsigned a = 0; a < kLaunchSlots && slot == kLaunchSlots; ++a)
            if (!t->slots[(t->launch_seq + a) % kLaunchSlots].growing) slot = (t->launch_seq + a) % kLaunchSlots;
        if (slot == kLaunchSlots)
            return fail(VR_ERR_HIP, "all %u launch slots are being resized by other threads", kLaunchSlots);
    }
    t->launch_seq++;
    LaunchSlot& ls = t->slots[slot];
    k.frames = t->slot_frames + (size_t)slot * vr::kMaxBatch;
    k.queue_head = t->slot_heads + kSlotWords * slot + 16;
    k.n_queues = tn.xcd_queues ? 8 : 1;
    k.chunk_max = tn.chunk_max;
    k.basis_words = basis_words_of(t);
    k.ray_tail_words = ray_tail_words_of(t);
    k.ray_vdir = ray_carries_vdir(t) ? 1 : 0;
    if (ls.ray_bytes < need) {
        // First use of the slot, or a larger batch than any before: (re)allocate.  This is the
        // one place where an enqueue-only call may block -- on THIS slot's previous launch
        // only, and hipFree/hipMalloc may synchronise the device; vr_reserve() / vr_reserve_tiles()
        // move it out of the render loop.
        // The wait, the free and the allocation run WITHOUT the launch mutex: the slot is marked
        // `growing` (nobody else picks it) and other threads keep enqueueing on the other slots.
        ls.growing = true;
        uint32_t* old_rays = ls.rays;
        const bool old_used = ls.used;
        ls.rays = nullptr;
        ls.ray_bytes = 0;
        guard.unlock();
        hipError_t ge = hipSuccess;
        if (old_rays) {
            // the slot's last launch must have finished before its buffer goes; if that wait
            // fails the buffer is still freed (hipFree synchronises by itself): only a failing
            // hipMalloc fails the call, and nothing is leaked either way
            if (old_used) (void)hipEventSynchronize(ls.done);
            (void)hipFree(old_rays);
            (void)hipGetLastError();
        }
        uint32_t* new_rays = nullptr;
        ge = hipMalloc((void**)&new_rays, need);
        guard.lock();
        ls.growing = false;
        refresh_instrumentation();  // (the mutex was dropped: see above)
        if (ge != hipSuccess)
            return fail(ge == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP,
                        "ray buffer of %zu bytes: %s", need, hipGetErrorString(ge));
        ls.rays = new_rays;
        ls.ray_bytes = need;
    }
    k.ray_buf_rw = ls.rays;
    k.ray_buf = k.ray_buf_rw;
    // whoever used this slot last (any stream) must have finished before its scratch is rewritten
    if (ls.used) HIP_TRY(hipStreamWaitEvent(hs, ls.done, 0));
    // From here on kernels of this launch may be in the stream: whatever happens below (a later
    // enqueue failing), the slot's event is recorded behind them and the slot is marked used, so
    // that the next user of the slot -- any stream -- waits for whatever did get enqueued.
    struct SlotSeal {
        LaunchSlot& ls;
        hipStream_t hs;
        ~SlotSeal() {
            if (hipEventRecord(ls.done, hs) == hipSuccess) {
                ls.used = true;
                ls.last_stream = hs;
            } else {
                (void)hipGetLastError();
            }
        }
    } seal{ls, hs};
    k.probe_coeffs = t->probe_buf + (size_t)slot * (size_t)t->desc.data_dim;
    if (k.enable_probe)  // launch_renderer's pre-kernel, volrend.cu:202-209
        HIP_TRY(vr::launch_probe(k, opt->probe, const_cast<float*>(k.probe_coeffs), hs));

    // frame table -> device memory, kTableChunk poses per (tiny) kernel
    for (int first = 0; first < n_frames; first += vr::kTableChunk) {
        vr::FrameTable tbl;
        memset(&tbl, 0, sizeof(tbl));
        tbl.first = first;
        tbl.n = n_frames - first < vr::kTableChunk ? n_frames - first : vr::kTableChunk;
        for (int i = 0; i < tbl.n; ++i) {
            memcpy(tbl.f[i].xf, cams[first + i].transform, sizeof(tbl.f[i].xf));
            tbl.f[i].rgba = static_cast<uint8_t*>(frames[first + i].rgba);
            tbl.f[i].accum = frames[first + i].accum;
            tbl.f[i].depth = frames[first + i].depth;
            tbl.f[i].counters = reinterpret_cast<unsigned long long*>(frames[first + i].counters);
        }
        HIP_TRY(vr::launch_prepare(k, tbl, hs));
    }
    // waves per ray-generation workgroup: 16 (one atomic per 1024 pixels) -- except launches of one or
    // two frames, the ones that run beside the tail of a neighbour on another stream: workgroups of
    // 4 waves find room there much earlier (vr_kernels.hip raygen_kernel; profiles/r06_raygen_waves.jsonl:
    // two streams -10 % / -6.5 % at one / two frames per launch, one stream +-0; from four frames on the
    // 4x atomics cost a lone launch 3-4 %, and one-wave workgroups 35 %)
    const int gen_waves = tn.raygen_waves > 0 ? tn.raygen_waves : (n_frames <= 2 ? 4 : 16);
    HIP_TRY(vr::launch_render(k, f->fp_mode, t->n_cus, tn.waves_per_cu, gen_waves, hs));
    return VR_OK;  // (`seal` records the slot's event)
}

int vr_reserve_tiles(vr_tree_t t, int width, int height, int n_frames, int tile_w, int tile_h,
                     int world, int n_slots) {
    if (!t) return fail(VR_ERR_INVALID_ARGUMENT, "tree is NULL");
    if (width < 1 || height < 1 || width > 65535 || height > 65535 || n_frames < 1 ||
        n_frames > VR_MAX_BATCH)
        return fail(VR_ERR_INVALID_ARGUMENT, "vr_reserve(%d x %d, %d frames) out of range", width,
                    height, n_frames);
    if (n_slots < 1 || n_slots > (int)kLaunchSlots)
        return fail(VR_ERR_INVALID_ARGUMENT, "n_slots=%d outside [1,%u]", n_slots, kLaunchSlots);
    if (world < 1) world = 1;
    // exactly the ray count vr_render_batch computes: the rank's tiles, rounded up to WHOLE tiles
    // (rank 0 holds the most)
    int tw, th, tx, ty;
    if (int rc = tile_geometry(width, height, tile_w, tile_h, &tw, &th, &tx, &ty)) return rc;
    const int64_t n_tiles = (int64_t)tx * ty;
    const int64_t local_tiles = (n_tiles + world - 1) / world;
    const int64_t total = local_tiles * (tw / 8) * (th / 8) * 64 * n_frames;
    if (total >= (1ll << 30))
        return fail(VR_ERR_INVALID_ARGUMENT, "batch of %lld rays exceeds the 2^30-ray queue",
                    (long long)total);
    const size_t need = ray_buffer_bytes((uint32_t)total, ray_tail_words_of(t));
    DeviceGuard device_guard(t->device);
    std::lock_guard<std::mutex> guard(t->launch_mutex);
    for (int i = 0; i < n_slots; ++i) {
        LaunchSlot& ls = t->slots[i];
        if (ls.ray_bytes >= need || ls.growing) continue;
        if (ls.rays) {
            if (ls.used) HIP_TRY(hipEventSynchronize(ls.done));
            HIP_TRY(hipFree(ls.rays));
            ls.rays = nullptr;
            ls.ray_bytes = 0;
        }
        HIP_TRY(hipMalloc((void**)&ls.rays, need));
        ls.ray_bytes = need;
    }
    return VR_OK;
}

// two slots of whole frames: what a render loop on one stream (one slot) or on two alternating
// streams needs
int vr_reserve(vr_tree_t t, int width, int height, int n_frames) {
    return vr_reserve_tiles(t, width, height, n_frames, 0, 0, 1, 2);
}

int vr_tree_status(vr_tree_t t, uint32_t* status, int reset) {
    if (!t || !status) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(t->device);
    HIP_TRY(hipMemcpy(status, t->status, sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (reset) HIP_TRY(hipMemset(t->status, 0, sizeof(uint32_t)));
    return VR_OK;
}

int vr_tree_status_on(vr_tree_t t, uint32_t* status, int reset, void* stream) {
    if (!t || !status) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(t->device);
    hipStream_t hs = static_cast<hipStream_t>(stream);
    // a pinned word per calling thread: the copy is asynchronous and ordered on `hs` alone
    thread_local uint32_t* pinned = nullptr;
    if (!pinned) HIP_TRY(hipHostMalloc((void**)&pinned, sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipMemcpyAsync(pinned, t->status, sizeof(uint32_t), hipMemcpyDeviceToHost, hs));
    if (reset) HIP_TRY(hipMemsetAsync(t->status, 0, sizeof(uint32_t), hs));
    HIP_TRY(hipStreamSynchronize(hs));
    *status = *pinned;
    return VR_OK;
}

int vr_render(vr_tree_t t, const VrCamera* cam, const VrRenderOptions* opt, const VrFrame* f,
              void* stream) {
    return vr_render_batch(t, 1, cam, opt, f, stream);
}

int vr_assemble_tiles(void* frame_rgba, int64_t pitch, const void* gathered, int width, int height,
                      int tile_w, int tile_h, int world, void* stream) {
    if (world < 1) world = 1;
    const int64_t rank_bytes = vr_compact_bytes(width, height, tile_w, tile_h, world);
    if (rank_bytes < 0) return VR_ERR_INVALID_ARGUMENT;
    return vr_assemble_tiles_batch(frame_rgba, 0, pitch, gathered, rank_bytes, 0, 1, width, height,
                                   tile_w, tile_h, world, stream);
}

int vr_assemble_tiles_batch(void* frames_rgba, int64_t frame_stride, int64_t pitch,
                            const void* gathered, int64_t rank_stride, int64_t in_frame_stride,
                            int n_frames, int width, int height, int tile_w, int tile_h, int world,
                            void* stream) {
    if (!frames_rgba || !gathered) return fail(VR_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (n_frames < 1) return fail(VR_ERR_INVALID_ARGUMENT, "n_frames must be positive");
    int tw, th, tx, ty;
    int rc = tile_geometry(width, height, tile_w, tile_h, &tw, &th, &tx, &ty);
    if (rc != VR_OK) return rc;
    if (world < 1) world = 1;
    HIP_TRY(vr::launch_assemble(static_cast<uint8_t*>(frames_rgba),
                                pitch ? pitch : (int64_t)width * 4,
                                static_cast<const uint8_t*>(gathered), width, height, tw, th, world,
                                n_frames, frame_stride, rank_stride, in_frame_stride,
                                static_cast<hipStream_t>(stream)));
    return VR_OK;
}

int vr_probe_coeffs(vr_tree_t t, const VrRenderOptions* opt, float* out_dev, void* stream) {
    if (!t || !opt || !out_dev) return fail(VR_ERR_INVALID_ARGUMENT, "NULL argument");
    DeviceGuard guard(t->device);
    vr::KParams k;
    memset(&k, 0, sizeof(k));
    fill_tree_params(k, t);
    HIP_TRY(vr::launch_probe(k, opt->probe, out_dev, static_cast<hipStream_t>(stream)));
    return VR_OK;
}

int vr_read_back(void* host_rgba, const void* dev_rgba, int64_t pitch, int width, int height,
                 void* stream) {
    if (!host_rgba || !dev_rgba) return fail(VR_ERR_INVALID_ARGUMENT, "NULL buffer");
    if (pitch == 0) pitch = (int64_t)width * 4;
    HIP_TRY(hipMemcpy2DAsync(host_rgba, (size_t)width * 4, dev_rgba, (size_t)pitch,
                             (size_t)width * 4, (size_t)height, hipMemcpyDeviceToHost,
                             static_cast<hipStream_t>(stream)));
    return VR_OK;
}

int vr_stream_sync(void* stream) {
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    return VR_OK;
}

}  // extern "C"
