/*
 * volrend_hip.h -- C ABI of the MI355X (gfx950) PlenOctree ray-march library
 * (libvolrend_hip.so).  Plain C: opaque handles, PODs, int error codes.
 *
 * This is the drop-in boundary for ONE path of sxyu/volrend: everything that
 * the reference does on the CUDA device for `volrend_headless`.  Each entry
 * point names the reference interface it replaces (paths relative to the
 * reference tree):
 *
 *   vr_tree_upload / vr_tree_free   N3Tree::load_cuda / free_cuda      src/cuda/n3tree.cu:9-49
 *   vr_tree_upload_quantized,       the codebook decode loop of
 *   vr_decode_quantized             N3Tree::load_npz                   src/n3tree.cpp:279-340
 *   VrTreeDesc                      internal::TreeSpec                 include/volrend/internal/data_spec.hpp:23-50
 *   VrCamera                        internal::CameraSpec + the 48-byte
 *                                   Camera::_update upload             data_spec.hpp:11-22, src/camera.cpp:67-75
 *   VrRenderOptions                 volrend::RenderOptions             include/volrend/render_options.hpp:11-53
 *   vr_render                       volrend::launch_renderer +
 *                                   device::render_kernel              include/volrend/cuda/renderer_kernel.hpp:9-12,
 *                                                                      src/cuda/volrend.cu:78-173,195-245
 *   vr_probe_coeffs                 retrieve_cursor_lumisphere_kernel  src/cuda/volrend.cu:175-191
 *   vr_read_back                    cudaMemcpy2DFromArrayAsync         main_headless.cpp:217-219
 *   vr_last_error / return codes    cuda_assert (print+exit) becomes
 *                                   an error code, the library never
 *                                   exits the process                  src/cuda/common.cu:8-21
 *
 * Asynchrony matches the reference: vr_render only enqueues on `stream`
 * (a hipStream_t passed as void*, NULL = the null stream) and returns.
 * The library owns device copies of trees; callers own output buffers,
 * streams and events.  A tree belongs to the device it was uploaded (or cloned)
 * to; calls that take a tree run on that device whatever the calling thread's
 * current device is and leave the thread's device unchanged, so one host thread
 * may drive the trees of several devices.  Streams and buffers passed with a
 * tree must belong to the tree's device.
 *
 * Launches in flight.  Unlike the reference's launch_renderer, a launch here
 * carries per-launch scratch in device memory (frame table, ray queue, ray
 * buffer, probe coefficients).  The library keeps 8 such launch slots per
 * tree; each slot remembers the last launch that used it with an event, and a
 * later launch that lands on the slot makes ITS stream wait for that event (a
 * device-side wait -- the host never blocks).  A launch takes the slot its own
 * stream used last, else one whose launch has finished, else queues up behind
 * the oldest.  So any number of launches on any number of streams and host
 * threads is safe; more than 8 un-finished launches on more than 8 streams of
 * one tree simply serialise.  The one host-blocking
 * step is the (re)allocation of a slot's ray buffer the first time a slot sees
 * a batch larger than any before -- call vr_reserve() once to take that out of
 * the render loop.
 */
#ifndef VOLREND_HIP_H_
#define VOLREND_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_ABI_VERSION 3

/* ---- error codes ------------------------------------------------------ */
enum {
    VR_OK = 0,
    VR_ERR_INVALID_ARGUMENT = 1,
    VR_ERR_HIP = 2,          /* a HIP runtime call failed; see vr_last_error() */
    VR_ERR_NO_DEVICE = 3,
    VR_ERR_BAD_TREE = 4,     /* child links out of range / cyclic / too deep */
    VR_ERR_UNSUPPORTED = 5,
    VR_ERR_OUT_OF_MEMORY = 6
};

/* DataFormat::format, include/volrend/data_format.hpp:9-15 */
enum { VR_FORMAT_RGBA = 0, VR_FORMAT_SH = 1, VR_FORMAT_SG = 2, VR_FORMAT_ASG = 3 };

/* Floating-point evaluation model of the kernel (DESIGN.md "FP contract").
 * STRICT evaluates the reference source with one rounding per operator and is
 * what the parity pin (host build of the reference) verifies bit for bit;
 * FMA fuses a*b+c the way an nvcc -fmad=true build plausibly does. */
enum { VR_FP_STRICT = 0, VR_FP_FMA = 1 };

#define VR_MAX_BASIS 25 /* VOLREND_GLOBAL_BASIS_MAX, render_options.hpp:6 */

typedef struct VrTreeOpaque* vr_tree_t;

/* Host (or device, see `memory`) view of a loaded tree.npz -- the arguments of
 * N3Tree::load_cuda.  child/data use the reference's flat layout:
 *   child[capacity * N^3]            int32, relative node offsets, 0 = leaf
 *   data [capacity * N^3 * data_dim] IEEE fp16, record = [R..,G..,B.., sigma] */
typedef struct VrTreeDesc {
    const int32_t* child;
    const uint16_t* data;
    const float* extra;      /* SG: basis_dim*4, ASG: basis_dim*11 floats; else NULL */
    uint64_t extra_count;    /* number of floats in `extra` */
    float offset[3];
    float scale[3];
    int32_t N;               /* branching factor per axis */
    int64_t capacity;        /* nodes */
    int32_t data_dim;
    int32_t format;          /* VR_FORMAT_* */
    int32_t basis_dim;       /* -1 for RGBA */
    float ndc_width;         /* <= 0 disables the LLFF NDC warp (data_spec.hpp:47) */
    float ndc_height;
    float ndc_focal;
    int32_t memory;          /* 0: pointers are host memory, 1: device memory */
} VrTreeDesc;

/* Median-cut compressed tree.npz (scripts/compress_octree.py:106-119): instead of
 * `data`, per basis function a 65536-entry RGB codebook + a 16-bit index per slot,
 * the densities, and optionally the first basis functions uncompressed.  Pointers
 * are host or device memory according to VrTreeDesc.memory.  n_slots = capacity*N^3. */
typedef struct VrQuantDesc {
    const uint16_t* quant_colors;  /* fp16 [n_quant, 65536, 3] */
    const uint16_t* quant_map;     /* u16  [n_quant, n_slots] */
    const uint16_t* sigma;         /* fp16 [n_slots] */
    const uint16_t* data_retained; /* fp16 [n_retained, n_slots, 3], or NULL */
    int32_t n_quant;
    int32_t n_retained;
} VrQuantDesc;

typedef struct VrTreeInfo {
    int64_t capacity;
    int32_t N, data_dim, format, basis_dim;
    int32_t max_depth;       /* deepest leaf: child words read = max_depth + 1 */
    int32_t device;
    uint64_t device_bytes;   /* total HBM held for this tree */
    uint64_t leaf_stride;    /* bytes between SH records in the device layout */
    int32_t query_mode;      /* VR_QUERY_LOOKUP / VR_QUERY_DESCENT: how a sample finds its leaf (below) */
    int32_t top_levels;      /* lookup structure built at upload: 2^top_levels cells per axis (0 = none) */
    int32_t brick_levels;    /* 2^brick_levels entries per axis of a brick (0 = no bricks) */
    int32_t brick_blocked;   /* bricks stored in 4 x 4 x 2 line blocks */
} VrTreeInfo;

/* How the kernels resolve query_single_from_root (n3tree_query.hpp:13-48) for a tree:
 *   VR_QUERY_LOOKUP  : N == 2, every leaf within 24 levels and fewer than 2^27 nodes -- integer
 *                      digits of the binary32 coordinate index a top grid + bricks (bit-identical);
 *   VR_QUERY_DESCENT : anything else the format allows (N != 2, deeper trees, larger arrays) -- the
 *                      literal float descent of the reference, level by level.
 * vr_query_mode_for is the rule vr_tree_upload applies (pure host arithmetic; max_depth as in
 * VrTreeInfo: the deepest leaf reads max_depth + 1 child words). */
#define VR_QUERY_LOOKUP 0
#define VR_QUERY_DESCENT 1
int vr_query_mode_for(int N, int max_depth, int64_t capacity);

/* CameraSpec: column-major 4x3 camera-to-world (right, up, back, centre) */
typedef struct VrCamera {
    float transform[12];
    int32_t width, height;
    float fx, fy;
} VrCamera;

/* Field for field volrend::RenderOptions (bools widened to int32). */
typedef struct VrRenderOptions {
    float step_size;
    float sigma_thresh;
    float stop_thresh;
    float background_brightness;
    float render_bbox[6];
    int32_t basis_minmax[2];
    float rot_dirs[3];
    int32_t show_grid;
    int32_t grid_max_depth;
    int32_t render_depth;
    int32_t enable_probe;
    float probe[3];
    int32_t probe_disp_size;
} VrRenderOptions;

/* Which pixels one call renders and where they land.
 * The frame is cut into tile_w x tile_h tiles (row-major tile order; both
 * multiples of 8, or 0/0 for "whole frame is one tile"); this call renders the
 * tiles t with t % world == rank -- the multi-GPU screen-tile shard.
 *   VR_LAYOUT_FRAME  : pixels go to their frame position, rgba + y*pitch + 4*x
 *   VR_LAYOUT_COMPACT: the rank's k-th tile (k = t / world) is stored densely at
 *                      rgba + k*tile_w*tile_h*4, row pitch tile_w*4 (the buffer
 *                      a gather collective sends); use vr_assemble_tiles on the
 *                      gathered buffer. */
enum { VR_LAYOUT_FRAME = 0, VR_LAYOUT_COMPACT = 1 };

typedef struct VrFrame {
    void* rgba;             /* device RGBA8 (byte order R,G,B,A; A = 255) */
    int64_t pitch;          /* bytes per row (FRAME layout); 0 = width*4 */
    const float* depth;     /* device R32F W*H mesh depth, or NULL */
    float* accum;           /* optional device float4 per frame pixel: trace_ray's
                               out[] before the background composite; or NULL */
    int32_t offscreen;      /* 1: composite over background_brightness (headless)
                               0: composite over the RGBA8 already in `rgba` */
    int32_t layout;         /* VR_LAYOUT_* */
    int32_t tile_w, tile_h;
    int32_t rank, world;    /* world <= 1: render everything */
    int32_t fp_mode;        /* VR_FP_* */
    int32_t reserved;
    uint64_t* counters;     /* optional device VrCounters (7 x u64, caller zeroes it):
                               switches to the instrumented kernel flavour; NULL in
                               production */
} VrFrame;

/* Access counters of one or more vr_render calls -- the algorithmic-bytes meter
 * (SURVEY.md 8(d)): what the REFERENCE algorithm reads for these rays, i.e.
 * per sample 4 bytes per child word from the root + 2 (sigma) + 2*(data_dim-1)
 * when sigma > sigma_thresh, plus 4 per pixel written. */
typedef struct VrCounters {
    uint64_t rays, rays_hit_box, samples, child_reads, hit_samples, alg_bytes, early_stops;
} VrCounters;

/* ---- library / device ------------------------------------------------- */
int vr_abi_version(void);
/* Thread-local description of the most recent failure on this thread. */
const char* vr_last_error(void);
int vr_device_count(int* count);
/* Select the device for subsequent calls on this thread (cudaSetDevice,
 * main_headless.cpp:108-111).  device < 0 keeps the current device. */
int vr_set_device(int device);
/* Fills name (<= name_len bytes) with the gcnArchName, e.g. "gfx950:..." */
int vr_device_name(int device, char* name, size_t name_len);

/* ---- tree ------------------------------------------------------------- */
void vr_default_tree_desc(VrTreeDesc* desc);
int vr_tree_upload(const VrTreeDesc* desc, vr_tree_t* out);
/* Same, for a quantised file: desc->data is ignored, the codebooks are decoded on the
 * device (only ~(2*n_quant + 6*n_retained + 2) instead of 2*data_dim bytes per slot
 * cross PCIe and no host loop runs).  The resulting tree is identical to uploading
 * the host-decoded data array. */
int vr_tree_upload_quantized(const VrTreeDesc* desc, const VrQuantDesc* quant, vr_tree_t* out);
/* The decode alone: writes the reference's flat data array [n_slots * data_dim] fp16 to
 * data_out (host or device memory per desc->memory; desc->child/data are not read). */
int vr_decode_quantized(const VrTreeDesc* desc, const VrQuantDesc* quant, uint16_t* data_out);
/* A second copy of an uploaded tree on another (or the same) device of this process: the device
 * layout is copied device to device (hipMemcpyPeer: xGMI between two GPUs of a node), so the
 * file crosses PCIe and is re-laid-out once however many GPUs render it.  The multi-GPU screen
 * tile shard (VrFrame.rank / world) renders from one such replica per GPU.  Synchronous. */
int vr_tree_clone(vr_tree_t src, int device, vr_tree_t* out);
int vr_tree_free(vr_tree_t tree);
int vr_tree_info(vr_tree_t tree, VrTreeInfo* info);

/* ---- render ----------------------------------------------------------- */
void vr_default_options(VrRenderOptions* opt);
void vr_default_frame(VrFrame* frame);
/* Number of bytes of the COMPACT buffer of one rank for the given sharding. */
int64_t vr_compact_bytes(int width, int height, int tile_w, int tile_h, int world);
int vr_render(vr_tree_t tree, const VrCamera* cam, const VrRenderOptions* opt,
              const VrFrame* frame, void* stream);
/* Several poses in ONE launch (the volrend_headless pose loop, main_headless.cpp:207-225,
 * known up front): cams[i] -> frames[i].  All entries must share image size,
 * intrinsics, layout, sharding and fp_mode; only the pose and the buffers differ.
 * A single 800x800 frame cannot fill 256 CUs; a batch can.  1 <= n <= VR_MAX_BATCH. */
#define VR_MAX_BATCH 512
int vr_render_batch(vr_tree_t tree, int n_frames, const VrCamera* cams,
                    const VrRenderOptions* opt, const VrFrame* frames, void* stream);
/* Pre-allocates the ray buffers of two launch slots for batches of up to n_frames whole
 * width x height frames (128-228 bytes per ray): a render loop on one stream lives in one slot,
 * two alternating streams in two, so no later vr_render / vr_render_batch of that size (or
 * smaller, or tile-sharded) on them allocates or blocks.  Optional; synchronous. */
int vr_reserve(vr_tree_t tree, int width, int height, int n_frames);
/* The same for a tile-sharded render loop: the ray buffer of a launch holds its rank's tiles
 * rounded up to WHOLE tiles (tile_w x tile_h as in VrFrame; 0, 0 = whole-frame tiles), which
 * can exceed the frame-rounded size vr_reserve assumes when tile_h does not divide the height.
 * n_slots launch slots are sized (1..8: one per stream that renders this tree concurrently). */
int vr_reserve_tiles(vr_tree_t tree, int width, int height, int n_frames, int tile_w, int tile_h,
                     int world, int n_slots);
/* Sticky device status word of the tree's launches: bit 0 = some ray hit the sample guard (a
 * wave marched 2^22 rounds -- tuning key "max_iter" -- since the last retire / refill pass in
 * which one of its rays retired; the reference would still be looping).  Synchronous; reset != 0 clears it.
 * vr_render* refuses step_size <= 0 / NaN (VR_ERR_INVALID_ARGUMENT), where the reference
 * hangs, so the bit only ever fires on pathological step_size / scene combinations.  A launch
 * that set it has WRONG pixels (the wave's marching rays were cut; which rays share a wave
 * depends on the scheduling knobs, so such a frame is not tuning-independent either): vr_render*
 * is enqueue-only and cannot report it, so every render loop checks this word once its last
 * launch has finished -- volrend_headless, TileShardRenderer::sync, VolumeRenderer::read_frame
 * and bench.py do, and fail loudly (the reference's abort-on-error convention,
 * src/cuda/common.cu:8-21). */
int vr_tree_status(vr_tree_t tree, uint32_t* status, int reset);
/* The same word read ON A STREAM: the copy (into pinned memory of the library) and, with
 * reset != 0, the clear are enqueued on `stream`; then THAT stream is waited for and the value
 * returned.  No other stream is waited for -- vr_tree_status copies on the legacy stream and so
 * waits for every blocking stream of the device: a render loop on two alternating streams that
 * checks frame k would wait for frame k + 1 as well.  The word is the TREE's: launches of the same
 * tree on other streams set the same bits, and whichever check comes first reports them. */
int vr_tree_status_on(vr_tree_t tree, uint32_t* status, int reset, void* stream);
/* Scheduling / layout knobs ("march_max", "refill_min", "waves_per_cu", "records_nt", "raygen_waves",
 * "top_levels", "brick_levels", "brick_blocked", ...); results never depend on them ("max_iter", the sample
 * guard above, is the exception by design: a launch that trips it says so in vr_tree_status).  Every tree carries its own
 * copy, taken at upload (or from the source of a clone) from the process defaults:
 *   vr_set_tuning       changes the DEFAULTS of trees uploaded afterwards (serialised);
 *   vr_tree_set_tuning  changes one tree (not the upload-time keys top_levels / brick_levels /
 *                       brick_blocked: VR_ERR_INVALID_ARGUMENT);
 *                       takes effect with that tree's next launch, any thread. */
int vr_set_tuning(const char* key, int value);
int vr_tree_set_tuning(vr_tree_t tree, const char* key, int value);
/* Scheduling tallies accumulated by instrumented launches (frames with counters):
 * [0] march rounds [1] lanes busy in them [2] shade rounds [3] lanes busy in them
 * [4] distinct leaves summed over shade rounds [5] retire rounds [6] rays retired in them [7] scheduler iterations.
 * Synchronous (copies from the device); reset != 0 clears them. */
int vr_sched_stats(vr_tree_t tree, uint64_t out[8], int reset);
/* Distinct-line meter (SURVEY.md 8(d) "B_unique", the compulsory-traffic lower bound): with
 * enable != 0 the tree gets one bit per 128-byte line of its device arrays, and every
 * INSTRUMENTED launch (frames with counters) sets the bits of the lines its accesses touch.
 * vr_touch_count returns the number of distinct lines touched since the last reset in
 * out[0..3] = coefficient records, child words, top grid, bricks (x 128 = bytes).  Both calls
 * are synchronous (device-wide).  Production launches never see the bitmaps. */
int vr_touch_enable(vr_tree_t tree, int enable);
int vr_touch_count(vr_tree_t tree, uint64_t out[4], int reset);
/* The bitmap of array `which` (0 records, 1 child words, 2 top grid, 3 bricks) itself, copied to
 * host memory: bit b of word w = granule 32 w + b of the array was touched; *granule_bytes (if
 * not NULL) receives the bytes one bit stands for (128; layout studies build the library with
 * a finer grain for the records).  Copies min(n_words, size of the bitmap) words and returns
 * the size of the bitmap in *bitmap_words (if not NULL).  Synchronous. */
int vr_touch_read(vr_tree_t tree, int which, uint32_t* host_words, uint64_t n_words,
                  uint64_t* bitmap_words, uint64_t* granule_bytes);
/* gathered = world consecutive COMPACT buffers (rank-major), all device memory
 * on the current device.  Writes the W x H frame. */
int vr_assemble_tiles(void* frame_rgba, int64_t pitch, const void* gathered, int width,
                      int height, int tile_w, int tile_h, int world, void* stream);
/* The same for n_frames at once (one launch): frame i is written at
 * frames_rgba + i*frame_stride; rank r's COMPACT buffer of frame i is read at
 * gathered + r*rank_stride + i*in_frame_stride -- e.g. a [world][n_frames][compact_bytes]
 * gather result has rank_stride = n_frames*compact_bytes, in_frame_stride = compact_bytes. */
int vr_assemble_tiles_batch(void* frames_rgba, int64_t frame_stride, int64_t pitch,
                            const void* gathered, int64_t rank_stride, int64_t in_frame_stride,
                            int n_frames, int width, int height, int tile_w, int tile_h, int world,
                            void* stream);
/* out_dev: device float[data_dim-1]; the lumisphere at opt->probe. */
int vr_probe_coeffs(vr_tree_t tree, const VrRenderOptions* opt, float* out_dev, void* stream);
/* Async D2H of a pitched RGBA8 frame into tightly packed host memory. */
int vr_read_back(void* host_rgba, const void* dev_rgba, int64_t pitch, int width, int height,
                 void* stream);
int vr_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VOLREND_HIP_H_ */
