/*
 * vr_oracle.h -- CPU ORACLE for the PlenOctree ray-march hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * renderer's per-pixel algorithm (sxyu/volrend):
 *     src/cuda/volrend.cu:22-173            (ray gen, NDC, rodrigues, composite, quantise)
 *     include/volrend/cuda/rt_core.cuh:17-196   (trace_ray, _dda_world, _dda_unit)
 *     include/volrend/internal/n3tree_query.hpp:13-48  (query_single_from_root)
 *     include/volrend/internal/lumisphere.hpp:9-87     (SH / SG / ASG basis)
 *     include/volrend/cuda/common.cuh:12-55            (_norm, _normalize, _mv3, ...)
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it.  Nothing in volrend_amd/ (the product) links, imports or calls it.
 *
 * PARITY PIN: the reference ships no tests / golden vectors and
 * its CUDA build cannot run here, so the oracle is pinned against the
 * reference's OWN device code compiled for the host (oracle/ref_build ->
 * oracle/_ref/libvolrend_ref.so, strict IEEE evaluation, same deterministic
 * expf): tests/test_oracle_vs_ref.py requires bit-equal fp32 accumulators and
 * RGBA8.  What stays unpinned is CUDA's own libdevice expf and nvcc's FMA
 * contraction choices; `VR_FP_FMA` models the latter.
 */
#ifndef VR_ORACLE_H_
#define VR_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* data_format.hpp:9-15 */
enum { OR_FORMAT_RGBA = 0, OR_FORMAT_SH = 1, OR_FORMAT_SG = 2, OR_FORMAT_ASG = 3 };

/* fp evaluation model */
enum {
    OR_FP_STRICT = 0, /* every C operator rounds once (source semantics, -ffp-contract=off) */
    OR_FP_FMA = 1     /* a*b+c fused where an LLVM/NVPTX-style contraction pass would (nvcc -fmad=true model) */
};

/* internal/data_spec.hpp:23-50 (TreeSpec), host pointers */
typedef struct OrTree {
    const int32_t* child;  /* [capacity*N^3] relative child offsets, 0 = leaf */
    const uint16_t* data;  /* [capacity*N^3*data_dim] IEEE fp16 bits */
    const float* extra;    /* SG: basis_dim*4 floats, ASG: basis_dim*11, else NULL */
    float offset[3];
    float scale[3];
    int32_t N;
    int32_t data_dim;
    int32_t format;        /* OR_FORMAT_* */
    int32_t basis_dim;     /* -1 for RGBA */
    float ndc_width;       /* <= 0: NDC off (data_spec.hpp:47) */
    float ndc_height;
    float ndc_focal;
} OrTree;

/* internal/data_spec.hpp:11-22 (CameraSpec); transform = column-major 4x3 c2w */
typedef struct OrCamera {
    float transform[12];
    int32_t width, height;
    float fx, fy;
} OrCamera;

/* render_options.hpp:11-53 */
typedef struct OrOptions {
    float step_size;
    float sigma_thresh;
    float stop_thresh;
    float background_brightness;
    float render_bbox[6];
    int32_t basis_minmax[2];
    float rot_dirs[3];
    int32_t show_grid;       /* carried, unused by the CUDA kernel */
    int32_t grid_max_depth;
    int32_t render_depth;
    int32_t enable_probe;
    float probe[3];
    int32_t probe_disp_size;
} OrOptions;

/* Access counters == the algorithmic-bytes meter of SURVEY.md 8(d). */
typedef struct OrCounters {
    uint64_t rays;         /* pixels traced */
    uint64_t rays_hit_box; /* rays that entered the march loop */
    uint64_t samples;      /* iterations of rt_core.cuh:108-188 */
    uint64_t child_reads;  /* 4-byte child words read (n3tree_query.hpp:36-37) */
    uint64_t hit_samples;  /* samples with sigma > sigma_thresh */
    uint64_t alg_bytes;    /* sum(4*L + 2 + hit*2*(data_dim-1)) + 4 per pixel */
    uint64_t early_stops;  /* rays ended by stop_thresh */
} OrCounters;

void or_default_options(OrOptions* opt);

/*
 * Render the pixel rectangle [x0,x0+w) x [y0,y0+h) of the frame.
 *   rgba       : full-frame RGBA8 buffer, row pitch = cam->width*4 (may be NULL)
 *   accum      : full-frame float[4] per pixel = trace_ray's out[] BEFORE the
 *                background composite (volrend.cu:150), may be NULL
 *   rgba_init  : existing colour for offscreen=0 compositing (volrend.cu:92-96), or NULL
 *   depth_init : mesh depth for offscreen=0 (volrend.cu:144-146), or NULL
 *   probe_coeffs: from or_probe_coeffs when opt->enable_probe, else NULL
 *   counters   : accumulated (+=) if non-NULL
 * Returns 0 on success.
 */
int or_render(const OrTree* tree, const OrCamera* cam, const OrOptions* opt,
              int fp_mode, int offscreen, int x0, int y0, int w, int h,
              uint8_t* rgba, float* accum, const uint8_t* rgba_init,
              const float* depth_init, const float* probe_coeffs,
              OrCounters* counters, int nthreads);

/* or_render + per-pixel work maps (full-frame uint32 arrays, either may be NULL): the number of
 * samples / of samples above sigma_thresh each pixel's ray took -- the per-ray cost the
 * scheduling studies need (tools/ray_length_study.py). */
int or_render_maps(const OrTree* tree, const OrCamera* cam, const OrOptions* opt,
                   int fp_mode, int offscreen, int x0, int y0, int w, int h,
                   uint8_t* rgba, float* accum, const uint8_t* rgba_init,
                   const float* depth_init, const float* probe_coeffs,
                   OrCounters* counters, int nthreads, uint32_t* samples_map, uint32_t* hits_map);

/* retrieve_cursor_lumisphere_kernel, volrend.cu:175-191: out[data_dim-1] */
void or_probe_coeffs(const OrTree* tree, const OrOptions* opt, float* out);

/* Point query (n3tree_query.hpp:13-48).  xyz is rewritten to leaf-local
 * coordinates; returns the leaf slot index (sub_ptr), cube_sz and depth. */
int64_t or_query(const OrTree* tree, float xyz[3], float* cube_sz, int* depth);

/* deterministic expf shared by spec with the HIP kernel (DESIGN.md "vr_expf") */
float or_expf(float x);
/* fp16 bits -> fp32, exact */
float or_half2float(uint16_t h);
/* SH/SG/ASG basis of a direction (lumisphere.hpp:9-87), out[25] */
void or_basis(const OrTree* tree, const float dir[3], int fp_mode, float out[25]);

#ifdef __cplusplus
}
#endif
#endif
