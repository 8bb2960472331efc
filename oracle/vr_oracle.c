/*
 * vr_oracle.c -- CPU ORACLE driver (TEST INFRASTRUCTURE, see vr_oracle.h).
 *
 * Build: gcc -O2 -std=c11 -ffp-contract=off -mfma -fPIC -shared -pthread
 *        (-ffp-contract=off is mandatory: gcc defaults to =fast and would
 *         fuse on its own; every fusion of the FMA model is an explicit fmaf.)
 */
#define _GNU_SOURCE
#include "vr_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "vr_detmath.h"

#define VR_FMA 0
#include "vr_oracle_core.inc"
#undef VR_FMA
#define VR_FMA 1
#include "vr_oracle_core.inc"
#undef VR_FMA

void or_default_options(OrOptions* o) {
    /* render_options.hpp:11-53 defaults */
    memset(o, 0, sizeof(*o));
    o->step_size = 1e-4f;
    o->sigma_thresh = 1e-2f;
    o->stop_thresh = 1e-2f;
    o->background_brightness = 1.f;
    o->render_bbox[3] = o->render_bbox[4] = o->render_bbox[5] = 1.f;
    o->basis_minmax[0] = 0;
    o->basis_minmax[1] = 24;
    o->grid_max_depth = 4;
    o->probe[2] = 1.f;
    o->probe_disp_size = 100;
}

typedef struct Job {
    const OrTree* tree;
    const OrCamera* cam;
    const OrOptions* opt;
    int fp_mode, offscreen, x0, y0, w, h;
    uint8_t* rgba;
    float* accum;
    const uint8_t* rgba_init;
    const float* depth_init;
    const float* probe_coeffs;
    uint32_t* samples_map; /* optional per-pixel sample / hit-sample counts (or_render_maps) */
    uint32_t* hits_map;
    long next_item; /* atomic work dispenser */
    pthread_mutex_t mu;
    OrCounters total;
} Job;

static void add_counters(OrCounters* a, const OrCounters* b) {
    a->rays += b->rays;
    a->rays_hit_box += b->rays_hit_box;
    a->samples += b->samples;
    a->child_reads += b->child_reads;
    a->hit_samples += b->hit_samples;
    a->alg_bytes += b->alg_bytes;
    a->early_stops += b->early_stops;
}

static void* worker(void* arg) {
    Job* j = (Job*)arg;
    OrCounters local;
    memset(&local, 0, sizeof(local));
    const int chunk = 32; /* pixels per work item */
    const int per_row = (j->w + chunk - 1) / chunk;
    const long n_items = (long)per_row * j->h;
    for (;;) {
        const long item = __atomic_fetch_add(&j->next_item, 1, __ATOMIC_RELAXED);
        if (item >= n_items) break;
        const int y = j->y0 + (int)(item / per_row);
        const int xs = j->x0 + (int)(item % per_row) * chunk;
        const int xe = xs + chunk < j->x0 + j->w ? xs + chunk : j->x0 + j->w;
        for (int x = xs; x < xe; ++x) {
            const uint64_t s0 = local.samples, h0 = local.hit_samples;
            if (j->fp_mode == OR_FP_FMA)
                render_pixel_fma(j->tree, j->cam, j->opt, j->offscreen, x, y, j->rgba, j->accum,
                                 j->rgba_init, j->depth_init, j->probe_coeffs, &local);
            else
                render_pixel_strict(j->tree, j->cam, j->opt, j->offscreen, x, y, j->rgba,
                                    j->accum, j->rgba_init, j->depth_init, j->probe_coeffs,
                                    &local);
            if (j->samples_map)
                j->samples_map[(size_t)y * j->cam->width + x] = (uint32_t)(local.samples - s0);
            if (j->hits_map)
                j->hits_map[(size_t)y * j->cam->width + x] = (uint32_t)(local.hit_samples - h0);
        }
    }
    pthread_mutex_lock(&j->mu);
    add_counters(&j->total, &local);
    pthread_mutex_unlock(&j->mu);
    return NULL;
}

int or_render(const OrTree* tree, const OrCamera* cam, const OrOptions* opt, int fp_mode,
              int offscreen, int x0, int y0, int w, int h, uint8_t* rgba, float* accum,
              const uint8_t* rgba_init, const float* depth_init, const float* probe_coeffs,
              OrCounters* counters, int nthreads) {
    return or_render_maps(tree, cam, opt, fp_mode, offscreen, x0, y0, w, h, rgba, accum, rgba_init,
                          depth_init, probe_coeffs, counters, nthreads, NULL, NULL);
}

int or_render_maps(const OrTree* tree, const OrCamera* cam, const OrOptions* opt, int fp_mode,
                   int offscreen, int x0, int y0, int w, int h, uint8_t* rgba, float* accum,
                   const uint8_t* rgba_init, const float* depth_init, const float* probe_coeffs,
                   OrCounters* counters, int nthreads, uint32_t* samples_map, uint32_t* hits_map) {
    if (!tree || !cam || !opt) return 1;
    if (x0 < 0 || y0 < 0 || w < 0 || h < 0 || x0 + w > cam->width || y0 + h > cam->height)
        return 2;
    if (opt->enable_probe && !probe_coeffs) return 3;
    Job job;
    memset(&job, 0, sizeof(job));
    job.tree = tree;
    job.cam = cam;
    job.opt = opt;
    job.fp_mode = fp_mode;
    job.offscreen = offscreen;
    job.x0 = x0;
    job.y0 = y0;
    job.w = w;
    job.h = h;
    job.rgba = rgba;
    job.accum = accum;
    job.rgba_init = rgba_init;
    job.depth_init = depth_init;
    job.probe_coeffs = probe_coeffs;
    job.samples_map = samples_map;
    job.hits_map = hits_map;
    pthread_mutex_init(&job.mu, NULL);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    if (nthreads == 1) {
        worker(&job);
    } else {
        pthread_t th[256];
        for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, worker, &job);
        for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
    }
    pthread_mutex_destroy(&job.mu);
    if (counters) add_counters(counters, &job.total);
    return 0;
}

void or_probe_coeffs(const OrTree* tree, const OrOptions* opt, float* out) {
    /* volrend.cu:175-191 */
    float cen[3];
    for (int i = 0; i < 3; ++i) cen[i] = tree->offset[i] + tree->scale[i] * opt->probe[i];
    float cube_sz;
    int levels;
    const int64_t leaf = query_strict(tree, cen, &cube_sz, &levels);
    const uint16_t* v = tree->data + leaf * tree->data_dim;
    for (int i = 0; i < tree->data_dim - 1; ++i) out[i] = vr_half_bits_to_float(v[i]);
}

int64_t or_query(const OrTree* tree, float xyz[3], float* cube_sz, int* depth) {
    int levels;
    const int64_t leaf = query_strict(tree, xyz, cube_sz, &levels);
    if (depth) *depth = levels - 1;
    return leaf;
}

float or_expf(float x) { return vr_det_expf(x); }
float or_half2float(uint16_t h) { return vr_half_bits_to_float(h); }

void or_basis(const OrTree* tree, const float dir[3], int fp_mode, float out[25]) {
    for (int i = 0; i < 25; ++i) out[i] = 0.f;
    if (fp_mode == OR_FP_FMA)
        precalc_basis_fma(tree, dir, out);
    else
        precalc_basis_strict(tree, dir, out);
}
