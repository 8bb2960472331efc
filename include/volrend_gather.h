/* volrend_gather.h -- C ABI of libvolrend_gather.so: the ONE collective of the screen-tile shard.
 *
 * Every rank renders its interleaved tiles of a launch's frames into a dense COMPACT buffer
 * (VrFrame.layout, include/volrend_hip.h); one gather per launch moves the buffers to the root
 * (rank 0): every peer sends its share straight to the root -- over its own xGMI link inside a
 * node -- as ONE grouped ncclSend / ncclRecv (RCCL); the root then de-interleaves the batch with
 * vr_assemble_tiles_batch.  The reference has no multi-GPU path (main_headless.cpp:108-111: one
 * cudaSetDevice); north_star asks for "an RCCL gather of the RGBA8 output over xGMI".
 *
 * Both product callers go through these entry points:
 *   volrend_headless --gpus N   one process drives N devices: vr_gather_init_all, and per launch
 *                               vr_gather_group_begin / vr_gather_tiles per rank / _group_end
 *                               (volrend::internal::TileShardRenderer);
 *   bench.py --gpus N           one process per GPU: rank 0 makes the id (vr_gather_unique_id),
 *                               the launcher's rendezvous carries its 128 bytes to the others,
 *                               every rank calls vr_gather_init_rank and, per launch,
 *                               vr_gather_tiles (volrend_amd/gather.py).
 * Enqueue-only like vr_render*: the transfer is ordered on the stream it is given.  Plain pointers
 * and sizes; errors as return codes (0 = ok) + vr_gather_last_error().  The library links RCCL
 * and the HIP runtime, nothing else: libvolrend_hip.so stays free of RCCL for single-GPU users. */
#ifndef VOLREND_GATHER_H
#define VOLREND_GATHER_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VR_GATHER_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct VrGatherOpaque* vr_gather_t;

const char* vr_gather_last_error(void);
/* RCCL's version code (ncclGetVersion), e.g. 22703 */
int vr_gather_version(void);

/* One process per rank: rank 0 creates the id, every rank (0 included) initialises with the same
 * 128 bytes.  `device` = the HIP device this rank renders on (collective call: returns once every
 * rank has joined). */
int vr_gather_unique_id(void* id_out);
int vr_gather_init_rank(const void* id, int rank, int world, int device, vr_gather_t* out);
/* One process, n ranks on devices[0..n): out[0..n) (ncclCommInitAll). */
int vr_gather_init_all(int n, const int* devices, vr_gather_t* out);
int vr_gather_free(vr_gather_t g);
int vr_gather_rank(vr_gather_t g);
int vr_gather_world(vr_gather_t g);

/* This rank's part of ONE gather to rank 0, enqueued on `stream` (a hipStream_t of the rank's
 * device):
 *   rank r > 0 : sends `bytes` bytes at `send` to the root;
 *   rank 0     : receives rank r's bytes at recv_base + r * rank_stride for r = 1..world-1; its
 *                own share needs no transfer (it renders straight into recv_base) unless
 *                self_transfer != 0: then it also sends `send` to itself and receives it at
 *                recv_base -- the one-rank run of the collective path (volrend_headless --gpus 1,
 *                VOLREND_FORCE_GATHER=1).
 * The calls of one rank form one RCCL group.  A process that drives several ranks brackets the
 * ranks' calls of one launch with vr_gather_group_begin / _end (ncclGroupStart / End). */
int vr_gather_tiles(vr_gather_t g, const void* send, void* recv_base, int64_t rank_stride,
                    int64_t bytes, int self_transfer, void* stream);
int vr_gather_group_begin(void);
int vr_gather_group_end(void);

#ifdef __cplusplus
}
#endif
#endif
