// volrend::internal::TileShardRenderer -- one process drives N MI355X: every frame is cut into
// interleaved screen tiles (VrFrame.tile_* / rank / world), rank r renders its tiles from its own
// replica of the tree (vr_tree_clone: device to device) into a COMPACT buffer, the RGBA8 tiles
// are gathered to the root GPU with RCCL (libvolrend_gather, include/volrend_gather.h:
// ncclCommInitAll, one grouped ncclSend / ncclRecv per launch: every peer sends its share straight
// over its xGMI link to the root -- the same entry points bench.py --gpus N drives with one
// process per rank) and the root de-interleaves the whole batch with one kernel
// (vr_assemble_tiles_batch).
//
// The reference has no multi-GPU path; this is the native counterpart of
// volrend_amd/dist.py + bench.py --gpus N for the kept volrend_headless CLI (--gpus / --tile).
//
// Pipelining: two buffer sets.  Launch j renders into set j & 1 on each rank's render stream;
// the transfer + assembly of launch j run on per-rank communication streams and overlap the
// rendering of launch j + 1.  Assembled frames of set s are complete on out_stream() in
// enqueue order -- consumers (vr_read_back) enqueue there.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "volrend/n3tree.hpp"
#include "volrend_hip.h"

namespace volrend {
namespace internal {

struct TileShardConfig {
    int n_ranks = 1;
    int first_device = 0;     // ranks use devices first_device .. first_device + n_ranks - 1
    bool share_device = false;  // REHEARSAL on a box with fewer GPUs: every rank on first_device,
                                // tiles move with hipMemcpyAsync instead of RCCL (never a measurement)
    int tile_rows = 8;        // rows per screen tile (multiple of 8); tiles span the image width
    int max_batch = 32;       // poses per launch
};

class TileShardRenderer {
   public:
    // `tree` must be uploaded (on any device); throws std::runtime_error on HIP / RCCL errors
    TileShardRenderer(const N3Tree& tree, int width, int height, const TileShardConfig& cfg);
    ~TileShardRenderer();
    TileShardRenderer(const TileShardRenderer&) = delete;
    TileShardRenderer& operator=(const TileShardRenderer&) = delete;

    // Enqueues launch number `seq` (0, 1, 2, ...): cams[0..n) -> frames(seq & 1)[0..n).
    void render(int seq, const VrCamera* cams, int n, const VrRenderOptions& opt, int fp_mode);
    // Assembled frames of buffer set `set` on the root device: frame i at + i * frame_bytes()
    uint8_t* frames(int set) const { return frames_[set]; }
    size_t frame_bytes() const { return (size_t)width_ * height_ * 4; }
    void* out_stream() const { return comm_stream_[0]; }
    int root_device() const { return device_[0]; }
    // "RCCL <version>, N ranks" or the rehearsal label
    const std::string& transport() const { return transport_; }
    void sync();

   private:
    void init(const N3Tree& tree, const TileShardConfig& cfg);
    void release();
    int n_, width_, height_, tile_w_, tile_h_, max_batch_;
    bool share_, rccl_self_;
    int64_t compact_bytes_ = 0;  // one rank's share of one frame
    std::vector<int> device_;
    std::vector<vr_tree_t> tree_;     // tree_[0] is borrowed when it already lives on device_[0]
    std::vector<bool> owns_tree_;
    std::vector<void*> render_stream_, comm_stream_;
    std::vector<void*> rendered_[2], released_[2];  // events per rank and set
    std::vector<bool> released_used_[2];
    std::vector<uint8_t*> compact_[2];  // per rank (peers; rank 0 only in the self-transfer mode)
    uint8_t* gather_[2] = {nullptr, nullptr};   // root: [n_ranks][max_batch][compact_bytes]
    uint8_t* frames_[2] = {nullptr, nullptr};   // root: [max_batch][H][W][4]
    std::vector<void*> comm_;  // vr_gather_t per rank
    std::string transport_, p2p_note_;
};

}  // namespace internal
}  // namespace volrend
