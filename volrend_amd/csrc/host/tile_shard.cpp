// tile_shard.cpp -- see include/volrend/internal/tile_shard.hpp.
#include "volrend/internal/tile_shard.hpp"

#include <hip/hip_runtime.h>

#include "volrend_gather.h"

#include <cstdlib>
#include <stdexcept>

namespace volrend {
namespace internal {
namespace {

void hip_ok(hipError_t e, const char* what) {
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void gather_ok(int rc, const char* what) {  // libvolrend_gather: the shard's RCCL collective
    if (rc != 0) throw std::runtime_error(std::string(what) + ": " + vr_gather_last_error());
}
void vr_ok(int rc, const char* what) {
    if (rc != VR_OK) throw std::runtime_error(std::string(what) + ": " + vr_last_error());
}
hipStream_t hs(void* s) { return static_cast<hipStream_t>(s); }
hipEvent_t he(void* e) { return static_cast<hipEvent_t>(e); }
vr_gather_t gc(void* c) { return static_cast<vr_gather_t>(c); }

}  // namespace

TileShardRenderer::TileShardRenderer(const N3Tree& tree, int width, int height,
                                     const TileShardConfig& cfg)
    : n_(cfg.n_ranks < 1 ? 1 : cfg.n_ranks),
      width_(width),
      height_(height),
      tile_w_((width + 7) / 8 * 8),
      tile_h_(cfg.tile_rows < 8 ? 8 : cfg.tile_rows / 8 * 8),
      max_batch_(cfg.max_batch < 1 ? 1 : (cfg.max_batch > VR_MAX_BATCH ? VR_MAX_BATCH : cfg.max_batch)),
      share_(cfg.share_device && n_ > 1),
      rccl_self_(n_ == 1) {
    try {
        init(tree, cfg);
    } catch (...) {
        release();  // a constructor that throws gets no destructor call
        throw;
    }
}

void TileShardRenderer::init(const N3Tree& tree, const TileShardConfig& cfg) {
    if (!tree.device) throw std::runtime_error("TileShardRenderer: the tree is not on a device");
    int n_dev = 0;
    hip_ok(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
    for (int r = 0; r < n_; ++r) device_.push_back(share_ ? cfg.first_device : cfg.first_device + r);
    if (cfg.first_device < 0 || device_.back() >= n_dev)
        throw std::runtime_error("TileShardRenderer: " + std::to_string(n_) + " ranks from device " +
                                 std::to_string(cfg.first_device) + " need more GPUs than the " +
                                 std::to_string(n_dev) + " visible (--share_gpu rehearses on one)");
    compact_bytes_ = vr_compact_bytes(width_, height_, tile_w_, tile_h_, n_);
    if (compact_bytes_ <= 0) throw std::runtime_error("TileShardRenderer: bad tile geometry");

    // The gather sends every peer's tiles straight to the root GPU: that needs peer access
    // (xGMI inside a node) between the root and each peer.  Without it RCCL falls back to
    // staging through host memory -- correct but no longer the path this class exists for --
    // so say so loudly instead of silently running 10x slower.  VOLREND_ALLOW_NO_P2P=1 proceeds.
    if (!share_) {
        std::string no_p2p;
        for (int r = 1; r < n_; ++r) {
            int to_root = 0, from_root = 0;
            hip_ok(hipDeviceCanAccessPeer(&to_root, device_[r], device_[0]), "hipDeviceCanAccessPeer");
            hip_ok(hipDeviceCanAccessPeer(&from_root, device_[0], device_[r]), "hipDeviceCanAccessPeer");
            if (!to_root || !from_root) no_p2p += " " + std::to_string(device_[r]);
        }
        const char* allow = getenv("VOLREND_ALLOW_NO_P2P");
        if (!no_p2p.empty() && !(allow && allow[0] == '1'))
            throw std::runtime_error(
                "TileShardRenderer: no peer access between the root GPU " + std::to_string(device_[0]) +
                " and GPU(s)" + no_p2p + " (hipDeviceCanAccessPeer = 0): the RGBA8 gather and the "
                "tree replicas would be staged through host memory.  Check `rocm-smi --showtopo`, "
                "IOMMU / ACS settings and HSA_ENABLE_IPC_MODE_LEGACY=0; set VOLREND_ALLOW_NO_P2P=1 to "
                "run anyway, or --share_gpu to rehearse on one device");
        p2p_note_ = no_p2p.empty() ? "peer access to the root: yes"
                                   : "NO peer access for GPU(s)" + no_p2p + " (host-staged)";
    }
    int prev = 0;
    hip_ok(hipGetDevice(&prev), "hipGetDevice");
    VrTreeInfo info;
    vr_ok(vr_tree_info(tree.device, &info), "vr_tree_info");
    tree_.assign(n_, nullptr);
    owns_tree_.assign(n_, false);
    render_stream_.assign(n_, nullptr);
    comm_stream_.assign(n_, nullptr);
    for (int s = 0; s < 2; ++s) {
        rendered_[s].assign(n_, nullptr);
        released_[s].assign(n_, nullptr);
        released_used_[s].assign(n_, false);
        compact_[s].assign(n_, nullptr);
    }
    for (int r = 0; r < n_; ++r) {
        hip_ok(hipSetDevice(device_[r]), "hipSetDevice");
        // one replica per rank; the caller's copy serves the root when it already lives there
        if (r == 0 && info.device == device_[0]) {
            tree_[r] = tree.device;
        } else {
            vr_ok(vr_tree_clone(tree.device, device_[r], &tree_[r]), "vr_tree_clone");
            owns_tree_[r] = true;
        }
        // the rank's tiles, rounded up to whole tiles, one slot (one render stream per rank)
        vr_ok(vr_reserve_tiles(tree_[r], width_, height_, max_batch_, tile_w_, tile_h_, n_, 1),
              "vr_reserve_tiles");
        hipStream_t st;
        hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        render_stream_[r] = st;
        hip_ok(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
        comm_stream_[r] = st;
        for (int s = 0; s < 2; ++s) {
            hipEvent_t ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            rendered_[s][r] = ev;
            hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
            released_[s][r] = ev;
            if (r > 0 || rccl_self_)
                hip_ok(hipMalloc((void**)&compact_[s][r], (size_t)compact_bytes_ * max_batch_),
                       "hipMalloc(compact)");
        }
    }
    hip_ok(hipSetDevice(device_[0]), "hipSetDevice");
    for (int s = 0; s < 2; ++s) {
        hip_ok(hipMalloc((void**)&gather_[s], (size_t)compact_bytes_ * max_batch_ * n_),
               "hipMalloc(gather)");
        hip_ok(hipMalloc((void**)&frames_[s], frame_bytes() * max_batch_), "hipMalloc(frames)");
    }
    if (!share_) {
        std::vector<vr_gather_t> comms(n_, nullptr);
        gather_ok(vr_gather_init_all(n_, device_.data(), comms.data()), "vr_gather_init_all");
        for (auto c : comms) comm_.push_back(c);
        transport_ = "RCCL " + std::to_string(vr_gather_version()) + ", " + std::to_string(n_) +
                     (n_ == 1 ? " rank (self send/recv)" : " ranks, grouped send/recv to the root") +
                     (n_ > 1 ? ", " + p2p_note_ : "");
    } else {
        transport_ = "REHEARSAL: " + std::to_string(n_) + " ranks share device " +
                     std::to_string(device_[0]) + ", tiles move with hipMemcpyAsync (no RCCL)";
    }
    hip_ok(hipSetDevice(prev), "hipSetDevice");
}

TileShardRenderer::~TileShardRenderer() { release(); }

void TileShardRenderer::release() {
    int prev = 0;
    (void)hipGetDevice(&prev);
    const int n = (int)device_.size();  // (a failed construction may have got this far only)
    for (int r = 0; r < n; ++r) {
        (void)hipSetDevice(device_[r]);
        (void)hipDeviceSynchronize();
    }
    for (void* c : comm_) (void)vr_gather_free(gc(c));
    comm_.clear();
    auto at = [](const auto& v, int r) { return r < (int)v.size() ? v[r] : nullptr; };
    for (int r = 0; r < n; ++r) {
        (void)hipSetDevice(device_[r]);
        for (int s = 0; s < 2; ++s) {
            if (at(compact_[s], r)) (void)hipFree(compact_[s][r]);
            if (at(rendered_[s], r)) (void)hipEventDestroy(he(rendered_[s][r]));
            if (at(released_[s], r)) (void)hipEventDestroy(he(released_[s][r]));
        }
        if (at(render_stream_, r)) (void)hipStreamDestroy(hs(render_stream_[r]));
        if (at(comm_stream_, r)) (void)hipStreamDestroy(hs(comm_stream_[r]));
        if (r < (int)owns_tree_.size() && owns_tree_[r] && tree_[r]) (void)vr_tree_free(tree_[r]);
    }
    for (int s = 0; s < 2; ++s) {
        compact_[s].clear();
        rendered_[s].clear();
        released_[s].clear();
    }
    render_stream_.clear();
    comm_stream_.clear();
    tree_.clear();
    owns_tree_.clear();
    if (n > 0) (void)hipSetDevice(device_[0]);
    for (int s = 0; s < 2; ++s) {
        if (gather_[s]) (void)hipFree(gather_[s]);
        if (frames_[s]) (void)hipFree(frames_[s]);
        gather_[s] = frames_[s] = nullptr;
    }
    device_.clear();
    (void)hipSetDevice(prev);
}

void TileShardRenderer::render(int seq, const VrCamera* cams, int n, const VrRenderOptions& opt,
                               int fp_mode) {
    if (n < 1 || n > max_batch_) throw std::invalid_argument("TileShardRenderer::render: batch size");
    const int s = seq & 1;
    const size_t share = (size_t)compact_bytes_;
    const size_t rank_stride = share * max_batch_;
    int prev = 0;
    hip_ok(hipGetDevice(&prev), "hipGetDevice");
    std::vector<VrFrame> frames((size_t)n);
    // 1. every rank renders its tiles of the n poses into its COMPACT buffer of set s
    for (int r = 0; r < n_; ++r) {
        hip_ok(hipSetDevice(device_[r]), "hipSetDevice");
        uint8_t* dst = (r == 0 && !rccl_self_) ? gather_[s] : compact_[s][r];
        // set s is free again once the transfer (root: the assembly) of launch seq - 2 is done
        if (released_used_[s][r])
            hip_ok(hipStreamWaitEvent(hs(render_stream_[r]), he(released_[s][r]), 0),
                   "hipStreamWaitEvent");
        for (int i = 0; i < n; ++i) {
            vr_default_frame(&frames[i]);
            frames[i].rgba = dst + share * i;
            frames[i].offscreen = 1;
            frames[i].layout = VR_LAYOUT_COMPACT;
            frames[i].tile_w = tile_w_;
            frames[i].tile_h = tile_h_;
            frames[i].rank = r;
            frames[i].world = n_;
            frames[i].fp_mode = fp_mode;
        }
        vr_ok(vr_render_batch(tree_[r], n, cams, &opt, frames.data(), render_stream_[r]),
              "vr_render_batch");
        hip_ok(hipEventRecord(he(rendered_[s][r]), hs(render_stream_[r])), "hipEventRecord");
        hip_ok(hipStreamWaitEvent(hs(comm_stream_[r]), he(rendered_[s][r]), 0), "hipStreamWaitEvent");
    }
    // 2. the tiles travel to the root: rank r's n shares land at gather + r * rank_stride
    if (!share_) {
        // (the same entry point bench.py --gpus N drives, one rank per process there: include/volrend_gather.h)
        gather_ok(vr_gather_group_begin(), "vr_gather_group_begin");
        for (int r = 0; r < n_; ++r)
            gather_ok(vr_gather_tiles(gc(comm_[r]), compact_[s][r], gather_[s], (int64_t)rank_stride,
                                      (int64_t)(share * n), rccl_self_ ? 1 : 0, comm_stream_[r]),
                      "vr_gather_tiles");
        gather_ok(vr_gather_group_end(), "vr_gather_group_end");
    } else {
        hip_ok(hipSetDevice(device_[0]), "hipSetDevice");
        for (int r = 1; r < n_; ++r) {
            // same device: the root's communication stream copies once rank r has rendered
            hip_ok(hipStreamWaitEvent(hs(comm_stream_[0]), he(rendered_[s][r]), 0),
                   "hipStreamWaitEvent");
            hip_ok(hipMemcpyAsync(gather_[s] + rank_stride * r, compact_[s][r], share * n,
                                  hipMemcpyDeviceToDevice, hs(comm_stream_[0])), "hipMemcpyAsync");
        }
    }
    // 3. the root de-interleaves the batch; 4. the buffers of set s are released
    hip_ok(hipSetDevice(device_[0]), "hipSetDevice");
    vr_ok(vr_assemble_tiles_batch(frames_[s], (int64_t)frame_bytes(), 0, gather_[s],
                                  (int64_t)rank_stride, (int64_t)share, n, width_, height_, tile_w_,
                                  tile_h_, n_, comm_stream_[0]),
          "vr_assemble_tiles_batch");
    for (int r = 0; r < n_; ++r) {
        hip_ok(hipSetDevice(device_[r]), "hipSetDevice");
        // shared device: rank r's buffer is read by the ROOT's stream
        hipStream_t after = hs(comm_stream_[share_ ? 0 : r]);
        hip_ok(hipEventRecord(he(released_[s][r]), after), "hipEventRecord");
        released_used_[s][r] = true;
    }
    hip_ok(hipSetDevice(prev), "hipSetDevice");
}

void TileShardRenderer::sync() {
    int prev = 0;
    hip_ok(hipGetDevice(&prev), "hipGetDevice");
    for (int r = 0; r < n_; ++r) {
        hip_ok(hipSetDevice(device_[r]), "hipSetDevice");
        hip_ok(hipStreamSynchronize(hs(render_stream_[r])), "hipStreamSynchronize");
        hip_ok(hipStreamSynchronize(hs(comm_stream_[r])), "hipStreamSynchronize");
    }
    hip_ok(hipSetDevice(prev), "hipSetDevice");
    // every launch has run: a rank whose rays hit the sample guard rendered wrong tiles.  ALL ranks'
    // words are read and cleared before anything is thrown -- a later sync() must not trip over
    // stale bits of these launches
    std::string failed;
    for (int r = 0; r < n_; ++r) {
        uint32_t status = 0;
        vr_ok(vr_tree_status(tree_[r], &status, 1), "vr_tree_status");
        if (status != 0)
            failed += (failed.empty() ? "" : ", ") + std::to_string(r) + " (0x" + std::to_string(status) + ")";
    }
    if (!failed.empty())
        throw std::runtime_error("tile shard: render status of rank(s) " + failed +
                                 " (rays hit the sample guard: step_size too small for this scene?); "
                                 "the frames are wrong");
}

}  // namespace internal
}  // namespace volrend
