// volrend_headless -- offscreen PlenOctree rendering on MI355X.
// Same command line, pose / intrinsics file formats, PNG naming and the two result lines
// ("%.10f ms per frame", "%.10f fps") as the reference's main_headless.cpp; the device
// work goes through the C ABI (include/volrend_hip.h).  Poses are known up front, so they
// are rendered in batches of --batch frames per launch (default 32).
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <deque>
#include <fstream>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <sys/stat.h>

#include <memory>

#include "volrend/internal/imwrite.hpp"
#include "volrend/internal/opts.hpp"
#include "volrend/internal/tile_shard.hpp"
#include "volrend/n3tree.hpp"
#include "volrend/renderer_kernel.hpp"

namespace {

#define HIP_OK(expr)                                                                   \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, \
                    __LINE__);                                                         \
            std::exit(1);                                                              \
        }                                                                              \
    } while (0)

std::string path_basename(const std::string& str) {
    const size_t p = str.find_last_of("/\\");
    return p == std::string::npos ? str : str.substr(p + 1);
}

std::string remove_ext(const std::string& str) {
    const size_t p = str.find_last_of('.');
    return p == std::string::npos ? str : str.substr(0, p);
}

// Every leading float of a text file (parsing stops at the first token that is not a number,
// which is where formatted stream extraction would stop, too).
std::vector<float> leading_floats(const std::string& path, const char* what) {
    std::ifstream ifs(path);
    if (!ifs) {
        fprintf(stderr, "ERROR: %s'%s' does not exist\n", what, path.c_str());
        std::exit(1);
    }
    std::vector<float> v;
    for (float x; ifs >> x;) v.push_back(x);
    return v;
}

// A pose file holds one or more row-major matrices (reference main_headless.cpp:40-63): twelve
// numbers are the three rows of a camera-to-world matrix; when more numbers follow, the next four
// are taken to be its 4th row and skipped (so stacked matrices must be 4x4).  A matrix whose
// first row is complete counts even if the file ends inside it (missing entries = 0).  Each
// becomes a column-major 4x3: right, up, back, centre.
int read_transform_matrices(const std::string& path, std::vector<glm::mat4x3>& out) {
    const std::vector<float> v = leading_floats(path, "");
    int cnt = 0;
    for (size_t at = 0; v.size() - at >= 4;) {
        const size_t have = std::min<size_t>(12, v.size() - at);
        glm::mat4x3 m{};
        for (size_t k = 0; k < have; ++k) m[(int)(k % 4)][(int)(k / 4)] = v[at + k];
        out.push_back(m);
        ++cnt;
        if (have < 12) break;
        at += 12 + std::min<size_t>(4, v.size() - at - 12);
    }
    return cnt;
}

// intrinsics.txt is a 4x4 K matrix; only K[0][0] and K[1][1] are used (main_headless.cpp:65-75).
void read_intrins(const std::string& path, float& fx, float& fy) {
    const std::vector<float> v = leading_floats(path, "intrin ");
    fx = v.empty() ? 0.f : v[0];
    if (v.size() >= 6) fy = v[5];
    else if (v.size() == 5) fy = 0.f;  // the extraction that runs into the end of file zeroes its target
}

// Frame egress (reference main_headless.cpp:216-222 writes each PNG inline and calls it
// "a huge bottleneck", README.md:128): PNG encoding runs on worker threads while the GPU
// renders the next batch; frames come back through pinned, double-buffered host memory.
class EncodePool {
   public:
    explicit EncodePool(unsigned n) {
        for (unsigned i = 0; i < n; ++i) workers_.emplace_back([this] { run(); });
    }
    ~EncodePool() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    void submit(int group, std::function<void()> fn) {
        {
            std::lock_guard<std::mutex> lk(mu_);
            pending_[group]++;
            tasks_.push_back({group, std::move(fn)});
        }
        cv_.notify_one();
    }
    void wait(int group) {  // until every task of this buffer set is done
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [&] { return pending_[group] == 0; });
    }

   private:
    struct Task {
        int group;
        std::function<void()> fn;
    };
    void run() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !tasks_.empty(); });
                if (tasks_.empty()) return;
                t = std::move(tasks_.front());
                tasks_.pop_front();
            }
            t.fn();
            {
                std::lock_guard<std::mutex> lk(mu_);
                pending_[t.group]--;
            }
            done_cv_.notify_all();
        }
    }
    std::vector<std::thread> workers_;
    std::deque<Task> tasks_;
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    int pending_[2] = {0, 0};
    bool stop_ = false;
};

void make_dirs(const std::string& path) {
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) mkdir(cur.c_str(), 0755);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

// How the pose list is cut into launches.  A launch pays one ray-chain latency of ramp-up + tail
// whatever it carries, and under --gpus N every rank only renders 1/N of each frame: the launch
// grows with N so that the work per rank and launch does not shrink (bench.py --gpus N does the
// same).  The list is cut into ceil(P / batch) EQUAL launches (sizes differ by at most one pose:
// no short last launch), the first `n_long` of them carry `batch` poses, the rest one less.
struct LaunchPlan {
    int batch = 1;          // poses of a long launch
    size_t n_launches = 0;
    size_t n_long = 0;
    int n_streams = 1;      // render streams the launches alternate between
};
LaunchPlan plan_launches(size_t n_poses, int batch_arg, int n_gpus, int streams_arg) {
    LaunchPlan p;
    int batch = batch_arg < 1 ? 1 : batch_arg;
    if (n_gpus > 1) batch = batch > VR_MAX_BATCH / n_gpus ? VR_MAX_BATCH : batch * n_gpus;
    if (batch > VR_MAX_BATCH) batch = VR_MAX_BATCH;
    if (n_poses == 0) return p;
    p.n_launches = (n_poses + (size_t)batch - 1) / (size_t)batch;
    p.batch = (int)((n_poses + p.n_launches - 1) / p.n_launches);
    p.n_long = n_poses - (size_t)(p.batch - 1) * p.n_launches;
    // A launch drains for ~0.25 ms while its longest rays finish; on a second stream the next
    // launch starts under that tail (auto: below 48 poses per launch; at 50 the two are a tie).
    p.n_streams = streams_arg <= 0 ? (p.batch < 48 ? 2 : 1) : (streams_arg > 2 ? 2 : streams_arg);
    if (n_gpus >= 1) p.n_streams = 1;  // the tile shard brings its own streams
    return p;
}

}  // namespace

int main(int argc, char* argv[]) {
    using namespace volrend;
    internal::Options args("volrend_headless",
                           "Headless PlenOctree volume rendering on MI355X (HIP)");
    internal::add_common_opts(args);
    args.add("write_images", 'o', false, "",
             "output directory of images; if empty, DOES NOT save (for timing only)");
    args.add("intrin", 'i', false, "", "intrinsics matrix 4x4; if set, overrides the fx/fy");
    args.add("reverse_yz", 'r', true, "", "use OpenCV camera space convention instead of NeRF");
    args.add("scale", 0, false, "1.0", "scaling to apply to image");
    args.add("max_imgs", 0, false, "0", "max images to render, default no limit");
    args.add("batch", 0, false, "64",
             "poses per launch and GPU (1..512): a launch carries batch x max(1, --gpus) poses, at "
             "most 512, and the pose list is cut into EQUAL launches (200 poses at 64: 4 launches "
             "of 50 -- no short last launch)");
    args.add("streams", 0, false, "0",
             "render streams the launches alternate between (1 or 2; 0 = auto: 2 when a launch "
             "carries fewer than 48 poses).  A launch drains for ~0.25 ms while its longest rays "
             "finish; on a second stream the next launch starts under that tail "
             "(profiles/r05_cli_bench.json: one pose per launch 0.53 -> 0.39 ms per frame, four "
             "0.31 -> 0.29, thirty-two 0.254 -> 0.250, fifty: a tie)");
    args.add("gpus", 0, false, "0",
             "render every frame on this many GPUs (devices --gpu .. --gpu+N-1): interleaved "
             "screen tiles, tree replicated device to device, RCCL gather of the RGBA8 tiles to "
             "the first GPU; 0 = plain single-GPU path");
    args.add("tile", 0, false, "8", "rows per screen tile of the --gpus shard (multiple of 8)");
    args.add("share_gpu", 0, true, "",
             "REHEARSAL of --gpus N on a box with fewer GPUs: all ranks on one device, no RCCL");
    args.add("fp", 0, false, "strict", "floating-point model: strict | fma");
    args.add("dump_poses", 0, true, "", "print the parsed poses / intrinsics and exit (no GPU needed)");
    args.add("host_decode", 0, true, "",
             "decode quantised trees with the host loop instead of on the device");
    try {
        internal::parse_options(args, argc, argv);
    } catch (const std::exception& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }

    const int device_id = args.as_int("gpu");
    if (device_id >= 0 && vr_set_device(device_id) != VR_OK) {
        fprintf(stderr, "ERROR: %s\n", vr_last_error());
        return 1;
    }

    // Load all transform matrices
    std::vector<glm::mat4x3> trans;
    std::vector<std::string> basenames;
    for (const std::string& path : args.unmatched()) {
        if (!path.empty() && path[0] == '-') continue;  // an unknown option, not a pose file
        const int cnt = read_transform_matrices(path, trans);
        const std::string fname = remove_ext(path_basename(path));
        if (cnt == 1) {
            basenames.push_back(fname);
        } else {
            for (int i = 0; i < cnt; ++i) {
                std::string tmp = std::to_string(i);
                while (tmp.size() < 6) tmp = "0" + tmp;
                basenames.push_back(fname + "_" + tmp);
            }
        }
    }
    if (args.as_bool("reverse_yz")) {
        puts("INFO: Use OpenCV camera convention\n");
        for (auto& t : trans) {  // c2w * diag(1, -1, -1, 1): flip the up and back columns
            t[1] = t[1] * -1.f;
            t[2] = t[2] * -1.f;
        }
    } else {
        puts("INFO: Use NeRF camera convention\n");
    }
    if (args.as_bool("dump_poses")) {
        for (size_t i = 0; i < trans.size(); ++i) {
            printf("pose %s", basenames[i].c_str());
            for (int c = 0; c < 4; ++c)
                printf(" %.9g %.9g %.9g", trans[i][c].x, trans[i][c].y, trans[i][c].z);
            printf("\n");
        }
        if (!args.str("intrin").empty()) {
            float fx = -1.f, fy = -1.f;
            read_intrins(args.str("intrin"), fx, fy);
            printf("intrin %.9g %.9g\n", fx, fy);
        }
        size_t n_poses = trans.size();
        if (args.as_int("max_imgs") > 0 && n_poses > (size_t)args.as_int("max_imgs"))
            n_poses = (size_t)args.as_int("max_imgs");
        const LaunchPlan lp = plan_launches(n_poses, args.as_int("batch"), args.as_int("gpus"),
                                            args.as_int("streams"));
        printf("plan poses %zu launches %zu long %zu batch %d streams %d\n", n_poses, lp.n_launches,
               lp.n_long, lp.batch, lp.n_streams);
        return 0;
    }
    if (trans.empty()) {
        fputs("WARNING: No camera poses specified, quitting\n", stderr);
        return 1;
    }
    const std::string out_dir = args.str("write_images");

    N3Tree tree;
    N3Tree::device_decode = !args.as_bool("host_decode");
    try {
        tree.open(args.str("file"));
    } catch (const std::exception& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    if (!tree.is_cuda_loaded()) return 1;

    int width = args.as_int("width"), height = args.as_int("height");
    float fx = args.as_float("fx");
    if (fx < 0) fx = 1111.11f;
    float fy = args.as_float("fy");
    if (fy < 0) fy = fx;
    if (!args.str("intrin").empty()) read_intrins(args.str("intrin"), fx, fy);
    {
        const float scale = args.as_float("scale");
        if (scale != 1.f) {
            const int owidth = width, oheight = height;
            width = (int)(width * scale);
            height = (int)(height * scale);
            fx *= (float)width / owidth;
            fy *= (float)height / oheight;
        }
    }
    {
        const int max_imgs = args.as_int("max_imgs");
        if (max_imgs > 0 && trans.size() > (size_t)max_imgs) {
            trans.resize(max_imgs);
            basenames.resize(max_imgs);
        }
    }
    const int n_gpus = args.as_int("gpus");
    const LaunchPlan plan = plan_launches(trans.size(), args.as_int("batch"), n_gpus, args.as_int("streams"));
    const int batch = plan.batch;
    const size_t n_long = plan.n_long;
    const int fp_mode = args.str("fp") == "fma" ? VR_FP_FMA : VR_FP_STRICT;

    const size_t frame_bytes = (size_t)width * height * 4;
    const RenderOptions options = internal::render_options_from_args(args);
    VrRenderOptions copt;
    vr_default_options(&copt);
    copt.step_size = options.step_size;
    copt.sigma_thresh = options.sigma_thresh;
    copt.stop_thresh = options.stop_thresh;
    copt.background_brightness = options.background_brightness;

    // --gpus N: the frames of a launch are rendered tile-sharded on N devices and assembled on
    // the first one (include/volrend/internal/tile_shard.hpp); otherwise one device renders
    // whole frames.  Either way `images[i]` below is frame i of the current launch on `out_dev`.
    std::unique_ptr<internal::TileShardRenderer> shard;
    if (n_gpus >= 1) {
        internal::TileShardConfig sc;
        sc.n_ranks = n_gpus;
        sc.first_device = device_id >= 0 ? device_id : 0;
        sc.share_device = args.as_bool("share_gpu");
        sc.tile_rows = args.as_int("tile");
        sc.max_batch = batch;
        try {
            shard.reset(new internal::TileShardRenderer(tree, width, height, sc));
        } catch (const std::exception& e) {
            fprintf(stderr, "ERROR: %s\n", e.what());
            return 1;
        }
        printf("INFO: %d-way screen-tile shard, %d-row tiles, %s\n", n_gpus,
               sc.tile_rows < 8 ? 8 : sc.tile_rows / 8 * 8, shard->transport().c_str());
        HIP_OK(hipSetDevice(shard->root_device()));
    }
    // Device frames: two sets of `batch` contiguous frames -- launch k renders into set k % 2
    // while the read-back of launch k - 1 drains the other one (the tile shard brings its own).
    uint8_t* image_sets[2] = {nullptr, nullptr};
    if (!shard)
        for (auto& is : image_sets) HIP_OK(hipMalloc((void**)&is, frame_bytes * batch));
    uint8_t* host_sets[2] = {nullptr, nullptr};  // pinned, one per in-flight batch
    std::unique_ptr<EncodePool> pool;
    hipStream_t copy_stream = nullptr;
    hipEvent_t rendered[2] = {nullptr, nullptr}, copied[2] = {nullptr, nullptr};
    // one event behind the copy of every frame of a set: a frame is handed to the encoders as soon
    // as ITS copy has landed, so that the encoding of the last launch runs under its own copies
    std::vector<hipEvent_t> frame_copied[2];
    if (!out_dir.empty()) {
        make_dirs(out_dir);
        for (auto& hs : host_sets) HIP_OK(hipHostMalloc((void**)&hs, frame_bytes * batch));
        unsigned nt = std::thread::hardware_concurrency();
        nt = nt == 0 ? 4 : (nt > 64 ? 64 : nt);
        pool.reset(new EncodePool(nt));
        HIP_OK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_OK(hipEventCreateWithFlags(&rendered[i], hipEventDisableTiming));
            HIP_OK(hipEventCreateWithFlags(&copied[i], hipEventDisableTiming));
            frame_copied[i].resize((size_t)batch);
            for (auto& e : frame_copied[i]) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
    }
    // Render streams: launch k runs on stream k % n_streams and writes image set k % 2 -- with two
    // streams every stream owns one image set (the tile shard brings its own streams).
    const int n_streams = shard ? 1 : plan.n_streams;
    hipStream_t streams[2] = {nullptr, nullptr};
    if (shard) streams[0] = static_cast<hipStream_t>(shard->out_stream());
    else
        for (int i = 0; i < n_streams; ++i) HIP_OK(hipStreamCreate(&streams[i]));
    hipEvent_t start, stop, joined;
    HIP_OK(hipEventCreate(&start));
    HIP_OK(hipEventCreate(&stop));
    HIP_OK(hipEventCreateWithFlags(&joined, hipEventDisableTiming));

    // Frame egress of launch `seq` (frames [first, first + n)): called one launch late, so the
    // host waits for the copies of launch k - 1 while the GPU renders launch k.
    auto submit_encodes = [&](int set, size_t first, int n) {
        for (int i = 0; i < n; ++i) {
            HIP_OK(hipEventSynchronize(frame_copied[set][(size_t)i]));
            const std::string fpath = out_dir + "/" + basenames[first + i] + ".png";
            const uint8_t* src = host_sets[set] + frame_bytes * i;
            pool->submit(set, [fpath, src, width, height] {
                internal::write_png_file(fpath, src, width, height);
            });
        }
    };

    // The launch slots' ray buffers are sized BEFORE the clock starts (one slot per render
    // stream; 76 bytes per ray of a launch -- 2.4 GB at 50 poses of 800 x 800): left to the first
    // launches, the allocations would sit inside the timed loop, as the reference's cudaArray
    // would if it were created behind cudaEventRecord(start) (main_headless.cpp:187-203).
    if (!shard && vr_reserve_tiles(tree.device, width, height, batch, 0, 0, 1, n_streams) != VR_OK) {
        fprintf(stderr, "ERROR: %s\n", vr_last_error());
        return 1;
    }
    HIP_OK(hipEventRecord(start, streams[0]));
    if (n_streams > 1) HIP_OK(hipStreamWaitEvent(streams[1], start, 0));  // the clock starts before any launch
    int seq = 0;
    size_t prev_first = 0;
    int prev_n = 0;
    for (size_t first = 0; first < trans.size(); ++seq) {
        const int n = (size_t)seq < n_long ? batch : batch - 1;
        const int set = seq & 1;
        hipStream_t stream = streams[seq % n_streams];
        std::vector<VrCamera> cams((size_t)n);
        std::vector<VrFrame> frames((size_t)n);
        for (int i = 0; i < n; ++i) {
            const float* m = glm::value_ptr(trans[first + i]);
            for (int k = 0; k < 12; ++k) cams[i].transform[k] = m[k];
            cams[i].width = width;
            cams[i].height = height;
            cams[i].fx = fx;
            cams[i].fy = fy;
            vr_default_frame(&frames[i]);
            frames[i].rgba = shard ? nullptr : image_sets[set] + frame_bytes * i;
            frames[i].offscreen = 1;
            frames[i].fp_mode = fp_mode;
        }
        // set `set` of the device frames was read back by launch seq - 2: that copy must be done
        // before this launch overwrites them (device-side wait, the host does not block)
        if (copy_stream && seq >= 2) HIP_OK(hipStreamWaitEvent(stream, copied[set], 0));
        const uint8_t* dev_frames = nullptr;
        if (shard) {
            try {
                shard->render(seq, cams.data(), n, copt, fp_mode);
            } catch (const std::exception& e) {
                fprintf(stderr, "ERROR: %s\n", e.what());
                return 1;
            }
            dev_frames = shard->frames(set);
        } else {
            if (vr_render_batch(tree.device, n, cams.data(), &copt, frames.data(), stream) != VR_OK) {
                fprintf(stderr, "ERROR: %s\n", vr_last_error());
                return 1;
            }
            dev_frames = image_sets[set];
        }
        if (!out_dir.empty()) {
            HIP_OK(hipEventRecord(rendered[set], stream));
            pool->wait(set);  // the encoders (launch seq - 2) are done with this host buffer set
            HIP_OK(hipStreamWaitEvent(copy_stream, rendered[set], 0));
            // frame by frame (2.56 MB at 800 x 800: large enough for the copy engine), an event
            // behind each
            for (int i = 0; i < n; ++i) {
                if (vr_read_back(host_sets[set] + frame_bytes * i, dev_frames + frame_bytes * i, 0, width,
                                 height, copy_stream) != VR_OK) {
                    fprintf(stderr, "ERROR: %s\n", vr_last_error());
                    return 1;
                }
                HIP_OK(hipEventRecord(frame_copied[set][(size_t)i], copy_stream));
            }
            HIP_OK(hipEventRecord(copied[set], copy_stream));
            if (prev_n > 0) submit_encodes(set ^ 1, prev_first, prev_n);
            prev_first = first;
            prev_n = n;
        }
        first += (size_t)n;
    }
    if (pool) {
        if (prev_n > 0) submit_encodes((seq - 1) & 1, prev_first, prev_n);
        pool->wait(0);
        pool->wait(1);
    }
    if (n_streams > 1) {  // the clock stops behind the last launch of BOTH streams
        HIP_OK(hipEventRecord(joined, streams[1]));
        HIP_OK(hipStreamWaitEvent(streams[0], joined, 0));
    }
    HIP_OK(hipEventRecord(stop, streams[0]));
    HIP_OK(hipEventSynchronize(stop));
    // Everything has run: what the launches found out on the device surfaces now -- a ray that
    // hit the sample guard means wrong frames, and the reference's convention for device errors
    // is message + non-zero exit (src/cuda/common.cu:8-21).
    try {
        if (shard) shard->sync();  // (checks the status word of every rank's tree)
        else check_render_status(tree);
    } catch (const std::exception& e) {
        fprintf(stderr, "ERROR: %s\n", e.what());
        return 1;
    }
    float milliseconds = 0;
    HIP_OK(hipEventElapsedTime(&milliseconds, start, stop));
    milliseconds = milliseconds / trans.size();

    printf("%.10f ms per frame\n", milliseconds);
    printf("%.10f fps\n", 1000.f / milliseconds);
    printf("%.4f Mrays/s\n", (double)width * height / (milliseconds * 1e3));

    pool.reset();
    for (auto& hs : host_sets)
        if (hs) HIP_OK(hipHostFree(hs));
    for (int i = 0; i < 2; ++i) {
        if (rendered[i]) HIP_OK(hipEventDestroy(rendered[i]));
        if (copied[i]) HIP_OK(hipEventDestroy(copied[i]));
        for (auto e : frame_copied[i]) HIP_OK(hipEventDestroy(e));
    }
    if (copy_stream) HIP_OK(hipStreamDestroy(copy_stream));
    if (!shard) {
        for (uint8_t* p : image_sets) HIP_OK(hipFree(p));
        for (int i = 0; i < n_streams; ++i) HIP_OK(hipStreamDestroy(streams[i]));
    }
    HIP_OK(hipEventDestroy(joined));
    shard.reset();
    return 0;
}
