// rt_bench — headless counterpart of the reference's interactive binary (src/bin/main.rs:8-57): builds the Game
// (six optional positional floats `x y z heading pitch sun_angle`, src/game/mod.rs:45-52), creates the renderer, then
// loops draw_frame and prints the rolling average / maximum frame time of the last 120 frames
// (RingBufferAverage, src/util.rs:175-221; printout main.rs:42-47) — followed by ONE JSON line with the metrics SURVEY.md 5
// asks of the bench binary (config, rays, ms, Mrays/s).
//
//   rt_bench [x y z heading pitch sun] [--width W] [--height H] [--spp N] [--depth D] [--frames F]
//            [--noise tests/golden/blue_noise_512.rgba] [--device I] [--gpus N] [--gather] [--overlap] [--post]
//
// --post: the reference's whole frame — ray trace, six denoise dispatches, finalize (pipeline.rs:86-123) — per draw_frame
// (Pipeline::enable_post_passes; one device only: the passes need the whole frame).
//
// --gpus N (one host thread per device, ncclCommInitAll through rt_comm_init_all): device i renders the tiles t % N == i and
// every frame ends with rt_gather_gbuffer to device 0, which assembles the full frame in the library's own planes.
// --gather runs that frame-end step with N = 1 too (a one-rank communicator).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "render.hpp"

namespace {

// RingBufferAverage — src/util.rs:175-221
class RingBufferAverage {
 public:
    explicit RingBufferAverage(size_t n) : buf_(n, 0.0), filled_(0), next_(0) {}
    void push_sample(double v) {
        buf_[next_] = v;
        next_ = (next_ + 1) % buf_.size();
        filled_ = std::min(filled_ + 1, buf_.size());
    }
    double average() const {
        double s = 0;
        for (size_t i = 0; i < filled_; i++) s += buf_[i];
        return filled_ ? s / (double)filled_ : 0.0;
    }
    double max() const {
        double m = 0;
        for (size_t i = 0; i < filled_; i++) m = std::max(m, buf_[i]);
        return m;
    }

 private:
    std::vector<double> buf_;
    size_t filled_, next_;
};

RtConfig make_config(int width, int height, int spp, int depth, int device, int rank, int world, uint32_t flags) {
    RtConfig cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.width = width; cfg.height = height; cfg.region = RT_ROOT_BLOCK_SIZE; cfg.spp = spp; cfg.depth = depth;
    cfg.device = device; cfg.tile_rank = rank; cfg.tile_world = world; cfg.kernel = RT_KERNEL_DEFAULT;
    cfg.flags = flags;
    return cfg;
}

// the host threads of --gpus N meet here once per frame (C++17 has no std::barrier)
class ThreadBarrier {
 public:
    explicit ThreadBarrier(int n) : n_(n) {}
    void arrive_and_wait() {
        std::unique_lock<std::mutex> lk(m_);
        const unsigned long gen = gen_;
        if (++count_ == n_) { count_ = 0; gen_++; cv_.notify_all(); return; }
        cv_.wait(lk, [&] { return gen_ != gen; });
    }
 private:
    std::mutex m_;
    std::condition_variable cv_;
    int n_, count_ = 0;
    unsigned long gen_ = 0;
};

}  // namespace

int main(int argc, char** argv) {
    int width = 1024, height = 1024;   // WINDOW_WIDTH / WINDOW_HEIGHT, src/render/constants.rs:9-10
    int spp = 1, depth = 2, frames = 240, device = 0, gpus = 1;
    bool gather = false, overlap = false, post = false;
    std::string noise_path = "tests/golden/blue_noise_512.rgba";
    std::vector<const char*> positional = {argv[0]};
    for (int i = 1; i < argc; i++) {
        auto want = [&](const char* flag) { return std::strcmp(argv[i], flag) == 0 && i + 1 < argc; };
        if (want("--width")) width = std::atoi(argv[++i]);
        else if (want("--height")) height = std::atoi(argv[++i]);
        else if (want("--spp")) spp = std::atoi(argv[++i]);
        else if (want("--depth")) depth = std::atoi(argv[++i]);
        else if (want("--frames")) frames = std::atoi(argv[++i]);
        else if (want("--device")) device = std::atoi(argv[++i]);
        else if (want("--gpus")) gpus = std::atoi(argv[++i]);
        else if (want("--noise")) noise_path = argv[++i];
        else if (std::strcmp(argv[i], "--gather") == 0) gather = true;
        else if (std::strcmp(argv[i], "--overlap") == 0) overlap = true;
        else if (std::strcmp(argv[i], "--post") == 0) post = true;
        else positional.push_back(argv[i]);
    }
    if (gpus < 1 || frames < 1) { std::fprintf(stderr, "--gpus and --frames must be >= 1\n"); return 2; }
    if (gpus > 1) gather = true;
    if (post && gpus > 1) { std::fprintf(stderr, "--post needs the whole frame on one device (gather first on several)\n"); return 2; }
    rt::game::Game game((int)positional.size(), positional.data());

    std::vector<uint8_t> noise(RT_NOISE_BYTES);
    FILE* fp = std::fopen(noise_path.c_str(), "rb");
    if (!fp || std::fread(noise.data(), 1, noise.size(), fp) != noise.size()) {
        std::fprintf(stderr, "cannot read the blue-noise table (512x512 RGBA8 raw) from %s\n", noise_path.c_str());
        return 2;
    }
    std::fclose(fp);

    std::printf("Creating renderer (and world.)\n");                 // main.rs:10
    auto t0 = std::chrono::steady_clock::now();
    game.generate_world(0x5EED);
    std::string err;

    // exact ray count of one frame (a counting context, outside the timed loop): the JSON line's Mrays/s is rays actually traced
    unsigned long long rays_per_frame = 0;
    {
        RtConfig ccfg = make_config(width, height, spp, depth, device, 0, 1, RT_FLAG_CACHE_PRIMARY | RT_FLAG_COUNTERS);
        rt::render::Pipeline* cp = rt::render::create_instance(ccfg, noise.data(), game, &err);
        if (!cp) { std::fprintf(stderr, "create_instance failed: %s\n", err.c_str()); return 1; }
        RtCounters cn{};
        if (cp->draw_frame(game) != RT_OK || cp->wait() != RT_OK || rt_get_counters(cp->context(), &cn) != RT_OK) {
            std::fprintf(stderr, "counting frame failed: %s\n", cp->last_error());
            delete cp;
            return 1;
        }
        rays_per_frame = cn.rays;
        delete cp;
    }

    std::vector<rt::render::Pipeline*> pipes((size_t)gpus, nullptr);
    std::vector<void*> comms((size_t)gpus, nullptr);
    std::vector<int> devices((size_t)gpus);
    for (int g = 0; g < gpus; g++) devices[(size_t)g] = device + g;
    for (int g = 0; g < gpus; g++) {
        RtConfig cfg = make_config(width, height, spp, depth, devices[(size_t)g], g, gpus, RT_FLAG_CACHE_PRIMARY);
        pipes[(size_t)g] = rt::render::create_instance(cfg, noise.data(), game, &err);
        if (!pipes[(size_t)g]) {
            std::fprintf(stderr, "create_instance failed on device %d: %s\n", devices[(size_t)g], err.c_str());
            return 1;
        }
    }
    if (post && pipes[0]->enable_post_passes(true) != RT_OK) { std::fprintf(stderr, "enable_post_passes failed\n"); return 1; }
    if (gather) {
        int rc = rt_comm_init_all(gpus, devices.data(), comms.data());
        if (rc != RT_OK) { std::fprintf(stderr, "rt_comm_init_all failed (%d): %s\n", rc, rt_last_error(nullptr)); return 1; }
    }
    std::printf("Created in %fs.\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());   // main.rs:13

    RingBufferAverage perf(120);                                      // main.rs:16
    ThreadBarrier frame_barrier(gpus);
    std::atomic<int> failed{0};
    double total_ms = 0.0;
    // one host thread per device; thread 0 keeps the reference's frame-time statistics (main.rs:42-47)
    auto worker = [&](int g) {
        rt::render::Pipeline* p = pipes[(size_t)g];
        auto frame_timer = std::chrono::steady_clock::now();
        const auto loop_start = frame_timer;
        for (int f = 0; f < frames; f++) {
            if (g == 0) {
                auto now = std::chrono::steady_clock::now();
                double millis = std::chrono::duration<double, std::milli>(now - frame_timer).count();
                frame_timer = now;
                if (f > 0) perf.push_sample(millis);
            }
            int rc = p->draw_frame(game);                             // main.rs:52
            if (rc != RT_OK) {
                std::fprintf(stderr, "frame %d failed on device %d (%d): %s\n", f, devices[(size_t)g], rc, p->last_error());
                failed.store(1);
            }
            if (gpus > 1) {
                // either every rank posts the frame's send/recv or none does: a rank that posted alone would wait for its peers for ever
                frame_barrier.arrive_and_wait();
                if (failed.load()) break;
            } else if (rc != RT_OK) break;
            if (gather) {
                rc = rt_gather_gbuffer(p->context(), comms[(size_t)g], 0, nullptr, overlap ? 1 : 0);
                if (rc != RT_OK) {
                    // the peers have posted theirs already and cannot be recalled: leave without waiting for them
                    std::fprintf(stderr, "gather of frame %d failed on device %d (%d): %s\n", f, devices[(size_t)g], rc, p->last_error());
                    std::fflush(nullptr);
                    if (gpus > 1) std::_Exit(1);
                    failed.store(1);
                    break;
                }
            }
        }
        if (failed.load()) return;     // nothing is waited for on the failure path: queued collectives may have no peer
        if (rt_sync(p->context()) != RT_OK) failed.store(1);
        if (g == 0) total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - loop_start).count();
    };
    std::vector<std::thread> threads;
    for (int g = 1; g < gpus; g++) threads.emplace_back(worker, g);
    worker(0);
    for (auto& t : threads) t.join();

    int exit_code = failed.load() ? 1 : 0;
    if (exit_code && gpus > 1) { std::fflush(nullptr); std::_Exit(1); }   // communicators may hold unmatched operations: no orderly teardown
    if (!exit_code) {
        std::printf("%.3fms / %.3fms\n", perf.average(), perf.max());    // main.rs:45-46: average / max
        // a checksum of the assembled frame's depth plane shows that the gather delivered pixels (0 without --gather)
        unsigned long long checksum = 0;
        if (gather) {
            std::vector<uint16_t> depth_plane((size_t)width * height);
            if (rt_frame_readback(pipes[0]->context(), RT_BUF_DEPTH_R16UI, depth_plane.data(), depth_plane.size() * 2) == RT_OK)
                for (uint16_t v : depth_plane) checksum += v;
            else { std::fprintf(stderr, "rt_frame_readback failed: %s\n", pipes[0]->last_error()); exit_code = 1; }
        }
        unsigned long long final_checksum = 0;   // --post: sum over the swapchain image's bytes (0 without)
        if (post) {
            std::vector<uint8_t> final_plane((size_t)width * height * 4);
            if (rt_readback(pipes[0]->context(), RT_BUF_FINAL_BGRA8, final_plane.data(), final_plane.size()) == RT_OK)
                for (uint8_t v : final_plane) final_checksum += v;
            else { std::fprintf(stderr, "rt_readback(final) failed: %s\n", pipes[0]->last_error()); exit_code = 1; }
        }
        const double ms = total_ms / frames;
        std::printf("{\"binary\": \"rt_bench\", \"config\": {\"width\": %d, \"height\": %d, \"spp\": %d, \"depth\": %d, \"gpus\": %d, "
                    "\"gather\": \"%s\", \"post_passes\": %s, \"pose\": [%g, %g, %g, %g, %g], \"sun_angle\": %g}, \"frames\": %d, \"rays_per_frame\": %llu, "
                    "\"ms_per_frame\": %.4f, \"avg_ms_last_120\": %.4f, \"max_ms_last_120\": %.4f, \"mrays_per_s\": %.2f, "
                    "\"depth_plane_checksum\": %llu, \"final_image_checksum\": %llu}\n",
                    width, height, spp, depth, gpus, gather ? (overlap ? "rccl-overlapped" : "rccl-serial") : "none", post ? "true" : "false",
                    game.camera.origin[0], game.camera.origin[1], game.camera.origin[2], game.camera.heading, game.camera.pitch,
                    game.sun_angle, frames, rays_per_frame, ms, perf.average(), perf.max(), (double)rays_per_frame / (ms * 1e3), checksum, final_checksum);
    }
    for (int g = 0; g < gpus; g++) {
        if (comms[(size_t)g]) rt_comm_destroy(comms[(size_t)g]);
        delete pipes[(size_t)g];
    }
    return exit_code;
}
