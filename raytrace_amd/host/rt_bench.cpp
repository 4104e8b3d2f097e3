// rt_bench — headless counterpart of the reference's interactive binary (src/bin/main.rs:8-57): builds the Game
// (six optional positional floats `x y z heading pitch sun_angle`, src/game/mod.rs:45-52), creates the renderer, then
// loops draw_frame and prints the rolling average / maximum frame time of the last 120 frames
// (RingBufferAverage, src/util.rs:175-221; printout main.rs:42-47).
//
//   rt_bench [x y z heading pitch sun] [--width W] [--height H] [--spp N] [--depth D] [--frames F]
//            [--noise tests/golden/blue_noise_512.rgba] [--device I]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

}  // namespace

int main(int argc, char** argv) {
    int width = 1024, height = 1024;   // WINDOW_WIDTH / WINDOW_HEIGHT, src/render/constants.rs:9-10
    int spp = 1, depth = 2, frames = 240, device = 0;
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
        else if (want("--noise")) noise_path = argv[++i];
        else positional.push_back(argv[i]);
    }
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
    RtConfig cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.width = width; cfg.height = height; cfg.region = RT_ROOT_BLOCK_SIZE; cfg.spp = spp; cfg.depth = depth;
    cfg.device = device; cfg.tile_rank = 0; cfg.tile_world = 1; cfg.kernel = RT_KERNEL_DEFAULT;
    cfg.flags = RT_FLAG_CACHE_PRIMARY;
    std::string err;
    rt::render::Pipeline* pipeline = rt::render::create_instance(cfg, noise.data(), game, &err);
    if (!pipeline) {
        std::fprintf(stderr, "create_instance failed: %s\n", err.c_str());
        return 1;
    }
    std::printf("Created in %fs.\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());   // main.rs:13

    RingBufferAverage perf(120);                                      // main.rs:16
    auto frame_timer = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++) {
        auto now = std::chrono::steady_clock::now();
        double millis = std::chrono::duration<double, std::milli>(now - frame_timer).count();
        frame_timer = now;
        if (f > 0) perf.push_sample(millis);
        int rc = pipeline->draw_frame(game);                          // main.rs:52
        if (rc != RT_OK) {
            std::fprintf(stderr, "draw_frame failed (%d): %s\n", rc, pipeline->last_error());
            delete pipeline;
            return 1;
        }
    }
    pipeline->wait();
    std::printf("%.3fms / %.3fms\n", perf.average(), perf.max());    // main.rs:45-46: average / max
    delete pipeline;
    return 0;
}
