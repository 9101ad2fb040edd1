// hpt_render -- the reference's command line on top of libhpt.so:
//     cargo run --release -- scene.xml out.png      ==>      hpt_render scene.xml out.png
// main.rs:34-41 (arguments), load_scene (load.rs:17), the pixel loop of main.rs:55-115 replaced by ONE hpt_render call
// (adaptive sampling with the constants of main.rs:27-32 unless --spp is given), color_as_u8_array (scene.rs:362) via
// hpt_tonemap_rgba8, save_img (main.rs:117-130).  Errors end the process with exit code 101 like a Rust panic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#include "hpt_host.h"

using namespace hpt_host;

static void usage() {
    std::fprintf(stderr,
        "usage: hpt_render <scene.xml> <out.png> [options]\n"
        "  --spp N            fixed N samples per pixel (default: the reference's adaptive 128 -> 512 sampling)\n"
        "  --depth D          bounce cap (default 8; the reference recursion is unbounded, see DESIGN.md)\n"
        "  --seed S           seed of the counter-based generator (default 1)\n"
        "  --size WxH         override <width>/<height> of the scene file\n"
        "  --device I         GPU index (default 0)\n"
        "  --gpus N           one call over the first N GPUs (0 = all): hpt_group_render -- samples split over the GPUs, RCCL\n"
        "                     sum-reduce onto GPU 0 (also: HPT_GPUS=N in the environment, as rust/main_render.rs reads it)\n"
        "  --library-bvh      let the library build its SAH tree instead of handing over the bvh.rs tree\n"
        "  --checkpoint FILE  (with --spp N) keep the accumulator + sample cursor of this render in FILE\n"
        "  --resume FILE      (with --spp M) continue the render of FILE with M more samples per pixel; the image is the one a\n"
        "                     single run of all the samples writes, byte for byte (integer accumulator)\n"
        "  --stats            print ray counts and timings to stderr\n"
        "  --dump-scene FILE  write the loaded scene (exact float bits + digests) and exit; needs no GPU\n");
}

int main(int argc, char** argv) {
    try {
        std::vector<std::string> pos;
        int spp = 0, depth = 8, device = 0; unsigned seed = 1; bool library_bvh = false, stats = false;
        int ow = 0, oh = 0; std::string dump, checkpoint, resume;
        int gpus = -1;      // -1: one context on --device; >= 0: a group
        if (const char* e = std::getenv("HPT_GPUS")) gpus = std::atoi(e);
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&]() -> const char* { if (i + 1 >= argc) throw Error("missing value after " + a); return argv[++i]; };
            if (a == "--spp") spp = std::atoi(val());
            else if (a == "--depth") depth = std::atoi(val());
            else if (a == "--seed") seed = (unsigned)std::strtoul(val(), nullptr, 10);
            else if (a == "--device") device = std::atoi(val());
            else if (a == "--gpus") gpus = std::atoi(val());
            else if (a == "--size") { if (std::sscanf(val(), "%dx%d", &ow, &oh) != 2 || ow <= 0 || oh <= 0) throw Error("--size wants WxH"); }
            else if (a == "--library-bvh") library_bvh = true;
            else if (a == "--stats") stats = true;
            else if (a == "--checkpoint") checkpoint = val();
            else if (a == "--resume") resume = val();
            else if (a == "--dump-scene") dump = val();
            else if (a == "-h" || a == "--help") { usage(); return 0; }
            else pos.push_back(a);
        }
        if (!dump.empty() && pos.size() >= 1) {
            const SceneDesc d = load_scene(pos[0], true);
            std::ofstream f(dump);
            if (!f) throw Error("cannot create " + dump);
            f << dump_scene(d);
            return 0;
        }
        if (pos.size() < 2) { usage(); throw Error("please provide a scene description file and an image output file"); }   // main.rs:36-38

        const bool accumulate = !checkpoint.empty() || !resume.empty();
        if (accumulate && spp <= 0) throw Error("--checkpoint / --resume need --spp N (a fixed number of samples per pixel for this run)");
        if (accumulate && gpus >= 0) throw Error("--checkpoint / --resume drive one GPU (--device); not available with --gpus");
        SceneDesc d = load_scene(pos[0], !library_bvh);
        if (ow) { d.camera.img_width = (uint32_t)ow; d.camera.img_height = (uint32_t)oh; }

        hpt_context* ctx = nullptr; hpt_scene* scene = nullptr; hpt_group* group = nullptr;
        if (gpus >= 0) {      // main.rs:57,72 use the whole machine: so does one hpt_group_render call
            std::vector<int> devs;
            for (int i = 0; i < gpus; ++i) devs.push_back(i);
            if (hpt_group_create(devs.empty() ? nullptr : devs.data(), (int)devs.size(), &group) != HPT_OK) throw Error(std::string("hpt_group_create: ") + hpt_last_error());
            ctx = hpt_group_context(group, 0);
        } else if (hpt_context_create(device, &ctx) != HPT_OK) throw Error(std::string("hpt_context_create: ") + hpt_last_error());
        if (hpt_scene_create(ctx, &scene) != HPT_OK) throw Error(std::string("hpt_scene_create: ") + hpt_last_error());
        hand_over(d, scene, library_bvh);

        hpt_render_params p; std::memset(&p, 0, sizeof p);
        p.adaptive = spp > 0 ? 0 : 1; p.spp = spp;
        p.initial_samples = 128; p.max_samples = 512;                       // main.rs:31-32
        p.thresholds[0] = 0.4f; p.thresholds[1] = 0.3f; p.thresholds[2] = 0.6f;   // main.rs:27-29
        p.max_depth = depth; p.seed = seed;
        if (stats) p.flags = HPT_RENDER_TIME_KERNELS;
        const size_t npix = (size_t)d.camera.img_width * d.camera.img_height;
        std::vector<float> rgb(npix * 3);
        std::vector<uint32_t> counts(npix);
        hpt_render_stats st; std::memset(&st, 0, sizeof st);
        if (group) {
            p.flags = 0;
            hpt_group_stats gs;
            if (hpt_group_render(group, scene, &p, rgb.data(), counts.data(), &gs) != HPT_OK) throw Error(std::string("hpt_group_render: ") + hpt_last_error());
            st.rays_primary = gs.rays_primary; st.rays_bounce = gs.rays_bounce; st.samples = gs.samples; st.seconds_total = gs.seconds_total;
            if (stats) std::fprintf(stderr, "%d GPU(s), %s: slowest render %.3f s, exchange steps %.4f s\n", gs.n_devices, gs.transport ? "peer copies" : "RCCL", gs.seconds_render_max, gs.seconds_reduce);
        } else if (accumulate) {
            // Resumable render: the checkpoint is the accumulator (hpt_accum per pixel: 64-bit fixed-point sums) + the sample
            // cursor + what the samples depend on (frame size, seed, depth cap).  Samples are keyed by (pixel, index) and the
            // sums are integers, so N samples now + M after a restart leave the bits one run of N + M leaves.
            struct Header { char magic[8]; uint32_t width, height, seed; int32_t depth; int64_t cursor; };
            std::vector<hpt_accum> acc(npix);
            std::memset(acc.data(), 0, npix * sizeof(hpt_accum));
            int64_t cursor = 0;
            if (!resume.empty()) {
                std::ifstream f(resume, std::ios::binary);
                Header h;
                if (!f || !f.read((char*)&h, sizeof h) || std::memcmp(h.magic, "HPTACC1", 8) != 0) throw Error("cannot read checkpoint " + resume);
                if (h.width != d.camera.img_width || h.height != d.camera.img_height || h.seed != seed || h.depth != depth)
                    throw Error("checkpoint " + resume + " was rendered with another frame size, seed or depth");
                if (!f.read((char*)acc.data(), (std::streamsize)(npix * sizeof(hpt_accum)))) throw Error("checkpoint " + resume + " is truncated");
                cursor = h.cursor;
            }
            p.adaptive = 0; p.first_sample = (int32_t)cursor; p.spp = spp;
            if (hpt_render_accum(scene, &p, acc.data(), &st) != HPT_OK) throw Error(std::string("hpt_render_accum: ") + hpt_last_error());
            cursor += spp;
            if (hpt_accum_resolve(acc.data(), (uint32_t)npix, nullptr, (uint32_t)cursor, rgb.data()) != HPT_OK) throw Error(std::string("hpt_accum_resolve: ") + hpt_last_error());
            if (!checkpoint.empty()) {
                Header h; std::memset(&h, 0, sizeof h);
                std::memcpy(h.magic, "HPTACC1", 8);
                h.width = d.camera.img_width; h.height = d.camera.img_height; h.seed = seed; h.depth = depth; h.cursor = cursor;
                std::ofstream f(checkpoint, std::ios::binary);
                if (!f || !f.write((const char*)&h, sizeof h) || !f.write((const char*)acc.data(), (std::streamsize)(npix * sizeof(hpt_accum)))) throw Error("cannot write checkpoint " + checkpoint);
            }
            if (stats) std::fprintf(stderr, "samples %lld..%lld of every pixel\n", (long long)(cursor - spp + 1), (long long)cursor);
        } else if (hpt_render(scene, &p, rgb.data(), counts.data(), &st) != HPT_OK) throw Error(std::string("hpt_render: ") + hpt_last_error());
        std::vector<uint8_t> rgba(npix * 4);
        if (hpt_tonemap_rgba8(rgb.data(), (uint32_t)npix, rgba.data()) != HPT_OK) throw Error(std::string("hpt_tonemap_rgba8: ") + hpt_last_error());
        write_png_rgba8(pos[1], d.camera.img_width, d.camera.img_height, rgba.data());   // main.rs:117
        if (stats) {
            const double rays = (double)(st.rays_primary + st.rays_bounce);
            std::fprintf(stderr, "%ux%u, %llu pixel-samples, %.0f rays in %.3f s = %.1f Mrays/s (extend %.3f s, shade %.3f s, %u passes)\n",
                         d.camera.img_width, d.camera.img_height, (unsigned long long)st.samples, rays, st.seconds_total, rays / st.seconds_total / 1e6,
                         st.seconds_extend, st.seconds_shade, st.passes);
        }
        hpt_scene_destroy(scene);
        if (group) hpt_group_destroy(group); else hpt_context_destroy(ctx);
        return 0;
    } catch (const std::exception& e) {
        std::fprintf(stderr, "hpt_render: %s\n", e.what());
        return 101;   // Rust's panic exit code: the reference panics on every load / IO error
    }
}
