// `path-tracer` command line — same surface as the reference binary for the
// render path (src/main.rs:14-57, src/config/mod.rs:10-52, README.md:28-60):
//
//   path-tracer render <INPUT> [-o/--output <OUTPUT>] [-q/--quiet] [-v/--viewer]
//                              [--debug-textures] [-p/--profile <PROFILE>]
//       env OUTPUT (default render.png), env PROFILE
//   path-tracer convert <INPUT> <OUTPUT>      (glTF 2.0 -> ISF, host/gltf_convert.cpp)
//
// Any error prints the message on stderr and exits with code 2 (main.rs:14-22).
// The render itself runs on the GPU through the C ABI of include/ptgpu.h;
// there is no CPU fallback.  Extras (not in the reference): --device N,
// --devices A,B,... (one host thread per GPU, each rendering its share of interleaved 32x32 tiles: the
// sharding of SURVEY 8-e inside one process; the KD-tree and the origin grids are built once and uploaded to every
// device, the slices are exchanged by one RCCL all-gather), --stats (one JSON line with timings on stderr).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <unistd.h>
#include <vector>

#include "ptgpu.h"
#include "pthost.h"

namespace {

[[noreturn]] void die(const std::string& msg) {
    fprintf(stderr, "%s\n", msg.c_str());
    exit(2);
}

void usage_render(FILE* f) {
    fputs("Path-trace awesome things\n\n"
          "Usage: path-tracer render [OPTIONS] <INPUT>\n\n"
          "Arguments:\n"
          "  <INPUT>  Input file name ISF format\n\n"
          "Options:\n"
          "  -o, --output <OUTPUT>    Output image name [env: OUTPUT=] [default: render.png]\n"
          "  -q, --quiet              No progress bar printed\n"
          "  -v, --viewer             Display a viewer (headless: the output file is refreshed after every sample batch)\n"
          "      --debug-textures     Generate debug textures\n"
          "  -p, --profile <PROFILE>  A path to the yaml file containing all the rendering profile information [env: PROFILE=]\n"
          "      --device <N>         HIP device ordinal [default: 0]\n"
          "      --devices <A,B,..>   Render on several GPUs (interleaved tiles, one thread per device)\n"
          "      --stats              Print timing statistics as JSON on stderr\n"
          "  -h, --help               Print help\n",
          f);
}

void usage_main(FILE* f) {
    fputs("Path-trace awesome things\n\n"
          "Usage: path-tracer <COMMAND>\n\n"
          "Commands:\n"
          "  render   Path-trace awesome things\n"
          "  convert  Convert scenes into ISF format\n"
          "  help     Print this message or the help of the given subcommand(s)\n\n"
          "Options:\n"
          "  -h, --help     Print help\n"
          "  -V, --version  Print version\n",
          f);
}

struct Progress {
    bool quiet;
    std::chrono::steady_clock::time_point start;
};

// -v / --viewer on a headless node: the viewer feed (renderer/mod.rs:133-141, renderer/viewer.rs) becomes a
// progressively refreshed output file — after every sample batch the image of the samples done so far.
struct Preview {
    std::string path;
    uint32_t width, height;
};
void on_preview(const uint8_t* rgb8, uint64_t n_pixels, uint32_t, uint32_t, void* user) {
    Preview* v = (Preview*)user;
    if (n_pixels == (uint64_t)v->width * v->height) pth_png_write_rgb8(v->path.c_str(), v->width, v->height, rgb8);
}

void on_progress(uint32_t done, uint32_t total, void* user) {
    Progress* p = (Progress*)user;
    if (p->quiet) return;
    int width = 40, filled = total ? (int)((uint64_t)done * width / total) : width;
    fprintf(stderr, "\rRendering: %u / %u [", done, total);
    for (int i = 0; i < width; ++i) fputc(i < filled ? '=' : (i == filled ? '>' : '-'), stderr);
    fprintf(stderr, "] %3u %%", total ? (unsigned)((uint64_t)done * 100 / total) : 100u);
    fflush(stderr);
}

int run_render(int argc, char** argv) {
    std::string input, output, profile_path;
    bool have_output = false, have_profile = false, quiet = false, debug_textures = false, stats = false, viewer = false;
    int device = 0;
    std::vector<int> devices;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&](const char* name) -> std::string {
            size_t eq = a.find('=');
            if (a.rfind("--", 0) == 0 && eq != std::string::npos) return a.substr(eq + 1);
            if (i + 1 >= argc) die(std::string("error: a value is required for '") + name + "' but none was supplied");
            return argv[++i];
        };
        if (a == "-h" || a == "--help") {
            usage_render(stdout);
            return 0;
        } else if (a == "-o" || a == "--output" || a.rfind("--output=", 0) == 0) {
            output = value("--output <OUTPUT>");
            have_output = true;
        } else if (a.rfind("-o", 0) == 0 && a.size() > 2 && a[1] == 'o') {
            output = a.substr(2);
            have_output = true;
        } else if (a == "-p" || a == "--profile" || a.rfind("--profile=", 0) == 0) {
            profile_path = value("--profile <PROFILE>");
            have_profile = true;
        } else if (a.rfind("-p", 0) == 0 && a.size() > 2 && a[1] == 'p') {
            profile_path = a.substr(2);
            have_profile = true;
        } else if (a == "-q" || a == "--quiet") quiet = true;
        else if (a == "-v" || a == "--viewer") viewer = true;  // no window system here: progressive output file
        else if (a == "-qv" || a == "-vq") quiet = viewer = true;
        else if (a == "--debug-textures") debug_textures = true;
        else if (a == "--stats") stats = true;
        else if (a == "--devices" || a.rfind("--devices=", 0) == 0) {
            std::string list = value("--devices <A,B,..>");
            devices.clear();
            for (size_t p = 0; p <= list.size();) {
                size_t q = list.find(',', p);
                if (q == std::string::npos) q = list.size();
                if (q == p) die("error: invalid value '" + list + "' for '--devices <A,B,..>'");
                devices.push_back(atoi(list.substr(p, q - p).c_str()));
                p = q + 1;
            }
        } else if (a == "--device" || a.rfind("--device=", 0) == 0) device = atoi(value("--device <N>").c_str());
        else if (a.size() > 1 && a[0] == '-' && a != "-")
            die("error: unexpected argument '" + a + "' found\n\nUsage: path-tracer render [OPTIONS] <INPUT>");
        else if (input.empty()) input = a;
        else die("error: unexpected argument '" + a + "' found\n\nUsage: path-tracer render [OPTIONS] <INPUT>");
    }
    if (input.empty()) die("error: the following required arguments were not provided:\n  <INPUT>\n\nUsage: path-tracer render [OPTIONS] <INPUT>");
    if (!have_output) {
        const char* e = getenv("OUTPUT");
        output = e && *e ? e : "render.png";
    }
    if (!have_profile) {
        const char* e = getenv("PROFILE");
        if (e && *e) {
            profile_path = e;
            have_profile = true;
        }
    }

    // Profile::load / Default (main.rs:33-36)
    pt_profile profile;
    if (pth_profile_load(have_profile ? profile_path.c_str() : nullptr, &profile) != PT_OK) die(pth_last_error());

    auto t0 = std::chrono::steady_clock::now();
    pth_scene* hscene = nullptr;  // load_internal (main.rs:38)
    if (pth_scene_load_isf(input.c_str(), &hscene) != PT_OK) die(pth_last_error());
    auto t1 = std::chrono::steady_clock::now();

    if (devices.size() == 1 || (debug_textures && !devices.empty())) {
        device = devices[0];
        devices.clear();
    }
    pt_scene* scene = nullptr;   // (several devices: every worker thread creates its own below)
    if (devices.empty() && pt_scene_create(pth_scene_desc(hscene), device, &scene) != PT_OK) die(pt_last_error());
    auto t2 = std::chrono::steady_clock::now();

    if (debug_textures) {  // debug_render(&scene, profile.resolution); return (main.rs:40-43)
        static const char* names[PT_DEBUG_PLANES] = {"normal", "albedo", "opacity", "metalness", "roughness", "emissive", "ior"};
        size_t plane = (size_t)profile.width * profile.height * 3;
        std::vector<uint8_t> planes(plane * PT_DEBUG_PLANES);
        int any_hit = 0;
        if (pt_debug_render(scene, profile.width, profile.height, planes.data(), &any_hit) != PT_OK) die(pt_last_error());
        if (any_hit)  // the reference creates the buffers on the first hit: no hit, no files
            for (int p = 0; p < PT_DEBUG_PLANES; ++p)
                if (pth_png_write_rgb8((std::string(names[p]) + ".png").c_str(), profile.width, profile.height,
                                       planes.data() + plane * p) != PT_OK)
                    die(pth_last_error());
        pt_scene_destroy(scene);
        pth_scene_free(hscene);
        return 0;
    }

    if (devices.size() > 1) {
        // One host thread per GPU.  The host work (KD-tree, origin grids) is done ONCE (pt_prep) and uploaded to every
        // device; tile (tx, ty) of the 32x32 grid goes to shard (tx + ty * s) mod N (diagonal stripes, pt_local_pixel_map;
        // the global pixel index stays in the RNG seed, so the assembled image equals the single-GPU one bit for bit).
        // Distinct devices exchange their packed u8 slices with one RCCL all-gather over xGMI and scatter them on the
        // device (pt_gather_tiles); a device list with repeats (several shards on one GPU: a rehearsal) is assembled on
        // the host, RCCL does not allow duplicates.
        // Two phases with a barrier between them: every worker uploads its scene first, and the collective phase is
        // entered only if ALL of them succeeded - a worker that failed would never arrive at the all-gather and the
        // others would wait for it forever.  A failure INSIDE the collective phase (after the peers may already be
        // enqueued in ncclAllGather) ends the process with the message and exit code 2 instead of joining.
        const uint32_t n = (uint32_t)devices.size();
        pt_prep* prep = nullptr;
        if (pt_prep_create(pth_scene_desc(hscene), &prep) != PT_OK) die(pt_last_error());
        bool distinct = true;
        for (uint32_t i = 0; i < n; ++i)
            for (uint32_t j = i + 1; j < n; ++j)
                if (devices[i] == devices[j]) distinct = false;
        std::vector<pt_comm*> comms(n, nullptr);
        if (distinct && pt_comm_create_all(devices.data(), (int)n, comms.data()) != PT_OK) die(pt_last_error());
        uint64_t slice_pixels = 0;
        std::vector<pt_opts> opts(n);
        for (uint32_t k = 0; k < n; ++k) {
            memset(&opts[k], 0, sizeof(pt_opts));
            opts[k].device = devices[k];
            opts[k].shard_rank = k;
            opts[k].shard_count = n;
            opts[k].tile_w = opts[k].tile_h = 32;
            slice_pixels = std::max<uint64_t>(slice_pixels, pt_local_pixel_count(&profile, &opts[k]));
        }
        std::vector<std::vector<uint8_t>> part(n);
        std::vector<std::vector<uint32_t>> map(n);
        std::vector<std::string> error(n);
        std::vector<uint8_t> rgb((size_t)profile.width * profile.height * 3);
        std::mutex bar_mutex;
        std::condition_variable bar_cv;
        uint32_t arrived = 0, failed = 0;
        auto worker = [&](uint32_t k) {
            pt_scene* sc = nullptr;
            auto check = [&](int rc) {   // (the message is thread-local: read it on this thread)
                if (rc != PT_OK && error[k].empty()) error[k] = pt_last_error();
                return rc == PT_OK;
            };
            const bool up = check(pt_scene_create_from_prep(prep, devices[k], &sc));
            {   // barrier: all scenes are resident, or nobody renders
                std::unique_lock<std::mutex> lock(bar_mutex);
                if (!up) ++failed;
                if (++arrived == n) bar_cv.notify_all();
                else bar_cv.wait(lock, [&] { return arrived == n; });
            }
            if (up && failed == 0) {
                if (distinct) {
                    if (!check(pt_render_gathered(sc, comms[k], &profile, &opts[k], slice_pixels, k == 0 ? rgb.data() : nullptr))) {
                        fprintf(stderr, "Error: %s\n", error[k].c_str());   // (the peers may be inside the all-gather: do not join them)
                        fflush(stderr);
                        _exit(2);
                    }
                } else {
                    uint64_t count = pt_local_pixel_count(&profile, &opts[k]);
                    map[k].resize(count);
                    part[k].resize(count * 3);
                    if (check(pt_local_pixel_map(&profile, &opts[k], map[k].data())))
                        check(pt_render(sc, &profile, &opts[k], part[k].data(), nullptr));
                }
            }
            if (sc) pt_scene_destroy(sc);
        };
        std::vector<std::thread> threads;
        for (uint32_t k = 0; k < n; ++k) threads.emplace_back(worker, k);
        for (auto& t : threads) t.join();
        for (pt_comm* c : comms) pt_comm_destroy(c);
        pt_prep_destroy(prep);
        for (uint32_t k = 0; k < n; ++k)
            if (!error[k].empty()) die(error[k]);
        if (!distinct)
            for (uint32_t k = 0; k < n; ++k)
                for (size_t i = 0; i < map[k].size(); ++i) memcpy(&rgb[(size_t)map[k][i] * 3], &part[k][i * 3], 3);
        auto t3 = std::chrono::steady_clock::now();
        if (!quiet)
            fprintf(stderr, "Done: %llds\n", (long long)std::chrono::duration_cast<std::chrono::seconds>(t3 - t2).count());
        size_t dot = output.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : output.substr(dot + 1);
        for (char& c : ext) c = (char)tolower(c);
        if (ext != "png") die("The image format could not be determined (only .png output is supported): " + output);
        if (pth_png_write_rgb8(output.c_str(), profile.width, profile.height, rgb.data()) != PT_OK) die(pth_last_error());
        if (stats) {
            double sec = std::chrono::duration<double>(t3 - t2).count();
            fprintf(stderr, "{\"devices\": %u, \"gather\": \"%s\", \"render_s\": %.3f, \"msamples_per_s\": %.2f}\n", n,
                    distinct ? "rccl" : "host", sec, (double)profile.width * profile.height * profile.samples / sec / 1e6);
        }
        pth_scene_free(hscene);
        return 0;
    }

    // Renderer::new + render (main.rs:46-47)
    Progress prog{quiet, std::chrono::steady_clock::now()};
    pt_opts opts;
    memset(&opts, 0, sizeof opts);
    opts.device = device;
    opts.flags = PT_FLAG_TIMING;
    if (!quiet) {
        opts.progress = on_progress;
        opts.progress_user = &prog;
    }
    if (!quiet || viewer) opts.sample_batch = profile.samples > 16 ? (profile.samples + 15) / 16 : 1;  // ~16 ticks
    Preview pv{output, profile.width, profile.height};
    if (viewer) {
        size_t dot = output.find_last_of('.');
        std::string ext = dot == std::string::npos ? "" : output.substr(dot + 1);
        if (ext == "png" || ext == "PNG") {
            opts.preview = on_preview;
            opts.preview_user = &pv;
        }
    }
    std::vector<uint8_t> rgb((size_t)profile.width * profile.height * 3);
    if (pt_render(scene, &profile, &opts, rgb.data(), nullptr) != PT_OK) die(pt_last_error());
    auto t3 = std::chrono::steady_clock::now();
    if (!quiet)
        fprintf(stderr, "\nDone: %llds\n", (long long)std::chrono::duration_cast<std::chrono::seconds>(t3 - t2).count());

    // rendered_image.save(output) (main.rs:50); the format follows the extension, PNG only here
    size_t dot = output.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : output.substr(dot + 1);
    for (char& c : ext) c = (char)tolower(c);
    if (ext != "png") die("The image format could not be determined (only .png output is supported): " + output);
    if (pth_png_write_rgb8(output.c_str(), profile.width, profile.height, rgb.data()) != PT_OK) die(pth_last_error());
    auto t4 = std::chrono::steady_clock::now();

    if (stats) {
        pt_timing tm{};
        pt_scene_info info{};
        pt_get_timing(scene, &tm);
        pt_scene_get_info(scene, &info);
        auto sec = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
        double samples = (double)profile.width * profile.height * profile.samples;
        fprintf(stderr,
                // load = ISF + textures; build = pt_scene_create (KD-tree and origin grids side by side on the host, upload);
                // render = pt_render (kernels + the copy back); png = the output file
                "{\"load_s\": %.3f, \"build_s\": %.3f, \"kd_build_s\": %.3f, \"grid_build_s\": %.3f, \"upload_s\": %.3f, "
                "\"render_s\": %.3f, \"kernel_ms\": %.3f, \"png_s\": %.3f, \"total_s\": %.3f, \"msamples_per_s\": %.2f, "
                "\"prims\": %llu, \"kd_nodes\": %llu, \"leaf_refs\": %llu, \"cam_grid_res\": %u, \"light_grids\": %u}\n",
                sec(t0, t1), sec(t1, t2), (double)info.kd_build_seconds, (double)info.grid_build_seconds, (double)info.upload_seconds,
                sec(t2, t3), (double)tm.integrate_ms, sec(t3, t4), sec(t0, t4), samples / sec(t2, t3) / 1e6,
                (unsigned long long)info.n_prims, (unsigned long long)info.n_kd_nodes, (unsigned long long)info.n_leaf_refs,
                info.cam_grid_res, info.light_grids);
    }
    pt_scene_destroy(scene);
    pth_scene_free(hscene);
    return 0;
}

// path-tracer convert <INPUT> <OUTPUT> (config/mod.rs:44-52, main.rs:54-57)
int run_convert(int argc, char** argv) {
    std::vector<std::string> pos;
    for (int i = 0; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "-h" || a == "--help") {
            fputs("Convert scenes into ISF format\n\n"
                  "Usage: path-tracer convert <INPUT> <OUTPUT>\n\n"
                  "Arguments:\n"
                  "  <INPUT>   Input file name gltf format\n"
                  "  <OUTPUT>  Output directory name\n\n"
                  "Options:\n"
                  "  -h, --help  Print help\n",
                  stdout);
            return 0;
        }
        if (a.size() > 1 && a[0] == '-') die("error: unexpected argument '" + a + "' found\n\nUsage: path-tracer convert <INPUT> <OUTPUT>");
        pos.push_back(a);
    }
    if (pos.size() < 2)
        die(std::string("error: the following required arguments were not provided:\n") + (pos.empty() ? "  <INPUT>\n" : "") +
            "  <OUTPUT>\n\nUsage: path-tracer convert <INPUT> <OUTPUT>");
    if (pos.size() > 2) die("error: unexpected argument '" + pos[2] + "' found\n\nUsage: path-tracer convert <INPUT> <OUTPUT>");
    if (pth_convert_gltf(pos[0].c_str(), pos[1].c_str()) != PT_OK) die(pth_last_error());
    return 0;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 2) {
        usage_main(stderr);
        return 2;
    }
    std::string cmd = argv[1];
    if (cmd == "render") return run_render(argc - 2, argv + 2);
    if (cmd == "convert") return run_convert(argc - 2, argv + 2);
    if (cmd == "-h" || cmd == "--help" || cmd == "help") {
        usage_main(stdout);
        return 0;
    }
    if (cmd == "-V" || cmd == "--version") {
        printf("path-tracer 0.1.0 (%s)\n", pt_version());
        return 0;
    }
    die("error: unrecognized subcommand '" + cmd + "'\n\nUsage: path-tracer <COMMAND>");
}
