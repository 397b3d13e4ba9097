// main.cpp -- command-line driver.  Like the reference's `./main [out.ppm]` (main.cpp:1212-1288) it
// builds a scene, renders it and writes an ASCII PPM; the scene, size and sampling are selectable:
//   rt_main [out.ppm] [--preset shells|analytic|teapot|teapot_head|textured] [--size WxH]
//           [--mode 1spp|aa9|gridN] [--depth D] [--assets DIR] [--brute] [--p6] [--device N]
//           [--frames N [--spin DEGREES]] [--gpus N | --devices a,b,...]
// Defaults reproduce the reference: preset shells, 1024x768, 1 spp, ./result.ppm.  An output name ending
// in .png is written as an 8-bit RGB PNG.
// --frames N renders N frames, turning every mesh by DEGREES (default 360 / N) about its own y axis
// per frame, and writes out_0000.ppm ...; the geometry is uploaded and its LBVHs are built once.
// --gpus N (devices 0 .. N-1) or --devices a,b,... renders every frame on several GPUs from this one process: a context
// and a host thread per device, each rendering its rank's tiles (rt_render_params.rank / world_size) into a buffer of
// its own; the buffers are merged on the host.  No collective is involved; bench.py's pipeline is the fast path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <stdexcept>
#include <thread>

#include "scene.hpp"

using namespace rta;

int main(int argc, const char *argv[]) {
    std::string out = "./result.ppm", preset = "shells", assets = "assets";
    int width = 1024, height = 768, mode = RT_MODE_REF_1SPP, grid = 1, depth = 8, device = 0, accel = RT_ACCEL_LBVH;
    bool p6 = false;
    int frames = 1;
    float spin = 0;
    std::vector<int> devices;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto next = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--preset") preset = next();
        else if (a == "--assets") assets = next();
        else if (a == "--size") { if (std::sscanf(next(), "%dx%d", &width, &height) != 2) { std::cerr << "bad --size\n"; return 2; } }
        else if (a == "--mode") {
            std::string m = next();
            if (m == "1spp") mode = RT_MODE_REF_1SPP;
            else if (m == "aa9") mode = RT_MODE_REF_AA9;
            else if (m.rfind("grid", 0) == 0) { mode = RT_MODE_GRID; grid = std::atoi(m.c_str() + 4); }
            else { std::cerr << "bad --mode\n"; return 2; }
        } else if (a == "--depth") depth = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--gpus") { int n = std::atoi(next()); devices.clear(); for (int d = 0; d < n; d++) devices.push_back(d); }
        else if (a == "--devices") {
            devices.clear();
            for (const char *q = next(); *q;) { devices.push_back(std::atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
        }
        else if (a == "--brute") accel = RT_ACCEL_BRUTE;
        else if (a == "--p6") p6 = true;
        else if (a == "--frames") frames = std::atoi(next());
        else if (a == "--spin") spin = (float)std::atof(next());
        else if (a.rfind("--", 0) == 0) { std::cerr << "unknown option " << a << "\n"; return 2; }
        else out = a;
    }
    try {
        auto t0 = std::chrono::steady_clock::now();
        std::cout << "Defining the scene..." << std::endl;
        PresetHolder holder;
        std::string err;
        if (!build_preset(preset, assets, holder, err)) throw std::runtime_error(err);
        if (devices.empty()) devices.push_back(device);
        const int world = (int)devices.size();
        std::vector<std::unique_ptr<Renderer>> renderers;
        for (int d : devices) {
            renderers.emplace_back(new Renderer(d));
            renderers.back()->upload(holder.scene);
        }
        std::cout << "Coloring each pixel on the screen..." << std::endl;
        rt_render_params p{};
        p.width = width; p.height = height; p.fov = 90; p.mode = mode; p.grid_m = grid; p.max_depth = depth;
        p.accel = accel; p.rank = 0; p.world_size = 1;
        std::vector<uint8_t> rgb8(3 * (size_t)width * height);
        if (frames < 1) frames = 1;
        if (spin == 0) spin = 360.0f / frames;
        // the meshes (behind their Box gates) and their poses in frame 0
        std::vector<Object *> movers;
        std::vector<mat4> base;
        for (Object *o : holder.scene.objects) {
            Object *target = o;
            if (Box *bx = dynamic_cast<Box *>(o)) target = bx->obj;
            if (dynamic_cast<Mesh *>(target)) { movers.push_back(target); base.push_back(target->transformation()); }
            if (target != o) { movers.push_back(o); base.push_back(o->transformation()); }
        }
        unsigned long long rays_before = 0;
        for (int f = 0; f < frames; f++) {
            if (f > 0) {
                float angle = (float)(spin * f * 3.14159265358979323846 / 180.0);
                for (size_t k = 0; k < movers.size(); k++) movers[k]->setTransformation(rotate(base[k], angle, vec3(0, 1, 0)));
                for (auto &r : renderers) r->update(holder.scene);
            }
            unsigned long long rays_total = 0;
            float trace_ms = 0, build_ms = 0;
            if (world == 1) {
                renderers[0]->render(p, nullptr, rgb8.data());
            } else {
                // every rank's tiles into a zeroed buffer of its own (rt_render leaves the other ranks' pixels as they are), then merged
                std::vector<std::vector<uint8_t>> part(world, std::vector<uint8_t>(rgb8.size(), 0));
                std::vector<std::string> errors(world);
                std::vector<std::thread> threads;
                for (int r = 0; r < world; r++)
                    threads.emplace_back([&, r]() {
                        try {
                            rt_render_params pr = p;
                            pr.rank = r; pr.world_size = world;
                            renderers[r]->render(pr, nullptr, part[r].data());
                        } catch (const std::exception &e) { errors[r] = e.what(); }
                    });
                for (auto &t : threads) t.join();
                for (int r = 0; r < world; r++) if (!errors[r].empty()) throw std::runtime_error("device " + std::to_string(devices[r]) + ": " + errors[r]);
                for (size_t q = 0; q < rgb8.size(); q++) {
                    uint8_t v = 0;
                    for (int r = 0; r < world; r++) v |= part[r][q];  // one rank owns the pixel, the others hold 0
                    rgb8[q] = v;
                }
            }
            for (auto &r : renderers) {
                rt_counters c = r->counters();
                rays_total += c.rays_primary + c.rays_shadow + c.rays_secondary;
                trace_ms = std::max(trace_ms, c.last_trace_ms); build_ms = std::max(build_ms, c.last_build_ms);
            }
            unsigned long long rays = rays_total - rays_before;
            rays_before = rays_total;  // the counters accumulate over frames
            std::cout << "frame " << f << ": trace kernel " << trace_ms << " ms" << (world > 1 ? " (slowest of " + std::to_string(world) + " devices)" : std::string()) << ", " << rays
                      << " rays, " << rays / (trace_ms * 1e3) << " Mrays/s; LBVH build " << build_ms << " ms (once)" << std::endl;
            std::string path = out;
            if (frames > 1) {
                char tag[16];
                std::snprintf(tag, sizeof tag, "_%04d", f);
                size_t dot = out.rfind('.');
                path = dot == std::string::npos ? out + tag : out.substr(0, dot) + tag + out.substr(dot);
            }
            Image image(width, height);
            image.setPixels8(rgb8.data());
            bool png = path.size() > 4 && path.compare(path.size() - 4, 4, ".png") == 0;
            if (png) { if (!image.writeImagePNG(path.c_str())) throw std::runtime_error("cannot write " + path); }
            else if (p6) image.writeImageBinary(path.c_str());
            else image.writeImage(path.c_str());
        }
        double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "It took " << secs << " seconds to render the image." << std::endl;
        std::cout << "I could render at " << frames / secs << " frames per second." << std::endl;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
