// Runs the CPU code that has no GPU in it -- the host object API (presets, OBJ and PNG loaders, flattening), the CPU twin of
// the per-ray code and the oracle -- on small frames, for a build with AddressSanitizer and UndefinedBehaviorSanitizer
// (tests/test_sanitizers.py builds and runs it).  Test infrastructure: nothing here is part of the product.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "scene.hpp"
extern "C" {
#include "rt_oracle.h"
}

extern "C" {
void *twin_create(const rt_scene_desc *sc, char *errbuf, int errcap);
void twin_destroy(void *h);
int twin_build_accel(void *h);
void twin_trace_rays(void *h, int64_t n, const float *o, const float *d, int max_depth, int accel, rt_hit_record *hits, float *radiance, int *max_stack);
void twin_render(void *h, const rt_render_params *p, int x0, int y0, int x1, int y1, float *rgb, uint8_t *rgb8, unsigned long long *counters);

// The object API's Renderer forwards to the GPU library, which this program neither links nor calls.
#define NO_GPU(name, ...) int name(__VA_ARGS__) { std::fprintf(stderr, #name " called in the sanitizer driver\n"); std::abort(); }
NO_GPU(rt_context_create, int, rt_ctx **)
NO_GPU(rt_scene_upload, rt_ctx *, const rt_scene_desc *)
NO_GPU(rt_scene_update, rt_ctx *, const rt_scene_desc *)
NO_GPU(rt_build_accel, rt_ctx *)
NO_GPU(rt_render, rt_ctx *, const rt_render_params *, float *, uint8_t *)
NO_GPU(rt_trace_rays, rt_ctx *, int64_t, const float *, const float *, int32_t, int32_t, rt_hit_record *, float *)
NO_GPU(rt_get_counters, rt_ctx *, rt_counters *)
NO_GPU(rt_intersect_object, rt_ctx *, int32_t, int64_t, const float *, const float *, int32_t, rt_hit_record *)
int rt_context_destroy(rt_ctx *) { return 0; }
const char *rt_last_error(const rt_ctx *) { return ""; }
}

static int fail(const std::string &what) { std::fprintf(stderr, "FAIL: %s\n", what.c_str()); return 1; }

int main(int argc, char **argv) {
    const std::string assets = argc > 1 ? argv[1] : "assets";
    const int w = 40, h = 30;
    std::vector<float> o(3 * w * h, 0.0f), d(3 * w * h);
    for (int j = 0; j < h; j++)
        for (int i = 0; i < w; i++) {
            float x = (i + 0.5f) / w * 2 - 1, y = 1 - (j + 0.5f) / h * 2, z = 1, l = std::sqrt(x * x + y * y + z * z);
            float *q = &d[3 * (j * w + i)];
            q[0] = x / l; q[1] = y * 0.75f / l; q[2] = z / l;
        }
    const char *presets[] = {"teapot", "analytic", "shells"};
    for (const char *name : presets) {
        rta::PresetHolder holder;
        std::string err;
        if (!rta::build_preset(name, assets, holder, err)) return fail(std::string(name) + ": " + err);
        rta::FlatScene flat;
        if (!holder.scene.flatten(flat, err)) return fail(std::string(name) + ": flatten: " + err);
        char buf[256] = "";
        void *tw = twin_create(&flat.desc, buf, sizeof buf);
        if (!tw) return fail(std::string(name) + ": twin_create: " + buf);
        if (twin_build_accel(tw) < 0) return fail(std::string(name) + ": twin_build_accel");
        std::vector<rt_hit_record> hits(w * h);
        std::vector<float> rad(3 * w * h), rad_brute(3 * w * h);
        int deepest = 0;
        twin_trace_rays(tw, w * h, o.data(), d.data(), 6, RT_ACCEL_LBVH, hits.data(), rad.data(), &deepest);
        twin_trace_rays(tw, w * h, o.data(), d.data(), 6, RT_ACCEL_BRUTE, nullptr, rad_brute.data(), nullptr);
        if (std::memcmp(rad.data(), rad_brute.data(), rad.size() * sizeof(float)) != 0 && std::strcmp(name, "shells") != 0)
            return fail(std::string(name) + ": tree and linear scan differ");
        rt_render_params p{};
        p.width = w; p.height = h; p.fov = 90; p.mode = RT_MODE_REF_AA9; p.grid_m = 1; p.max_depth = 6; p.accel = RT_ACCEL_LBVH; p.world_size = 1;
        std::vector<uint8_t> rgb8(3 * w * h);
        unsigned long long counters[16] = {0};
        twin_render(tw, &p, 0, 0, w, h, nullptr, rgb8.data(), counters);
        twin_destroy(tw);
        if (std::strcmp(name, "shells") == 0) continue;  // its oracle scene needs the texture sets; the twin alone ran above
        orc_scene *os = orc_scene_new();
        if (orc_build_preset(os, name, assets.c_str(), nullptr) != 0) return fail(std::string(name) + ": orc_build_preset");
        std::vector<orc_hit> oh(w * h);
        orc_find_closest_hit(os, w * h, o.data(), d.data(), oh.data());
        std::vector<float> orad(3 * w * h);
        std::vector<uint64_t> rays(w * h);
        orc_trace_rays(os, w * h, o.data(), d.data(), 6, orad.data(), rays.data());
        orc_scene_free(os);
        int bad = 0;
        for (int r = 0; r < w * h; r++) {
            if (oh[r].hit != hits[r].hit || (oh[r].hit && oh[r].object != hits[r].object)) bad++;
            for (int c = 0; c < 3; c++) {
                float a = orad[3 * r + c], b = rad[3 * r + c];
                if (std::fabs(a - b) > 1e-4f * std::fmax(std::fabs(a), 1e-3f)) bad++;
            }
        }
        if (bad > 2) return fail(std::string(name) + ": " + std::to_string(bad) + " rays differ from the oracle");
        std::printf("%s ok (stack %d)\n", name, deepest);
    }
    std::printf("sanitizer driver: all presets ran clean\n");
    return 0;
}
