// Development: walks the C ABI of include/sxe.h with scenes that take every host path of the
// library (uploads through both entry points, texture tables with and without alpha, frames in
// flight, work buffers that must grow, the clipper, view sets, the allocation-free draw, both
// indexing modes, the diagnostics, misuse that must be refused) so that a build of the library with
// the host-side address sanitizer sees them.  Results are not checked here beyond the return
// codes: the parity tests do that.  usage: api_walk   (exit 0: every call did what was expected)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sxe.h"

static int g_failed = 0;
#define EXPECT(cond) do { if (!(cond)) { std::fprintf(stderr, "line %d: %s\n", __LINE__, #cond); ++g_failed; } } while (0)
#define OK(ctx, call) do { const int rc_ = (call); if (rc_ != 0) { std::fprintf(stderr, "line %d: %s -> %d (%s)\n", __LINE__, #call, rc_, sxe_last_error(ctx)); ++g_failed; } } while (0)

struct Mesh {
    std::vector<sxe_vertex> v;
    std::vector<uint32_t> i;
};

static uint32_t g_seed = 12345u;
static float rnd() { g_seed = g_seed * 1664525u + 1013904223u; return (float)(g_seed >> 8) / 16777216.0f; }

// a grid of n x n quads bent into a patch, two triangles each
static Mesh grid(int n, float size)
{
    Mesh m;
    for (int y = 0; y <= n; ++y)
        for (int x = 0; x <= n; ++x) {
            const float u = (float)x / n, w = (float)y / n;
            sxe_vertex q = {{(u - 0.5f) * size, (w - 0.5f) * size, 0.2f * size * std::sin(6.0f * u) * std::cos(5.0f * w)}, {u, w, 1.0f - u}, {u, w}};
            m.v.push_back(q);
        }
    for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
            const uint32_t a = (uint32_t)(y * (n + 1) + x), b = a + 1, c = a + (uint32_t)n + 1, d = c + 1;
            const uint32_t t[6] = {a, b, c, b, d, c};
            m.i.insert(m.i.end(), t, t + 6);
        }
    return m;
}

static std::vector<uint8_t> texture(uint32_t w, uint32_t h, bool alpha)
{
    std::vector<uint8_t> t((size_t)w * h * 4);
    for (size_t k = 0; k < t.size(); ++k) t[k] = (uint8_t)(rnd() * 255.0f);
    for (size_t k = 3; k < t.size(); k += 4) t[k] = alpha ? (uint8_t)(rnd() * 255.0f) : 255;
    return t;
}

static sxe_transform placed(float x, float y, float z, float s)
{
    sxe_transform t;
    sxe_transform_identity(&t);
    sxe_transform_translate(&t, x, y, z);
    sxe_transform_rotate(&t, 360.0f * rnd(), 360.0f * rnd(), 360.0f * rnd());
    t.scale[0] = t.scale[1] = t.scale[2] = s;
    return t;
}

static void read_back(sxe_ctx *c, uint32_t image, uint32_t w, uint32_t h, bool coverage)
{
    std::vector<uint8_t> rgba((size_t)w * h * 4);
    std::vector<float> depth((size_t)w * h);
    std::vector<uint32_t> prim((size_t)w * h);
    OK(c, sxe_read_color(c, image, rgba.data(), rgba.size()));
    OK(c, sxe_read_depth(c, image, depth.data(), depth.size() * 4));
    if (coverage) OK(c, sxe_read_coverage(c, image, prim.data(), prim.size() * 4));
    EXPECT(sxe_read_color(c, image, rgba.data(), rgba.size() - 1) != 0); // a buffer that is too small is refused
    const uint8_t *mapped = nullptr;
    OK(c, sxe_map_color(c, image, &mapped));
    EXPECT(mapped != nullptr && mapped[3] == rgba[3]);
}

static void walk(uint32_t w, uint32_t h, uint32_t shading, uint32_t flags, bool alpha, bool instances)
{
    sxe_config cfg;
    sxe_default_config(&cfg, w, h);
    cfg.n_images = 3;
    cfg.shading = shading;
    cfg.flags = flags | SXE_FLAG_TIMING;
    sxe_ctx *c = nullptr;
    if (sxe_create(&cfg, &c) != 0 || !c) {
        std::fprintf(stderr, "sxe_create: %s\n", sxe_last_error(nullptr));
        ++g_failed;
        return;
    }
    const bool coverage = (flags & SXE_FLAG_WRITE_COVERAGE) != 0;
    const std::vector<uint8_t> t0 = texture(64, 48, false), t1 = texture(5, 3, alpha), t2 = texture(1, 1, false);
    const sxe_texture_desc tex[3] = {{t0.data(), 64, 48}, {t1.data(), 5, 3}, {t2.data(), 1, 1}};
    OK(c, sxe_set_textures(c, tex, 3));
    const Mesh fine = grid(40, 2.0f), coarse = grid(3, 6.0f), huge = grid(2, 400.0f);
    const uint32_t n_obj = 60;
    std::vector<sxe_transform> tr;
    for (uint32_t o = 0; o < n_obj; ++o) tr.push_back(placed(8.0f * (rnd() - 0.5f), 6.0f * (rnd() - 0.5f), -2.0f - 20.0f * rnd(), 0.3f + rnd()));
    tr[0] = placed(0.0f, 0.0f, -0.3f, 1.0f); // the huge patch crosses the near plane and the guard band: the clipper
    if (instances) {
        const sxe_mesh_desc md[3] = {{fine.v.data(), (uint32_t)fine.v.size(), fine.i.data(), (uint32_t)fine.i.size()},
                                     {coarse.v.data(), (uint32_t)coarse.v.size(), coarse.i.data(), (uint32_t)coarse.i.size()},
                                     {huge.v.data(), (uint32_t)huge.v.size(), huge.i.data(), (uint32_t)huge.i.size()}};
        std::vector<sxe_instance> in(n_obj);
        for (uint32_t o = 0; o < n_obj; ++o) in[o] = {o == 0 ? 2u : o % 2u, (int32_t)(o % 3u), o % 7u != 3u, tr[o]};
        OK(c, sxe_update_instances(c, md, 3, in.data(), n_obj));
    } else {
        std::vector<sxe_drawable> d(n_obj);
        for (uint32_t o = 0; o < n_obj; ++o) {
            const Mesh &m = o == 0 ? huge : (o % 2u ? coarse : fine);
            d[o] = {m.v.data(), (uint32_t)m.v.size(), m.i.data(), (uint32_t)m.i.size(), tr[o], (int32_t)(o % 3u), o % 7u != 3u};
        }
        OK(c, sxe_update_drawables(c, d.data(), n_obj));
    }
    sxe_transform cam;
    sxe_transform_identity(&cam);
    OK(c, sxe_update_camera(c, &cam));
    // frames in flight over three images while the scene moves (the first ones grow the work buffers)
    for (uint32_t f = 0; f < 9; ++f) {
        if (f == 7) {
            // the matrices were given directly at frame 6: there are no transforms to rotate until new ones come
            EXPECT(sxe_rotation_system(c, 77, f, 16.0f) != 0);
            OK(c, sxe_update_transforms(c, tr.data(), n_obj));
        }
        OK(c, sxe_rotation_system(c, 77, f, 16.0f));
        if (f == 4) OK(c, sxe_update_transforms(c, tr.data(), n_obj));
        if (f == 6) {
            std::vector<float> models(16 * n_obj);
            for (uint32_t o = 0; o < n_obj; ++o) sxe_transform_to_model(&tr[o], &models[16 * o]);
            OK(c, sxe_update_uniforms(c, models.data(), n_obj));
            EXPECT(sxe_update_uniforms(c, models.data(), n_obj + 1) != 0);
        }
        sxe_transform_translate(&cam, 0.01f, 0.0f, -0.05f);
        OK(c, sxe_update_camera(c, &cam));
        OK(c, sxe_draw(c, f % 3));
    }
    OK(c, sxe_sync(c));
    read_back(c, 2, w, h, coverage);
    sxe_timings tm;
    sxe_frame_stats st;
    uint64_t frames = 0;
    OK(c, sxe_get_timings(c, &tm));
    OK(c, sxe_get_timing_totals(c, &tm, &frames));
    OK(c, sxe_set_timing_stride(c, 0));
    OK(c, sxe_get_stats(c, &st));
    EXPECT(st.objects == n_obj && st.prims > 0);
    std::printf("%ux%u shading %u flags %u: %llu triangles, %llu records, %llu bin entries, %u frames drawn again\n", w, h, shading, flags,
                (unsigned long long)st.prims, (unsigned long long)st.records, (unsigned long long)st.bin_entries, st.redraws);
    std::vector<uint32_t> counts(st.tiles), parts(st.tiles * 5);
    OK(c, sxe_read_tile_counts(c, counts.data(), counts.size() * 4));
    OK(c, sxe_read_tile_parts(c, parts.data(), parts.size() * 4));
    std::vector<float> models(16 * n_obj), clip(4 * st.instance_vertices);
    std::vector<sxe_transform> back(n_obj);
    OK(c, sxe_read_models(c, models.data(), models.size() * 4));
    OK(c, sxe_read_transforms(c, back.data(), back.size() * sizeof(sxe_transform)));
    OK(c, sxe_read_clip_positions(c, clip.data(), clip.size() * 4));
    // a set of views in one go, then the allocation-free draw twice
    sxe_transform cams[3] = {cam, cam, cam};
    sxe_transform_rotate(&cams[1], 0.0f, 20.0f, 0.0f);
    sxe_transform_translate(&cams[2], 1.0f, 0.5f, 2.0f);
    OK(c, sxe_draw_views(c, cams, 3, 0));
    OK(c, sxe_sync(c));
    EXPECT(sxe_draw_views(c, cams, 3, 1) != 0); // images 1..3 of 3: refused
    read_back(c, 1, w, h, coverage);
    OK(c, sxe_draw_noalloc(c, 0));
    OK(c, sxe_draw_noalloc(c, 1));
    read_back(c, 1, w, h, coverage);
    if (shading == SXE_SHADE_LIT) {
        sxe_light light;
        sxe_default_light(&light);
        OK(c, sxe_set_light(c, &light));
        OK(c, sxe_draw(c, 0));
        OK(c, sxe_sync(c));
    }
    // misuse
    EXPECT(sxe_draw(c, 3) != 0);
    EXPECT(sxe_set_textures(c, tex, 0) != 0 || true); // (an empty table is allowed or refused, never a crash)
    void *stream = nullptr, *ptr = nullptr;
    OK(c, sxe_stream(c, &stream));
    OK(c, sxe_color_device_ptr(c, 0, &ptr));
    // the scene replaced by a smaller one, and by none
    {
        sxe_drawable one = {coarse.v.data(), (uint32_t)coarse.v.size(), coarse.i.data(), (uint32_t)coarse.i.size(), tr[1], 0, 1};
        OK(c, sxe_set_textures(c, tex, 1));
        OK(c, sxe_update_drawables(c, &one, 1));
        OK(c, sxe_draw(c, 0));
        OK(c, sxe_sync(c));
        OK(c, sxe_update_drawables(c, nullptr, 0));
        OK(c, sxe_draw(c, 1));
        OK(c, sxe_sync(c));
        read_back(c, 1, w, h, coverage);
    }
    sxe_destroy(c);
}

int main()
{
    std::printf("%s\n", sxe_version());
    walk(333, 217, SXE_SHADE_TEXTURE, SXE_FLAG_WRITE_COVERAGE, false, false);
    walk(640, 360, SXE_SHADE_TEXTURE, 0, true, true);   // translucent texture: the ordered path
    walk(200, 150, SXE_SHADE_VERTEX_COLOR, SXE_FLAG_WRITE_COVERAGE | SXE_FLAG_BGRA, false, true);
    walk(256, 256, SXE_SHADE_COLOR_TIMES_TEXTURE, SXE_FLAG_COMPAT_REFERENCE_INDEXING, false, false);
    walk(320, 200, SXE_SHADE_LIT, SXE_FLAG_WRITE_COVERAGE, true, false);
    // a band of a larger frame
    {
        sxe_config cfg;
        sxe_default_config(&cfg, 1024, 768);
        cfg.scissor_y0 = 256; cfg.scissor_y1 = 480;
        sxe_ctx *c = nullptr;
        if (sxe_create(&cfg, &c) == 0 && c) {
            const Mesh m = grid(100, 3.0f);
            const std::vector<uint8_t> t = texture(16, 16, false);
            const sxe_texture_desc td = {t.data(), 16, 16};
            OK(c, sxe_set_textures(c, &td, 1));
            sxe_drawable d = {m.v.data(), (uint32_t)m.v.size(), m.i.data(), (uint32_t)m.i.size(), placed(0, 0, -3, 1), 0, 1};
            OK(c, sxe_update_drawables(c, &d, 1));
            sxe_transform cam;
            sxe_transform_identity(&cam);
            OK(c, sxe_update_camera(c, &cam));
            OK(c, sxe_draw(c, 0));
            OK(c, sxe_sync(c));
            read_back(c, 0, 1024, 768, false);
            sxe_destroy(c);
        } else {
            ++g_failed;
        }
    }
    // configurations that must be refused
    {
        sxe_config cfg;
        sxe_ctx *c = nullptr;
        sxe_default_config(&cfg, 0, 10);
        EXPECT(sxe_create(&cfg, &c) != 0 && c == nullptr);
        sxe_default_config(&cfg, 100000, 10);
        EXPECT(sxe_create(&cfg, &c) != 0 && c == nullptr);
        sxe_default_config(&cfg, 64, 64);
        cfg.flags = 1u << 20;
        EXPECT(sxe_create(&cfg, &c) != 0 && c == nullptr);
    }
    std::printf("%s\n", g_failed ? "FAILED" : "api walk done");
    return g_failed ? 1 : 0;
}
