// sxe_drawer.hpp -- C++17 host side above the C ABI of sxe.h, shaped like the reference's seam.
//
// The reference is Rust; where no Rust toolchain exists this header gives the same interface in
// C++: the `Drawer` trait (src/renderer/drawer.rs:26-31) with its four methods, the value types
// that cross it (Drawable src/primitives/drawable.rs:7-12, Mesh src/components/mesh.rs:4-9,
// Texture src/components/texture.rs:4-6, Transform src/components/transform.rs:16-80,
// ObjectUniformBufferObject src/primitives/uniform_buffer_object.rs:42-53) and the part of
// `Presenter` (src/renderer/presenter.rs:16-22) that makes sense without a swapchain.  Method
// names, argument meaning and error behaviour follow the reference: the update_* methods return a
// Result-like Status (the reference's Result<(), String>), draw() has no result and throws where
// the reference panics (src/renderer/drawer.rs:453).
//
// Header-only; link with libsxe_hip.so.  Nothing here computes: every call ends in the C ABI.
#ifndef SXE_DRAWER_HPP
#define SXE_DRAWER_HPP

#include "sxe.h"

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <map>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace sxe_host {

// Result<(), String>: empty error means Ok(()).
struct Status {
    std::string error;
    bool ok() const { return error.empty(); }
    explicit operator bool() const { return ok(); }
    static Status Ok() { return Status{}; }
    static Status Err(std::string e) { return Status{std::move(e)}; }
    // .unwrap() of the reference's call sites (src/main.rs:191-192)
    void unwrap() const
    {
        if (!ok()) throw std::runtime_error(error);
    }
};

using Vertex = sxe_vertex; // src/primitives/vertex.rs:2-6, same 32-byte layout

struct ObjectUniformBufferObject { // src/primitives/uniform_buffer_object.rs:42-53
    float model[16];               // column-major
};

struct Transform { // src/components/transform.rs:16-21
    sxe_transform raw;
    Transform() { sxe_transform_identity(&raw); }                           // Transform::new, :24-30
    void translate(float x, float y, float z) { sxe_transform_translate(&raw, x, y, z); } // :33-35
    void rotate(float x, float y, float z) { sxe_transform_rotate(&raw, x, y, z); }       // :37-46, degrees
    ObjectUniformBufferObject to_ubo() const                                // :69-80
    {
        ObjectUniformBufferObject u;
        sxe_transform_to_model(&raw, u.model);
        return u;
    }
};

struct Mesh { // src/components/mesh.rs:4-9
    std::string key;
    std::vector<Vertex> vertices;
    std::vector<uint32_t> indices;
    bool rendered = true;
};

struct Texture { // src/components/texture.rs:4-6; ordered by path like the reference's Ord impl
    std::string path;
    bool operator<(const Texture &o) const { return path < o.path; }
    bool operator==(const Texture &o) const { return path == o.path; }
};

struct Drawable { // src/primitives/drawable.rs:7-12 (the colour field is never read by the drawer)
    Mesh mesh;
    Transform transform;
    std::optional<Texture> texture;
};

// Decoded texels of one texture file: RGBA8, sRGB-encoded colour, rows tightly packed
// (what TextureData::load hands to the allocator, src/renderer/types.rs:158-188).
struct Image {
    uint32_t width = 0, height = 0;
    std::vector<uint8_t> rgba;
};

struct Viewport { // the fields of hal::pso::Viewport that presenter.rs:373-383 sets
    uint32_t x = 0, y = 0, w = 0, h = 0;
    float depth_near = 0.0f, depth_far = 1.0f;
};

// trait Drawer (src/renderer/drawer.rs:26-31).  The semaphore arguments of draw() belong to the
// Vulkan queue and have no counterpart on a HIP stream, which orders the frames by itself.
class Drawer {
  public:
    virtual ~Drawer() = default;
    virtual void draw(size_t image_index) = 0;
    virtual Status update_drawables(std::vector<Drawable> drawables) = 0;
    virtual Status update_uniforms(std::vector<ObjectUniformBufferObject> uniforms) = 0;
    virtual Status update_camera(Transform transform) = 0;
};

// GfxDrawer's stand-in (src/renderer/drawer.rs:33-160).
class HipDrawer final : public Drawer {
  public:
    // GfxDrawer::new (src/renderer/drawer.rs:55): viewport extent, number of swapchain images.
    static Status create(const Viewport &viewport, uint32_t n_images, int device, std::unique_ptr<HipDrawer> &out,
                         uint32_t shading = SXE_SHADE_TEXTURE, uint32_t flags = 0)
    {
        sxe_config cfg;
        sxe_default_config(&cfg, viewport.w, viewport.h);
        cfg.device = device;
        cfg.n_images = n_images;
        cfg.shading = shading;
        cfg.flags = flags;
        sxe_ctx *ctx = nullptr;
        const int rc = sxe_create(&cfg, &ctx);
        if (rc != SXE_OK) return Status::Err(message(nullptr, rc));
        out.reset(new HipDrawer(ctx, viewport, shading));
        return Status::Ok();
    }

    ~HipDrawer() override { sxe_destroy(ctx_); } // Drop, src/renderer/drawer.rs:360-388
    HipDrawer(const HipDrawer &) = delete;
    HipDrawer &operator=(const HipDrawer &) = delete;

    // The reference opens the file behind Texture::path itself (src/renderer/types.rs:164); file
    // decoding is outside this path, so the caller supplies the decoded texels per path.
    void provide_image(const std::string &path, Image image) { images_[path] = std::move(image); }

    // src/renderer/drawer.rs:458-494: uniforms from the drawables' transforms, the texture table
    // from their distinct textures, one draw per mesh, grouped by texture in path order
    // (the BTreeMap at :481-487), caller order inside a group.
    Status update_drawables(std::vector<Drawable> drawables) override
    {
        std::map<Texture, int32_t> ids;
        for (const Drawable &d : drawables)
            if (d.texture) ids.emplace(*d.texture, 0);
        std::vector<sxe_texture_desc> table;
        for (auto &kv : ids) {
            auto it = images_.find(kv.first.path);
            if (it == images_.end()) return Status::Err("no image was provided for texture '" + kv.first.path + "'");
            const Image &im = it->second;
            if ((size_t)im.width * im.height * 4 != im.rgba.size())
                return Status::Err("image '" + kv.first.path + "' does not hold width * height RGBA8 texels");
            kv.second = (int32_t)table.size();
            table.push_back(sxe_texture_desc{im.rgba.data(), im.width, im.height});
        }
        int rc = sxe_set_textures(ctx_, table.data(), (uint32_t)table.size());
        if (rc != SXE_OK) return Status::Err(message(ctx_, rc));
        std::vector<sxe_drawable> flat(drawables.size());
        for (size_t i = 0; i < drawables.size(); ++i) {
            const Drawable &d = drawables[i];
            sxe_drawable &f = flat[i];
            f.vertices = d.mesh.vertices.data();
            f.n_vertices = (uint32_t)d.mesh.vertices.size();
            f.indices = d.mesh.indices.data();
            f.n_indices = (uint32_t)d.mesh.indices.size();
            f.transform = d.transform.raw;
            f.texture = d.texture ? ids[*d.texture] : -1;
            f.rendered = d.mesh.rendered ? 1u : 0u;
        }
        rc = sxe_update_drawables(ctx_, flat.data(), (uint32_t)flat.size());
        if (rc != SXE_OK) return Status::Err(message(ctx_, rc));
        last_drawables_ = std::move(drawables); // :490
        return Status::Ok();
    }

    // src/renderer/drawer.rs:496-505; array order is update_drawables' order (src/main.rs:237-263).
    Status update_uniforms(std::vector<ObjectUniformBufferObject> uniforms) override
    {
        static_assert(sizeof(ObjectUniformBufferObject) == 64, "one column-major mat4");
        const int rc = sxe_update_uniforms(ctx_, uniforms.empty() ? nullptr : uniforms[0].model, (uint32_t)uniforms.size());
        return rc == SXE_OK ? Status::Ok() : Status::Err(message(ctx_, rc));
    }

    // src/renderer/drawer.rs:507-518
    Status update_camera(Transform transform) override
    {
        const int rc = sxe_update_camera(ctx_, &transform.raw);
        return rc == SXE_OK ? Status::Ok() : Status::Err(message(ctx_, rc));
    }

    // src/renderer/drawer.rs:412-456; asynchronous, like a queue submit.
    void draw(size_t image_index) override
    {
        const int rc = sxe_draw(ctx_, (uint32_t)image_index);
        if (rc != SXE_OK) throw std::runtime_error("sxe_draw: " + message(ctx_, rc));
    }

    // Systems that keep the ECS state on the device (src/systems/rotation.rs:26-37,
    // src/main.rs:237-244): not part of the trait, offered next to it.
    Status update_transforms(const std::vector<Transform> &transforms)
    {
        std::vector<sxe_transform> raw(transforms.size());
        for (size_t i = 0; i < raw.size(); ++i) raw[i] = transforms[i].raw;
        const int rc = sxe_update_transforms(ctx_, raw.data(), (uint32_t)raw.size());
        return rc == SXE_OK ? Status::Ok() : Status::Err(message(ctx_, rc));
    }
    Status rotation_system(uint64_t seed, uint32_t frame, float delta_ms)
    {
        const int rc = sxe_rotation_system(ctx_, seed, frame, delta_ms);
        return rc == SXE_OK ? Status::Ok() : Status::Err(message(ctx_, rc));
    }

    // The light of SXE_SHADE_LIT: not part of the trait either, and without a counterpart in the
    // reference, whose shaders have no lighting.
    Status set_light(const sxe_light &light)
    {
        const int rc = sxe_set_light(ctx_, &light);
        return rc == SXE_OK ? Status::Ok() : Status::Err(message(ctx_, rc));
    }

    sxe_ctx *context() const { return ctx_; }
    const Viewport &viewport() const { return viewport_; }
    const std::vector<Drawable> &last_drawables() const { return last_drawables_; }

  private:
    HipDrawer(sxe_ctx *ctx, const Viewport &vp, uint32_t shading) : ctx_(ctx), viewport_(vp), shading_(shading) {}
    static std::string message(const sxe_ctx *ctx, int rc)
    {
        const char *m = sxe_last_error(ctx);
        return "sxe error " + std::to_string(rc) + ": " + (m ? m : "");
    }
    sxe_ctx *ctx_;
    Viewport viewport_;
    uint32_t shading_;
    std::map<std::string, Image> images_;
    std::vector<Drawable> last_drawables_;
};

// trait Presenter (src/renderer/presenter.rs:16-22) without a window: images are the drawer's own
// render targets, acquire_image hands them out in turn (the swapchain's behaviour in the steady
// state, presenter.rs:386-413), present waits for the frame and reads it back
// (presenter.rs:415-430 queues it to the display instead).
class ReadbackPresenter {
  public:
    ReadbackPresenter(HipDrawer &drawer, uint32_t n_images) : drawer_(drawer), n_images_(n_images ? n_images : 1) {}
    Viewport viewport() const { return drawer_.viewport(); }
    uint32_t acquire_image(Status *status = nullptr)
    {
        if (status) *status = Status::Ok();
        current_ = next_;
        next_ = (next_ + 1) % n_images_;
        return current_;
    }
    Status present()
    {
        const Viewport vp = drawer_.viewport();
        frame_.width = vp.w;
        frame_.height = vp.h;
        frame_.rgba.resize((size_t)vp.w * vp.h * 4);
        int rc = sxe_sync(drawer_.context());
        if (rc == SXE_OK) rc = sxe_read_color(drawer_.context(), current_, frame_.rgba.data(), frame_.rgba.size());
        if (rc != SXE_OK) {
            const char *m = sxe_last_error(drawer_.context());
            return Status::Err("present: " + std::string(m ? m : ""));
        }
        return Status::Ok();
    }
    const Image &frame() const { return frame_; } // the image presented last
    // present() without the copy into an Image: the pixels stay in the library's pinned buffer
    // (valid until this image is presented this way again).
    Status present_view(const uint8_t *&rgba)
    {
        const int rc = sxe_map_color(drawer_.context(), current_, &rgba);
        if (rc != SXE_OK) {
            const char *m = sxe_last_error(drawer_.context());
            return Status::Err("present: " + std::string(m ? m : ""));
        }
        return Status::Ok();
    }

  private:
    HipDrawer &drawer_;
    uint32_t n_images_, next_ = 0, current_ = 0;
    Image frame_;
};

// Binary PPM (P6) of an RGBA8 image, alpha dropped: the simplest container every viewer opens.
inline Status write_ppm(const std::string &path, const Image &im)
{
    if ((size_t)im.width * im.height * 4 != im.rgba.size()) return Status::Err("image does not hold width * height RGBA8 pixels");
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return Status::Err("cannot open '" + path + "' for writing");
    std::fprintf(f, "P6\n%u %u\n255\n", im.width, im.height);
    std::vector<uint8_t> row((size_t)im.width * 3);
    bool good = true;
    for (uint32_t y = 0; y < im.height && good; ++y) {
        const uint8_t *src = im.rgba.data() + (size_t)y * im.width * 4;
        for (uint32_t x = 0; x < im.width; ++x) {
            row[3 * x] = src[4 * x];
            row[3 * x + 1] = src[4 * x + 1];
            row[3 * x + 2] = src[4 * x + 2];
        }
        good = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    good = (std::fclose(f) == 0) && good;
    return good ? Status::Ok() : Status::Err("short write to '" + path + "'");
}

// PNG (8-bit RGBA, colour type 6) with stored, i.e. uncompressed, deflate blocks: every decoder
// reads it and it needs no compression library.  The pixels are sRGB-encoded, as the render target
// is (src/renderer/presenter.rs:472-478), which is what PNG viewers assume.
namespace detail {
inline uint32_t crc32(uint32_t crc, const uint8_t *p, size_t n)
{
    crc = ~crc;
    for (size_t i = 0; i < n; ++i) {
        crc ^= p[i];
        for (int k = 0; k < 8; ++k) crc = (crc >> 1) ^ (0xEDB88320u & (0u - (crc & 1u)));
    }
    return ~crc;
}
inline void put32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
inline void chunk(std::vector<uint8_t> &out, const char type[4], const std::vector<uint8_t> &data)
{
    put32(out, (uint32_t)data.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put32(out, crc32(0, out.data() + start, out.size() - start));
}
} // namespace detail

inline Status write_png(const std::string &path, const Image &im)
{
    if (im.width == 0 || im.height == 0 || (size_t)im.width * im.height * 4 != im.rgba.size())
        return Status::Err("image does not hold width * height RGBA8 pixels");
    std::vector<uint8_t> raw; // scanlines, each behind a filter byte 0 (none)
    raw.reserve(((size_t)im.width * 4 + 1) * im.height);
    for (uint32_t y = 0; y < im.height; ++y) {
        raw.push_back(0);
        const uint8_t *row = im.rgba.data() + (size_t)y * im.width * 4;
        raw.insert(raw.end(), row, row + (size_t)im.width * 4);
    }
    std::vector<uint8_t> z = {0x78, 0x01}; // zlib header, then stored blocks of at most 65535 bytes
    uint32_t a = 1, b = 0;                 // Adler-32 of the raw data
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)n); z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)~n); z.push_back((uint8_t)(~n >> 8));
        z.insert(z.end(), raw.begin() + (std::ptrdiff_t)pos, raw.begin() + (std::ptrdiff_t)(pos + n));
        for (size_t i = pos; i < pos + n; ++i) {
            a = (a + raw[i]) % 65521u;
            b = (b + a) % 65521u;
        }
        pos += n;
    }
    detail::put32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    detail::put32(ihdr, im.width);
    detail::put32(ihdr, im.height);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0}); // bit depth, RGBA, deflate, adaptive filtering, no interlace
    detail::chunk(out, "IHDR", ihdr);
    detail::chunk(out, "sRGB", {0}); // rendering intent: perceptual
    detail::chunk(out, "IDAT", z);
    detail::chunk(out, "IEND", {});
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) return Status::Err("cannot open '" + path + "' for writing");
    const bool good = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0 && good) ? Status::Ok() : Status::Err("short write to '" + path + "'");
}

} // namespace sxe_host

#endif
