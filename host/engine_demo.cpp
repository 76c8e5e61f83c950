// engine_demo.cpp -- the reference's frame loop (start_engine, src/main.rs:140-199) on the C++
// Drawer/Presenter mirror of include/sxe_drawer.hpp.
//
//   engine_demo <scene file> <output .rgba> [output .ppm or .png]
//
// The scene file is written by tests/util.py:write_scene_file (little endian):
//   "SXES" u32 version(1) | u32 width height n_images n_frames | f32 rotate_deg[3] | f32 camera[10]
//   u32 n_textures { u32 path_len, path bytes, u32 w h, w*h*4 bytes }   (w = h = 0: no bytes follow,
//                    the path names a baseline JPEG file relative to the scene file's directory)
//   u32 n_drawables { u32 n_vertices n_indices, vertices (32 B each), indices (u32),
//                     f32 transform[10], i32 texture (index above or -1), u32 rendered }
// Every frame all drawables are rotated by rotate_deg (the stand-in for the Rotation system,
// src/systems/rotation.rs:26-37, with fixed angles), then the loop body of main.rs:189-197 runs:
// update_uniforms, update_camera, acquire_image, draw, present.  The last presented frame is
// written out.
#include "sxe_drawer.hpp"
#include "sxe_png.hpp"

#include <cstring>
#include <fstream>
#include <iostream>

using namespace sxe_host;

namespace {

struct Reader {
    std::ifstream in;
    template <class T> T get()
    {
        T v;
        in.read(reinterpret_cast<char *>(&v), sizeof(T));
        if (!in) throw std::runtime_error("scene file is truncated");
        return v;
    }
    void bytes(void *dst, size_t n)
    {
        in.read(reinterpret_cast<char *>(dst), (std::streamsize)n);
        if (!in) throw std::runtime_error("scene file is truncated");
    }
};

Transform read_transform(Reader &r)
{
    Transform t;
    r.bytes(&t.raw, sizeof(t.raw));
    return t;
}

// fetch_uniforms (src/main.rs:237-244)
std::vector<ObjectUniformBufferObject> fetch_uniforms(const std::vector<Drawable> &world)
{
    std::vector<ObjectUniformBufferObject> u;
    u.reserve(world.size());
    for (const Drawable &d : world) u.push_back(d.transform.to_ubo());
    return u;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::cerr << "usage: engine_demo <scene file> <output .rgba> [output .ppm or .png]\n";
        return 2;
    }
    try {
        Reader r;
        r.in.open(argv[1], std::ios::binary);
        if (!r.in) throw std::runtime_error(std::string("cannot open ") + argv[1]);
        std::string scene_dir = argv[1];
        scene_dir = scene_dir.find_last_of('/') == std::string::npos ? std::string() : scene_dir.substr(0, scene_dir.find_last_of('/') + 1);
        char magic[4];
        r.bytes(magic, 4);
        if (std::memcmp(magic, "SXES", 4) != 0 || r.get<uint32_t>() != 1u) throw std::runtime_error("not a version 1 scene file");
        const uint32_t width = r.get<uint32_t>(), height = r.get<uint32_t>(), n_images = r.get<uint32_t>(), n_frames = r.get<uint32_t>();
        float rot[3];
        r.bytes(rot, sizeof(rot));
        const Transform camera = read_transform(r);

        Viewport viewport;
        viewport.w = width;
        viewport.h = height;
        std::unique_ptr<HipDrawer> drawer;
        HipDrawer::create(viewport, n_images, 0, drawer).unwrap();

        std::vector<Texture> textures(r.get<uint32_t>());
        for (Texture &t : textures) {
            t.path.resize(r.get<uint32_t>());
            r.bytes(&t.path[0], t.path.size());
            Image im;
            im.width = r.get<uint32_t>();
            im.height = r.get<uint32_t>();
            if (im.width == 0 && im.height == 0) {
                // no texels in the scene file: the path names an image file next to it (JPEG or PNG),
                // loaded the way TextureData::load does (image::open(path).to_rgba(), src/renderer/types.rs:158-188)
                const Status st = load_image(scene_dir + t.path, im);
                if (!st) throw std::runtime_error(st.error);
            } else {
                im.rgba.resize((size_t)im.width * im.height * 4);
                r.bytes(im.rgba.data(), im.rgba.size());
            }
            drawer->provide_image(t.path, std::move(im));
        }
        std::vector<Drawable> world(r.get<uint32_t>());
        for (size_t i = 0; i < world.size(); ++i) {
            Drawable &d = world[i];
            d.mesh.key = "mesh" + std::to_string(i);
            d.mesh.vertices.resize(r.get<uint32_t>());
            d.mesh.indices.resize(r.get<uint32_t>());
            r.bytes(d.mesh.vertices.data(), d.mesh.vertices.size() * sizeof(Vertex));
            r.bytes(d.mesh.indices.data(), d.mesh.indices.size() * sizeof(uint32_t));
            d.transform = read_transform(r);
            const int32_t tex = r.get<int32_t>();
            if (tex >= 0) {
                if ((size_t)tex >= textures.size()) throw std::runtime_error("texture index out of range in scene file");
                d.texture = textures[(size_t)tex];
            }
            d.mesh.rendered = r.get<uint32_t>() != 0;
        }

        ReadbackPresenter presenter(*drawer, n_images);
        // src/main.rs:165: the result is ignored there; a bad scene would only show later
        const Status scene_status = drawer->update_drawables(world);
        if (!scene_status) throw std::runtime_error("update_drawables: " + scene_status.error);
        for (uint32_t frame = 0; frame < n_frames; ++frame) {
            for (Drawable &d : world) d.transform.rotate(rot[0], rot[1], rot[2]);
            drawer->update_uniforms(fetch_uniforms(world)).unwrap(); // main.rs:191
            drawer->update_camera(camera).unwrap();                  // main.rs:192
            const uint32_t image = presenter.acquire_image();        // main.rs:193
            drawer->draw(image);                                     // main.rs:195
            presenter.present().unwrap();                            // main.rs:196
        }
        const Image &last = presenter.frame();
        std::ofstream out(argv[2], std::ios::binary);
        out.write(reinterpret_cast<const char *>(last.rgba.data()), (std::streamsize)last.rgba.size());
        if (!out) throw std::runtime_error(std::string("cannot write ") + argv[2]);
        if (argc > 3) {
            const std::string picture = argv[3];
            const bool png = picture.size() > 4 && picture.compare(picture.size() - 4, 4, ".png") == 0;
            (png ? write_png(picture, last) : write_ppm(picture, last)).unwrap();
        }
        std::cout << "frames " << n_frames << " size " << last.width << "x" << last.height << "\n";
    } catch (const std::exception &e) {
        std::cerr << "engine_demo: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
