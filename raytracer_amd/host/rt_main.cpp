// rt_main.cpp -- the host program: counterpart of raytracer_cpp/main.cpp:21-63,230-258 on top of the
// MI355X library.  Same options (-p/--partitions, -a/--antialiased), same scene, same "Time:" line,
// same image.bmp.  Extra options select the size, depth, device, output path, a scene file
// (rt_scenefile.hpp) instead of InitScene's scene, and the library's mode / sweep.
#include <algorithm>
#include <chrono>
#include <iostream>
#include <stdexcept>
#include <string>

#include "rt_cmdline.hpp"
#include "rt_draw.hpp"
#include "rt_scenefile.hpp"

std::vector<std::shared_ptr<SceneObject>> sceneObjects;
std::shared_ptr<Camera> pCamera;

static void InitScene(int width, int height)
{
    pCamera = std::make_shared<Camera>(vec3(0.0f, 6.0f, 8.0f), vec3(0.0f, -.8f, -1.0f), 60.0f, width, (unsigned)height);

    struct Ball { vec3 albedo, specular; float reflectance; vec3 emissive, center; float radius; };
    const Ball balls[] = {
        {{.7f, .1f, .1f}, {.9f, .1f, .1f}, .5f, {0, 0, 0}, {0.0f, 2.0f, 0.f}, 2.0f},          // red
        {{.7f, 0.f, .7f}, {.9f, .9f, .8f}, .5f, {0, 0, 0}, {-2.5f, 1.0f, 2.f}, 1.0f},         // purple
        {{0.f, .3f, 1.f}, {0.f, 0.f, 1.f}, 0.f, {0, 0, 0}, {-0.0f, 0.5f, 3.f}, 0.5f},         // blue
        {{1.f, 1.f, 1.f}, {0.f, 0.f, 0.f}, 0.f, {1, 1, 0}, {2.8f, 0.8f, 2.0f}, 0.8f},         // white ball (emits yellow)
        {{0.f, .8f, 0.f}, {0.f, 0.f, 0.f}, 0.f, {1, 1, 1}, {-10.8f, 8.4f, 10.0f}, 0.4f},      // white light
    };
    for (const Ball &b : balls) {
        Material m;
        m.albedo = b.albedo;
        m.specular = b.specular;
        m.reflectance = b.reflectance;
        m.emissive = b.emissive;
        sceneObjects.push_back(std::make_shared<Sphere>(m, b.center, b.radius));
    }
    sceneObjects.push_back(std::make_shared<TiledPlane>(vec3(0.0f, 0.0f, 0.0f), normalize(vec3(0.0f, 1.0f, 0.0f))));
}

int main(int argc, char **argv)
{
    cli::Parser parser(argc, argv);
    parser.set_optional<int>("p", "partitions", 2, "thread partitions of the CPU reference (accepted, unused on the GPU)");
    parser.set_optional<int>("a", "antialiased", 1, "Antialias each pixel");
    parser.set_optional<int>("W", "width", 1024, "image width");
    parser.set_optional<int>("H", "height", 768, "image height");
    parser.set_optional<int>("d", "depth", 5, "MAX_DEPTH");
    parser.set_optional<std::string>("g", "gpu", "0", "HIP device, or a comma-separated list: the frame is shared between them");
    parser.set_optional<std::string>("o", "output", "image.bmp", "BMP file to write");
    parser.set_optional<std::string>("s", "scene", "", "scene description file (default: the reference's InitScene scene)");
    parser.set_optional<int>("f", "frames", 1, "progressive mode: this many jittered single-sample frames, running mean");
    parser.set_optional<float>("l", "lens", 0.0f, "thin-lens radius (depth of field), e.g. 0.14; needs --focus");
    parser.set_optional<float>("F", "focus", 0.0f, "distance from the camera to the focal point");
    parser.set_optional<int>("m", "mode", 0, "0 = bit-identical to the reference, 1 = fast (radiance to rounding)");
    parser.set_optional<int>("w", "sweep", 0, "0 = auto, 1 = flat (every ray tests every sphere), 2 = grouped, 3 = sorted (grouped, one launch per bounce level)");
    if (!parser.run()) return 2;

    const int partitions = parser.get<int>("p");
    const int antialias = parser.get<int>("a");
    const int width = parser.get<int>("W"), height = parser.get<int>("H");
    if (width <= 0 || height <= 0 || partitions < 1) {
        std::cerr << "width, height and partitions must be positive" << std::endl;
        return 2;
    }

    Bitmap *pBitmap = CreateBitmap(width, height);
    ClearBitmap(pBitmap, Color{127, 127, 127});
    const std::string sceneFile = parser.get<std::string>("s");
    try {
        if (sceneFile.empty()) InitScene(width, height);
        else LoadSceneFile(sceneFile, width, height);
    } catch (const std::exception &e) {
        std::cerr << e.what() << std::endl;
        DestroyBitmap(pBitmap);
        return 2;
    }

    DrawSceneOptions opt;
    opt.maxDepth = parser.get<int>("d");
    {
        const std::string list = parser.get<std::string>("g");
        size_t at = 0;
        try {
            while (at <= list.size()) {
                const size_t comma = std::min(list.find(',', at), list.size());
                size_t used = 0;
                const std::string item = list.substr(at, comma - at);
                opt.devices.push_back(std::stoi(item, &used));
                if (used != item.size() || opt.devices.back() < 0) throw std::invalid_argument(item);
                at = comma + 1;
            }
        } catch (const std::exception &) {
            std::cerr << "-g wants a device index or a comma-separated list of them, got '" << list << "'" << std::endl;
            DestroyBitmap(pBitmap);
            return 2;
        }
    }
    opt.frames = parser.get<int>("f");
    opt.lensRadius = parser.get<float>("l");
    opt.focalLength = parser.get<float>("F");
    opt.mode = parser.get<int>("m");
    opt.sweep = parser.get<int>("w");
    double kernelMs = 0.0;
    opt.kernelMs = &kernelMs;

    const auto start = std::chrono::high_resolution_clock::now();
    try {
        DrawScene(pBitmap, partitions, antialias == 1, opt);
    } catch (const DrawSceneError &e) {
        std::cerr << "DrawScene failed (rt_status " << e.status << "): " << e.what() << std::endl;
        DestroyBitmap(pBitmap);
        return 1;
    }
    const auto end = std::chrono::high_resolution_clock::now();

    std::cout << "Time: " << std::chrono::duration<double, std::milli>(end - start).count() << " ms" << std::endl;
    std::cout << "Kernel: " << kernelMs << " ms" << std::endl;
    WriteBitmap(pBitmap, parser.get<std::string>("o").c_str());
    DestroyBitmap(pBitmap);
    return 0;
}
