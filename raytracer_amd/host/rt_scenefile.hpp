// rt_scenefile.hpp -- a plain-text scene description for the host program: the
// reference hard-codes its scene in InitScene (raytracer_cpp/main.cpp:21-63); this loader fills the
// same globals (sceneObjects, pCamera) through the same constructors (Sphere, TiledPlane, Camera),
// so any scene can be fed through the reference's API.  One statement per line, '#' starts a
// comment, objects keep their file order (list order is semantic: tie-breaks, light order):
//
//   camera      px py pz   dx dy dz   fov_degrees
//   sphere      cx cy cz r   albedo(r g b)   specular(r g b)   reflectance   [emissive(r g b)]
//   tiledplane  ox oy oz   nx ny nz   [white: albedo(3) specular(3) reflectance   black: the same 7]
//
// The plane normal is normalised like InitScene does (main.cpp:61).  raytracer_amd/capi.py has the
// same parser (load_scene_file), which the tests use to check the rendered file.
#pragma once
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_camera.hpp"
#include "rt_scene.hpp"

extern std::vector<std::shared_ptr<SceneObject>> sceneObjects;
extern std::shared_ptr<Camera> pCamera;

inline void LoadSceneFile(const std::string &path, int width, int height)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open scene file " + path);
    sceneObjects.clear();
    pCamera.reset();
    std::string line;
    int lineno = 0;
    auto bad = [&](const std::string &what) {
        throw std::runtime_error(path + ":" + std::to_string(lineno) + ": " + what);
    };
    while (std::getline(in, line)) {
        lineno++;
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        std::istringstream ss(line);
        std::string kind;
        if (!(ss >> kind)) continue;
        std::vector<float> v;
        for (float x; ss >> x;) v.push_back(x);
        if (!ss.eof()) bad("not a number");
        if (kind == "camera") {
            if (v.size() != 7) bad("camera takes 7 numbers");
            pCamera = std::make_shared<Camera>(vec3(v[0], v[1], v[2]), vec3(v[3], v[4], v[5]), v[6], width, (unsigned)height);
        } else if (kind == "sphere") {
            if (v.size() != 11 && v.size() != 14) bad("sphere takes 11 or 14 numbers");
            Material m;
            m.albedo = vec3(v[4], v[5], v[6]);
            m.specular = vec3(v[7], v[8], v[9]);
            m.reflectance = v[10];
            m.emissive = v.size() == 14 ? vec3(v[11], v[12], v[13]) : vec3(0.0f, 0.0f, 0.0f);
            sceneObjects.push_back(std::make_shared<Sphere>(m, vec3(v[0], v[1], v[2]), v[3]));
        } else if (kind == "tiledplane") {
            if (v.size() != 6 && v.size() != 20) bad("tiledplane takes 6 or 20 numbers");
            auto p = std::make_shared<TiledPlane>(vec3(v[0], v[1], v[2]), normalize(vec3(v[3], v[4], v[5])));
            if (v.size() == 20) {
                p->whiteMat.albedo = vec3(v[6], v[7], v[8]);
                p->whiteMat.specular = vec3(v[9], v[10], v[11]);
                p->whiteMat.reflectance = v[12];
                p->blackMat.albedo = vec3(v[13], v[14], v[15]);
                p->blackMat.specular = vec3(v[16], v[17], v[18]);
                p->blackMat.reflectance = v[19];
            }
            sceneObjects.push_back(p);
        } else {
            bad("unknown statement '" + kind + "'");
        }
    }
    if (!pCamera) throw std::runtime_error(path + ": no camera statement");
    if (sceneObjects.empty()) throw std::runtime_error(path + ": no objects");
}
