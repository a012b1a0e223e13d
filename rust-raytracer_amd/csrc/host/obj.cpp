// Wavefront OBJ reader with the semantics Mesh::load_obj asks of tobj 3.1.0
// (mesh.rs:150-158: LoadOptions{single_index: true, triangulate: true}, models[0]):
//   * coordinates are parsed as f32 and widened to f64 (mesh.rs:160-172 `i[0] as f64`);
//   * single_index: one vertex per distinct (v, vt, vn) triple, numbered by first appearance;
//   * triangulate: polygons become a fan (0, k, k+1);
//   * only the first object/group with faces is returned (models[0]).
// tobj itself is not vendored in /root/reference; this follows its documented behaviour.
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

#include "scene.h"

namespace rtamd {

static double f32_widen(const std::string& tok) { return (double)strtof(tok.c_str(), nullptr); }

ObjMesh load_obj_file(const char* path) {
    std::ifstream f(path);
    if (!f) throw RtError(RT_ERR_IO, std::string("Failed to load OBJ file. (") + path + ")");
    std::vector<double> vs, vns;
    std::map<std::tuple<long, long, long>, uint32_t> uniq;
    ObjMesh m;
    bool all_have_n = true, any_face = false, model_closed = false;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream in(line);
        std::string tag;
        if (!(in >> tag) || tag[0] == '#') continue;
        if (tag == "v" || tag == "vn") {
            std::string a, b, c;
            if (!(in >> a >> b >> c)) throw RtError(RT_ERR_IO, "OBJ: malformed vertex line");
            auto& dst = (tag == "v") ? vs : vns;
            dst.push_back(f32_widen(a));
            dst.push_back(f32_widen(b));
            dst.push_back(f32_widen(c));
        } else if (tag == "o" || tag == "g") {
            if (any_face) model_closed = true;  // models[0] only
        } else if (tag == "f") {
            if (model_closed) continue;
            any_face = true;
            std::vector<uint32_t> face;
            std::string ft;
            while (in >> ft) {
                long vi = 0, ti = 0, ni = 0;
                size_t s1 = ft.find('/');
                vi = atol(ft.substr(0, s1).c_str());
                if (s1 != std::string::npos) {
                    size_t s2 = ft.find('/', s1 + 1);
                    std::string t = ft.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                    if (!t.empty()) ti = atol(t.c_str());
                    if (s2 != std::string::npos && s2 + 1 < ft.size()) ni = atol(ft.substr(s2 + 1).c_str());
                }
                long nv = (long)(vs.size() / 3), nn = (long)(vns.size() / 3);
                if (vi < 0) vi = nv + vi + 1;
                if (ni < 0) ni = nn + ni + 1;
                if (vi < 1 || vi > nv || ni > nn) throw RtError(RT_ERR_IO, "OBJ: face index out of range");
                auto key = std::make_tuple(vi, ti, ni);
                auto it = uniq.find(key);
                if (it == uniq.end()) {
                    it = uniq.emplace(key, (uint32_t)(m.pos.size() / 3)).first;
                    for (int k = 0; k < 3; k++) m.pos.push_back(vs[3 * (vi - 1) + k]);
                    if (ni >= 1) {
                        for (int k = 0; k < 3; k++) m.nrm.push_back(vns[3 * (ni - 1) + k]);
                    } else {
                        all_have_n = false;
                        for (int k = 0; k < 3; k++) m.nrm.push_back(0.0);
                    }
                }
                face.push_back(it->second);
            }
            if (face.size() < 3) throw RtError(RT_ERR_IO, "OBJ: face with fewer than 3 vertices");
            for (size_t k = 1; k + 1 < face.size(); k++) {
                m.idx.push_back(face[0]);
                m.idx.push_back(face[k]);
                m.idx.push_back(face[k + 1]);
            }
        }
    }
    if (m.idx.empty()) throw RtError(RT_ERR_IO, std::string("OBJ: no faces in ") + path);
    m.has_normals = all_have_n;
    if (!all_have_n) m.nrm.clear();
    return m;
}

}  // namespace rtamd
