// Shape "plymesh": a PLY reader with the semantics of pbrt-v3's CreatePLYMesh (shapes/plymesh.cpp:158-290, which drives
// the third-party rply parser): element "vertex" with float-convertible x y z, optional nx ny nz, optional texture
// coordinates named (u,v) | (s,t) | (texture_u,texture_v) | (texture_s,texture_t); element "face" with the list property
// "vertex_indices" holding triangles or quads -- a quad (a b c d) becomes the triangles (a b c) and (d a c)
// (plymesh.cpp:139-147) -- other face sizes are skipped with a warning.  ASCII and binary little/big endian.  Every
// value goes through double and is narrowed to float, as rply's ply_get_argument_value does for the reference.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include "api.h"
#include "error.h"
#include "scene.h"

namespace pbrt {
int MeshAlphaMask(const ParamSet &params);  // api.cpp
namespace {
enum PlyType { T_I8, T_U8, T_I16, T_U16, T_I32, T_U32, T_F32, T_F64, T_BAD };
int typeSize(PlyType t) { static const int sz[] = {1, 1, 2, 2, 4, 4, 4, 8, 0}; return sz[t]; }
PlyType parseType(const std::string &s) {
    static const struct { const char *name; PlyType t; } names[] = {
        {"char", T_I8}, {"int8", T_I8}, {"uchar", T_U8}, {"uint8", T_U8}, {"short", T_I16}, {"int16", T_I16},
        {"ushort", T_U16}, {"uint16", T_U16}, {"int", T_I32}, {"int32", T_I32}, {"uint", T_U32}, {"uint32", T_U32},
        {"float", T_F32}, {"float32", T_F32}, {"double", T_F64}, {"float64", T_F64}};
    for (auto &n : names) if (s == n.name) return n.t;
    return T_BAD;
}
struct Property { std::string name; bool isList = false; PlyType countType = T_BAD, type = T_BAD; };
struct Element { std::string name; long count = 0; std::vector<Property> props; };

struct Reader {
    std::ifstream in;
    enum { ASCII, LE, BE } format = ASCII;
    bool ok = true;
    double readValue(PlyType t) {
        if (format == ASCII) {
            std::string tok;
            if (!(in >> tok)) { ok = false; return 0; }
            return strtod(tok.c_str(), nullptr);
        }
        unsigned char b[8] = {0};
        const int n = typeSize(t);
        in.read((char *)b, n);
        if (!in) { ok = false; return 0; }
        const uint16_t probe = 1;
        const bool hostLE = *(const unsigned char *)&probe == 1;
        if ((format == LE) != hostLE) for (int i = 0; i < n / 2; ++i) std::swap(b[i], b[n - 1 - i]);
        switch (t) {
            case T_I8: { int8_t v; memcpy(&v, b, 1); return v; }
            case T_U8: { uint8_t v; memcpy(&v, b, 1); return v; }
            case T_I16: { int16_t v; memcpy(&v, b, 2); return v; }
            case T_U16: { uint16_t v; memcpy(&v, b, 2); return v; }
            case T_I32: { int32_t v; memcpy(&v, b, 4); return v; }
            case T_U32: { uint32_t v; memcpy(&v, b, 4); return v; }
            case T_F32: { float v; memcpy(&v, b, 4); return v; }
            case T_F64: { double v; memcpy(&v, b, 8); return v; }
            default: ok = false; return 0;
        }
    }
};
int findProp(const Element &e, const char *name) {
    for (size_t i = 0; i < e.props.size(); ++i) if (e.props[i].name == name && !e.props[i].isList) return (int)i;
    return -1;
}
}  // namespace

std::shared_ptr<TriangleMesh> BuildTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTriangles, const int *indices,
                                                int nVertices, const Float *P, const Float *S, const Float *N, const Float *UV);

std::shared_ptr<TriangleMesh> CreatePLYMesh(const Transform &o2w, bool reverseOrientation, const ParamSet &params) {
    const std::string filename = AbsolutePath(ResolveFilename(params.FindOneString("filename", "")));
    Reader rd;
    rd.in.open(filename, std::ios::binary);
    if (!rd.in) { Error("Couldn't open PLY file \"%s\"", filename.c_str()); return nullptr; }
    // ---- header
    std::string line;
    std::vector<Element> elements;
    bool sawMagic = false, sawFormat = false, sawEnd = false;
    while (std::getline(rd.in, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ls(line);
        std::string kw;
        if (!(ls >> kw)) continue;
        if (!sawMagic) { if (kw != "ply") break; sawMagic = true; continue; }
        if (kw == "format") {
            std::string f;
            ls >> f;
            if (f == "ascii") rd.format = Reader::ASCII;
            else if (f == "binary_little_endian") rd.format = Reader::LE;
            else if (f == "binary_big_endian") rd.format = Reader::BE;
            else break;
            sawFormat = true;
        } else if (kw == "comment" || kw == "obj_info") continue;
        else if (kw == "element") {
            Element e;
            ls >> e.name >> e.count;
            elements.push_back(e);
        } else if (kw == "property") {
            if (elements.empty()) break;
            Property p;
            std::string t;
            ls >> t;
            if (t == "list") {
                std::string ct, it;
                ls >> ct >> it >> p.name;
                p.isList = true; p.countType = parseType(ct); p.type = parseType(it);
                if (p.countType == T_BAD) { sawFormat = false; break; }
            } else { p.type = parseType(t); ls >> p.name; }
            if (p.type == T_BAD) { sawFormat = false; break; }
            elements.back().props.push_back(p);
        } else if (kw == "end_header") { sawEnd = true; break; }
    }
    if (!sawMagic || !sawFormat || !sawEnd) { Error("Unable to read the header of PLY file \"%s\"", filename.c_str()); return nullptr; }
    long vertexCount = 0, faceCount = 0;
    for (const Element &e : elements) {
        if (e.name == "vertex") vertexCount = e.count;
        else if (e.name == "face") faceCount = e.count;
    }
    if (vertexCount == 0 || faceCount == 0) { Error("%s: PLY file is invalid! No face/vertex elements found!", filename.c_str()); return nullptr; }
    // ---- which vertex properties feed which buffer (plymesh.cpp:205-247)
    std::vector<Float> P, N, UV;
    std::vector<int> indices;
    bool error = false;
    for (const Element &e : elements) {
        if (e.name == "vertex") {
            const int ix = findProp(e, "x"), iy = findProp(e, "y"), iz = findProp(e, "z");
            if (ix < 0 || iy < 0 || iz < 0) { Error("%s: Vertex coordinate property not found!", filename.c_str()); return nullptr; }
            const int inx = findProp(e, "nx"), iny = findProp(e, "ny"), inz = findProp(e, "nz");
            const bool hasN = inx >= 0 && iny >= 0 && inz >= 0;
            int iu = -1, iv = -1;
            static const char *uvNames[4][2] = {{"u", "v"}, {"s", "t"}, {"texture_u", "texture_v"}, {"texture_s", "texture_t"}};
            for (auto &nm : uvNames) {
                int a = findProp(e, nm[0]), b = findProp(e, nm[1]);
                if (a >= 0 && b >= 0) { iu = a; iv = b; break; }
            }
            P.assign(3 * (size_t)e.count, 0.f);
            if (hasN) N.assign(3 * (size_t)e.count, 0.f);
            if (iu >= 0) UV.assign(2 * (size_t)e.count, 0.f);
            for (long i = 0; i < e.count && rd.ok; ++i)
                for (int k = 0; k < (int)e.props.size(); ++k) {
                    const Property &p = e.props[k];
                    if (p.isList) { long n = (long)rd.readValue(p.countType); for (long j = 0; j < n; ++j) rd.readValue(p.type); continue; }
                    const float v = (float)rd.readValue(p.type);
                    if (k == ix) P[3 * i] = v; else if (k == iy) P[3 * i + 1] = v; else if (k == iz) P[3 * i + 2] = v;
                    if (hasN) { if (k == inx) N[3 * i] = v; else if (k == iny) N[3 * i + 1] = v; else if (k == inz) N[3 * i + 2] = v; }
                    if (iu >= 0) { if (k == iu) UV[2 * i] = v; else if (k == iv) UV[2 * i + 1] = v; }
                }
        } else if (e.name == "face") {
            indices.reserve(6 * (size_t)e.count);
            for (long i = 0; i < e.count && rd.ok; ++i)
                for (const Property &p : e.props) {
                    if (!p.isList) { rd.readValue(p.type); continue; }
                    const long n = (long)rd.readValue(p.countType);
                    if (p.name != "vertex_indices") { for (long j = 0; j < n; ++j) rd.readValue(p.type); continue; }
                    int face[4] = {0, 0, 0, 0};
                    for (long j = 0; j < n; ++j) {
                        const int value = (int)rd.readValue(p.type);
                        if (n != 3 && n != 4) continue;
                        if (value < 0 || value >= vertexCount) {
                            Error("plymesh: Vertex reference %i is out of bounds! Valid range is [0..%i)", value, (int)vertexCount);
                            error = true;
                        }
                        face[j] = value;
                    }
                    if (n != 3 && n != 4) {
                        Warning("plymesh: Ignoring face with %i vertices (only triangles and quads are supported!)", (int)n);
                        continue;
                    }
                    indices.insert(indices.end(), {face[0], face[1], face[2]});
                    if (n == 4) indices.insert(indices.end(), {face[3], face[0], face[2]});
                }
        } else {  // skip unknown elements
            for (long i = 0; i < e.count && rd.ok; ++i)
                for (const Property &p : e.props) {
                    if (p.isList) { long n = (long)rd.readValue(p.countType); for (long j = 0; j < n; ++j) rd.readValue(p.type); }
                    else rd.readValue(p.type);
                }
        }
    }
    if (!rd.ok) { Error("%s: unable to read the contents of PLY file", filename.c_str()); return nullptr; }
    if (error) return nullptr;
    const int alphaMask = MeshAlphaMask(params);  // plymesh.cpp:259-287
    auto mesh = BuildTriangleMesh(o2w, reverseOrientation, (int)indices.size() / 3, indices.data(), (int)vertexCount, P.data(), nullptr,
                                  N.empty() ? nullptr : N.data(), UV.empty() ? nullptr : UV.data());
    mesh->alphaMask = alphaMask;
    return mesh;
}
}  // namespace pbrt
