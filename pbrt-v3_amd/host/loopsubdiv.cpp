// Shape "loopsubdiv": Loop subdivision surfaces, restating pbrt-v3's LoopSubdivide (shapes/loopsubdiv.cpp:150-419) with
// index-linked vertex / face records instead of arena pointers.  The reference orders edge end points by pointer value;
// that order only decides which of two commutative float additions comes first (loopsubdiv.cpp:266-278), so results do
// not depend on it.  Everything that does fix the output -- face and vertex visiting order, one-ring order, the float
// expressions of the even / odd / boundary / limit rules and of the limit-surface tangents -- follows the reference.
// The refined mesh is emitted through CreateTriangleMesh with limit positions and normals Cross(S, T).
#include <cmath>
#include <map>
#include <utility>
#include "api.h"
#include "error.h"
#include "scene.h"

namespace pbrt {
namespace {
inline int NEXT(int i) { return (i + 1) % 3; }
inline int PREV(int i) { return (i + 2) % 3; }
struct SDVertex { Point3f p; int startFace = -1, child = -1; bool regular = false, boundary = false; };
struct SDFace { int v[3] = {-1, -1, -1}, f[3] = {-1, -1, -1}, children[4] = {-1, -1, -1, -1}; };
typedef std::pair<int, int> SDEdge;
inline SDEdge makeEdge(int a, int b) { return a < b ? SDEdge(a, b) : SDEdge(b, a); }

struct Level {  // one level of the mesh tree: vertices and faces refer to each other by index within the level
    std::vector<SDVertex> V;
    std::vector<SDFace> F;
    int vnum(int f, int vert) const {
        // a walk around a vertex of a non-manifold mesh arrives at a face that does not hold the vertex, or at no face at all:
        // the reference aborts here (LOG(FATAL), loopsubdiv.cpp:92-97); the library reports the mesh
        if (f >= 0 && f < (int)F.size()) for (int i = 0; i < 3; ++i) if (F[f].v[i] == vert) return i;
        Error("Basic logic error in SDFace::vnum(): the loopsubdiv control mesh is not a manifold");
        Fatal();
    }
    // every walk around a vertex visits a face at most once on a manifold; on anything else it can circle without ever
    // meeting its start again (the reference then hangs): `steps` is reset by the walks' callers and checked here
    mutable size_t steps = 0;
    void step() const { if (++steps > 2 * F.size() + 8) { Error("loopsubdiv: a walk around a vertex does not close: the control mesh is not a manifold"); Fatal(); } }
    int nextFace(int f, int vert) const { step(); return F[f].f[vnum(f, vert)]; }
    int prevFace(int f, int vert) const { step(); return F[f].f[PREV(vnum(f, vert))]; }
    int nextVert(int f, int vert) const { return F[f].v[NEXT(vnum(f, vert))]; }
    int prevVert(int f, int vert) const { return F[f].v[PREV(vnum(f, vert))]; }
    int otherVert(int f, int v0, int v1) const {
        for (int i = 0; i < 3; ++i) if (F[f].v[i] != v0 && F[f].v[i] != v1) return F[f].v[i];
        Error("Basic logic error in SDVertex::otherVert()");
        return F[f].v[0];
    }
    int valence(int vi) const {  // loopsubdiv.cpp:121-136
        steps = 0;
        const SDVertex &v = V[vi];
        int f = v.startFace;
        if (!v.boundary) {
            int nf = 1;
            while ((f = nextFace(f, vi)) != v.startFace) ++nf;
            return nf;
        }
        int nf = 1;
        while ((f = nextFace(f, vi)) != -1) ++nf;
        f = v.startFace;
        while ((f = prevFace(f, vi)) != -1) ++nf;
        return nf + 1;
    }
    void oneRing(int vi, Point3f *p) const {  // loopsubdiv.cpp:435-453
        steps = 0;
        const SDVertex &v = V[vi];
        if (!v.boundary) {
            int face = v.startFace;
            do {
                *p++ = V[nextVert(face, vi)].p;
                face = nextFace(face, vi);
            } while (face != v.startFace);
        } else {
            int face = v.startFace, f2;
            while ((f2 = nextFace(face, vi)) != -1) face = f2;
            *p++ = V[nextVert(face, vi)].p;
            do {
                *p++ = V[prevVert(face, vi)].p;
                face = prevFace(face, vi);
            } while (face != -1);
        }
    }
    Point3f weightOneRing(int vi, Float beta) const {  // loopsubdiv.cpp:424-433
        int val = valence(vi);
        std::vector<Point3f> pRing(val);
        oneRing(vi, pRing.data());
        Point3f p = V[vi].p * (1 - val * beta);
        for (int i = 0; i < val; ++i) p = p + pRing[i] * beta;
        return p;
    }
    Point3f weightBoundary(int vi, Float beta) const {  // loopsubdiv.cpp:455-465
        int val = valence(vi);
        std::vector<Point3f> pRing(val);
        oneRing(vi, pRing.data());
        Point3f p = V[vi].p * (1 - 2 * beta);
        p = p + pRing[0] * beta;
        p = p + pRing[val - 1] * beta;
        return p;
    }
};
inline Float beta(int valence) { return valence == 3 ? 3.f / 16.f : 3.f / (8.f * valence); }          // :138-143
inline Float loopGamma(int valence) { return 1.f / (valence + 3.f / (8.f * beta(valence))); }           // :145-147
}  // namespace

std::shared_ptr<TriangleMesh> BuildTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTriangles, const int *indices,
                                                int nVertices, const Float *P, const Float *S, const Float *N, const Float *UV);

std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &o2w, bool reverseOrientation, const ParamSet &params) {
    int nLevels = params.FindOneInt("levels", params.FindOneInt("nlevels", 3));
    const std::vector<int> *vi = params.FindInt("indices");
    const std::vector<Float> *P = params.FindPoint3f("P");
    if (!vi) { Error("Vertex indices \"indices\" not provided for LoopSubdiv shape."); return nullptr; }
    if (!P) { Error("Vertex positions \"P\" not provided for LoopSubdiv shape."); return nullptr; }
    params.FindOneString("scheme", "loop");
    const int nVertices = (int)P->size() / 3, nFaces = (int)vi->size() / 3;
    for (int idx : *vi) if (idx < 0 || idx >= nVertices) { Error("loopsubdiv has out of-bounds vertex index %d (%d \"P\" values were given", idx, nVertices); return nullptr; }

    Level cur;
    cur.V.resize(nVertices);
    cur.F.resize(nFaces);
    for (int i = 0; i < nVertices; ++i) cur.V[i].p = Point3f((*P)[3 * i], (*P)[3 * i + 1], (*P)[3 * i + 2]);
    // face -> vertex links; a vertex remembers the LAST face that names it (loopsubdiv.cpp:166-174)
    for (int i = 0; i < nFaces; ++i)
        for (int j = 0; j < 3; ++j) { int v = (*vi)[3 * i + j]; cur.F[i].v[j] = v; cur.V[v].startFace = i; }
    // neighbour links through an edge table that forgets an edge once two faces share it (loopsubdiv.cpp:176-197)
    {
        std::map<SDEdge, std::pair<int, int>> edges;  // edge -> (first face, its edge number)
        for (int i = 0; i < nFaces; ++i)
            for (int edgeNum = 0; edgeNum < 3; ++edgeNum) {
                SDEdge e = makeEdge(cur.F[i].v[edgeNum], cur.F[i].v[NEXT(edgeNum)]);
                auto it = edges.find(e);
                if (it == edges.end()) edges[e] = std::make_pair(i, edgeNum);
                else {
                    cur.F[it->second.first].f[it->second.second] = i;
                    cur.F[i].f[edgeNum] = it->second.first;
                    edges.erase(it);
                }
            }
    }
    // boundary / regular flags (loopsubdiv.cpp:199-213)
    for (int i = 0; i < nVertices; ++i) {
        SDVertex &v = cur.V[i];
        if (v.startFace < 0) { Error("loopsubdiv: vertex %d is not used by any face", i); return nullptr; }
        int f = v.startFace;
        cur.steps = 0;
        do { f = cur.nextFace(f, i); } while (f != -1 && f != v.startFace);
        v.boundary = (f == -1);
        if (!v.boundary && cur.valence(i) == 6) v.regular = true;
        else if (v.boundary && cur.valence(i) == 4) v.regular = true;
        else v.regular = false;
    }

    for (int level = 0; level < nLevels; ++level) {  // loopsubdiv.cpp:219-321
        Level nxt;
        const int nv = (int)cur.V.size(), nf = (int)cur.F.size();
        nxt.V.reserve(nv + 3 * nf / 2 + 8);
        // children: even vertices first, in order; four faces per face
        for (int i = 0; i < nv; ++i) {
            cur.V[i].child = (int)nxt.V.size();
            SDVertex c;
            c.regular = cur.V[i].regular; c.boundary = cur.V[i].boundary;
            nxt.V.push_back(c);
        }
        nxt.F.resize(4 * (size_t)nf);
        for (int i = 0; i < nf; ++i) for (int k = 0; k < 4; ++k) cur.F[i].children[k] = 4 * i + k;
        // even vertices (loopsubdiv.cpp:240-253)
        for (int i = 0; i < nv; ++i) {
            const SDVertex &v = cur.V[i];
            if (!v.boundary) nxt.V[v.child].p = v.regular ? cur.weightOneRing(i, 1.f / 16.f) : cur.weightOneRing(i, beta(cur.valence(i)));
            else nxt.V[v.child].p = cur.weightBoundary(i, 1.f / 8.f);
        }
        // odd (edge) vertices, created on first encounter in face order (loopsubdiv.cpp:255-285)
        std::map<SDEdge, int> edgeVerts;
        for (int fi = 0; fi < nf; ++fi)
            for (int k = 0; k < 3; ++k) {
                const int a = cur.F[fi].v[k], b = cur.F[fi].v[NEXT(k)];
                SDEdge edge = makeEdge(a, b);
                if (edgeVerts.count(edge)) continue;
                SDVertex vert;
                vert.regular = true;
                vert.boundary = (cur.F[fi].f[k] == -1);
                vert.startFace = cur.F[fi].children[3];
                if (vert.boundary) {
                    vert.p = cur.V[edge.first].p * 0.5f;
                    vert.p = vert.p + cur.V[edge.second].p * 0.5f;
                } else {
                    vert.p = cur.V[edge.first].p * (3.f / 8.f);
                    vert.p = vert.p + cur.V[edge.second].p * (3.f / 8.f);
                    vert.p = vert.p + cur.V[cur.otherVert(fi, a, b)].p * (1.f / 8.f);
                    vert.p = vert.p + cur.V[cur.otherVert(cur.F[fi].f[k], a, b)].p * (1.f / 8.f);
                }
                edgeVerts[edge] = (int)nxt.V.size();
                nxt.V.push_back(vert);
            }
        // even vertices' start faces (loopsubdiv.cpp:289-293)
        for (int i = 0; i < nv; ++i) {
            int vertNum = cur.vnum(cur.V[i].startFace, i);
            nxt.V[cur.V[i].child].startFace = cur.F[cur.V[i].startFace].children[vertNum];
        }
        // face neighbour links (loopsubdiv.cpp:295-309)
        for (int fi = 0; fi < nf; ++fi) {
            const SDFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nxt.F[face.children[3]].f[j] = face.children[NEXT(j)];
                nxt.F[face.children[j]].f[NEXT(j)] = face.children[3];
                int f2 = face.f[j];
                nxt.F[face.children[j]].f[j] = f2 != -1 ? cur.F[f2].children[cur.vnum(f2, face.v[j])] : -1;
                f2 = face.f[PREV(j)];
                nxt.F[face.children[j]].f[PREV(j)] = f2 != -1 ? cur.F[f2].children[cur.vnum(f2, face.v[j])] : -1;
            }
        }
        // face vertex links (loopsubdiv.cpp:311-325)
        for (int fi = 0; fi < nf; ++fi) {
            const SDFace &face = cur.F[fi];
            for (int j = 0; j < 3; ++j) {
                nxt.F[face.children[j]].v[j] = cur.V[face.v[j]].child;
                int vert = edgeVerts[makeEdge(face.v[j], face.v[NEXT(j)])];
                nxt.F[face.children[j]].v[NEXT(j)] = vert;
                nxt.F[face.children[NEXT(j)]].v[j] = vert;
                nxt.F[face.children[3]].v[j] = vert;
            }
        }
        cur = std::move(nxt);
    }

    // limit surface positions (loopsubdiv.cpp:328-337)
    const int nv = (int)cur.V.size();
    std::vector<Point3f> pLimit(nv);
    for (int i = 0; i < nv; ++i)
        pLimit[i] = cur.V[i].boundary ? cur.weightBoundary(i, 1.f / 5.f) : cur.weightOneRing(i, loopGamma(cur.valence(i)));
    for (int i = 0; i < nv; ++i) cur.V[i].p = pLimit[i];
    // limit surface tangents -> normals (loopsubdiv.cpp:339-378)
    std::vector<Float> Ns(3 * (size_t)nv), Pout(3 * (size_t)nv);
    std::vector<Point3f> pRing(16);
    for (int i = 0; i < nv; ++i) {
        Vector3f S(0, 0, 0), T(0, 0, 0);
        int valence = cur.valence(i);
        if (valence > (int)pRing.size()) pRing.resize(valence);
        cur.oneRing(i, pRing.data());
        auto vec = [](const Point3f &p) { return Vector3f(p.x, p.y, p.z); };
        if (!cur.V[i].boundary) {
            for (int j = 0; j < valence; ++j) {
                S = S + vec(pRing[j]) * std::cos(2 * Pi * j / valence);
                T = T + vec(pRing[j]) * std::sin(2 * Pi * j / valence);
            }
        } else {
            const Point3f &vp = cur.V[i].p;
            S = pRing[valence - 1] - pRing[0];
            if (valence == 2) T = vec(pRing[0] + pRing[1] - vp * 2);
            else if (valence == 3) T = pRing[1] - vp;
            else if (valence == 4) T = vec(pRing[0] * -1 + pRing[1] * 2 + pRing[2] * 2 + pRing[3] * -1 + vp * -2);
            else {
                Float theta = Pi / float(valence - 1);
                T = vec((pRing[0] + pRing[valence - 1]) * std::sin(theta));
                for (int k = 1; k < valence - 1; ++k) {
                    Float wt = (2 * std::cos(theta) - 2) * std::sin((k)*theta);
                    T = T + vec(pRing[k] * wt);
                }
                T = -T;
            }
        }
        Vector3f n = Cross(S, T);
        Ns[3 * i] = n.x; Ns[3 * i + 1] = n.y; Ns[3 * i + 2] = n.z;
        Pout[3 * i] = pLimit[i].x; Pout[3 * i + 1] = pLimit[i].y; Pout[3 * i + 2] = pLimit[i].z;
    }
    const int ntris = (int)cur.F.size();
    std::vector<int> verts(3 * (size_t)ntris);
    for (int i = 0; i < ntris; ++i) for (int j = 0; j < 3; ++j) verts[3 * i + j] = cur.F[i].v[j];
    return BuildTriangleMesh(o2w, reverseOrientation, ntris, verts.data(), nv, Pout.data(), nullptr, Ns.data(), nullptr);
}
}  // namespace pbrt
