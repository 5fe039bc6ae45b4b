// Shapes the reference turns into triangle meshes at load time: "heightfield" (shapes/heightfield.cpp:41-87) and "nurbs"
// (shapes/nurbs.cpp:43-308, a fixed 30 x 30 dicing of the rational B-spline surface with analytic normals).  Both end in
// CreateTriangleMesh, i.e. BuildTriangleMesh here; the evaluation keeps the reference's float operation order so the
// vertices, normals and uv are the reference's bit for bit.
#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>
#include "api.h"
#include "error.h"
#include "scene.h"

namespace pbrt {
std::shared_ptr<TriangleMesh> BuildTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTriangles, const int *indices,
                                                int nVertices, const Float *P, const Float *S, const Float *N, const Float *UV);

// Two triangles per cell of an nu x nv vertex grid (row-major, u fastest): (a, b, c) and (a, c, d) with a = (u, v),
// b = (u+1, v), c = (u+1, v+1), d = (u, v+1) -- the winding of heightfield.cpp:69-81 and nurbs.cpp:286-299
static void GridIndices(int nu, int nv, std::vector<int> *idx) {
    idx->clear();
    idx->reserve((size_t)6 * (nu - 1) * (nv - 1));
    for (int v = 0; v + 1 < nv; ++v)
        for (int u = 0; u + 1 < nu; ++u) {
            const int a = v * nu + u, b = a + 1, c = a + nu + 1, d = a + nu;
            const int tri[6] = {a, b, c, a, c, d};
            idx->insert(idx->end(), tri, tri + 6);
        }
}

std::shared_ptr<TriangleMesh> CreateHeightfield(const Transform &o2w, bool reverseOrientation, const ParamSet &params) {
    const int nx = params.FindOneInt("nu", -1), ny = params.FindOneInt("nv", -1);
    const std::vector<Float> *z = params.FindFloat("Pz");
    if (nx < 2 || ny < 2 || !z || (int)z->size() != nx * ny) {  // the reference CHECK-fails (aborts) on these
        Error("Heightfield: \"nu\" x \"nv\" (both >= 2) height values \"Pz\" are required; ignoring the shape.");
        return nullptr;
    }
    std::vector<Float> P((size_t)3 * nx * ny), uv((size_t)2 * nx * ny);
    for (int y = 0, pos = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x, ++pos) {
            P[3 * pos] = uv[2 * pos] = (float)x / (float)(nx - 1);
            P[3 * pos + 1] = uv[2 * pos + 1] = (float)y / (float)(ny - 1);
            P[3 * pos + 2] = (*z)[pos];
        }
    std::vector<int> idx;
    GridIndices(nx, ny, &idx);
    return BuildTriangleMesh(o2w, reverseOrientation, (int)idx.size() / 3, idx.data(), nx * ny, P.data(), nullptr, nullptr, uv.data());
}

namespace {
struct H4 { Float x = 0, y = 0, z = 0, w = 0; };  // homogeneous control point
H4 blend(const H4 &a, const H4 &b, Float alpha) {  // a * alpha + b * (1 - alpha), component by component
    H4 r;
    r.x = a.x * alpha + b.x * (1 - alpha); r.y = a.y * alpha + b.y * (1 - alpha);
    r.z = a.z * alpha + b.z * (1 - alpha); r.w = a.w * alpha + b.w * (1 - alpha);
    return r;
}
// index of the knot span containing t (nurbs.cpp:43-52)
int KnotSpan(const Float *knot, int order, Float t) {
    int span = order - 1;
    while (t > knot[span + 1]) ++span;
    return span;
}
// de Boor evaluation of a rational B-spline curve whose control points are cp[i * stride], i = 0..np-1; optionally the
// derivative of the projected curve (nurbs.cpp:69-118)
H4 EvalCurve(int order, const Float *knot, const H4 *cp, int stride, Float t, Vector3f *deriv) {
    const int span = KnotSpan(knot, order, t);
    const Float *k = knot + span;
    const int first = span - order + 1;
    std::vector<H4> work((size_t)order);
    for (int i = 0; i < order; ++i) work[i] = cp[(first + i) * stride];
    for (int i = 0; i < order - 2; ++i)
        for (int j = 0; j < order - 1 - i; ++j) {
            const Float alpha = (k[1 + j] - t) / (k[1 + j] - k[j + 2 - order + i]);
            work[j] = blend(work[j], work[j + 1], alpha);
        }
    const Float alpha = (k[1] - t) / (k[1] - k[0]);
    const H4 val = blend(work[0], work[1], alpha);
    if (deriv) {
        const Float factor = (order - 1) / (k[1] - k[0]);
        H4 delta;
        delta.x = (work[1].x - work[0].x) * factor; delta.y = (work[1].y - work[0].y) * factor;
        delta.z = (work[1].z - work[0].z) * factor; delta.w = (work[1].w - work[0].w) * factor;
        deriv->x = delta.x / val.w - (val.x * delta.w / (val.w * val.w));
        deriv->y = delta.y / val.w - (val.y * delta.w / (val.w * val.w));
        deriv->z = delta.z / val.w - (val.z * delta.w / (val.w * val.w));
    }
    return val;
}
// surface point and both partial derivatives (nurbs.cpp:120-147): iso-curves in v through the u-span's control columns, then
// the u curve through them; the other way round for dpdv
Point3f EvalSurface(int uOrder, const Float *uKnot, int ucp, Float u, int vOrder, const Float *vKnot, Float v, const H4 *cp, Vector3f *dpdu,
                    Vector3f *dpdv) {
    const int uFirst = KnotSpan(uKnot, uOrder, u) - uOrder + 1, vFirst = KnotSpan(vKnot, vOrder, v) - vOrder + 1;
    // iso[uFirst + i] lines up with control column uFirst + i, as the reference's `iso - uFirstCp` indexing does
    std::vector<H4> iso((size_t)std::max(uFirst + uOrder, vFirst + vOrder));
    for (int i = 0; i < uOrder; ++i) iso[uFirst + i] = EvalCurve(vOrder, vKnot, &cp[uFirst + i], ucp, v, nullptr);
    const H4 P = EvalCurve(uOrder, uKnot, iso.data(), 1, u, dpdu);
    for (int i = 0; i < vOrder; ++i) iso[vFirst + i] = EvalCurve(uOrder, uKnot, &cp[(vFirst + i) * ucp], 1, u, nullptr);
    (void)EvalCurve(vOrder, vKnot, iso.data(), 1, v, dpdv);
    return Point3f(P.x / P.w, P.y / P.w, P.z / P.w);
}
}  // namespace

std::shared_ptr<TriangleMesh> CreateNURBS(const Transform &o2w, bool reverseOrientation, const ParamSet &params) {
    struct Dir { int n, order; const std::vector<Float> *knots; Float t0, t1; } d[2];
    const char *names[2][5] = {{"nu", "uorder", "uknots", "u0", "u1"}, {"nv", "vorder", "vknots", "v0", "v1"}};
    for (int k = 0; k < 2; ++k) {  // the reference reads nu, uorder, uknots, u0, u1, then the v set (nurbs.cpp:153-211)
        const char axis = k == 0 ? 'u' : 'v';
        d[k].n = params.FindOneInt(names[k][0], -1);
        if (d[k].n == -1) { Error("Must provide number of control points \"%s\" with NURBS shape.", names[k][0]); return nullptr; }
        d[k].order = params.FindOneInt(names[k][1], -1);
        if (d[k].order == -1) { Error("Must provide %c order \"%s\" with NURBS shape.", axis, names[k][1]); return nullptr; }
        d[k].knots = params.FindFloat(names[k][2]);
        if (!d[k].knots) { Error("Must provide %c knot vector \"%s\" with NURBS shape.", axis, names[k][2]); return nullptr; }
        if ((int)d[k].knots->size() != d[k].n + d[k].order) {
            Error("Number of knots in %c knot vector %d doesn't match sum of number of %c control points %d and %c order %d.", axis, (int)d[k].knots->size(), axis,
                  d[k].n, axis, d[k].order);
            return nullptr;
        }
        if (d[k].order < 2 || d[k].n < d[k].order) { Error("NURBS shape: %c order %d with %d control points cannot be evaluated.", axis, d[k].order, d[k].n); return nullptr; }
        d[k].t0 = params.FindOneFloat(names[k][3], (*d[k].knots)[d[k].order - 1]);
        d[k].t1 = params.FindOneFloat(names[k][4], (*d[k].knots)[d[k].n]);
    }
    const int nu = d[0].n, nv = d[1].n;
    std::vector<H4> Pw((size_t)nu * nv);
    if (const std::vector<Float> *P = params.FindPoint3f("P")) {
        if ((int)P->size() / 3 != nu * nv) { Error("NURBS shape was expecting %dx%d=%d control points, was given %d", nu, nv, nu * nv, (int)P->size() / 3); return nullptr; }
        for (int i = 0; i < nu * nv; ++i) { Pw[i].x = (*P)[3 * i]; Pw[i].y = (*P)[3 * i + 1]; Pw[i].z = (*P)[3 * i + 2]; Pw[i].w = 1.; }
    } else if (const std::vector<Float> *Ph = params.FindFloat("Pw")) {
        if (Ph->size() % 4) { Error("Number of \"Pw\" control points provided to NURBS shape must be multiple of four"); return nullptr; }
        if ((int)Ph->size() / 4 != nu * nv) { Error("NURBS shape was expecting %dx%d=%d control points, was given %d", nu, nv, nu * nv, (int)Ph->size() / 4); return nullptr; }
        for (int i = 0; i < nu * nv; ++i) { Pw[i].x = (*Ph)[4 * i]; Pw[i].y = (*Ph)[4 * i + 1]; Pw[i].z = (*Ph)[4 * i + 2]; Pw[i].w = (*Ph)[4 * i + 3]; }
    } else { Error("Must provide control points via \"P\" or \"Pw\" parameter to NURBS shape."); return nullptr; }

    const int dice = 30;  // diceu = dicev = 30, nurbs.cpp:239
    std::vector<Float> ueval(dice), veval(dice);
    for (int i = 0; i < dice; ++i) {
        const Float t = (float)i / (float)(dice - 1);  // Lerp(t, a, b) = (1 - t) * a + t * b
        ueval[i] = (1 - t) * d[0].t0 + t * d[0].t1;
        veval[i] = (1 - t) * d[1].t0 + t * d[1].t1;
    }
    std::vector<Float> P((size_t)3 * dice * dice), N((size_t)3 * dice * dice), uv((size_t)2 * dice * dice);
    for (int v = 0; v < dice; ++v)
        for (int u = 0; u < dice; ++u) {
            const int k = v * dice + u;
            uv[2 * k] = ueval[u]; uv[2 * k + 1] = veval[v];
            Vector3f dpdu, dpdv;
            const Point3f pt = EvalSurface(d[0].order, d[0].knots->data(), nu, ueval[u], d[1].order, d[1].knots->data(), veval[v], Pw.data(), &dpdu, &dpdv);
            P[3 * k] = pt.x; P[3 * k + 1] = pt.y; P[3 * k + 2] = pt.z;
            const Vector3f n = Normalize(Cross(dpdu, dpdv));
            N[3 * k] = n.x; N[3 * k + 1] = n.y; N[3 * k + 2] = n.z;
        }
    std::vector<int> idx;
    GridIndices(dice, dice, &idx);
    return BuildTriangleMesh(o2w, reverseOrientation, (int)idx.size() / 3, idx.data(), dice * dice, P.data(), nullptr, N.data(), uv.data());
}
}  // namespace pbrt
