// Prediction for VERDICT r04 item 1 (two-level records): dependent record fetches per closest-hit ray on BASELINE config 3 under
//   A  today's child-pair records (one 64-B fetch per expanded interior node),
//   Q  128-B records at even depths holding the four grandchildren (a step expands the node and, if the ray goes on into it, its
//      near child; a popped odd-depth node re-reads its parent's record and expands alone),
//   Q' a 128-B record for EVERY interior node (a popped odd-depth node also gets two levels per fetch; twice the bytes).
// Rays: camera rays of the config-3 view + 4 cosine-distributed bounces (statistics only: Moeller-Trumbore, not the watertight
// test).  Input: nodes.bin (PgBVHNode[]), tris.bin (9 floats per triangle, BVH order) written by sim_quad_records.py.
// gcc -O2 -o sim_quad_records sim_quad_records.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef struct { float bmin[3], bmax[3]; int offset; uint16_t nprims; uint8_t axis, pad; } Node;
static Node *nodes; static float *tris; static int nn, nt;
static int *depthOf;
static uint64_t rng = 88172645463325252ull;
static double urand(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (rng >> 11) * (1.0 / 9007199254740992.0); }
static int box(const Node *n, const float *o, const float *inv, const int *neg, float tMax, float *tminOut) {
    float t0 = 0, t1 = tMax;
    for (int a = 0; a < 3; ++a) {
        float tn = ((neg[a] ? n->bmax[a] : n->bmin[a]) - o[a]) * inv[a], tf = ((neg[a] ? n->bmin[a] : n->bmax[a]) - o[a]) * inv[a];
        tf *= 1 + 2 * 1.7881393e-7f;
        if (tn > t0) t0 = tn; if (tf < t1) t1 = tf;
        if (t0 > t1) return 0;
    }
    *tminOut = t0; return 1;
}
static int tri(int k, const float *o, const float *d, float tMax, float *t) {
    const float *p = tris + 9 * (size_t)k;
    float e1[3], e2[3], h[3], s[3], q[3];
    for (int a = 0; a < 3; ++a) { e1[a] = p[3 + a] - p[a]; e2[a] = p[6 + a] - p[a]; s[a] = o[a] - p[a]; }
    h[0] = d[1] * e2[2] - d[2] * e2[1]; h[1] = d[2] * e2[0] - d[0] * e2[2]; h[2] = d[0] * e2[1] - d[1] * e2[0];
    float det = e1[0] * h[0] + e1[1] * h[1] + e1[2] * h[2];
    if (fabsf(det) < 1e-20f) return 0;
    float f = 1 / det, u = f * (s[0] * h[0] + s[1] * h[1] + s[2] * h[2]);
    if (u < 0 || u > 1) return 0;
    q[0] = s[1] * e1[2] - s[2] * e1[1]; q[1] = s[2] * e1[0] - s[0] * e1[2]; q[2] = s[0] * e1[1] - s[1] * e1[0];
    float v = f * (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]);
    if (v < 0 || u + v > 1) return 0;
    float tt = f * (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]);
    if (tt <= 1e-4f || tt >= tMax) return 0;
    *t = tt; return 1;
}
typedef struct { uint64_t rays, exp, tri, stepsQ, stepsQ2, popsPassOdd, popsPassEven, chains, maxStackA, maxStackQ, stackSumA, stackSumQ; uint64_t chainHist[32]; } Stats;
static Stats S;
// returns hit prim or -1
static int trace(const float *o, const float *d, float *tHit) {
    float inv[3] = {1 / d[0], 1 / d[1], 1 / d[2]}; int neg[3] = {inv[0] < 0, inv[1] < 0, inv[2] < 0};
    float tMax = INFINITY; int hit = -1;
    int stack[64]; int sp = 0, cur = 0;
    // chain bookkeeping: a chain = maximal run of expansions linked by near-descents
    int chainLen = 0, chainStartDepth = 0;
    int maxSp = 0;
    uint64_t exp = 0, stepsQ = 0, stepsQ2 = 0;
#define END_CHAIN() do { if (chainLen) { S.chains++; S.chainHist[chainLen < 31 ? chainLen : 31]++; \
        stepsQ2 += (chainLen + 1) / 2; \
        stepsQ += (chainStartDepth & 1) ? 1 + chainLen / 2 : (chainLen + 1) / 2; chainLen = 0; } } while (0)
    int fromPop = 1;
    for (;;) {
        const Node *n = &nodes[cur];
        float tmin;
        if (box(n, o, inv, neg, tMax, &tmin)) {
            if (n->nprims) {
                END_CHAIN();
                for (int k = 0; k < n->nprims; ++k) { float t; S.tri++; if (tri(n->offset + k, o, d, tMax, &t)) { tMax = t; hit = n->offset + k; } }
                if (!sp) break; cur = stack[--sp]; fromPop = 1;
            } else {
                ++exp;
                if (fromPop) { END_CHAIN(); chainStartDepth = depthOf[cur]; if (cur != 0) { if (depthOf[cur] & 1) S.popsPassOdd++; else S.popsPassEven++; } }
                ++chainLen; fromPop = 0;
                if (neg[n->axis]) { stack[sp++] = cur + 1; cur = n->offset; } else { stack[sp++] = n->offset; cur = cur + 1; }
                if (sp > maxSp) maxSp = sp;
            }
        } else { END_CHAIN(); if (!sp) break; cur = stack[--sp]; fromPop = 1; }
    }
    END_CHAIN();
    S.rays++; S.exp += exp; S.stepsQ += stepsQ; S.stepsQ2 += stepsQ2; S.stackSumA += maxSp; if ((uint64_t)maxSp > S.maxStackA) S.maxStackA = maxSp;
    *tHit = tMax; return hit;
}
static void norm3(float *v) { float l = sqrtf(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); nn = sz / 32; nodes = malloc(sz); if (fread(nodes, 1, sz, f) != (size_t)sz) return 1; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0, SEEK_SET); nt = sz / 36; tris = malloc(sz); if (fread(tris, 1, sz, f) != (size_t)sz) return 1; fclose(f);
    depthOf = calloc(nn, sizeof(int));
    for (int i = 0; i < nn; ++i) if (!nodes[i].nprims) { depthOf[i + 1] = depthOf[i] + 1; depthOf[nodes[i].offset] = depthOf[i] + 1; }
    int nInt = 0, nIntEven = 0; for (int i = 0; i < nn; ++i) if (!nodes[i].nprims) { ++nInt; if (!(depthOf[i] & 1)) ++nIntEven; }
    printf("nodes %d interior %d (even depth %d) tris %d\n", nn, nInt, nIntEven, nt);
    // camera: LookAt 0 -2.6 1.4 -> 0 0 0 up 0 0 1, fov 40 (of the shorter axis), 1920x1080
    float eye[3] = {0, -2.6f, 1.4f}, fw[3] = {0, 2.6f, -1.4f}; norm3(fw);
    float up[3] = {0, 0, 1}, rt[3] = {fw[1] * up[2] - fw[2] * up[1], fw[2] * up[0] - fw[0] * up[2], fw[0] * up[1] - fw[1] * up[0]}; norm3(rt);
    float u2[3] = {rt[1] * fw[2] - rt[2] * fw[1], rt[2] * fw[0] - rt[0] * fw[2], rt[0] * fw[1] - rt[1] * fw[0]};
    const float th = tanf(20 * 3.14159265f / 180), aspect = 1920.f / 1080.f;
    const int N = argc > 3 ? atoi(argv[3]) : 200000;
    Stats perDepth[6] = {0};
    for (int i = 0; i < N; ++i) {
        float sx = (2 * urand() - 1) * th * aspect, sy = (2 * urand() - 1) * th;
        float o[3] = {eye[0], eye[1], eye[2]}, d[3];
        for (int a = 0; a < 3; ++a) d[a] = fw[a] + sx * rt[a] + sy * u2[a];
        norm3(d);
        for (int b = 0; b < 5; ++b) {
            Stats before = S; float t;
            int h = trace(o, d, &t);
            perDepth[b].rays += S.rays - before.rays; perDepth[b].exp += S.exp - before.exp; perDepth[b].stepsQ += S.stepsQ - before.stepsQ; perDepth[b].stepsQ2 += S.stepsQ2 - before.stepsQ2; perDepth[b].tri += S.tri - before.tri;
            if (h < 0) break;
            const float *p = tris + 9 * (size_t)h;
            float e1[3], e2[3], n[3];
            for (int a = 0; a < 3; ++a) { e1[a] = p[3 + a] - p[a]; e2[a] = p[6 + a] - p[a]; }
            n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0]; norm3(n);
            if (n[0] * d[0] + n[1] * d[1] + n[2] * d[2] > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
            for (int a = 0; a < 3; ++a) o[a] = o[a] + t * d[a] + 1e-4f * n[a];
            // cosine hemisphere about n
            float r1 = urand(), r2 = urand(), r = sqrtf(r1), ph = 6.2831853f * r2, lx = r * cosf(ph), ly = r * sinf(ph), lz = sqrtf(1 - r1);
            float tx[3] = {fabsf(n[0]) > 0.5f ? 0 : 1, fabsf(n[0]) > 0.5f ? 1 : 0, 0}, sxx[3], tyy[3];
            sxx[0] = n[1] * tx[2] - n[2] * tx[1]; sxx[1] = n[2] * tx[0] - n[0] * tx[2]; sxx[2] = n[0] * tx[1] - n[1] * tx[0]; norm3(sxx);
            tyy[0] = n[1] * sxx[2] - n[2] * sxx[1]; tyy[1] = n[2] * sxx[0] - n[0] * sxx[2]; tyy[2] = n[0] * sxx[1] - n[1] * sxx[0];
            for (int a = 0; a < 3; ++a) d[a] = lx * sxx[a] + ly * tyy[a] + lz * n[a];
        }
    }
    printf("rays %llu: expansions/ray %.2f  tri tests/ray %.2f  | fetch rounds/ray: A %.2f  Q %.2f (%.3f of A)  Q' %.2f (%.3f of A)\n", (unsigned long long)S.rays, (double)S.exp / S.rays,
           (double)S.tri / S.rays, (double)(S.exp + S.tri) / S.rays, (double)(S.stepsQ + S.tri) / S.rays, (double)(S.stepsQ + S.tri) / (S.exp + S.tri),
           (double)(S.stepsQ2 + S.tri) / S.rays, (double)(S.stepsQ2 + S.tri) / (S.exp + S.tri));
    printf("interior steps only: A %.2f  Q %.2f (%.3f)  Q' %.2f (%.3f); chains/ray %.2f, passed pops at odd depth %.2f even %.2f per ray; mean max stack %.2f, max %llu\n", (double)S.exp / S.rays, (double)S.stepsQ / S.rays,
           (double)S.stepsQ / S.exp, (double)S.stepsQ2 / S.rays, (double)S.stepsQ2 / S.exp, (double)S.chains / S.rays, (double)S.popsPassOdd / S.rays, (double)S.popsPassEven / S.rays, (double)S.stackSumA / S.rays, (unsigned long long)S.maxStackA);
    for (int b = 0; b < 5; ++b) if (perDepth[b].rays) printf(" bounce %d: rays %llu exp %.2f tri %.2f Q %.2f Q' %.2f\n", b, (unsigned long long)perDepth[b].rays, (double)perDepth[b].exp / perDepth[b].rays, (double)perDepth[b].tri / perDepth[b].rays,
                                      (double)perDepth[b].stepsQ / perDepth[b].rays, (double)perDepth[b].stepsQ2 / perDepth[b].rays);
    printf("chain length histogram:"); for (int k = 1; k < 32; ++k) if (S.chainHist[k]) printf(" %d:%.3f", k, (double)S.chainHist[k] / S.chains); printf("\n");
    return 0;
}
