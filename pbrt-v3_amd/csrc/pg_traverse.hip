// pg_traverse.hip -- BVH traversal kernels of the MI355X path tracer (gfx950):
// BVHAccel::Intersect (closest hit, bvh.cpp:662-700) and BVHAccel::IntersectP
// (any hit, bvh.cpp:702-738) with Triangle::Intersect[P] (triangle.cpp:188-572)
// over an SoA ray queue, one ray per lane.
//
// Design (DESIGN.md "k_trace"):
//  * child-pair node records (64 B): an interior record holds the boxes and
//    references of BOTH children, so one dependent fetch replaces the
//    reference's two node fetches per level, and a leaf's (first prim, count)
//    sits in its parent's record -- a leaf costs no node fetch at all;
//  * per-ray visiting order, tMax shrinking and tie-breaking are exactly the
//    reference's: the far child's slab interval is computed when its parent is
//    read, its entry tMin goes on the stack, and the reference's later
//    `tMin < ray.tMax` decision is re-taken with the then-current tMax at pop
//    time (the other terms of Bounds3::IntersectP do not depend on ray.tMax);
//  * the reference's node-visit count (one per nodes[cur] read, bvh.cpp:672/710)
//    is reproduced exactly: it defines the algorithmic bytes of the roofline;
//  * one step per iteration for the whole wave -- either every lane holding an
//    interior record expands it or every lane holding a leaf tests its next
//    triangle (the larger group goes) -- instead of nested per-lane loops whose
//    trip counts diverge across 64 lanes;
//  * persistent waves: a wave takes chunks of rays from its XCD's region of the
//    queue (one atomic per chunk) and refills lanes whose ray has finished, so
//    lane occupancy does not decay to the longest ray of the first 64;
//  * traversal stack: first `depth` entries per lane in LDS ([entry][lane],
//    8 B entries, bank-conflict-free), the rest of pbrt's 64 in scratch;
//  * any-hit rays in two orders (KIND): the reference's (its node and triangle
//    counters reproduced exactly) and a free one -- the child the ray enters
//    first -- whose answer is the same by construction: ray.tMax never changes
//    during IntersectP, so which nodes pass their box test, and with them which
//    primitives can be met at all, does not depend on the order of the visits.
#include "pg_device.h"
#include "pg_sphere.h"
#include <algorithm>
#include <cstring>
#include "pg_kernels.h"
#include "pg_texture.h"
#include "pg_motion.h"

#ifndef TR_BLOCK
#define TR_BLOCK 256
#endif
#define TR_NONE ((int)0x80000000)
#define TR_STACK_TOTAL 64
// 256-thread blocks resident per CU = min(8, floor(800 / (ceil(sgpr/16)*16 + 16))) (MI355X_MICROARCH.md): 106 SGPRs admit 6,
// <= 96 admit 7.  The cap makes the compiler keep the excess in VGPR lanes (measured: +2.5 % whole-frame).
#ifndef TR_SGPR_ATTR
#define TR_SGPR_ATTR __attribute__((amdgpu_num_sgpr(96)))
#endif
#ifndef TR_MIN_WAVES
#define TR_MIN_WAVES 2
#endif
// instanced triangle scenes: 7 waves per SIMD (72 VGPRs) since round 6 -- the kernels needed 85 - 87 registers (5 waves; held to 80 they spilled
// 7 - 10 and lost, profiles/r03d_xprim_register_diet.txt) until the diet described at k_trace; each resident wave is worth 5 - 13 % of this
// latency-bound kernel's time (4 / 5 / 6 / 7 waves: 335 / 292 / 259 / 244 ms of closest-hit traversal per frame of the config-4 stand-in,
// profiles/r06_trace_inst_ab.txt).  The free-order any-hit kernel fits 8 waves (63 registers) but then needs the stack depth 10 that costs more.
#ifndef TR_FREE_EXTRA_WAVES
#define TR_FREE_EXTRA_WAVES 0
#endif
#ifndef TR_INST_WAVES
#define TR_INST_WAVES 7
#endif
#ifndef TR_FLAT_WAVES  // triangle-only scenes (XP 0, XP_ALPHA)
#define TR_FLAT_WAVES TR_MIN_WAVES
#endif
#define TR_MAX_ACCEPTED 4096  // (1+2^-24)^(3*4096) < 1+2^-10
#ifndef TR_DEFAULT_DEPTH
#define TR_DEFAULT_DEPTH 11
#endif

// TR_STATS (tools/trace_step_stats.py; never defined in the product build): where the lanes of a wave go -- per step kind the number of
// wave steps and of lanes that took part, and what the other lanes were doing meanwhile.  Wave-uniform counts, added up once per wave.
#ifdef TR_STATS
enum { TS_ITER, TS_INT_STEPS, TS_INT_LANES, TS_TRI_STEPS, TS_TRI_LANES, TS_ALPHA_STEPS, TS_ALPHA_LANES, TS_ENTER_STEPS, TS_ENTER_LANES, TS_EXIT_STEPS,
       TS_EXIT_LANES, TS_REFILL_STEPS, TS_REFILL_LANES, TS_WAIT_ENTER, TS_WAIT_EXIT, TS_IDLE, TS_INST_PRIM_LANES, TS_RAYS, TS_IN_INST_INT, TS_IN_INST_TRI,
       TS_ALPHA_WAIT, TS_N };
__device__ unsigned long long g_trStats[3][TS_N];
#define TS_ADD(k, v) (ts[k] += (unsigned long long)(v))
extern "C" int pg_debug_trace_stats(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_trStats), sizeof(unsigned long long) * 3 * TS_N) != hipSuccess) return -1;
    if (reset) { static const unsigned long long z[3][TS_N] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_trStats), z, sizeof(z)) != hipSuccess) return -1; }
    return TS_N;
}
#else
#define TS_ADD(k, v) ((void)0)
#endif

// Bounds3::IntersectP(ray, invDir, dirIsNeg), geometry.h:1412-1438, split into
// the part that does not depend on ray.tMax (returns ok, tMin) and the final
// `tMin < ray.tMax`, which the caller evaluates when the reference would.
PG_DEV bool slab_interval(float lox, float hix, float loy, float hiy, float loz, float hiz, float ox, float oy, float oz,
                          float ix, float iy, float iz, bool nx, bool ny, bool nz, float &tMinOut) {
    float tMin = ((nx ? hix : lox) - ox) * ix;
    float tMax = ((nx ? lox : hix) - ox) * ix;
    float tyMin = ((ny ? hiy : loy) - oy) * iy;
    float tyMax = ((ny ? loy : hiy) - oy) * iy;
    const float widen = 1 + 2 * pgamma(3);
    tMax *= widen;
    tyMax *= widen;
    bool ok = !(tMin > tyMax || tyMin > tMax);
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((nz ? hiz : loz) - oz) * iz;
    float tzMax = ((nz ? loz : hiz) - oz) * iz;
    tzMax *= widen;
    ok = ok && !(tMin > tzMax || tzMin > tMax);
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    tMinOut = tMin;
    return ok && (tMax > 0);
}

// The same test for the lanes of a wave at once, as a lane mask: every comparison is balloted on its own and the masks are
// combined with scalar instructions (a per-lane `bool` that is the AND of several comparisons costs two vector
// instructions each time it is selected or balloted; the masks cost none).
PG_DEV unsigned long long slab_mask(float lox, float hix, float loy, float hiy, float loz, float hiz, float ox, float oy, float oz, float ix,
                                    float iy, float iz, bool nx, bool ny, bool nz, float &tMinOut) {
    float tMin = ((nx ? hix : lox) - ox) * ix;
    float tMax = ((nx ? lox : hix) - ox) * ix;
    float tyMin = ((ny ? hiy : loy) - oy) * iy;
    float tyMax = ((ny ? loy : hiy) - oy) * iy;
    const float widen = 1 + 2 * pgamma(3);
    tMax *= widen;
    tyMax *= widen;
    unsigned long long ok = __ballot(!(tMin > tyMax)) & __ballot(!(tyMin > tMax));
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((nz ? hiz : loz) - oz) * iz;
    float tzMax = ((nz ? loz : hiz) - oz) * iz;
    tzMax *= widen;
    ok &= __ballot(!(tMin > tzMax)) & __ballot(!(tzMin > tMax));
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    tMinOut = tMin;
    return ok & __ballot(tMax > 0);
}

PG_DEV unsigned long long tr_wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// XP: what the scene has besides plain triangles, one instantiation per combination that occurs, so that a scene pays (in
// registers, i.e. resident waves) only for what it contains: XP_INST object instances (TransformedPrimitive over an object
// definition's own BVH), XP_QUADRIC spheres / cylinders / disks / cones / paraboloids / hyperboloids (EFloat arithmetic),
// XP_ALPHA alpha / shadow-alpha masks that are constants or plain image maps (DAlphaTex: four texels, inline), XP_ALPHATEX masks
// of any texture type through the general evaluator (a function call: 180+ registers and 1 KB of scratch for the whole kernel).
// 0 = the lean triangle-only kernel.
#define XP_INST 1
#define XP_QUADRIC 2
#define XP_ALPHA 4
#define XP_ALPHATEX 8
#define XP_ANIM 16  // moving instances (PgInstance::animated): their transform is interpolated at the ray's time (with XP_INST only)
#define XP_GENERAL (XP_INST | XP_QUADRIC | XP_ALPHATEX)
// ABI 29: TransformedPrimitives inside object definitions (a moving shape between ObjectBegin and ObjectEnd, api.cpp:1386-1419): a lane keeps a SECOND saved
// context (the rest of the instance's leaf, the instance ray's tMax, the visit bookkeeping, the stack floor) while it walks the inner object, and comes back
// to the instance's ray by deriving it again from the queue entry and the outer transform.  With XP_ANIM | XP_GENERAL only: the feature's own instantiation.
#define XP_NEST 32

// MIPMap<Float>::Lookup(st, 0 width) of a DAlphaTex = `triangle(0, st)` (mipmap.h:231-243) with Texel's wrap modes (:189-212): the
// operations of mip_triangle / mip_texel (pg_texture.h) on the red channel, in the same order.
PG_DEV float alpha_texel(const DScene &sc, const DAlphaTex &a, int s, int t) {
    if (a.wrap == 0) { s = mod_i(s, a.width); t = mod_i(t, a.height); }  // (mod_i skips its division where the texel is inside the map)
    else if (a.wrap == 2) { s = s < 0 ? 0 : (s > a.width - 1 ? a.width - 1 : s); t = t < 0 ? 0 : (t > a.height - 1 ? a.height - 1 : t); }
    else if (s < 0 || s >= a.width || t < 0 || t >= a.height) return 0.f;
    return sc.texels[a.offset + ((size_t)t * a.width + s)];
}
PG_DEV float alpha_lookup(const DScene &sc, const DAlphaTex &a, float u, float v) {
    if (a.image < 0) return a.constant;
    const float st0 = a.su * u + a.du, st1 = a.sv * v + a.dv;  // UVMapping2D::Map, texture.cpp:93-100
    const float s = st0 * a.width - 0.5f, t = st1 * a.height - 0.5f;
    const int s0 = (int)floorf(s), t0 = (int)floorf(t);
    const float ds = s - s0, dt = t - t0;
    return alpha_texel(sc, a, s0, t0) * ((1 - ds) * (1 - dt)) + alpha_texel(sc, a, s0, t0 + 1) * ((1 - ds) * dt) +
           alpha_texel(sc, a, s0 + 1, t0) * (ds * (1 - dt)) + alpha_texel(sc, a, s0 + 1, t0 + 1) * (ds * dt);
}
// Ray::time of entry `ray` of the traced queue(s); the unit entry points (pg_intersect / pg_intersect_p: origins, directions and tMax only) trace
// rays of time 0, Ray's default (geometry.h:865-866)
PG_DEV float trace_ray_time(const DScene &sc, const RayQueue &q0, const RayQueue &q1, int ray, int hitOffset1) {
    if (!sc.rayTimes) return 0.f;
    if (q1.regionCap > 0 && ray >= hitOffset1) return PG_QUEUE_TIMES(sc, q1)[ray - hitOffset1];
    return PG_QUEUE_TIMES(sc, q0)[ray];
}
// KIND: 0 closest hit (BVHAccel::Intersect), 1 any hit in the reference's order (BVHAccel::IntersectP, counters exact),
// 2 any hit in free order (same occlusion answers; the counters then say what THIS traversal read)
// What keeps the instanced kernels' registers at 72 (7 resident waves; 87 and 5 before round 6: profiles/r06_trace_inst_ab.txt).  The triangle-only
// kernels sit at 7 / 8 waves either way and are a little FASTER without the first two (config 3: 216.2 vs 219.2 ms per frame), so these are
// per-instantiation switches:
//   HITSTORE  the closest-hit record goes to memory when a hit is accepted (the last one stays) instead of waiting in four registers for the retire
//   NOSZ      Sz = 1 / d[kz] of the triangle test is one of the slab test's three reciprocals: no register and no division of its own
//   WCNT      the two statistics are counted per wave in scalar registers (a population count per step), not per lane
#ifndef TR_FLAT_DIET
#define TR_FLAT_DIET 0
#endif
#ifndef TR_INST_GROUP
#define TR_INST_GROUP 8  // lanes waiting for an instance entry / exit step before the wave runs it (measured 4 / 8 / 12 / 20 / 32: profiles/r03x_*)
#endif
template <int KIND, int XP>
__global__ __launch_bounds__(TR_BLOCK, ((XP == XP_INST || XP == (XP_INST | XP_ALPHA)) ? (KIND == 2 ? TR_INST_WAVES + TR_FREE_EXTRA_WAVES : TR_INST_WAVES) : ((XP == 0 || XP == XP_ALPHA) ? TR_FLAT_WAVES : TR_MIN_WAVES))) TR_SGPR_ATTR void k_trace(DScene sc, RayQueue q0, RayQueue q1, float4 *__restrict__ hits,
                                                    int hitOffset1, float *__restrict__ tOut, int *__restrict__ occluded,
                                                    TraceCounters *cn, int *__restrict__ cursors, int depth, int chunk, int refillAt, int triW,
                                                    float cullK, int *cullGuard, int maxAccepted) {
    constexpr bool ANYHIT = KIND != 0, FREE = KIND == 2, NEST = (XP & XP_NEST) != 0;
    constexpr bool DIET = (XP & XP_INST) || (TR_FLAT_DIET & 1), HITSTORE = !ANYHIT && ((XP & XP_INST) || (TR_FLAT_DIET & 2)), NOSZ = (XP & XP_INST) || (TR_FLAT_DIET & 4);
    extern __shared__ uint2 ldsStack[];  // [depth][TR_BLOCK]
    uint2 spill[TR_STACK_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // Work distribution: the queue is PG_REGIONS sub-queues, one per XCD (block b runs on XCD b % 8, and the producers
    // appended from that XCD, so each private L2 sees one coherent part of the queue); a wave takes `chunk` rays at a time
    // from its region's cursor and moves on to the next region when its own is drained.  Waves are persistent: the grid
    // only has to fill the chip.  One launch can drain two queues (closest hit: the next bounce's rays and the previous
    // bounce's MIS rays): regions 0-7 are q0's, 8-15 are q1's; q1's results go to hits[hitOffset1 + i].
    const int nRegions = q1.regionCap > 0 ? 2 * PG_REGIONS : PG_REGIONS;
    int region = blockIdx.x & (PG_REGIONS - 1), regionsTried = 0;
    bool fresh = false;  // moved to a region this wave has not taken a chunk from yet: look before the atomic
    int next = 0, segEnd = 0, curQ = 0;  // the wave's current chunk [next, segEnd) of queue curQ
    bool exhausted = false;
    int idleThreshold = refillAt;
    // per-lane ray state.  A lane is in exactly one of three states:
    //   cur >= 0                 : holds an interior record to expand (its box test already passed)
    //   cur == NONE, triLeft > 0 : holds a leaf with triLeft untested triangles starting at triNext
    //   cur == NONE, triLeft == 0: idle (ray finished or none assigned)
    int ray = -1, cur = TR_NONE, triNext = 0, triLeft = 0;
    float ox = 0, oy = 0, oz = 0, ix = 1, iy = 1, iz = 1, tMax = 0;
    TriRay tr = {2, 0.f, 0.f, 1.f};  // Triangle::Intersect's per-ray permutation and shear (triangle.cpp:205-220)
    bool nx = false, ny = false, nz = false;
    unsigned negBits = 0;  // nx | ny << 1 | nz << 2: dirIsNeg[] indexed by a node's split axis
    int hitPrim = -1;
    float hb0 = 0, hb1 = 0, hb2 = 0;
    int sp = 0;                       // real stack entries
    int vd = 0;                       // ANYHIT: reference stack depth (real + culled entries)
    unsigned long long vmask = 0;     // ANYHIT: bit i set = the reference's entry at depth i was culled early
    unsigned int nodeVisits = 0, triTests = 0;
    // every step is one kind of step for the whole wave, so what it adds to the two statistics is a population count of the step's lane mask:
    // scalar registers and scalar adds instead of two vector registers and two vector adds per lane (only the reference-order any-hit walk counts
    // a per-lane number of visits inside its pop loop and keeps the lane counter)
    constexpr bool WCNT = KIND != 1 && DIET;
    unsigned long long wNodes = 0, wTris = 0;
    int nAccepted = 0;  // hits accepted by this lane's current ray (bounds the rounding growth of tMax, see cullK below)
    // XP_INST: object instances.  While a lane traverses an instance's BVH its ray registers hold the instance-space ray
    // (TransformedPrimitive::Intersect, primitive.cpp:76-96); the world ray, the rest of the world leaf and (any-hit) the
    // world BVH's visit bookkeeping wait here.  Entries of the instance's traversal sit on the same stack above spBase.
    // Of the world ray only tMax is kept (it shrinks with every hit): origin, reciprocal direction and the triangle shear are
    // derived again from the queue entry when the lane comes back -- the same operations on the same inputs as at the refill,
    // so the same bits -- which spares ten registers for the whole kernel.  (Round 6 measured the alternative of parking them in
    // LDS behind the stack, 6 / 9 / 11 words per lane: the exit step loses its loads and divisions, the kernel a resident block or
    // two stack entries; slower: profiles/r06_trace_inst_ab.txt.)
    int inInst = -1, hitInstCur = -1, spBase = 0, wvd = 0;
    // Entering and leaving an instance are steps of their own (like "expand an interior record" and "test a triangle"): a lane
    // that meets an instance record in a leaf waits with cur = -2 - (the instance's index) -- between iterations cur is otherwise an
    // interior record (>= 0) or TR_NONE --, a lane that has finished an instance's BVH waits as it is, and the wave runs the entry /
    // exit code when TR_INST_GROUP lanes wait for it or nothing else is left to do.  Run inline where they arose, the two blocks
    // executed in half of all wave iterations of the instanced 5 M-triangle scene with 2.6 (exit) and 5.3 (entry) active lanes
    // (profiles/r03w_trace_step_statistics.txt; the gain is 3 % closest hit, 10 % any hit).  The alpha lookup as a third such step
    // (candidate hits waiting in the stack slots above the lane's top) LOSES 3 - 6 % at group sizes 2 / 4 / 8: r06_trace_inst_ab.txt.
    unsigned wLeaf = 0;  // the rest of the world leaf: next primitive << (leafBits + 1) | primitives left
    // XP_NEST: the same one level further in -- the inner TransformedPrimitive's index, and what waits of the instance around it
    int inInst2 = -1, hitInst2Cur = -1, spBase1 = 0, w2vd = 0;
    unsigned w2Leaf = 0;
    float w2tMax = 0;
    bool instHit1 = false;
    unsigned long long w2vmask = 0;
    bool instHit = false;
    float wtMax = 0;
    unsigned long long wvmask = 0;
    const int leafBits = sc.leafBits, leafMask = (1 << leafBits) - 1;
#ifdef TR_STATS
    unsigned long long ts[TS_N] = {};
#endif

#define TR_PUSH(ref_, t_) do { uint2 e_ = make_uint2((unsigned)(ref_), __float_as_uint(t_)); \
        if (sp < depth) ldsStack[sp * TR_BLOCK + tid] = e_; else spill[sp - depth] = e_; ++sp; } while (0)
    // Pops are written as one loop per address space (scratch part first, then LDS): a single `sp < depth ? lds : spill`
    // expression makes the compiler build a generic pointer and issue slow flat loads.
    // The reference's "pop or finish" (bvh.cpp:694-697): next node whose deferred `tMin < ray.tMax` test passes.
#define TR_POP() do { cur = TR_NONE; \
        if (!ANYHIT) { \
            while (sp > depth && sp > spBase) { --sp; const uint2 e_ = spill[sp - depth]; if (__uint_as_float(e_.y) < tMax) { cur = (int)e_.x; break; } } \
            if (cur == TR_NONE) while (sp > spBase && sp <= depth) { --sp; const uint2 e_ = ldsStack[sp * TR_BLOCK + tid]; if (__uint_as_float(e_.y) < tMax) { cur = (int)e_.x; break; } } \
        } else if (FREE) { if (sp > spBase) { --sp; if (sp >= depth) cur = (int)spill[sp - depth].x; else cur = (int)ldsStack[sp * TR_BLOCK + tid].x; } \
        } else { while (vd > 0) { --vd; ++nodeVisits; if ((vmask >> vd) & 1ull) { vmask &= ~(1ull << vd); continue; } \
                                 --sp; if (sp >= depth) cur = (int)spill[sp - depth].x; else cur = (int)ldsStack[sp * TR_BLOCK + tid].x; break; } } } while (0)
    // A negative reference is a leaf: unpack (first prim, count) into the lane's triangle state.
#define TR_SETTLE() do { if (cur < 0 && cur != TR_NONE) { const int code_ = ~cur; triNext = code_ >> leafBits; triLeft = (code_ & leafMask) + 1; cur = TR_NONE; } } while (0)

    // the accepted hit, written with selects (in place: a branch here makes the compiler copy the whole hit record aside before the test and
    // back after it); HITSTORE: the record goes to memory now (the last accepted one stays), not from registers when the ray retires
#define TR_ACCEPT_CLOSEST(hit_, prim_, t_, b0_, b1_, b2_) do { tMax = (hit_) ? (t_) : tMax; \
        if (HITSTORE) { \
            if (hit_) { hits[ray] = make_float4(__int_as_float(prim_), b0_, b1_, b2_); \
                        if ((XP & XP_INST) && sc.hitInst) sc.hitInst[ray] = (NEST && inInst >= 0) ? inInst + sc.nInstances * (inInst2 + 1) : inInst; } \
            if (XP & XP_ANIM) hitInstCur = (hit_) ? inInst : hitInstCur; \
            if (NEST) hitInst2Cur = (hit_) ? inInst2 : hitInst2Cur; \
        } else { \
            hb0 = (hit_) ? (b0_) : hb0; hb1 = (hit_) ? (b1_) : hb1; hb2 = (hit_) ? (b2_) : hb2; \
            if (XP & XP_INST) hitInstCur = (hit_) ? inInst : hitInstCur; \
        } } while (0)
#define TR_ACCEPT(hit_, prim_, t_, b0_, b1_, b2_) do { hitPrim = (hit_) ? (prim_) : hitPrim; \
        if (ANYHIT) {  /* bvh.cpp:717: return true */ \
            triLeft = (hit_) ? 0 : triLeft; sp = (hit_) ? 0 : sp; vd = (hit_) ? 0 : vd; \
            if (XP & XP_INST) inInst = (hit_) ? -1 : inInst; \
            if (NEST) inInst2 = (hit_) ? -1 : inInst2; \
        } else {  /* primitive.cpp:123: r.tMax = tHit */ \
            TR_ACCEPT_CLOSEST(hit_, prim_, t_, b0_, b1_, b2_); \
            if (XP & XP_INST) instHit = instHit || (hit_); \
            nAccepted += (hit_) ? 1 : 0; \
            if ((hit_) && nAccepted == maxAccepted) atomicOr(cullGuard, 1); \
        } } while (0)

    for (;;) {
        // ---- retire finished rays and refill idle lanes from this wave's segment.  A finished ray's result stays in its lane
        //      until the wave refills (>= refillAt idle lanes) or runs dry, so the stores run with many lanes active instead of
        //      once per iteration for one or two lanes.  Measured and NOT kept: copying the hit triangle's 48-B record next to
        //      the hit here to spare the shading kernel its gather (+12 ms per frame here, no gain in k_shade); handing invDir
        //      and the triangle test's shear over from the ray's producer instead of deriving them at refill (no gain:
        //      the refill costs 9 % of this kernel's issue slots, but the extra 32 B per ray cost as much as the divisions).
        const bool wantExit = (XP & XP_INST) && inInst >= 0 && cur == TR_NONE && triLeft == 0;  // finished an instance's BVH
        const bool wantEnter = (XP & XP_INST) && cur < 0 && cur != TR_NONE;
        const bool idle = cur == TR_NONE && triLeft == 0 && !wantExit && !wantEnter;
        const unsigned long long idleMask = __ballot(idle);
        const int nIdle = __popcll(idleMask);
        TS_ADD(TS_ITER, 1); TS_ADD(TS_IDLE, nIdle);
        // one scalar comparison decides between the step and the (rarer) retire / refill path: the threshold is refillAt
        // while the queues still have rays and 64 -- every lane idle -- once they are exhausted
        if (nIdle >= idleThreshold) {
            if (idle && ray >= 0) {
                if (ANYHIT) occluded[ray] = hitPrim >= 0 ? 1 : 0;
                else {
                    if (!HITSTORE) {
                        hits[ray] = make_float4(__int_as_float(hitPrim), hb0, hb1, hb2);
                        if ((XP & XP_INST) && sc.hitInst) sc.hitInst[ray] = hitInstCur;
                    }
                    if ((XP & XP_ANIM) && sc.animXf && hitInstCur >= 0 && sc.instances[hitInstCur].animated) {
                        // InterpolatedPrimToWorld of the accepted hit's instance, for *isect = InterpolatedPrimToWorld(*isect) in the shading kernels
                        float xf[PG_XF_STRIDE];
                        instance_matrices_at(sc.instances[hitInstCur], trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                        for (int k = 0; k < 33; ++k) sc.animXf[(size_t)PG_XF_STRIDE * ray + k] = xf[k];
                    }
                    if (NEST && sc.animXf && hitInst2Cur >= 0 && sc.instances[hitInst2Cur].animated) {  // the inner transform of a hit two levels deep: the buffer's second half
                        float xf[PG_XF_STRIDE];
                        instance_matrices_at(sc.instances[hitInst2Cur], trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                        for (int k = 0; k < 33; ++k) sc.animXf[(size_t)PG_XF_STRIDE * ((size_t)sc.nestXfOff + ray) + k] = xf[k];
                    }
                    if (tOut) tOut[ray] = tMax;
                }
                ray = -1;
            }
            if (exhausted) break;
        }
        if (nIdle >= idleThreshold) {  // (refill: the queues were not exhausted on entry)
            if (next >= segEnd) {  // wave-uniform: take the next chunk
                for (;;) {
                    const int qsel = region >> 3, rr = region & (PG_REGIONS - 1);
                    const RayQueue &cq = qsel ? q1 : q0;
                    const int count = cq.counts[rr * PG_COUNT_STRIDE];
                    int *cursor = &cursors[region * PG_COUNT_STRIDE];
                    // a drained region costs a load, not an atomic: cursors only grow, so a stale value can only under-report
                    // (measured and removed, round 6: guided self-scheduling -- a region's last round handed out in shares of what is left, 16 / 32 /
                    // 64 rays at least, instead of full chunks -- makes a 1 / 8 film shard SLOWER, 55.8 / 54.6 / 52.9 against 52.2 ms: a launch's
                    // tail is its longest ray, not its last chunk; profiles/r06f_guided_chunks.txt)
                    bool drained = fresh && __hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= count;
                    if (!drained) {
                        int base = 0;
                        if (lane == 0) base = atomicAdd(cursor, chunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < count) {
                            next = rr * cq.regionCap + base; segEnd = next + min(chunk, count - base);
                            curQ = qsel; fresh = false;
                            break;
                        }
                    }
                    // next region: the other regions of this queue first (same XCD-local L2 contents last), then the other queue
                    ++regionsTried;
                    if (regionsTried == nRegions) { exhausted = true; idleThreshold = 64; break; }
                    const int own = blockIdx.x & (PG_REGIONS - 1);
                    region = ((regionsTried >> 3) << 3) | ((own + regionsTried) & (PG_REGIONS - 1));
                    fresh = true;
                }
            }
            if (!exhausted) {
                int idx = next + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(idleMask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idleMask, 0u));  // + the idle lanes below this one (v_mbcnt: no lane-mask registers)
                next += nIdle;
                TS_ADD(TS_REFILL_STEPS, 1); TS_ADD(TS_REFILL_LANES, __popcll(__ballot(idle && idx < segEnd)));
                if (WCNT && sc.nNodes > 0) wNodes += __popcll(__ballot(idle && idx < segEnd));  // nodes[0]
                if (idle && idx < segEnd) {
                    const float4 o4 = curQ ? q1.o[idx] : q0.o[idx], d4 = curQ ? q1.d[idx] : q0.d[idx];
                    ray = idx + (curQ ? hitOffset1 : 0);  // index of this ray's result
                    if (HITSTORE) { hits[ray] = make_float4(__int_as_float(-1), 0.f, 0.f, 0.f); if ((XP & XP_INST) && sc.hitInst) sc.hitInst[ray] = -1; }  // "no hit" until one is accepted
                    ox = o4.x; oy = o4.y; oz = o4.z; tMax = o4.w;
                    tr = tri_ray_setup(mk(d4.x, d4.y, d4.z));
                    ix = 1 / d4.x; iy = 1 / d4.y; iz = 1 / d4.z;   // bvh.cpp:666
                    nx = ix < 0; ny = iy < 0; nz = iz < 0; negBits = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);   // bvh.cpp:667
                    hitPrim = -1; hb0 = hb1 = hb2 = 0; nAccepted = 0;
                    inInst = -1; hitInstCur = -1; spBase = 0;
                    if (NEST) { inInst2 = -1; hitInst2Cur = -1; }
                    sp = 0; vd = 0; vmask = 0;
                    if (sc.nNodes > 0) {
                        if (!WCNT) ++nodeVisits;  // nodes[0]
                        float t0;
                        if (slab_interval(sc.rootBox[0], sc.rootBox[3], sc.rootBox[1], sc.rootBox[4], sc.rootBox[2], sc.rootBox[5], ox, oy, oz,
                                          ix, iy, iz, nx, ny, nz, t0) && t0 < tMax) {
                            cur = sc.rootRef;
                            TR_SETTLE();
                        }
                    }
                }
            }
            continue;  // lanes that finished at once (root miss) retire at the next refill
        }

        // ---- one step for the wave: either every lane holding an interior record expands it, or every lane
        //      holding a leaf tests its next triangle.  The larger group goes first (weighted by triW/16), so
        //      at least about half of the busy lanes are active in every step and neither group starves.
        // ---- instance steps: leave, then enter, when enough lanes wait or no lane can do anything else
        if (XP & XP_INST) {
            const int nExit = __popcll(__ballot(wantExit)), nEnter = __popcll(__ballot(wantEnter));
            TS_ADD(TS_WAIT_EXIT, nExit); TS_ADD(TS_WAIT_ENTER, nEnter);
            if (nExit | nEnter) {
                const bool nothingElse = __ballot(cur >= 0 || triLeft > 0) == 0;
                if (nExit > 0 && (nExit >= TR_INST_GROUP || nothingElse)) {
                    TS_ADD(TS_EXIT_STEPS, 1); TS_ADD(TS_EXIT_LANES, nExit);
                    if (NEST && wantExit && inInst2 >= 0) {
                        // out of the inner TransformedPrimitive, back to the ray of the instance around it: the world ray carried into that instance
                        // again (the entry step's operations on the entry step's inputs: the same bits), its tMax the saved one or the hit's
                        if (!ANYHIT && instHit) w2tMax = tMax;  // r.tMax = ray.tMax (primitive.cpp:84)
                        const bool fromQ1 = q1.regionCap > 0 && ray >= hitOffset1;
                        const float4 o4 = fromQ1 ? q1.o[ray - hitOffset1] : q0.o[ray], d4 = fromQ1 ? q1.d[ray - hitOffset1] : q0.d[ray];
                        float xf[PG_XF_STRIDE];
                        const PgInstance &in1 = sc.instances[inInst];
                        V3 o1, d1;
                        float dt1;
                        if (in1.animated) {
                            instance_matrices_at(in1, trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                            instance_ray(xf + 16, mk(o4.x, o4.y, o4.z), mk(d4.x, d4.y, d4.z), o1, d1, dt1);
                        } else instance_ray(in1.w2i, mk(o4.x, o4.y, o4.z), mk(d4.x, d4.y, d4.z), o1, d1, dt1);
                        ox = o1.x; oy = o1.y; oz = o1.z; tMax = w2tMax;
                        tr = tri_ray_setup(d1);
                        ix = 1 / d1.x; iy = 1 / d1.y; iz = 1 / d1.z;
                        nx = ix < 0; ny = iy < 0; nz = iz < 0; negBits = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                        if (ANYHIT) { vd = w2vd; vmask = w2vmask; }
                        triNext = (int)(w2Leaf >> (leafBits + 1)); triLeft = (int)(w2Leaf & ((2u << leafBits) - 1u)); inInst2 = -1; spBase = spBase1;
                        instHit = instHit1 || instHit;  // a hit inside is a hit of the instance around it
                        if (triLeft == 0) { TR_POP(); TR_SETTLE(); }
                    } else if (wantExit) {  // back to the world ray (primitive.cpp:83-88)
                        if (!ANYHIT && instHit) wtMax = tMax;  // r.tMax = ray.tMax
                        {
                            const bool fromQ1 = q1.regionCap > 0 && ray >= hitOffset1;
                            const float4 o4 = fromQ1 ? q1.o[ray - hitOffset1] : q0.o[ray], d4 = fromQ1 ? q1.d[ray - hitOffset1] : q0.d[ray];
                            ox = o4.x; oy = o4.y; oz = o4.z; tMax = wtMax;
                            tr = tri_ray_setup(mk(d4.x, d4.y, d4.z));
                            ix = 1 / d4.x; iy = 1 / d4.y; iz = 1 / d4.z;
                        }
                        nx = ix < 0; ny = iy < 0; nz = iz < 0; negBits = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                        if (ANYHIT) { vd = wvd; vmask = wvmask; }
                        triNext = (int)(wLeaf >> (leafBits + 1)); triLeft = (int)(wLeaf & ((2u << leafBits) - 1u)); inInst = -1; spBase = 0;
                        if (triLeft == 0) { TR_POP(); TR_SETTLE(); }
                    }
                    continue;
                }
                if (nEnter > 0 && (nEnter >= TR_INST_GROUP || nothingElse)) {
                    TS_ADD(TS_ENTER_STEPS, 1); TS_ADD(TS_ENTER_LANES, nEnter);
                    bool enteredBVH = false;
                    if (wantEnter) {
                        // TransformedPrimitive::Intersect[P]: carry the ray into the instance's space (Transform::operator()(Ray),
                        // transform.h:249-262) and start on its BVH
                        const bool fromQ1 = q1.regionCap > 0 && ray >= hitOffset1;
                        float4 d4 = fromQ1 ? q1.d[ray - hitOffset1] : q0.d[ray];
                        const int idx = -2 - cur;
                        // one 96-byte record per instance (DInstEntry): the matrix rows the ray needs, the object's root box and references
                        const DInstEntry *obp = sc.instEntry + idx;
                        const DInstEntry &ob = *obp;
                        const bool inner = NEST && inInst >= 0;  // a TransformedPrimitive met inside an instance: the ray to carry over is the instance's
                        if (inner) {
                            w2tMax = tMax;
                            if (ANYHIT) { w2vd = vd; w2vmask = vmask; vd = 0; vmask = 0; }
                            // its origin is in the registers; its direction is the world direction under the outer transform, once more
                            const PgInstance &in1 = sc.instances[inInst];
                            V3 d1;
                            if (in1.animated) {
                                float xf1[PG_XF_STRIDE];
                                instance_matrices_at(in1, trace_ray_time(sc, q0, q1, ray, hitOffset1), xf1);
                                d1 = m4_vec(xf1 + 16, mk(d4.x, d4.y, d4.z));
                            } else d1 = m4_vec(in1.w2i, mk(d4.x, d4.y, d4.z));
                            d4.x = d1.x; d4.y = d1.y; d4.z = d1.z;
                        } else {
                            wtMax = tMax;
                            if (ANYHIT) { wvd = vd; wvmask = vmask; vd = 0; vmask = 0; }
                        }
                        V3 oErr, o, dd;
                        if ((XP & XP_ANIM) && sc.instances[idx].animated) {  // PrimitiveToWorld.Interpolate(r.time, ...), primitive.cpp:78-80 / :99-101
                            float xf[PG_XF_STRIDE];
                            instance_matrices_at(sc.instances[idx], trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                            o = m4_point_err(xf + 16, mk(ox, oy, oz), oErr);
                            dd = m4_vec(xf + 16, mk(d4.x, d4.y, d4.z));
                        } else {
                            // Transform::operator()(Point3f, Vector3f *pError) with the homogeneous weight from row 3's constants (0, 0, 0, 1)
                            // where that is the row (the same four products and three sums), from the PgInstance otherwise
                            float wp;
                            if (ob.affineStill) wp = (0.f * ox + 0.f * oy) + (0.f * oz + 1.f);
                            else { const float *r3 = sc.instances[idx].w2i + 12; wp = (r3[0] * ox + r3[1] * oy) + (r3[2] * oz + r3[3]); }
                            // row by row (Transform::operator() of the point with its error, transform.h:277-300, and of the vector, :233-239: the
                            // same products and sums per component): a row is loaded when the previous one has been consumed, so four matrix
                            // entries are live at a time instead of twelve
                            float op[3], dp[3], ab[3];
                            const float *mp = obp->w2i;
                            for (int r = 0; r < 3; ++r) {
                                const float4 row = *(const float4 *)(mp + 4 * r);
                                dp[r] = row.x * d4.x + row.y * d4.y + row.z * d4.z;
                                op[r] = (row.x * ox + row.y * oy) + (row.z * oz + row.w);
                                ab[r] = (fabsf(row.x * ox) + fabsf(row.y * oy) + fabsf(row.z * oz) + fabsf(row.w));
                                asm volatile("" : "+v"(mp), "+v"(dp[r]), "+v"(op[r]), "+v"(ab[r]));  // (the next row's address waits for this row's results)
                            }
                            oErr = mk(ab[0], ab[1], ab[2]) * pgamma(3);
                            dd = mk(dp[0], dp[1], dp[2]);
                            if (wp == 1) o = mk(op[0], op[1], op[2]);
                            else { o.x = op[0] / wp; PG_SCHED_BARRIER(); o.y = op[1] / wp; PG_SCHED_BARRIER(); o.z = op[2] / wp; }
                        }
                        const float lengthSquared = lensq(dd);
                        if (lengthSquared > 0) {
                            const float dt = dot(vabs(dd), oErr) / lengthSquared;
                            o = o + dd * dt;
                            tMax -= dt;
                        }
                        ox = o.x; oy = o.y; oz = o.z;
                        tr = tri_ray_setup(dd);
                        ix = 1 / dd.x; iy = 1 / dd.y; iz = 1 / dd.z;
                        nx = ix < 0; ny = iy < 0; nz = iz < 0; negBits = (nx ? 1u : 0u) | (ny ? 2u : 0u) | (nz ? 4u : 0u);
                        if (inner) { inInst2 = idx; spBase1 = spBase; instHit1 = instHit; }
                        else inInst = idx;
                        spBase = sp; instHit = false;
                        triLeft = 0; cur = TR_NONE;
                        // the record's second half (root box, references) is read HERE: without the barrier the compiler issues all six loads of the
                        // record at the top of the step and the matrix and the box are live together -- the register peak of the whole kernel
                        asm volatile("" : "+v"(obp));
                        const DInstEntry &ob2 = *obp;
                        if (ob2.nNodes == 0) { triNext = ob2.firstPrim; triLeft = 1; }  // a lone primitive, no accelerator (api.cpp:1567)
                        else {
                            if (!WCNT) ++nodeVisits;  // the instance BVH's nodes[0]
                            enteredBVH = true;
                            float t0;
                            if (slab_interval(ob2.box[0], ob2.box[3], ob2.box[1], ob2.box[4], ob2.box[2], ob2.box[5], ox, oy, oz, ix, iy, iz, nx, ny, nz, t0) &&
                                t0 < tMax) {
                                cur = ob2.rootRef;
                                TR_SETTLE();
                            }
                        }
                    }
                    if (WCNT) wNodes += __popcll(__ballot(enteredBVH));
                    continue;
                }
            }
        }
        bool needPop = false, settle = false;
        const int nInt = __popcll(__ballot(cur >= 0));
        const int nTri = __popcll(__ballot(triLeft > 0));
        if (nTri > 0 && (nInt == 0 || nTri * 16 >= nInt * triW)) {
            TS_ADD(TS_TRI_STEPS, 1); TS_ADD(TS_TRI_LANES, nTri);
            bool notTri = false;  // an instance or a quadric: not a triangle test for the reference's counter
#ifdef TR_STATS
            if (XP & XP_INST) TS_ADD(TS_IN_INST_TRI, __popcll(__ballot(triLeft > 0 && inInst >= 0)));
            bool tsAlpha = false;
#endif
            if (triLeft > 0) {  // Triangle::Intersect[P] on the leaf's next primitive, in order (bvh.cpp:677-680)
                const int prim = triNext;
                const float4 a = sc.tris[PG_TRI_STRIDE * prim], b = sc.tris[PG_TRI_STRIDE * prim + 1], c = sc.tris[PG_TRI_STRIDE * prim + 2];
                if (!WCNT) ++triTests;
                ++triNext; --triLeft;
                const uint32_t pflags = __float_as_uint(a.w);
                if ((XP & XP_INST) && (pflags & PG_PRIM_INSTANCE)) {
                    // TransformedPrimitive::Intersect[P]: not a triangle test for the reference's counter; the lane waits for the
                    // wave's next entry step with the rest of the world leaf put aside
                    if (!WCNT) --triTests;
                    notTri = true;
                    if (NEST && inInst >= 0) w2Leaf = ((unsigned)triNext << (leafBits + 1)) | (unsigned)triLeft;
                    else wLeaf = ((unsigned)triNext << (leafBits + 1)) | (unsigned)triLeft;
                    triLeft = 0;
                    cur = -2 - __float_as_int(a.x);
                } else {
                    float t, b0, b1, b2;
                    bool hit;
                    if ((XP & XP_QUADRIC) && (pflags & PG_PRIM_SPHERE)) {
                        // Sphere::Intersect[P] (sphere.cpp:48-106): not a triangle test for the reference's counter; the ray's
                        // direction is not kept in registers (only its reciprocal and the triangle shear), so it is re-read
                        if (!WCNT) --triTests;
                        notTri = true;
                        const bool fromQ1 = q1.regionCap > 0 && ray >= hitOffset1;
                        const float4 d4 = fromQ1 ? q1.d[ray - hitOffset1] : q0.d[ray];
                        V3 dd = mk(d4.x, d4.y, d4.z);
                        if ((XP & XP_ANIM) && inInst >= 0 && sc.instances[inInst].animated) {
                            float xf[PG_XF_STRIDE];
                            instance_matrices_at(sc.instances[inInst], trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                            dd = m4_vec(xf + 16, dd);
                        } else if (inInst >= 0) dd = m4_vec(sc.instances[inInst].w2i, dd);
                        if (NEST && inInst2 >= 0) {
                            if (sc.instances[inInst2].animated) {
                                float xf[PG_XF_STRIDE];
                                instance_matrices_at(sc.instances[inInst2], trace_ray_time(sc, q0, q1, ray, hitOffset1), xf);
                                dd = m4_vec(xf + 16, dd);
                            } else dd = m4_vec(sc.instances[inInst2].w2i, dd);
                        }
                        hit = sphere_test(sc.spheres[__float_as_int(a.x)], mk(ox, oy, oz), dd, tMax, t);
                        b0 = t; b1 = 0; b2 = 0;  // the hit record of a sphere carries tHit
                    } else {
                        TriRay trz = tr;
                        if (NOSZ) trz.Sz = tr.kz == 0 ? ix : (tr.kz == 1 ? iy : iz);  // 1 / d[kz]: the same quotient as the slab test's reciprocal
                        hit = tri_test_pre(mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), mk(ox, oy, oz), trz, tMax, t, b0, b1, b2) &&
                              !(pflags & PG_TRI_BOGUS);
                    }
                    if ((XP & (XP_ALPHA | XP_ALPHATEX)) && hit && (pflags & PG_TRI_ALPHA)) {
                        // the mesh's alpha / shadow-alpha textures at the hit (triangle.cpp:333-338, :531-569): point, (u, v),
                        // no differentials; a value of exactly 0 rejects the hit and the ray goes on
#ifdef TR_STATS
                        tsAlpha = true;
#endif
                        float uv[6] = {0, 0, 1, 0, 1, 1};
                        if (sc.alphaUV && (pflags & PG_TRI_HAS_UV)) for (int k = 0; k < 6; ++k) uv[k] = sc.alphaUV[6 * prim + k];
                        const float hu = b0 * uv[0] + b1 * uv[2] + b2 * uv[4], hv = b0 * uv[1] + b1 * uv[3] + b2 * uv[5];
                        if (XP & XP_ALPHATEX) {
                            const PgAlphaMask &am = sc.alphas[sc.triAlpha[prim]];
                            TexHit th;
                            th.p = mk(a.x, a.y, a.z) * b0 + mk(b.x, b.y, b.z) * b1 + mk(c.x, c.y, c.z) * b2;
                            th.u = hu; th.v = hv;
                            th.dpdx = th.dpdy = mk(0, 0, 0);
                            th.dudx = th.dvdx = th.dudy = th.dvdy = 0;
                            if (am.has_alpha && TexEval<PG_TEX_DEPTH>::f(*sc.self, am.alpha, th) == 0) hit = false;
                            if (ANYHIT && hit && am.has_shadow_alpha && TexEval<PG_TEX_DEPTH>::f(*sc.self, am.shadow_alpha, th) == 0) hit = false;
                        } else {
                            const DAlphaTex *at = sc.alphaTex + 2 * sc.triAlpha[prim];
                            if (at[0].image != -2 && alpha_lookup(sc, at[0], hu, hv) == 0) hit = false;
                            if (ANYHIT && hit && at[1].image != -2 && alpha_lookup(sc, at[1], hu, hv) == 0) hit = false;
                        }
                    }
                    TR_ACCEPT(hit, prim, t, b0, b1, b2);
                    needPop = triLeft == 0 && !(ANYHIT && hitPrim >= 0);
                }
            }
            if (WCNT) wTris += nTri - (((XP & (XP_INST | XP_QUADRIC)) != 0) ? __popcll(__ballot(notTri)) : 0);
#ifdef TR_STATS
            { const int na = __popcll(__ballot(tsAlpha)); if (na) { TS_ADD(TS_ALPHA_STEPS, 1); TS_ADD(TS_ALPHA_LANES, na); } }
#endif
        } else {
          // (wave-uniform: every lane reaches this point)
          TS_ADD(TS_INT_STEPS, 1); TS_ADD(TS_INT_LANES, nInt);
          if (WCNT) wNodes += 2 * nInt;  // closest hit: near now, far when the reference pops it (it always does); free order: both boxes were read
          if (cur >= 0) {
#ifdef TR_STATS
            if (XP & XP_INST) TS_ADD(TS_IN_INST_INT, __popcll(__ballot(inInst >= 0)));
#endif
            const float4 *rec = sc.wnodes + 4 * (size_t)cur;
            const float4 bx = rec[0], by = rec[1], bz = rec[2];
            const float4 rf = rec[3];
            float t0, t1;
            const unsigned long long ok0 = slab_mask(bx.x, bx.y, by.x, by.y, bz.x, bz.y, ox, oy, oz, ix, iy, iz, nx, ny, nz, t0);
            const unsigned long long ok1 = slab_mask(bx.z, bx.w, by.z, by.w, bz.z, bz.w, ox, oy, oz, ix, iy, iz, nx, ny, nz, t1);
            const int axis = __float_as_int(rf.z);
            const bool neg = ((negBits >> axis) & 1u) != 0;  // bvh.cpp:686: near child first (dirIsNeg[axis])
            const int nearRef = __float_as_int(neg ? rf.y : rf.x), farRef = __float_as_int(neg ? rf.x : rf.y);
            const float farT = neg ? t0 : t1;
            // the four comparisons as lane masks, picked by `neg` with scalar mask arithmetic (instead of selecting the
            // operands per lane first)
            const float tMaxK = tMax * cullK;
            const unsigned long long mNeg = __ballot(neg);
            const unsigned long long mNear = (mNeg & ok1 & __ballot(t1 < tMax)) | (~mNeg & ok0 & __ballot(t0 < tMax));
            const unsigned long long mFar = (mNeg & ok0 & __ballot(t0 < tMaxK)) | (~mNeg & ok1 & __ballot(t1 < tMaxK));
            const bool nearHit = __builtin_amdgcn_inverse_ballot_w64(mNear);
            // Early cull of the far child.  ray.tMax is not monotone: Triangle::Intersect accepts tScaled <= tMax*det and then
            // returns t = tScaled*invDet, which can round to a few ulps ABOVE the old tMax (triangle.cpp:262-283), so an entry
            // that fails `tMin < tMax` now could still pass when the reference pops it.  Each accepted hit raises tMax by at
            // most three roundings, (1+2^-24)^3, so after <= TR_MAX_ACCEPTED accepted hits tMax < (1+2^-10) * any earlier
            // value: cull only beyond cullK = 1+2^-10 (any-hit: tMax is constant, cullK = 1); survivors are re-tested
            // exactly at pop time, and a ray that accepts more hits than that raises cullGuard so the host fails loudly.
            const bool farMaybe = __builtin_amdgcn_inverse_ballot_w64(mFar);
            if (!ANYHIT) {
                if (!WCNT) nodeVisits += 2;  // near now, far when the reference pops it (it always does)
                if (farMaybe) TR_PUSH(farRef, farT);
            } else if (!FREE) {
                nodeVisits += 1;
                if (farMaybe) TR_PUSH(farRef, farT); else vmask |= 1ull << vd;
                ++vd;
            }
            if (!FREE) {
                if (nearHit) cur = nearRef;
                needPop = !nearHit;
            } else {
                // free order: of the children whose box the ray crosses, the one it enters first; the other waits on the stack
                // (no entry distance needed: tMax is constant, a box that passed now passes at any later time)
                const bool h0 = __builtin_amdgcn_inverse_ballot_w64(ok0 & __ballot(t0 < tMax)), h1 = __builtin_amdgcn_inverse_ballot_w64(ok1 & __ballot(t1 < tMax));
                const int r0 = __float_as_int(rf.x), r1 = __float_as_int(rf.y);
                const bool first0 = h0 && (!h1 || t0 <= t1);
                if (h0 && h1) TR_PUSH(first0 ? r1 : r0, 0.f);
                if (!WCNT) nodeVisits += 2;  // both children's boxes were read and tested
                if (h0 || h1) cur = first0 ? r0 : r1;
                needPop = !(h0 || h1);
            }
            settle = true;
          }
        }
        // the reference's "pop or finish" and the unpacking of a leaf reference, once for both kinds of step
        if (needPop) { TR_POP(); settle = true; }
        if (settle) TR_SETTLE();
    }
#undef TR_ACCEPT
#undef TR_ACCEPT_CLOSEST
#undef TR_PUSH
#undef TR_TOP
#undef TR_POP
#undef TR_SETTLE
#ifdef TR_STATS
    if (lane == 0) for (int k = 0; k < TS_N; ++k) if (ts[k]) atomicAdd(&g_trStats[KIND][k], ts[k]);
#endif
    unsigned long long nv = tr_wave_sum(nodeVisits), nt = tr_wave_sum(triTests);
    if (WCNT) { nv = wNodes; nt = wTris; }
    if (lane == 0 && cn) {
        atomicAdd(&cn->node_visits, nv);
        atomicAdd(&cn->tri_tests, nt);
    }
}

// depth 11: 7 resident 256-thread blocks x 22.5 KB of stack fill the 160 KB LDS
TraceConfig default_trace_config() {
    TraceConfig tc = {TR_DEFAULT_DEPTH, 128, 16, 8, 1.0009765625f, 2048, TR_MAX_ACCEPTED, 32, 16, 1, /* instanced scenes: */ 8, 12, 16};
    if (const char *e = getenv("PG_TRACE_DEPTH")) { int v = atoi(e); if (v >= 0 && v <= 64) tc.depth = v; }
    if (const char *e = getenv("PG_TRACE_SEG")) { int v = atoi(e); if (v >= 64) tc.segRays = v; }
    if (const char *e = getenv("PG_TRACE_REFILL")) { int v = atoi(e); if (v >= 1 && v <= 64) tc.refillAt = tc.refillAtInst = v; }
    if (const char *e = getenv("PG_TRACE_GRID")) { int v = atoi(e); if (v >= 8) tc.gridBlocks = v; }
    if (const char *e = getenv("PG_TRACE_TRIW")) { int v = atoi(e); if (v >= 0) tc.triW = tc.triWInst = v; }
    if (const char *e = getenv("PG_TRACE_REFILL_ANY")) { int v = atoi(e); if (v >= 1 && v <= 64) tc.refillAtAny = tc.refillAtAnyInst = v; }
    if (const char *e = getenv("PG_TRACE_TRIW_ANY")) { int v = atoi(e); if (v >= 0) tc.triWAny = v; }
    if (const char *e = getenv("PG_TRACE_MAXACC")) { int v = atoi(e); if (v >= 1 && v <= 4096) tc.maxAccepted = v; }  // tests: provoke the exact fallback
    if (const char *e = getenv("PG_TRACE_CULLK")) { float v = (float)atof(e); if (v >= 1.f) tc.cullK = v < 3e38f ? v : 3e38f; }  // finite: 0*inf would be NaN
    // PG_ANYHIT_ORDER=reference: shadow rays visit the BVH in the reference's order, which reproduces its triangle-test statistic
    // (and this library's node-visit count) exactly; the default free order gives the same occlusion answers sooner
    if (const char *e = getenv("PG_ANYHIT_ORDER")) tc.anyhitFree = strcmp(e, "reference") != 0;
    const size_t ldsMax = 160 * 1024;  // one block must fit the CU's LDS in any case
    if (sizeof(uint2) * (size_t)tc.depth * TR_BLOCK > ldsMax) tc.depth = (int)(ldsMax / (sizeof(uint2) * TR_BLOCK));
    return tc;
}

template <int KIND>
static void launch_trace(const DScene &sc, const TraceConfig &c, RayQueue q0, RayQueue q1, float4 *hits, int hitOffset1, float *tOut, int *occluded,
                         TraceCounters *cn, int *cursors, int *cullGuard, hipStream_t s, const int *cursorInit = nullptr) {
    if (q0.regionCap <= 0) return;
    // persistent grid: enough blocks to fill 256 CUs (gridBlocks counts 256-thread blocks), never more than the queues can feed
    long long need = ((long long)(q0.regionCap + q1.regionCap) * PG_REGIONS + c.segRays - 1) / c.segRays;  // chunks
    int nblk = (int)std::min<long long>((need + TR_BLOCK / 64 - 1) / (TR_BLOCK / 64), (long long)c.gridBlocks * 256 / TR_BLOCK);
    nblk = ((nblk + 7) / 8) * 8;
    size_t lds = sizeof(uint2) * (size_t)c.depth * TR_BLOCK;
    // experiments only (profiles/r04g_*): unused LDS per block, to hold the kernel at fewer resident waves than its registers allow
    if (const char *e = getenv("PG_TRACE_LDS_PAD")) { const long v = atol(e); if (v > 0 && lds + (size_t)v <= 160 * 1024) lds += (size_t)v; }
    (void)hipMemsetAsync(cursors, 0, 2 * PG_REGIONS * PG_COUNT_STRIDE * sizeof(int), s);
    // a "tail" launch: q0's cursors start where its regions ended before the latest entries were appended
    if (cursorInit) (void)hipMemcpyAsync(cursors, cursorInit, PG_REGIONS * PG_COUNT_STRIDE * sizeof(int), hipMemcpyDeviceToDevice, s);
    int xp = (sc.nInstances > 0 ? XP_INST : 0) | (sc.nSpheres > 0 ? XP_QUADRIC : 0) | (sc.hasAlpha ? (sc.alphaTex ? XP_ALPHA : XP_ALPHATEX) : 0);
    // moving instances: two instantiations only -- with quadrics, and the general one (any alpha mask) -- a moving scene pays for both features
    if (sc.hasMotion) xp = XP_ANIM | (sc.hasAlpha ? XP_GENERAL : (XP_INST | XP_QUADRIC));
    if (sc.hasNest) xp = XP_NEST | XP_ANIM | XP_GENERAL;  // TransformedPrimitives inside object definitions: one instantiation with everything
    // scenes with instances wait for fewer idle lanes before a refill and weigh the triangle step higher (entry / exit steps take lanes out of
    // the two main steps: profiles/r06_trace_inst_ab.txt)
    const bool inst = (xp & XP_INST) != 0;
    const int refill = KIND ? (inst ? c.refillAtAnyInst : c.refillAtAny) : (inst ? c.refillAtInst : c.refillAt), triW = KIND ? c.triWAny : (inst ? c.triWInst : c.triW);
#define TR_LAUNCH(XPV) hipLaunchKernelGGL((k_trace<KIND, XPV>), dim3(nblk), dim3(TR_BLOCK), lds, s, sc, q0, q1, hits, hitOffset1, tOut, occluded, cn, \
                                          cursors, c.depth, c.segRays, refill, triW, KIND ? 1.f : c.cullK, cullGuard, c.maxAccepted)
    switch (xp) {
    case 0: TR_LAUNCH(0); break;
    case XP_INST: TR_LAUNCH(XP_INST); break;
    case XP_ALPHA: TR_LAUNCH(XP_ALPHA); break;
    case XP_INST | XP_ALPHA: TR_LAUNCH(XP_INST | XP_ALPHA); break;
    // quadrics without the general texture evaluator: a sphere light (BASELINE config 0) does not cost a triangle scene that evaluator's registers
    case XP_QUADRIC: TR_LAUNCH(XP_QUADRIC); break;
    case XP_QUADRIC | XP_INST: TR_LAUNCH(XP_QUADRIC | XP_INST); break;
    case XP_QUADRIC | XP_ALPHA: TR_LAUNCH(XP_QUADRIC | XP_ALPHA); break;
    case XP_QUADRIC | XP_INST | XP_ALPHA: TR_LAUNCH(XP_QUADRIC | XP_INST | XP_ALPHA); break;
    case XP_ANIM | XP_INST | XP_QUADRIC: TR_LAUNCH(XP_ANIM | XP_INST | XP_QUADRIC); break;
    case XP_ANIM | XP_GENERAL: TR_LAUNCH(XP_ANIM | XP_GENERAL); break;
    case XP_NEST | XP_ANIM | XP_GENERAL: TR_LAUNCH(XP_NEST | XP_ANIM | XP_GENERAL); break;
    default: TR_LAUNCH(XP_GENERAL); break;  // masks that need the general texture evaluator
    }
#undef TR_LAUNCH
}
static RayQueue noQueue() { RayQueue q; q.o = q.d = nullptr; q.counts = nullptr; q.regionCap = 0; return q; }
void launch_closest(const DScene &sc, const TraceConfig &c, RayQueue q, float4 *hits, float *tOut, TraceCounters *cn, int *cursors, int *cullGuard, hipStream_t s,
                    const int *cursorInit) {
    launch_trace<0>(sc, c, q, noQueue(), hits, 0, tOut, nullptr, cn, cursors, cullGuard, s, cursorInit);
}
void launch_closest2(const DScene &sc, const TraceConfig &c, RayQueue q0, RayQueue q1, float4 *hits, int hitOffset1, TraceCounters *cn, int *cursors, int *cullGuard,
                     hipStream_t s, float *tOut, const int *cursorInit) {
    launch_trace<0>(sc, c, q0, q1, hits, hitOffset1, tOut, nullptr, cn, cursors, cullGuard, s, cursorInit);
}
void launch_anyhit(const DScene &sc, const TraceConfig &c, RayQueue q, int *occluded, TraceCounters *cn, int *cursors, hipStream_t s) {
    if (c.anyhitFree) launch_trace<2>(sc, c, q, noQueue(), nullptr, 0, nullptr, occluded, cn, cursors, nullptr, s);
    else launch_trace<1>(sc, c, q, noQueue(), nullptr, 0, nullptr, occluded, cn, cursors, nullptr, s);
}
