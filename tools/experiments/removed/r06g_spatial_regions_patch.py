p='pbrt-v3_amd/csrc/pg_kernels.hip'
s=open(p).read()
a='''// The queue entry this thread consumes (or -1): block b walks region b & 7.'''
assert s.count(a)==1
s=s.replace(a,'''// Spatial regions (PG_SPATIAL_REGIONS=1; round 6's A/B of VERDICT r05 item 4): the NQ rays a thread pushes start at one point (a hit point); the
// REGION they are appended to is the point's cell (0 .. 7: four slabs along the world bound's longest axis x two along the next) instead of
// the producing block's XCD, so that each XCD's traversal waves -- which start on region XCD -- walk one part of the scene and its L2 holds
// that part's records.  Regions then fill unevenly: they have twice the room (regionCapFor's slack), and what does not fit in a cell's
// region goes, whole, to the next region that takes it.  One atomic instruction (NQ x 8 lanes) reserves all of a block's entries.
template <int NQ, int BLOCK>
PG_DEV void block_push_spatial(const RayQueue *q, const bool *pred, int *pos, int cell) {
    constexpr int NW = BLOCK / 64;
    __shared__ int s_n[NQ][8][NW];       // pushes per queue, cell, wave
    __shared__ int s_at[NQ][8][2];       // where a (queue, cell) group's entries go: [0] the first `fits` of them, [1] the rest
    __shared__ int s_fits[NQ][8];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long mine[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        mine[k] = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const unsigned long long m = __ballot(pred[k] && cell == c);
            if (lane == 0) s_n[k][c][wave] = __popcll(m);
            if (cell == c) mine[k] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x < NQ * 8) {
        const int k = threadIdx.x >> 3, c = threadIdx.x & 7;
        int n = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) n += s_n[k][c][w];
        int *const cnt = q[k].counts;
        const int cap = q[k].regionCap;
        int fits = n, at0 = 0, at1 = 0;
        if (n) {
            const int old = atomicAdd(&cnt[c * PG_COUNT_STRIDE], n);
            fits = min(n, max(0, cap - old));
            at0 = c * cap + old;
            const int excess = n - fits;
            if (excess) {  // give the part that does not fit back and find it a region that takes it whole
                atomicSub(&cnt[c * PG_COUNT_STRIDE], excess);
                for (int r = (c + 1) & 7;; r = (r + 1) & 7) {
                    const int o2 = atomicAdd(&cnt[r * PG_COUNT_STRIDE], excess);
                    if (o2 + excess <= cap) { at1 = r * cap + o2; break; }
                    atomicSub(&cnt[r * PG_COUNT_STRIDE], excess);
                }
            }
        }
        s_at[k][c][0] = at0; s_at[k][c][1] = at1; s_fits[k][c] = fits;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        pos[k] = -1;
        if (pred[k]) {
            int rank = __popcll(mine[k] & ((1ull << lane) - 1ull));
            for (int w = 0; w < wave; ++w) rank += s_n[k][cell][w];
            const int fits = s_fits[k][cell];
            pos[k] = rank < fits ? s_at[k][cell][0] + rank : s_at[k][cell][1] + (rank - fits);
        }
    }
}
// the cell of a point (DScene::cell*)
PG_DEV int spatial_cell(const DScene &sc, float4 o) {
    const float a = sc.cellAxisA == 0 ? o.x : (sc.cellAxisA == 1 ? o.y : o.z), b = sc.cellAxisB == 0 ? o.x : (sc.cellAxisB == 1 ? o.y : o.z);
    const int ia = (int)pmin(3.f, pmax(0.f, (a - sc.cellA0) * sc.cellAInv)), ib = (int)pmin(1.f, pmax(0.f, (b - sc.cellB0) * sc.cellBInv));
    return ia | (ib << 2);
}
'''+a)
a='''    block_push<3, true, PG_SHADE_BLOCK>(outQ, outPred, outPos, nextBin);
    const int posNext = outPos[0], posShadow = outPos[1], posMis = outPos[2];
    if (pushNext) { qnext.o[posNext]'''
assert s.count(a)==2
i=s.index(a)
s=s[:i]+'''    if (sc.spatialRegions) block_push_spatial<3, PG_SHADE_BLOCK>(outQ, outPred, outPos, spatial_cell(sc, pushNext ? s_ray[0][0][tid] : (pushShadow ? s_ray[1][0][tid] : s_ray[2][0][tid])));
    else block_push<3, true, PG_SHADE_BLOCK>(outQ, outPred, outPos, nextBin);
    const int posNext = outPos[0], posShadow = outPos[1], posMis = outPos[2];
    if (pushNext) { qnext.o[posNext]'''+s[i+len(a):]
open(p,'w').write(s)
p='pbrt-v3_amd/csrc/pg_kernels.h'
s=open(p).read()
a='    float rootBox[6];  // nodes[0].bounds: (min.xyz, max.xyz)\n'
assert a in s
s=s.replace(a,a+'''    // PG_SPATIAL_REGIONS (experiment, round 6): the shading kernel appends a vertex's rays to the region of the hit point's cell -- 4 slabs along the
    // world bound's longest axis (cellAxisA) x 2 along the next (cellAxisB) -- instead of the producing block's XCD (block_push_spatial)
    int spatialRegions, cellAxisA, cellAxisB;
    float cellA0, cellAInv, cellB0, cellBInv;
''')
open(p,'w').write(s)
p='pbrt-v3_amd/csrc/pg_abi.hip'
s=open(p).read()
s=s.replace('regionCapFor(capacity, s->d.sparseLights != 0)','regionCapFor(capacity, s->d.sparseLights != 0 || s->d.spatialRegions != 0)')
s=s.replace('regionCapFor(rp.capacity, s->d.sparseLights != 0)','regionCapFor(rp.capacity, s->d.sparseLights != 0 || s->d.spatialRegions != 0)')
a='''        d.leafBits = leafBits;
'''
assert s.count(a)==1
s=s.replace(a,a+'''        {   // PG_SPATIAL_REGIONS=1: see DScene::spatialRegions
            const char *e = getenv("PG_SPATIAL_REGIONS");
            d.spatialRegions = (e && atoi(e) != 0 && nn > 0) ? 1 : 0;
            float ext[3] = {0, 0, 0};
            if (nn > 0) for (int c = 0; c < 3; ++c) ext[c] = desc->nodes[0].bmax[c] - desc->nodes[0].bmin[c];
            int a = 0; for (int c = 1; c < 3; ++c) if (ext[c] > ext[a]) a = c;
            int b = a == 0 ? 1 : 0; for (int c = 0; c < 3; ++c) if (c != a && ext[c] > ext[b]) b = c;
            d.cellAxisA = a; d.cellAxisB = b;
            d.cellA0 = nn > 0 ? desc->nodes[0].bmin[a] : 0.f; d.cellAInv = ext[a] > 0 ? 4.f / ext[a] : 0.f;
            d.cellB0 = nn > 0 ? desc->nodes[0].bmin[b] : 0.f; d.cellBInv = ext[b] > 0 ? 2.f / ext[b] : 0.f;
        }
''')
open(p,'w').write(s)
