// SOURCE-BOX variant of the fused F1 pass (3-D) for transforms far from the identity - an experiment of round 3, measured and SHELVED
// (numbers at the end of this comment and in DESIGN.md 8.2; profiles/r03d_source_box_experiment.txt).  Included by tools/sbench.hip inside namespace trx.
//
// The tile kernels cut the OUTPUT volume into tiles and stage the bounding box of each tile's pre-image: for a general rotation that
// box holds 5-7 floats per output voxel (a rotated tile fills a quarter of its bounding box, and every short box row touches two L2
// lines), and staging it is what bounds the pass (614 us per 8 x 256^3 launch, of which the staging alone 613).
// Here the roles are swapped: a block owns an axis-aligned box of the SOURCE (moving) volume - 64 x 16 x 15 interpolation cells, staged
// once, in the volume's own row order (68 x 17 x 16 floats: 1.2 floats per cell, three full L2 lines per row) - and processes every
// output voxel whose sample point falls into one of its cells, wherever that voxel lies in the output volume:
//   * cell(o) = floor(sample point of o), clamped to [-1, S-1] per axis, decides the owner; the boxes tile the cell lattice, so every
//     output voxel has exactly one owner; the test uses the very coordinates the gather uses, so the partition is exact in fp32.
//     Voxels whose sample lies outside the padded volume belong to the nearest border box and contribute w = 0 (zeros padding);
//   * the owned voxels form a parallelepiped in the output volume.  A wave takes one output plane oz of it at a time: lane = row oy, the
//     row's ox interval comes from clipping the line against the box's three slabs (enlarged by 0.02 voxels: a superset - the exact
//     ownership test discards the few extra candidates); a prefix sum turns the <= 64 intervals into a list of candidates, and the wave
//     works through the list 64 candidates at a time.  Candidate -> (row, ox): a bit mask of row starts in LDS (one ds_or per row), then
//     per chunk one 64-bit read, v_mbcnt and one ds_bpermute;
//   * per candidate: coordinates, 4 ds_read2_b32 from the box, trilinear value + gradient, one target load (consecutive candidates of a
//     row are consecutive addresses), and all 41 sums in registers - (x, y, z) change from candidate to candidate, so nothing folds.
// Cost model: ~100 vector instructions per output voxel-wave whatever the transform (the z-streaming body next to the identity: 51), no
// dependence on the angle.
// Measured (8 x 256^3, MI355X, tools/sbench.hip): 780 us per launch at EVERY pose (identity, R(.5,.4,.3), R(.7,.8,.6)) against 370 / 617 / 685 us
// for the tile kernels: the sums agree with theirs (moments to 1e-8, gradient sums to the rounding of the different summation order), the
// partition is exact (sum(y) is the tile kernels' to 1e-8), but the pass executes 105 vector instructions per 64 candidates (fetch +
// mapping 18, coordinates / cells / ownership 30, interpolation 14, the 41 unfolded sums 28, ...): enumeration and fetch alone 210 us,
// + arithmetic 365 us (issue-bound: 4 waves per SIMD), + target loads and box reads 205 us.  It would have to halve to beat the rotated
// tile kernels, whose per-voxel arithmetic is 60 instructions because a thread keeps (x, z) fixed - the staging they pay for that is the
// smaller price.  Not offered by the launcher.

#ifndef TRX_SB_DBG
#define TRX_SB_DBG 0   // development ablation (tools/sbench.hip): bits: 1 = no box DMA, 2 = no target loads, 4 = no box reads, 8 = enumeration only
#endif

struct SBox {
    static constexpr int CX = 64, CY = 16, CZ = 15;            // interpolation cells per box
    static constexpr int BW = 68, BH = 17, BD = 16, BW4 = 17;  // staged floats: cells + 1, x rounded up to float4
    static constexpr int Threads = 512, Waves = 8;
    static constexpr int PlaneSlots = BW4 * BH, Slots = PlaneSlots * BD;   // float4 slots
    static constexpr int Pieces = (Slots + 63) / 64;           // LDS-DMA instructions per box (one per wave and 64 slots)
    static constexpr int BoxFloats = BW * BH * BD;
    static constexpr int XTab = 512;                           // base x coordinates (W <= 512)
    static constexpr int MaskWords = 128;                      // per wave: row-start bits of up to 4096 candidates
    static constexpr int MaxCand = MaskWords * 32;
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static constexpr int Alloc = BoxFloats + XTab + Waves * MaskWords + 4;   // floats: 80 144 B, two blocks per CU (+ the words a fetch past the last chunk reads)
    static_assert(ReduceScratch <= BoxFloats && Alloc * 4 * 2 <= 160 * 1024, "LDS budget");
};

struct SGeom {
    int nbx, nby, nbz;        // boxes per axis
    int c0x, c0y, c0z;        // first cell of box 0 (<= -1; c0x % 4 == 0: a box's first cell is also the float4-aligned origin of its window)
    int zgroup, nzg;          // boxes per work item along z, groups per column
    int items_per_pair;
};

#ifndef TRX_SB_ZGROUP
#define TRX_SB_ZGROUP 4
#endif

static SGeom sb_geom(const trx_volumes &v)
{
    using C = SBox;
    SGeom g;
    // cells -1 .. S-1 per axis
    g.nbx = (v.W + 1 + 3 + C::CX - 1) / C::CX;
    const int slackx = g.nbx * C::CX - (v.W + 1);
    g.c0x = -4 * ((slackx / 2 + 4) / 4);
    g.nby = (v.H + 1 + C::CY - 1) / C::CY;
    g.c0y = -1 - (g.nby * C::CY - (v.H + 1)) / 2;
    g.nbz = (v.D + 1 + C::CZ - 1) / C::CZ;
    g.c0z = -1 - (g.nbz * C::CZ - (v.D + 1)) / 2;
    g.zgroup = TRX_SB_ZGROUP;
    g.nzg = (g.nbz + g.zgroup - 1) / g.zgroup;
    g.items_per_pair = g.nbx * g.nby * g.nzg;
    return g;
}

static bool sb_shape_ok(const trx_volumes &v)
{
    if (v.ndim != 3 || v.W % 4 || v.W > SBox::XTab || v.H > 1023 || v.D < 2 || v.H < 2 || v.W < 8) return false;
    return (size_t)v.D * v.H * v.W < ((size_t)1 << 29);   // 32-bit byte offsets inside one volume
}

// Is theta a map this body handles?  Finite, not wildly scaled (the enumeration inverts the voxel-space matrix in fp32).
__device__ __forceinline__ bool sb_theta_ok(const float *__restrict__ th, float fD, float fH, float fW)
{
    float mx = 0.f;
#pragma unroll
    for (int i = 0; i < 12; i++) mx = fmaxf(mx, fabsf(th[i]));
    if (!(mx < 8.0f)) return false;   // (NaN compares false)
    const float a = th[0], b = th[1], c = th[2], d = th[4], e = th[5], f = th[6], g = th[8], h = th[9], k = th[10];
    const float det = a * (e * k - f * h) - b * (d * k - f * g) + c * (d * h - e * g);
    return fabsf(det) > 0.2f && fabsf(det) < 5.0f;
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <int MODE>
__device__ __forceinline__ void srcbox_body(const trx_volumes &vol, const float *__restrict__ theta, const SGeom &sg, float *__restrict__ partials,
                                            float *lds, int bx, int by, int rows_per_pair, int wave_in)
{
    using C = SBox;
    constexpr bool kGrad = (MODE == 0 || MODE == 4);
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 4 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? 41 : (MODE == 1 ? 5 : 13);
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int lane = trx_lane_id(), tid = wave * 64 + lane;
    const int D = vol.D, H = vol.H, W = vol.W;
    const float fD = (float)D, fH = (float)H, fW = (float)W;
    const float *__restrict__ th = theta + (size_t)by * TRX_PSTRIDE;
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)by * vol.moving_stride);
    const float *__restrict__ tgt = vol.target + (size_t)by * vol.target_stride;
    const float *__restrict__ ytab = vol.yn, *__restrict__ ztab = vol.zn;
    float *box = lds, *xtab = lds + C::BoxFloats;
    unsigned *wmask = reinterpret_cast<unsigned *>(lds + C::BoxFloats + C::XTab) + wave * C::MaskWords;
    typedef const __attribute__((address_space(3))) f2u *lds_f2;

    int t = bx;
    const int ibx = t % sg.nbx; t /= sg.nbx;
    const int iby = t % sg.nby;
    const int izg = t / sg.nby;

    float T[12];
#pragma unroll
    for (int i = 0; i < 12; i++) T[i] = uni(th[i]);
    // the map in voxel units, p = M o + t0 (o = output index, p = sample point in source index units), and its inverse: only the
    // enumeration uses them (the samples themselves follow the normalised-coordinate arithmetic of the other kernels)
    const float S[3] = {fW, fH, fD};
    float M[3][3], t0[3], N[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int j = 0; j < 3; j++) M[i][j] = T[i * 4 + j] * S[i] / S[j];
        t0[i] = 0.5f * S[i] * (T[i * 4 + 0] * (1.0f / fW - 1.0f) + T[i * 4 + 1] * (1.0f / fH - 1.0f) + T[i * 4 + 2] * (1.0f / fD - 1.0f) + T[i * 4 + 3] + 1.0f) - 0.5f;
    }
    {
        const float c00 = M[1][1] * M[2][2] - M[1][2] * M[2][1], c01 = M[1][2] * M[2][0] - M[1][0] * M[2][2], c02 = M[1][0] * M[2][1] - M[1][1] * M[2][0];
        const float idet = 1.0f / (M[0][0] * c00 + M[0][1] * c01 + M[0][2] * c02);
        N[0][0] = c00 * idet; N[1][0] = c01 * idet; N[2][0] = c02 * idet;
        N[0][1] = (M[0][2] * M[2][1] - M[0][1] * M[2][2]) * idet; N[1][1] = (M[0][0] * M[2][2] - M[0][2] * M[2][0]) * idet; N[2][1] = (M[0][1] * M[2][0] - M[0][0] * M[2][1]) * idet;
        N[0][2] = (M[0][1] * M[1][2] - M[0][2] * M[1][1]) * idet; N[1][2] = (M[0][2] * M[1][0] - M[0][0] * M[1][2]) * idet; N[2][2] = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) * idet;
    }
    float inv0[3];
    bool flat0[3];   // |d p_i / d ox| too small to clip a row with
#pragma unroll
    for (int i = 0; i < 3; i++) {
        flat0[i] = fabsf(M[i][0]) < 1.0e-6f;
        inv0[i] = flat0[i] ? 0.f : 1.0f / M[i][0];
    }

    for (int i = tid; i < W; i += C::Threads) xtab[i] = vol.xn[i];

    // running sums: moments, and per (q, c) the pairs (g, g yn), (g xn, g zn) of sum(q g_c)
    f2 M01 = (f2)(0.f), M23 = (f2)(0.f);
    float M4 = 0.f;
    f2 GA[NQ > 0 ? NQ : 1][3], GB[NQ > 0 ? NQ : 1][3];
#pragma unroll
    for (int q = 0; q < (NQ > 0 ? NQ : 1); q++)
#pragma unroll
        for (int c = 0; c < 3; c++) GA[q][c] = GB[q][c] = (f2)(0.f);
    bool ok = true;

    const unsigned box_lds = (unsigned)(uintptr_t)box;
    const float kEps = 0.02f;
    // the box being processed: window origin (= first lattice cell) and the owned local cells [llo, lhi) per axis
    int bx0 = 0, by0 = 0, bz0 = 0, llo0 = 0, lhi0 = 0, llo1 = 0, lhi1 = 0, llo2 = 0, lhi2 = 0;

    // ---- one list of candidates: lane = row with the interval [xlo, xlo + len) of ox, the row's yn / zn and the offset of its first voxel
    // in the target.  kBox: the rows of one plane against the staged box (zn uniform); otherwise rows of the outside pass (exact test
    // only, y and y^2 of the voxels no box owns).
    auto run_list = [&](auto box_tag, int len, int xlo, float yn_l, float zn_l, unsigned rowoff_l) {
        constexpr bool kBox = decltype(box_tag)::value;
        int incl = len;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        if (total == 0) return;
        const int st = incl - len;
        // row-start bits of the candidate list
        wmask[lane] = 0u; wmask[lane + 64] = 0u;
        if (len > 0) atomicOr(&wmask[st >> 5], 1u << (st & 31));
        // non-empty rows, compacted to the low lanes: (row lane, first ox, first candidate)
        const unsigned long long ne = __builtin_amdgcn_ballot_w64(len > 0);
        const int below = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(ne >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)ne, 0u));
        const int nne = __builtin_popcountll(ne);
        const int kpos = len > 0 ? below : nne + (lane - below);
        const int info = (lane << 22) | (xlo << 12) | st;
        const int infoC = __builtin_amdgcn_ds_permute(kpos << 2, len > 0 ? info : 0);
        const float znu = kBox ? uni(zn_l) : 0.f;
        int nb = 0;   // row starts before the chunk being fetched
        // candidate of this lane in the chunk that starts at c: its row, ox, and the loads that do not depend on the box (issued one
        // chunk ahead: the target load is a trip to L2 / HBM and a wave has only three others to hide behind)
        struct Cand { float yn, zn, xn, yv; bool act; };
        auto fetch = [&](int c) -> Cand {
            const unsigned long long Hm = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)wmask[(c >> 5) + 1]) << 32) |
                                          (unsigned)__builtin_amdgcn_readfirstlane((int)wmask[c >> 5]);
            const unsigned long long Hs = Hm >> 1;
            const int kbase = nb + (int)(Hm & 1ull) - 1;
            const int kk = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(Hs >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)Hs, (unsigned)kbase));
            nb += __builtin_popcountll(Hm);
            const int q = c + lane;
            const int inf = __builtin_amdgcn_ds_bpermute(kk << 2, infoC);
            const int rl4 = (inf >> 20) & 252;
            Cand k;
            k.act = q < total;
            const int ox = k.act ? ((inf >> 12) & 1023) + (q - (inf & 4095)) : 0;
            k.yn = __int_as_float(__builtin_amdgcn_ds_bpermute(rl4, __float_as_int(yn_l)));
            k.zn = kBox ? znu : __int_as_float(__builtin_amdgcn_ds_bpermute(rl4, __float_as_int(zn_l)));
            const unsigned off = (unsigned)__builtin_amdgcn_ds_bpermute(rl4, (int)rowoff_l) + (unsigned)ox;
            k.xn = xtab[ox];
            k.yv = 0.f;
            if (!(TRX_SB_DBG & 2)) { if (k.act) k.yv = tgt[off]; }
            return k;
        };
        auto process = [&](const Cand &k) {
            if (!k.act) return;
            const float xn = k.xn, yn = k.yn, zn = k.zn;
            const float ix = unnorm<3>(fmaf(T[0], xn, fmaf(T[1], yn, fmaf(T[2], zn, T[3]))), fW);
            const float iy = unnorm<3>(fmaf(T[4], xn, fmaf(T[5], yn, fmaf(T[6], zn, T[7]))), fH);
            const float iz = unnorm<3>(fmaf(T[8], xn, fmaf(T[9], yn, fmaf(T[10], zn, T[11]))), fD);
            const int cxi = floor_to_int(ix), cyi = floor_to_int(iy), czi = floor_to_int(iz);
            const float yv = k.yv;
            if constexpr (!kBox) {
                const bool inside = ((unsigned)(cxi + 1) <= (unsigned)W) & ((unsigned)(cyi + 1) <= (unsigned)H) & ((unsigned)(czi + 1) <= (unsigned)D);
                if (!inside) {
                    if constexpr (MODE == 4) {
                        M4 = fmaf(yv, yv, M4);
                    } else {
                        M01.x += yv;
                        M23.x = fmaf(yv, yv, M23.x);
                    }
                }
            } else {
                const unsigned lx = (unsigned)(cxi - bx0), ly = (unsigned)(cyi - by0), lz = (unsigned)(czi - bz0);
                if ((lx - (unsigned)llo0 < (unsigned)(lhi0 - llo0)) & (ly - (unsigned)llo1 < (unsigned)(lhi1 - llo1)) & (lz - (unsigned)llo2 < (unsigned)(lhi2 - llo2))) {
                    const unsigned a0 = box_lds + ((lz * C::BH + ly) * C::BW + lx) * 4u, a1 = a0 + C::BH * C::BW * 4u;
                    f2 r00, r01, r10, r11;
                    if (TRX_SB_DBG & 4) {
                        r00 = r01 = r10 = r11 = (f2){xn, yn};
                    } else {
                        r00 = *(lds_f2)a0; r01 = *(lds_f2)(a0 + C::BW * 4u); r10 = *(lds_f2)a1; r11 = *(lds_f2)(a1 + C::BW * 4u);
                    }
                    const Samp3 sm = lerp3_pairs<kGrad>(r00, r01, r10, r11, __builtin_amdgcn_fractf(ix), __builtin_amdgcn_fractf(iy), __builtin_amdgcn_fractf(iz));
                    if constexpr (MODE == 4) {
                        const float dd = sm.v - yv;
                        M4 = fmaf(dd, dd, M4);
                        const float gq[3] = {sm.dx, sm.dy, sm.dz};
#pragma unroll
                        for (int cc = 0; cc < 3; cc++) {
                            const f2 ga = {gq[cc], yn * gq[cc]}, gb = {xn * gq[cc], zn * gq[cc]};
                            GA[0][cc] = ga * dd + GA[0][cc];
                            GB[0][cc] = gb * dd + GB[0][cc];
                        }
                    } else {
                        const f2 yw = {yv, sm.v};
                        M01 += yw;
                        M23 = yw * yw + M23;
                        M4 = fmaf(yv, sm.v, M4);
                        if constexpr (MODE == 0) {
                            const float gq[3] = {sm.dx, sm.dy, sm.dz};
#pragma unroll
                            for (int cc = 0; cc < 3; cc++) {
                                const f2 ga = {gq[cc], yn * gq[cc]}, gb = {xn * gq[cc], zn * gq[cc]};
                                GA[0][cc] += ga; GB[0][cc] += gb;
                                GA[1][cc] = ga * yv + GA[1][cc]; GB[1][cc] = gb * yv + GB[1][cc];
                                GA[2][cc] = ga * sm.v + GA[2][cc]; GB[2][cc] = gb * sm.v + GB[2][cc];
                            }
                        }
                    }
                }
            }
        };
        Cand cur = fetch(0);
        for (int c = 0; c < total; c += 64) {
            const Cand nxt = fetch(c + 64);   // (past the end: no active lane, no loads)
            if (!(TRX_SB_DBG & 8)) process(cur);
            cur = nxt;
        }
    };
    // the ox interval of row (oy, oz) whose sample points have p_i in [qlo_i, qhi_i] for every axis: [ceil(lo), floor(hi)], empty when lo > hi
    auto clip_row = [&](float foy, float foz, const float (&qlo)[3], const float (&qhi)[3], int &xa, int &xb) {
        float lo = 0.f, hi = fW - 1.0f;
        bool none = false;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const float r = fmaf(M[i][1], foy, fmaf(M[i][2], foz, t0[i]));
            if (flat0[i]) {
                none = none || !(r >= qlo[i] - 0.01f && r <= qhi[i] + 0.01f);
            } else {
                const float a = (qlo[i] - r) * inv0[i], b = (qhi[i] - r) * inv0[i];
                lo = fmaxf(lo, fminf(a, b)); hi = fminf(hi, fmaxf(a, b));
            }
        }
        xa = (int)ceilf(lo); xb = (int)floorf(hi);
        if (none || xa > xb) { xa = W; xb = W - 1; }
    };

    for (int zb = 0; zb < sg.zgroup; zb++) {
        const int ibz = izg * sg.zgroup + zb;
        if (ibz >= sg.nbz) break;
        bx0 = sg.c0x + ibx * C::CX; by0 = sg.c0y + iby * C::CY; bz0 = sg.c0z + ibz * C::CZ;
        // the cells this box owns: its lattice cells inside [-1, S-1] (a sample point beyond the padded volume has no owner here: the
        // outside pass below counts it), as local indices relative to the window origin, and as the p-range of the enumeration
        llo0 = max(0, -1 - bx0); lhi0 = min(C::CX, W - bx0); llo1 = max(0, -1 - by0); lhi1 = min(C::CY, H - by0); llo2 = max(0, -1 - bz0); lhi2 = min(C::CZ, D - bz0);
        float plo[3], phi[3];
        plo[0] = (float)(bx0 + llo0) - kEps; phi[0] = (float)(bx0 + lhi0) + kEps;
        plo[1] = (float)(by0 + llo1) - kEps; phi[1] = (float)(by0 + lhi1) + kEps;
        plo[2] = (float)(bz0 + llo2) - kEps; phi[2] = (float)(bz0 + lhi2) + kEps;
        const bool box_empty = (lhi0 <= llo0) || (lhi1 <= llo1) || (lhi2 <= llo2);
        // bounding box of the owned voxels in the output volume: o_j = sum_i N_ji (p_i - t0_i)
        float olo[3], ohi[3];
#pragma unroll
        for (int j = 0; j < 3; j++) {
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int i = 0; i < 3; i++) {
                const float a = N[j][i] * (plo[i] - t0[i]), b = N[j][i] * (phi[i] - t0[i]);
                lo += fminf(a, b); hi += fmaxf(a, b);
            }
            olo[j] = lo; ohi[j] = hi;
        }
        const int exlo = (int)fminf(fmaxf(floorf(olo[0]) - 1.0f, 0.f), fW - 1.0f), exhi = (int)fmaxf(fminf(ceilf(ohi[0]) + 1.0f, fW - 1.0f), -1.0f);
        const int oylo = (int)fminf(fmaxf(floorf(olo[1]) - 1.0f, 0.f), fH), oyhi = (int)fmaxf(fminf(ceilf(ohi[1]) + 1.0f, fH - 1.0f), -1.0f);
        const int ozlo = (int)fminf(fmaxf(floorf(olo[2]) - 1.0f, 0.f), fD), ozhi = (int)fmaxf(fminf(ceilf(ohi[2]) + 1.0f, fD - 1.0f), -1.0f);
        int maxlen = exhi - exlo + 1 + 4;
        maxlen = maxlen > W ? W : maxlen;
        const int rpi = maxlen <= 64 ? 64 : (maxlen <= 128 ? 32 : (maxlen <= 256 ? 16 : 8));   // rows per wave item: rpi * maxlen <= 4096 candidates
        const int lencap = C::MaxCand / rpi;
        const int nrows = oyhi - oylo + 1, nplanes = ozhi - ozlo + 1;
        const int nchunks = nrows > 0 ? (nrows + rpi - 1) / rpi : 0;
        const int nitems = (nplanes > 0 && !box_empty) ? nplanes * nchunks : 0;

        __syncthreads();   // every wave is done with the previous box (and with xtab's fill the first time)
        if (nitems > 0) {
            // ---- stage the window [bx0, bx0 + 68) x [by0, by0 + 17) x [bz0, bz0 + 16): LDS-DMA for float4 slots inside the volume, zeros elsewhere
            for (int k = wave; k < C::Pieces; k += C::Waves) {
                const int slot = k * 64 + lane;
                const bool valid = slot < C::Slots;
                const int sz = slot / C::PlaneSlots, r = slot - sz * C::PlaneSlots;
                const int sy = r / C::BW4, sx4 = r - sy * C::BW4;
                const int gz = bz0 + sz, gy = by0 + sy, gx = bx0 + 4 * sx4;
                const bool inb = valid && ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H) && (gx >= 0) && (gx + 4 <= W);
                const unsigned off = inb ? (unsigned)(((gz * H + gy) * W + gx) * 4) : 0u;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(inb);
                if (mk != 0ull && !(TRX_SB_DBG & 1)) {
                    unsigned long long sv;
                    unsigned m0s;
                    const unsigned dst = box_lds + (unsigned)k * 1024u;
                    asm volatile("s_mov_b64 %[sv], exec\n\t"
                                 "s_mov_b32 %[m0s], m0\n\t"
                                 "s_mov_b32 m0, %[l0]\n\t"
                                 "s_mov_b64 exec, %[k0]\n\t"
                                 "global_load_lds_dwordx4 %[o0], %[b0]\n\t"
                                 "s_mov_b64 exec, %[sv]\n\t"
                                 "s_mov_b32 m0, %[m0s]"
                                 : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                                 : [l0] "s"(__builtin_amdgcn_readfirstlane(dst)), [b0] "s"(mov), [o0] "v"(off), [k0] "s"(mk)
                                 : "memory");
                }
                if (valid && !inb) *reinterpret_cast<float4 *>(box + slot * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();

        // ---- the owned output voxels, one (plane oz, group of rpi rows) per wave at a time
        for (int it = wave; it < nitems; it += C::Waves) {
            const int pz = it / nchunks, ch = it - pz * nchunks;
            const int oz = ozlo + pz, oyb = oylo + ch * rpi;
            const int oy = oyb + lane;
            const bool rowok = (lane < rpi) && (oy <= oyhi);
            int xa, xb;
            clip_row((float)oy, (float)oz, plo, phi, xa, xb);
            int len = rowok ? xb - xa + 1 : 0;
            if (len > lencap) { len = lencap; ok = false; }
            const int oyc = oy < H ? oy : H - 1;
            run_list(std::true_type{}, len, xa, ytab[oyc], ztab[oz], (unsigned)((oz * H + oyc) * W));
        }
    }

    // ---- voxels whose sample point lies beyond the padded volume (a cell < -1 or > S-1 on some axis): w = 0 and no gradient, they only
    // count in sum(y), sum(y^2).  The items of a pair share the output rows (oy, oz) evenly; per row the ox interval that is certainly
    // inside comes from the same slab clipping, the rest of the row - two intervals - goes through the same candidate lists with the
    // exact test: the negation of the boxes' ownership test, on the same coordinates.
    {
        const int nrows_all = D * H;
        const int per = (nrows_all + sg.items_per_pair - 1) / sg.items_per_pair;
        const int r0 = bx * per, r1 = min(r0 + per, nrows_all);
        const int rpo = W <= 64 ? 64 : (W <= 128 ? 32 : (W <= 256 ? 16 : 8));   // rows per list: rpo * W <= 4096 candidates
        float qlo[3], qhi[3];
#pragma unroll
        for (int i = 0; i < 3; i++) { qlo[i] = -1.0f + kEps; qhi[i] = S[i] - kEps; }
        const int nlists = (r1 - r0 + rpo - 1) / rpo;
        for (int it = wave; it < 2 * nlists; it += C::Waves) {
            const int side = it & 1, rb = r0 + (it >> 1) * rpo;
            const int r = rb + lane;
            const bool rowok = (lane < rpo) && (r < r1);
            const int rc = r < nrows_all ? r : nrows_all - 1;
            const int roz = rc / H, roy = rc - roz * H;
            int xa, xb;   // certainly inside: [xa, xb]
            clip_row((float)roy, (float)roz, qlo, qhi, xa, xb);
            const int lo = side ? xb + 1 : 0, hi = side ? W - 1 : min(xa, W) - 1;
            const int len = rowok ? max(hi - lo + 1, 0) : 0;
            run_list(std::false_type{}, len, lo, ytab[roy], ztab[roz], (unsigned)(rc * W));
        }
    }

    __syncthreads();   // the box becomes the reduction scratch

    if (__syncthreads_or(!ok)) {
        if (tid < NP) partials[((size_t)by * rows_per_pair + bx) * NP + tid] = __builtin_nanf("");
        return;
    }
    float vals[NP];
    int o = 0;
    if constexpr (MODE == 4) {
        vals[0] = M4;
        o = 1;
    } else {
        vals[0] = M01.x; vals[1] = M01.y; vals[2] = M23.x; vals[3] = M23.y; vals[4] = M4;
        o = 5;
    }
    if constexpr (kGrad) {
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                vals[o++] = GB[q][c].x; vals[o++] = GA[q][c].y; vals[o++] = GB[q][c].y; vals[o++] = GA[q][c].x;
            }
    }
    block_reduce_store_nw<NP, C::Waves>(vals, partials + ((size_t)by * rows_per_pair + bx) * NP, box, wave);
}

#pragma clang diagnostic pop

template <int MODE>
__global__ __launch_bounds__(SBox::Threads, 4) void affine_srcbox_kernel(trx_volumes vol, const float *__restrict__ theta, SGeom sg,
                                                                         float *__restrict__ partials, int rows_per_pair)
{
    __shared__ __attribute__((aligned(16))) float lds[SBox::Alloc];
    if ((int)blockIdx.x >= sg.items_per_pair) return;
    srcbox_body<MODE>(vol, theta, sg, partials, lds, blockIdx.x, blockIdx.y, rows_per_pair, trx_wave_index());
}
