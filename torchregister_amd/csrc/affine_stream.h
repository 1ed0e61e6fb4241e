// Y-STREAMING variant of the F1 pass (3-D, MODE 0: moments + sum(qJ)) for transforms near the identity - where every affine run
// starts and, after a rigid pre-alignment, stays.  Included by affine.hip inside namespace trx.
//
// Why: the tile kernel (tile_body) stages the whole pre-image box of every 32 x 16 x 8 tile, waits for it, gathers, and starts over.
// Its L2-side request count is 2.24x the algorithmic bytes (x halo: a 32-voxel row + neighbour touches 2-3 lines of 128 B; y halo:
// 19 rows staged per 16 computed) and the memory pipe idles whenever both co-resident blocks gather (profiles/r01g_l2_counters.txt).
// Here a block owns a (64 x, 8 z) column and STREAMS it along y:
//   * the source rows live in an LDS RING indexed by (source row & 15): a row is fetched once per column and stays until the walk
//     has passed it - no y halo re-fetch; 64-voxel rows need 1.65 lines per 32 voxels instead of 2.3-3;
//   * rows for step s + 2 (a step = 4 output rows) are requested by LDS-DMA while step s is gathered: requests are in flight all the
//     time, one barrier per step, counted s_waitcnt vmcnt (never 0 inside the loop);
//   * the box origin (ox, oz) is fixed for a SEGMENT of steps, so the per-tile geometry of the tile kernel (readlanes, mask refresh,
//     address base) disappears from the loop; when the pre-image drifts out of the 76 x 14 window (rotation) the pipeline drains and
//     re-anchors.
// A pair whose theta does not fit the window for at least a few steps (stream_fits, a function of theta only) is left to the tile
// kernels: both kernels evaluate the same predicate and exactly one of them writes the pair's partial rows.
// Numerics: identical per-voxel arithmetic to tile_body's fast loop (coordinates bitwise ATen's at the identity); only the order in
// which a thread's voxels are added differs (one thread: one (x, z) column along the whole segment).

struct StreamCfg {
    static constexpr int TX = 64, TZ = 8, SR = 4, Threads = 512, Waves = 8;
    static constexpr int R = 16;                     // ring rows (slot = source row & 15); slot 16 duplicates slot 0 so that row + 1 is always at + RowBytes
    static constexpr int NP = 14, BW = 76, BW4 = 19; // planes and floats (float4 slots) per ring row
    static constexpr int PPL = 3;                    // planes per DMA piece: 3 x 19 = 57 float4 slots <= 64 lanes, contiguous in LDS
    static constexpr int NPIECE = (NP + PPL - 1) / PPL;
    static constexpr int RowFloats = NP * BW, RowBytes = RowFloats * 4;
    static constexpr int RingFloats = (R + 1) * RowFloats;                 // 72 352 B: two blocks per CU
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static constexpr int BoxAlloc = RingFloats > ReduceScratch ? RingFloats : ReduceScratch;
    static constexpr int Ahead = 2;                  // steps of look-ahead of the loader
    static constexpr int Chunk = 16;                 // steps per geometry refresh (64 output rows: one row per lane)
};

struct StreamGeom {
    int ntx, ntz, nsteps, nseg, steps_per_seg, blocks_per_pair;
};

static StreamGeom stream_geom(const trx_volumes &v)
{
    StreamGeom g;
    g.ntx = (v.W + StreamCfg::TX - 1) / StreamCfg::TX;
    g.ntz = (v.D + StreamCfg::TZ - 1) / StreamCfg::TZ;
    g.nsteps = (v.H + StreamCfg::SR - 1) / StreamCfg::SR;
    const long cols = (long)v.B * g.ntx * g.ntz;
    // 512 block slots (2 per CU); a segment pays ~2 steps of pipeline fill, so at least 8 steps each
    int nseg = cols >= 512 ? 1 : (int)((512 + cols - 1) / cols);
    const int cap = g.nsteps / 8 > 1 ? g.nsteps / 8 : 1;
    if (nseg > cap) nseg = cap;
    g.steps_per_seg = (g.nsteps + nseg - 1) / nseg;
    g.nseg = (g.nsteps + g.steps_per_seg - 1) / g.steps_per_seg;
    g.blocks_per_pair = g.ntx * g.ntz * g.nseg;
    return g;
}

// Host-side part of the decision (sizes only; the theta part is stream_fits on the device): rows of whole float4, a column at
// least one tile wide, and enough blocks to fill the chip (small problems are launch-bound and stay on the tile kernel).
static bool stream_shape_ok(const trx_volumes &v, bool force)
{
    if (v.ndim != 3 || (v.W & 3) || v.W < 16 || v.H < 8) return false;
    if ((size_t)v.H * v.W >= ((size_t)1 << 28)) return false;   // 32-bit byte offsets inside three planes
    if (force) return true;
    return v.W >= StreamCfg::TX && (long)stream_geom(v).blocks_per_pair * v.B >= 256;
}

// Does the pre-image of a (64 x, 8 z) column fit the ring window for a useful number of steps?  theta-only (the map is affine: extents
// do not depend on the position), so every block of a pair - and the tile kernels, which take the pair otherwise - agree.
__device__ __forceinline__ bool stream_fits(const float *__restrict__ th, float fD, float fH, float fW)
{
    const float s00 = th[0], s01 = th[1] * fW / fH, s02 = th[2] * fW / fD;
    const float s10 = th[4] * fH / fW, s11 = th[5], s12 = th[6] * fH / fD;
    const float s20 = th[8] * fD / fW, s21 = th[9] * fD / fH, s22 = th[10];
    const float ex = (float)(StreamCfg::TX - 1), ez = (float)(StreamCfg::TZ - 1);
    // source rows alive at once: lowest row of step s .. highest row of step s + Ahead (+1 neighbour, +1 floor, slack)
    const float rows = (float)(StreamCfg::SR * (StreamCfg::Ahead + 1) - 1) * s11 + fabsf(s10) * ex + fabsf(s12) * ez + 3.2f;
    // x / z window: the column's span + neighbour + floor (+3 of float4 alignment in x) + room for 16 rows of drift
    const float xs = fabsf(s00) * ex + fabsf(s02) * ez + 16.f * fabsf(s01) + 2.2f + 3.f;
    const float zs = fabsf(s22) * ez + fabsf(s20) * ex + 16.f * fabsf(s21) + 2.2f;
    return (s11 > 0.3f) && (rows <= (float)StreamCfg::R) && (xs <= (float)StreamCfg::BW) && (zs <= (float)StreamCfg::NP);   // NaN compares false
}

// s_waitcnt vmcnt(n) for a wave-uniform n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt(int n)
{
    n = n > 31 ? 31 : n;
#define TRX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        TRX_W(0) TRX_W(1) TRX_W(2) TRX_W(3) TRX_W(4) TRX_W(5) TRX_W(6) TRX_W(7) TRX_W(8) TRX_W(9) TRX_W(10) TRX_W(11) TRX_W(12) TRX_W(13) TRX_W(14) TRX_W(15)
        TRX_W(16) TRX_W(17) TRX_W(18) TRX_W(19) TRX_W(20) TRX_W(21) TRX_W(22) TRX_W(23) TRX_W(24) TRX_W(25) TRX_W(26) TRX_W(27) TRX_W(28) TRX_W(29) TRX_W(30)
    default: asm volatile("s_waitcnt vmcnt(31)" ::: "memory"); break;
    }
#undef TRX_W
}

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
template <int MODE>
__global__ __launch_bounds__(StreamCfg::Threads, 4) void affine_stream_kernel(trx_volumes vol, const float *__restrict__ theta, StreamGeom sg,
                                                                               float *__restrict__ partials, int rows_per_pair)
{
    static_assert(MODE == 0, "the streaming kernel implements the optimiser step (moments + sum(qJ))");
    using C = StreamCfg;
    constexpr int NP41 = np_full(3);
    __shared__ __attribute__((aligned(16))) float box[C::BoxAlloc];
    const int b = blockIdx.y, bx = blockIdx.x;
    const int D = vol.D, H = vol.H, W = vol.W;
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float *__restrict__ th = uni_ptr(theta + (size_t)b * TRX_PSTRIDE);
    if (!stream_fits(th, fD, fH, fW)) return;           // the tile kernels own this pair
    if (bx >= sg.blocks_per_pair) return;
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)b * vol.moving_stride);
    const float *__restrict__ tgt = uni_ptr(vol.target + (size_t)b * vol.target_stride);
    const float *__restrict__ xtab = uni_ptr(vol.xn), *__restrict__ ytab = uni_ptr(vol.yn), *__restrict__ ztab = uni_ptr(vol.zn);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    const float t00 = th[0], t01 = th[1], t02 = th[2], t03 = th[3];
    const float t10 = th[4], t11 = th[5], t12 = th[6], t13 = th[7];
    const float t20 = th[8], t21 = th[9], t22 = th[10], t23 = th[11];
    const float sx = uni(hW * t01), sy = uni(hH * (t11 - 1.0f)), sz = uni(hD * t21);

    // column of this block (XCD-aware order as in tile_body) and its y segment
    const int ncol = sg.ntx * sg.ntz;
    const int seg = bx / ncol, cb = bx - seg * ncol;
    int col = cb;
    if ((ncol & 7) == 0) col = (cb & 7) * (ncol >> 3) + (cb >> 3);
    const int X0 = (col % sg.ntx) * C::TX, Z0 = (col / sg.ntx) * C::TZ;
    const int nx = min(C::TX, W - X0), nz = min(C::TZ, D - Z0);
    const bool wave_on = wave < nz;                       // a wave = one z plane of the column
    const bool act = (lane < nx) && wave_on;
    const int x = X0 + (lane < nx ? lane : 0), z = Z0 + (wave_on ? wave : 0);
    const float xn = xtab[x], zn = ztab[z];
    const float base_x = unnorm<3>(xn, fW) + hW * fmaf(t00 - 1.0f, xn, fmaf(t02, zn, t03));
    const float base_y = hH * fmaf(t10, xn, fmaf(t12, zn, t13));
    const float base_z = unnorm<3>(zn, fD) + hD * fmaf(t20, xn, fmaf(t22 - 1.0f, zn, t23));

    // pre-image of the (x, z) rectangle of the column relative to the image of its (X0, Z0) corner
    float elo[3], ehi[3];
    {
        const float exs[2] = {(float)(C::TX - 1), (float)(C::TZ - 1)};
        const float slope[3][2] = {{t00, t02 * fW / fD}, {t10 * fH / fW, t12 * fH / fD}, {t20 * fD / fW, t22}};
#pragma unroll
        for (int c = 0; c < 3; c++) {
            float lo = 0.f, hi = 0.f;
#pragma unroll
            for (int a = 0; a < 2; a++) {
                const float e = slope[c][a] * exs[a];
                lo += fminf(e, 0.f); hi += fmaxf(e, 0.f);
            }
            elo[c] = uni(lo); ehi[c] = uni(hi);
        }
    }
    const float cxn = xtab[X0], czn = ztab[Z0];
    const float corner_x = uni(unnorm<3>(cxn, fW) + hW * fmaf(t00 - 1.0f, cxn, fmaf(t02, czn, t03)));
    const float corner_y = uni(hH * fmaf(t10, cxn, fmaf(t12, czn, t13)));
    const float corner_z = uni(unnorm<3>(czn, fD) + hD * fmaf(t20, cxn, fmaf(t22 - 1.0f, czn, t23)));

    // LDS-DMA slot of a lane inside a piece (3 planes x 19 float4): plane pz, float4 dx4; byte offset inside the volume
    const int pz = lane / C::BW4, dx4 = lane - pz * C::BW4;
    const unsigned rb0 = (unsigned)((pz * H) * W + dx4 * 4) * 4u;
    const unsigned box_lds = (unsigned)(uintptr_t)box;
    const unsigned toffb = (unsigned)((z * H) * W + x) * 4u;     // this thread's target offset inside a row block

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;

    float sxv, syv, szv;   // VGPR copies of the uniform slopes (the per-row yn is then the single SGPR operand)
    asm("v_mov_b32 %0, %1" : "=v"(sxv) : "s"(sx));
    asm("v_mov_b32 %0, %1" : "=v"(syv) : "s"(sy));
    asm("v_mov_b32 %0, %1" : "=v"(szv) : "s"(sz));
    int rs_s, ps_s;        // LDS strides (bytes) in SGPRs: ring row, plane
    asm("s_mov_b32 %0, %1" : "=s"(rs_s) : "i"(C::RowBytes));
    asm("s_mov_b32 %0, %1" : "=s"(ps_s) : "i"(C::BW * 4));

    const int s_begin = seg * sg.steps_per_seg, s_end = min(s_begin + sg.steps_per_seg, sg.nsteps);

    // ---- geometry of a chunk of steps, one STEP per lane (lanes 0 .. Chunk + Ahead + 1), and the row tables, one ROW per lane
    int g_rlo = 0, g_rhi = 0, g_xlo = 0, g_xhi = 0, g_zlo = 0, g_zhi = 0;
    float yn_l = 0.f, yid_l = 0.f;
    int chunk0 = -1 << 30;
    const float slack = 0.05f;
    auto chunk_geometry = [&](int s0) {
        chunk0 = s0;
        const int y0 = min((s0 + lane) * C::SR, H - 1), y3 = min((s0 + lane) * C::SR + C::SR - 1, H - 1);
        const float a0 = ytab[y0], a3 = ytab[y3];
        const float cy0 = unnorm<3>(a0, fH) + fmaf(sy, a0, corner_y), cy3 = unnorm<3>(a3, fH) + fmaf(sy, a3, corner_y);
        g_rlo = (int)floorf(cy0 + elo[1] - slack);
        g_rhi = (int)floorf(cy3 + ehi[1] + slack) + 1;
        const float cx0 = fmaf(sx, a0, corner_x), cx3 = fmaf(sx, a3, corner_x);
        g_xlo = (int)floorf(fminf(cx0, cx3) + elo[0] - slack);
        g_xhi = (int)floorf(fmaxf(cx0, cx3) + ehi[0] + slack) + 1;
        const float cz0 = fmaf(sz, a0, corner_z), cz3 = fmaf(sz, a3, corner_z);
        g_zlo = (int)floorf(fminf(cz0, cz3) + elo[2] - slack);
        g_zhi = (int)floorf(fmaxf(cz0, cz3) + ehi[2] + slack) + 1;
        yn_l = ytab[min(s0 * C::SR + lane, H - 1)];
        yid_l = unnorm<3>(yn_l, fH);
    };
    auto rl = [&](int v, int s) { return __builtin_amdgcn_readlane(v, s - chunk0); };   // geometry of step s (chunk0 <= s < chunk0 + 20)

    // ---- loader state (wave-uniform)
    int ox = 0, oz = 0;            // window origin of the current segment (ox % 4 == 0)
    int Lrow = 0;                  // source rows < Lrow have been requested under the current anchor
    int seg_end = s_end;           // first step that does not fit the current anchor (re-anchor there)
    unsigned long long m_piece[C::NPIECE];   // exec masks of the DMA pieces: slot needed by the steps being loaded AND inside the volume
    int m_key = -1;
#pragma unroll
    for (int p = 0; p < C::NPIECE; p++) m_piece[p] = 0;

    auto dma = [&](const char *gbase, unsigned lds_addr, unsigned long long mask) {
        unsigned long long sv;
        unsigned m0s;
        asm volatile("s_mov_b64 %[sv], exec\n\t"
                     "s_mov_b32 %[m0s], m0\n\t"
                     "s_mov_b32 m0, %[lds]\n\t"
                     "s_mov_b64 exec, %[mk]\n\t"
                     "global_load_lds_dwordx4 %[off], %[base]\n\t"
                     "s_mov_b64 exec, %[sv]\n\t"
                     "s_mov_b32 m0, %[m0s]"
                     : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                     : [lds] "s"(lds_addr), [base] "s"(gbase), [off] "v"(rb0), [mk] "s"(mask)
                     : "memory");
    };

    // Request the source rows step t needs and has not got yet (rows of whole 76-float x 14-plane slabs, masked to what steps t and
    // t + 1 can touch).  Returns the number of vector-memory instructions THIS wave issued.  Piece q = row * NPIECE + p goes to wave q & 7.
    auto issue_rows = [&](int t) -> int {
        if (t >= seg_end) return 0;
        const int xlo = rl(g_xlo, t), xhi = rl(g_xhi, t), zlo = rl(g_zlo, t), zhi = rl(g_zhi, t);
        if (xlo < ox || xhi > ox + C::BW - 1 || zlo < oz || zhi > oz + C::NP - 1) {   // the window no longer holds this step: drain and re-anchor there
            seg_end = t;
            return 0;
        }
        // slots that steps t .. t + 2 can touch (the rows requested now are read by those steps at most: stream_fits bounds the rows
        // alive at once), clamped to the window
        const int t1 = min(t + 1, s_end - 1), t2 = min(t + 2, s_end - 1);
        const int nx0 = max(min(xlo, min(rl(g_xlo, t1), rl(g_xlo, t2))), ox) - ox, nx1 = min(max(xhi, max(rl(g_xhi, t1), rl(g_xhi, t2))), ox + C::BW - 1) - ox;
        const int nz0 = max(min(zlo, min(rl(g_zlo, t1), rl(g_zlo, t2))), oz) - oz, nz1 = min(max(zhi, max(rl(g_zhi, t1), rl(g_zhi, t2))), oz + C::NP - 1) - oz;
        const int key = (nx0 >> 2) | ((nx1 >> 2) << 8) | (nz0 << 16) | (nz1 << 24);
        if (key != m_key) {
            m_key = key;
            const int gx = ox + dx4 * 4;
            const bool xok = (dx4 >= (nx0 >> 2)) && (dx4 <= (nx1 >> 2)) && (gx >= 0) && (gx + 4 <= W);
#pragma unroll
            for (int p = 0; p < C::NPIECE; p++) {
                const int pl = p * C::PPL + pz;
                const int gz = oz + pl;
                const bool ok = (lane < C::PPL * C::BW4) && (pl < C::NP) && xok && (pl >= nz0) && (pl <= nz1) && (gz >= 0) && (gz < D);
                m_piece[p] = __builtin_amdgcn_ballot_w64(ok);
            }
        }
        const int rhi = rl(g_rhi, t);
        int n = 0;
        // piece p of row `row` goes to wave (row * NPIECE + p) & 7: at most one piece of a row per wave (NPIECE <= 8)
        for (int row = max(Lrow, rl(g_rlo, t)); row <= rhi; row++) {
            const int p = (wave - row * C::NPIECE) & 7;
            if (p >= C::NPIECE) continue;
            const int slot = row & (C::R - 1);
            const unsigned loff = (unsigned)(slot * C::RowBytes + p * (C::PPL * C::BW * 4));
            if ((unsigned)row < (unsigned)H) {
                unsigned long long mk = 0;
#pragma unroll
                for (int k = 0; k < C::NPIECE; k++) mk = (k == p) ? m_piece[k] : mk;
                if (mk) {
                    const char *gb = reinterpret_cast<const char *>(mov + ((ptrdiff_t)(oz + p * C::PPL) * H + row) * W + ox);
                    dma(gb, box_lds + loff, mk);
                    n++;
                    if (slot == 0) { dma(gb, box_lds + loff + C::R * C::RowBytes, mk); n++; }
                }
            } else if (lane < C::PPL * C::BW4 && (p * C::PPL + pz) < C::NP) {   // a row outside the volume: zero padding
                float4 *d = reinterpret_cast<float4 *>(reinterpret_cast<char *>(box) + loff) + lane;
                *d = make_float4(0.f, 0.f, 0.f, 0.f);
                if (slot == 0) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(d) + C::R * C::RowBytes) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        Lrow = max(Lrow, rhi + 1);
        return n;
    };
    // target values of step t for this thread (4 rows) -> tv; rows past the end of the volume are clamped (and skipped by the gather)
    auto issue_targets = [&](int t, float (&tv)[C::SR]) -> int {
        if (t >= s_end) return 0;
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            const int y = min(t * C::SR + j, H - 1);
            asm volatile("global_load_dword %0, %1, %2" : "=v"(tv[j]) : "v"(toffb), "s"(tgt + (size_t)y * W) : "memory");
        }
        return C::SR;
    };

    typedef const __attribute__((address_space(3))) f2u *lds_f2;
    // ---- the 4 rows of step s for this thread, gathered from the ring
    auto gather_step = [&](int s, float (&tv)[C::SR]) {
        const int bpb = (int)box_lds - (oz * C::BW + ox) * 4;      // LDS byte address of (x = 0, ring slot 0, z = 0)
        const int rbase = s * C::SR - chunk0 * C::SR;               // lane of the step's first row in the row tables
        float yn_r[C::SR], yid_r[C::SR];
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            yn_r[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yn_l), rbase + j));
            yid_r[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yid_l), rbase + j));
        }
        const int nrow = min(C::SR, H - s * C::SR);                 // (uniform) rows of this step inside the volume
        struct Fetch { f2 r00, r01, r10, r11; float fx, fy, fz; };
        auto fetch = [&](int j) -> Fetch {
            const float yn = yn_r[j];
            const float ix = fmaf(sxv, yn, base_x);
            const float iy = yid_r[j] + fmaf(syv, yn, base_y);
            const float iz = fmaf(szv, yn, base_z);
            int a0, a1, a2, a3, ry;
            asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(a0) : "v"(floor_to_int(ix)), "s"(bpb));
            asm("v_and_b32 %0, 15, %1" : "=v"(ry) : "v"(floor_to_int(iy)));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a1) : "v"(ry), "s"(rs_s), "v"(a0));
            asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(a2) : "v"(floor_to_int(iz)), "s"(ps_s), "v"(a1));
            asm("v_add_u32 %0, %1, %2" : "=v"(a3) : "s"(rs_s), "v"(a2));
            Fetch f;
            f.r00 = *(lds_f2)(unsigned)a2; f.r10 = *(lds_f2)(unsigned)(a2 + C::BW * 4);      // (y0, z0), (y0, z0 + 1)
            f.r01 = *(lds_f2)(unsigned)a3; f.r11 = *(lds_f2)(unsigned)(a3 + C::BW * 4);      // (y0 + 1, z0), (y0 + 1, z0 + 1)
            f.fx = __builtin_amdgcn_fractf(ix); f.fy = __builtin_amdgcn_fractf(iy); f.fz = __builtin_amdgcn_fractf(iz);
            return f;
        };
        if (!wave_on) return;                                        // (uniform per wave) a z plane past the volume
        Fetch cur = fetch(0);
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            Fetch nxt;
            if (j + 1 < C::SR) nxt = fetch(j + 1);
            if (j < nrow) {
                const Samp3 sm = lerp3_pairs<true>(cur.r00, cur.r01, cur.r10, cur.r11, cur.fx, cur.fy, cur.fz);
                f1_accumulate_pk<0>(sm, tv[j], yn_r[j], acc);
            }
            if (j + 1 < C::SR) cur = nxt;
        }
    };

    // ---- one step: wait for its rows (everything but the youngest batch), barrier, request the rows of step s + Ahead, gather
    int last_batch = 0;
    auto step = [&](int s, float (&use)[C::SR], float (&load)[C::SR]) {
        wait_vmcnt(last_batch);
#pragma unroll
        for (int j = 0; j < C::SR; j++) asm volatile("" : "+v"(use[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                             // every wave's pieces of step s are in LDS; everyone has left step s - 1
        last_batch = issue_rows(s + C::Ahead) + issue_targets(s + C::Ahead, load);
        gather_step(s, use);
    };

    float tvA[C::SR], tvB[C::SR], tvC[C::SR];   // target values of steps s, s + 1, s + 2 (rotating roles, no copies: three call sites of step)
#pragma unroll
    for (int j = 0; j < C::SR; j++) tvA[j] = tvB[j] = tvC[j] = 0.f;

    int s = s_begin;
    while (s < s_end) {
        // ---------------- anchor a segment at step s: window origin, zero padding, fill the pipeline (steps s and s + 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // nobody gathers from the ring any more, nothing is in flight
        if (s - chunk0 < 0 || s - chunk0 >= C::Chunk) chunk_geometry(s);
        bool zero_ring;
        {
            const int xlo = rl(g_xlo, s), xhi = rl(g_xhi, s), zlo = rl(g_zlo, s), zhi = rl(g_zhi, s);
            // leave the window's slack on the side the pre-image drifts to as y grows
            ox = (sx >= 0.f) ? (xlo & ~3) : ((xhi - (C::BW - 1) + 3) & ~3);
            if (ox > (xlo & ~3)) ox = xlo & ~3;
            oz = (sz >= 0.f) ? zlo : zhi - (C::NP - 1);
            if (oz > zlo) oz = zlo;
            // cells of the window outside the volume are never written by a DMA: they hold the zero padding
            zero_ring = (ox < 0) || (ox + C::BW > W) || (oz < 0) || (oz + C::NP > D) || (rl(g_rlo, s) < 0);
        }
        seg_end = s_end;
        m_key = -1;
        Lrow = -(1 << 28);
        if (zero_ring) {
            for (int i = tid; i < C::RingFloats / 4; i += C::Threads) reinterpret_cast<float4 *>(box)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        issue_rows(s);
        issue_targets(s, tvA);
        last_batch = issue_rows(s + 1) + issue_targets(s + 1, tvB);
        if (seg_end <= s) seg_end = s + 1;                           // (cannot happen while stream_fits holds; never spin)
        // ---------------- the steps of the segment; the target registers rotate A -> B -> C through three call sites
        int ph = 0;
        while (s < seg_end) {
            if (s - chunk0 >= C::Chunk) chunk_geometry(s);           // (the lanes of the old chunk were valid up to step chunk0 + 63 for the loader)
            if (ph == 0) step(s, tvA, tvC);
            else if (ph == 1) step(s, tvB, tvA);
            else step(s, tvC, tvB);
            ph = (ph == 2) ? 0 : ph + 1;
            s++;
        }
        // seg_end < s_end: the window was left; loop back and re-anchor at step s == seg_end
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (!act) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
        acc.M01 = acc.M23 = (f2)(0.f);
        acc.M4 = 0.f;
    }
    float vals[NP41];
    vals[0] = acc.M01.x; vals[1] = acc.M01.y; vals[2] = acc.M23.x; vals[3] = acc.M23.y; vals[4] = acc.M4;
    int o = 5;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float a = acc.AB[q][c].x;
            vals[o++] = xn * a; vals[o++] = acc.AB[q][c].y; vals[o++] = zn * a; vals[o++] = a;
        }
    block_reduce_store_nw<NP41, C::Waves>(vals, partials + ((size_t)b * rows_per_pair + bx) * NP41, box);
}
#pragma clang diagnostic pop
