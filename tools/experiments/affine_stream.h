// EXPERIMENT (round 2) - NOT part of libtrx.so.  Build: hipcc ... -DTRX_DEV -DTRX_EXPERIMENT_STREAM tools/kbench.hip.
// Result on MI355X, 8 x 256^3 (profiles/r02a_stream_v4_ablation.txt; tile kernel in the same runs: 300-316 us per launch):
//   all 41 sums agree with the tile kernel to 2e-8 ... 1e-7 relative, but the full kernel needs 360-385 us.
//   staging alone (ring DMA + targets, no gather; earlier symmetric variant): 204-208 us - the y ring and the 64-wide rows do cut the
//   memory side to the rate of a plain read of the same bytes (tile kernel, staging alone: 234 us);
//   bookkeeping alone: 100-112 us; gather + bookkeeping with NO memory traffic: 258-272 us.  A step is only 4 rows per thread
//   (the ring must hold the rows of three steps in 72 KB), so one barrier, one counted wait, the plan read-back and the DMA issue
//   are paid every 4 rows where the tile kernel pays its per-tile costs every 8.  Variants tried: per-wave scalar bookkeeping
//   (221 us of it: the CU's single scalar pipe saturates), a dedicated producer wave (one wave cannot issue 20 LDS-DMA per step:
//   ~200 cycles each, 463 us), rows dealt to waves (unbalanced barrier arrival), pieces dealt round-robin (this file).
// Kept because the loader half is sound and measured; what would make it win is a step of >= 8 rows, i.e. a ring of 32 rows, which
// does not fit two blocks per CU.
//
// Y-STREAMING variant of the F1 pass (3-D, MODE 0: moments + sum(qJ)) for transforms near the identity - where every affine run
// starts and, after a rigid pre-alignment, stays.  Included by tools/kbench.hip inside namespace trx.
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

#ifndef TRX_STREAM_DBG
#define TRX_STREAM_DBG 0   // development ablation (tools/kbench.hip): bits: 1 = no box DMA, 2 = no target loads, 4 = no gather
#endif

struct StreamCfg {
    static constexpr int TX = 64, TZ = 8, SR = 4, Waves = 8, Threads = Waves * 64;   // a wave = one z plane of the column
    static constexpr int R = 16;                     // ring rows (slot = source row & 15); slot 16 duplicates slot 0 so that row + 1 is always at + RowBytes
    static constexpr int NP = 14, BW = 76, BW4 = 19; // planes and floats (float4 slots) per ring row
    static constexpr int PPL = 3;                    // planes per DMA piece: 3 x 19 = 57 float4 slots <= 64 lanes, contiguous in LDS
    static constexpr int NPIECE = (NP + PPL - 1) / PPL;
    static constexpr int RowFloats = NP * BW, RowBytes = RowFloats * 4;
    static constexpr int RingFloats = (R + 1) * RowFloats;                 // 72 352 B: two blocks per CU
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static constexpr int BoxAlloc = RingFloats > ReduceScratch ? RingFloats : ReduceScratch;
    static constexpr int Ahead = 2;                  // steps of look-ahead of the loader
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
#define TRX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
    switch (n) {
        TRX_W(0) TRX_W(1) TRX_W(2) TRX_W(3) TRX_W(4) TRX_W(5) TRX_W(6) TRX_W(7) TRX_W(8) TRX_W(9) TRX_W(10) TRX_W(11) TRX_W(12) TRX_W(13) TRX_W(14) TRX_W(15)
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
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
    const int s_begin = seg * sg.steps_per_seg, s_end = min(s_begin + sg.steps_per_seg, sg.nsteps);
    const unsigned box_lds = (unsigned)(uintptr_t)box;

    const bool wave_on = wave < nz;                       // a wave = one z plane of the column
    const bool act = (lane < nx) && wave_on;
    const int x = X0 + (lane < nx ? lane : 0), z = Z0 + (wave_on ? wave : 0);
    const float xn = xtab[x], zn = ztab[z];
    const float base_x = unnorm<3>(xn, fW) + hW * fmaf(t00 - 1.0f, xn, fmaf(t02, zn, t03));
    const float base_y = hH * fmaf(t10, xn, fmaf(t12, zn, t13));
    const float base_z = unnorm<3>(zn, fD) + hD * fmaf(t20, xn, fmaf(t22 - 1.0f, zn, t23));
    const unsigned toffb = (unsigned)((z * H) * W + x) * 4u;         // this thread's target offset inside a row block

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

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;
    int rs_s, ps_s;        // LDS strides (bytes) in SGPRs: ring row, plane
    asm("s_mov_b32 %0, %1" : "=s"(rs_s) : "i"(C::RowBytes));
    asm("s_mov_b32 %0, %1" : "=s"(ps_s) : "i"(C::BW * 4));

    // ---- The PLAN of a run of steps is computed lane-parallel, ONE STEP PER LANE (lane k = step plan0 + k; every wave keeps its own
    // copy), and read per step with a few v_readlane: wave-uniform bookkeeping done step by step on the scalar unit costs eight waves x
    // two blocks of it per CU and saturates the CU's one scalar pipe (measured: 221 us per launch of bookkeeping alone).
    int g_rlo, g_rhi, g_xlo, g_xhi, g_zlo, g_zhi;   // source rows / x / z range of the step (inclusive)
    int g_new = 0, g_cnt = 0, g_key = 0;             // first row the step has to request, number of them; packed slot extents of its rows
    float g_yn[C::SR], g_yid[C::SR];                 // row constants of the step's 4 output rows
    int plan0 = 0;
    const float slack = 0.05f;
    auto plan_geometry = [&](int s0) {
        plan0 = s0;
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            g_yn[j] = ytab[min((s0 + lane) * C::SR + j, H - 1)];
            g_yid[j] = unnorm<3>(g_yn[j], fH);
        }
        const float a0 = g_yn[0], a3 = g_yn[C::SR - 1];
        const float cy0 = g_yid[0] + fmaf(sy, a0, corner_y), cy3 = g_yid[C::SR - 1] + fmaf(sy, a3, corner_y);
        g_rlo = (int)floorf(cy0 + elo[1] - slack);
        g_rhi = (int)floorf(cy3 + ehi[1] + slack) + 1;
        const float cx0 = fmaf(sx, a0, corner_x), cx3 = fmaf(sx, a3, corner_x);
        g_xlo = (int)floorf(fminf(cx0, cx3) + elo[0] - slack);
        g_xhi = (int)floorf(fmaxf(cx0, cx3) + ehi[0] + slack) + 1;
        const float cz0 = fmaf(sz, a0, corner_z), cz3 = fmaf(sz, a3, corner_z);
        g_zlo = (int)floorf(fminf(cz0, cz3) + elo[2] - slack);
        g_zhi = (int)floorf(fmaxf(cz0, cz3) + ehi[2] + slack) + 1;
    };
    auto rl = [&](int v, int st) { return __builtin_amdgcn_readlane(v, st - plan0); };   // value of step st (plan0 <= st < plan0 + 64)

    int ox = 0, oz = 0;            // window origin of the current segment (ox % 4 == 0)
    int seg_end = s_end;           // first step outside the current segment (window left, plan exhausted, or end of the block)
    // exec masks and LDS / plane offsets of the DMA pieces that are not empty, compacted (np_act of them)
    unsigned long long m_act[C::NPIECE];
    int p_act[C::NPIECE];
    int np_act = 0, m_key = -1;
#pragma unroll
    for (int p = 0; p < C::NPIECE; p++) { m_act[p] = 0; p_act[p] = 0; }

    // After (ox, oz) is chosen: which steps of the plan the window holds, the rows each step must request (rows below were requested
    // by its predecessor) and the slot extents its rows must cover (steps t .. t + 2 read them), all lane-parallel.
    auto plan_segment = [&](int s_anchor) {
        const bool holds = (g_xlo >= ox) && (g_xhi <= ox + C::BW - 1) && (g_zlo >= oz) && (g_zhi <= oz + C::NP - 1);
        const int k0 = s_anchor - plan0;
        const unsigned long long bad = (__builtin_amdgcn_ballot_w64(!holds) | (1ull << 61)) >> k0 << k0;   // the plan serves 61 steps (t + 2 is read)
        seg_end = min(s_end, plan0 + (int)__builtin_ctzll(bad));
        const int prev_hi = __shfl_up(g_rhi, 1);
        g_new = (lane == k0) ? g_rlo : max(g_rlo, prev_hi + 1);
        g_cnt = max(0, g_rhi - g_new + 1);
        const int x0a = min(g_xlo, min(__shfl_down(g_xlo, 1), __shfl_down(g_xlo, 2))), x1a = max(g_xhi, max(__shfl_down(g_xhi, 1), __shfl_down(g_xhi, 2)));
        const int z0a = min(g_zlo, min(__shfl_down(g_zlo, 1), __shfl_down(g_zlo, 2))), z1a = max(g_zhi, max(__shfl_down(g_zhi, 1), __shfl_down(g_zhi, 2)));
        const int nx0 = max(x0a, ox) - ox, nx1 = min(x1a, ox + C::BW - 1) - ox, nz0 = max(z0a, oz) - oz, nz1 = min(z1a, oz + C::NP - 1) - oz;
        g_key = (nx0 >> 2) | ((nx1 >> 2) << 8) | (nz0 << 16) | (nz1 << 24);
    };
    auto dma = [&](const char *gbase, unsigned lds_addr, unsigned long long mask) {
        if (TRX_STREAM_DBG & 1) return;
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
    // Request the source rows step t needs and has not got yet: whole 76-float x 3-plane pieces, masked to the slots steps t .. t + 2
    // can touch.  The (row, piece) pairs of the step are numbered q = 0, 1, ... and pair q goes to wave q & 7, so every wave issues the
    // same number of DMAs (+-1): the step's barrier then waits for nobody in particular.  Returns this wave's vector-memory instructions.
    auto issue_rows = [&](int t) -> int {
        if (t >= seg_end) return 0;
        const int key = rl(g_key, t);
        if (key != m_key) {
            m_key = key;
            const int nx0 = key & 0xff, nx1 = (key >> 8) & 0xff, nz0 = (key >> 16) & 0xff, nz1 = (key >> 24) & 0xff;
            const int gx = ox + dx4 * 4;
            const bool xok = (dx4 >= nx0) && (dx4 <= nx1) && (gx >= 0) && (gx + 4 <= W);
            np_act = 0;
#pragma unroll
            for (int p = 0; p < C::NPIECE; p++) {
                const int pl = p * C::PPL + pz;
                const int gz = oz + pl;
                const bool ok = (lane < C::PPL * C::BW4) && (pl < C::NP) && xok && (pl >= nz0) && (pl <= nz1) && (gz >= 0) && (gz < D);
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(ok);
                if (mk) {
#pragma unroll
                    for (int k = 0; k < C::NPIECE; k++)
                        if (k == np_act) { m_act[k] = mk; p_act[k] = p; }
                    np_act++;
                }
            }
        }
        const int lo = rl(g_new, t), cnt = rl(g_cnt, t);
        // pairs per row: the non-empty pieces; when a row of the step lies outside the volume (zero padding: every piece) all of them
        const int npp = (lo >= 0 && lo + cnt <= H) ? np_act : C::NPIECE;
        const int total = cnt * npp;
        int n = 0;
        int row = lo, k = wave;                          // pair q = wave, wave + 8, ... -> (row, piece index k)
        for (int q = wave; q < total; q += 8) {
            while (k >= npp) { k -= npp; row++; }
            const int slot = row & (C::R - 1);
            const unsigned loff = (unsigned)(slot * C::RowBytes);
            if ((unsigned)row < (unsigned)H) {
                if (k < np_act) {
                    unsigned long long mk = 0;
                    int pp = 0;
#pragma unroll
                    for (int i = 0; i < C::NPIECE; i++) { mk = (i == k) ? m_act[i] : mk; pp = (i == k) ? p_act[i] : pp; }
                    const char *gb = reinterpret_cast<const char *>(mov + ((ptrdiff_t)(oz + pp * C::PPL) * H + row) * W + ox);
                    dma(gb, box_lds + loff + pp * (C::PPL * C::BW * 4), mk);
                    n++;
                    if (slot == 0) { dma(gb, box_lds + loff + pp * (C::PPL * C::BW * 4) + C::R * C::RowBytes, mk); n++; }
                }
            } else if (lane < C::PPL * C::BW4 && (k * C::PPL + pz) < C::NP) {   // a row outside the volume: zero padding (LDS stores)
                float4 *d = reinterpret_cast<float4 *>(reinterpret_cast<char *>(box) + loff + k * (C::PPL * C::BW * 4)) + lane;
                *d = make_float4(0.f, 0.f, 0.f, 0.f);
                if (slot == 0) *reinterpret_cast<float4 *>(reinterpret_cast<char *>(d) + C::R * C::RowBytes) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            k += 8;
        }
        return n;
    };
    // target values of step t for this thread (4 rows) -> tv; rows past the end of the volume are clamped (and skipped by the gather)
    auto issue_targets = [&](int t, float (&tv)[C::SR]) {
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            const int y = min(t * C::SR + j, H - 1);
            if (TRX_STREAM_DBG & 2) { tv[j] = 1.f; continue; }
            asm volatile("global_load_dword %0, %1, %2" : "=v"(tv[j]) : "v"(toffb), "s"(tgt + (size_t)y * W) : "memory");
        }
    };
    typedef const __attribute__((address_space(3))) f2u *lds_f2;
    int bpb = 0;                   // LDS byte address of (x = 0, ring slot 0, z = 0) under the current window
    auto gather_step = [&](int s, float (&tv)[C::SR]) {
        float yn_r[C::SR], yid_r[C::SR];
#pragma unroll
        for (int j = 0; j < C::SR; j++) {
            yn_r[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_yn[j]), s - plan0));
            yid_r[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g_yid[j]), s - plan0));
        }
        if (!wave_on || (TRX_STREAM_DBG & 4)) return;                // (uniform per wave) a z plane past the volume
        const int nrow = min(C::SR, H - s * C::SR);                  // (uniform) rows of this step inside the volume
        struct Fetch { f2 r00, r01, r10, r11; float fx, fy, fz; };
        auto fetch = [&](int j) -> Fetch {
            const float yn = yn_r[j];
            const float ix = fmaf(sx, yn, base_x);
            const float iy = yid_r[j] + fmaf(sy, yn, base_y);
            const float iz = fmaf(sz, yn, base_z);
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

    // ---- one step.  Issue order of a wave:  ... R(s+1) | T(s+1)  [wait: all but these]  barrier  R(s+2) | gather(s) ...
    //   T(s+1): this thread's target values of the NEXT step into the other register set (needs no barrier),
    //   wait:   everything older than R(s+1) + T(s+1) has landed, i.e. this wave's share of the rows and its targets of step s,
    //   R(s+2): rows of step s + 2 into ring slots every wave has left (they were read in step s - 1 at the latest).
    int nrows_last = 0;            // vector-memory instructions of this wave in R(s + 1)
    auto step = [&](int s, float (&use)[C::SR], float (&load)[C::SR]) {
        issue_targets(s + 1, load);
        wait_vmcnt(nrows_last + ((TRX_STREAM_DBG & 2) ? 0 : C::SR));
#pragma unroll
        for (int j = 0; j < C::SR; j++) asm volatile("" : "+v"(use[j]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        nrows_last = issue_rows(s + C::Ahead);
        gather_step(s, use);
    };

    float tvA[C::SR], tvB[C::SR];   // target values of the step being gathered / of the next one (ping-pong: two call sites of step)
#pragma unroll
    for (int j = 0; j < C::SR; j++) tvA[j] = tvB[j] = 0.f;

    int s = s_begin;
    while (s < s_end) {
        // ---------------- anchor a segment at step s: plan, window origin, zero padding, fill the pipeline (rows of steps s and s + 1)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                             // nobody gathers from the ring any more, nothing is in flight
        plan_geometry(s);
        {
            const int xlo = rl(g_xlo, s), xhi = rl(g_xhi, s), zlo = rl(g_zlo, s), zhi = rl(g_zhi, s);
            // leave the window's slack on the side the pre-image drifts to as y grows
            ox = (sx >= 0.f) ? (xlo & ~3) : ((xhi - (C::BW - 1) + 3) & ~3);
            if (ox > (xlo & ~3)) ox = xlo & ~3;
            oz = (sz >= 0.f) ? zlo : zhi - (C::NP - 1);
            if (oz > zlo) oz = zlo;
        }
        bpb = (int)box_lds - (oz * C::BW + ox) * 4;
        plan_segment(s);
        if (seg_end <= s) seg_end = s + 1;                           // (cannot happen while stream_fits holds; never spin)
        m_key = -1;
        // cells of the window outside the volume are never written by a DMA: they hold the zero padding
        if ((ox < 0) || (ox + C::BW > W) || (oz < 0) || (oz + C::NP > D) || (rl(g_rlo, s) < 0)) {
            for (int i = tid; i < C::RingFloats / 4; i += C::Threads) reinterpret_cast<float4 *>(box)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __syncthreads();
        }
        issue_rows(s);
        issue_targets(s, tvA);
        nrows_last = issue_rows(s + 1);
        // ---------------- the steps of the segment, two per trip (the target registers ping-pong)
        while (s < seg_end) {
            step(s, tvA, tvB);
            s++;
            if (s >= seg_end) break;   // odd number of steps: the next anchor reloads the targets into A
            step(s, tvB, tvA);
            s++;
        }
        // seg_end < s_end: the window was left (or the plan exhausted); loop back and re-anchor at step s == seg_end
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                 // the ring becomes the reduction scratch

    float vals[NP41];
#pragma unroll
    for (int i = 0; i < NP41; i++) vals[i] = 0.f;
    if (act) {
        vals[0] = acc.M01.x; vals[1] = acc.M01.y; vals[2] = acc.M23.x; vals[3] = acc.M23.y; vals[4] = acc.M4;
        int o = 5;
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a = acc.AB[q][c].x;
                vals[o++] = xn * a; vals[o++] = acc.AB[q][c].y; vals[o++] = zn * a; vals[o++] = a;
            }
    }
    block_reduce_store_nw<NP41, C::Waves>(vals, partials + ((size_t)b * rows_per_pair + bx) * NP41, box, wave);
}
#pragma clang diagnostic pop
