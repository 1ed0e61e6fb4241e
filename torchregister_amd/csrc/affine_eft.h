// EXACT-FOOTPRINT variant of the fused F1 pass (3-D) for ROTATED transforms.  Included by affine.hip inside namespace trx.
//
// Why a third kernel family.  Counters of the tile kernels at theta = R(0.5, 0.4, 0.3) diag(1.05, 0.95, 1.02), 8 x 256^3
// (profiles/r04a_pose_pmc.txt): 620 us per launch, 46.5 M L2 requests of 128 B = 5.5 x the algorithmic bytes, 41 % of them misses
// (HBM side 2.4 GB = 2.2 x), waves parked 44 % of their cycles.  A rotated 16 x 16 x 8 tile stages the AXIS-ALIGNED bounding box of its
// pre-image (5.3 floats per voxel, of which a third is read), as a burst that nothing overlaps inside the block.  Here:
//   * a block stages what the tile can TOUCH: per source row (y, z) of the pre-image the x-window [wlo, whi] that any voxel of the tile may
//     read, whatever the fractional position of the tile's corner - a function of theta and the tile shape only (the map is affine), so the
//     PLAN (row windows, their packing, which thread fetches which 16 bytes) is made once per block and every tile of the column re-uses it
//     with its own integer origin R = floor(image of the tile's corner).  ~2.0 floats per voxel for a general rotation of a 16^3 tile;
//   * rows are packed back to back in LDS in 16-byte granules that start AT wlo (unaligned 16-byte global loads: no alignment slack), and a
//     row table E[(y, z)] = byte address of x = 0 of that row turns the gather's address into  E + 4 floor(x)  - two more LDS reads and two
//     more vector instructions per voxel than a dense box, for a third of its bytes;
//   * ~33 KB per tile instead of 78.6 KB, so a block holds TWO buffers: the granules of tile t + 1 travel global -> VGPR -> the other buffer
//     while tile t is gathered - by LDS-DMA straight into the other buffer, a whole tile ahead - one wait and one barrier per tile.
// Per-voxel arithmetic is tile_body's (same coordinates up to the rounding of the origin shift, same accumulators, same partial-row layout
// as GeomRD: 16 x 16 x 16 tiles, column (x-tile, z-tile) walking y), so the finalise kernel and the tests see another choice of the same pass.

#ifndef TRX_EF_DBG
#define TRX_EF_DBG 0   // development ablation (tools/ebench.hip): bits: 1 = no staging loads, 2 = no target loads, 4 = no gather
#endif
#ifndef TRX_EF_STAGES
#define TRX_EF_STAGES 2   // rows in flight in the gather (3: also the table reads of row j + 2 - measured alternative)
#endif
#ifndef TRX_EF_STAMP
#define TRX_EF_STAMP 0   // development (tools/ebench.hip): per-wave s_memtime sums of a tile step's phases and the block's start / end -> trx_ef_stamps
#endif
#if TRX_EF_STAMP
__device__ unsigned long long trx_ef_stamps2[1024 * 8 * 8];   // the prologue in detail: windows + wave scans, barrier, totals + barrier, table + descriptors + barrier, granule registers + barrier, origins + barrier, head tiles, first tile requested
#define TRX_EF_ST2(k) do { if (trx_lane_id() == 0) trx_ef_stamps2[((size_t)((by * rows_stride + bx) & 1023) * 8 + wave) * 8 + (k)] = __builtin_amdgcn_s_memtime() - ef_t0; } while (0)
__device__ unsigned long long trx_ef_blocks[1024 * 8];   // flat step kernel: per block the 100 MHz clock at its start and after each of its items
__device__ unsigned long long trx_ef_stamps[1024 * 8 * 8];   // [item][wave][issue, gather, wait (sums over the tiles), plan done, tail done | all done << 32, tiles | hwid << 16 | xcc << 32, first tile landed, walk done] in s_memtime ticks since the item began
#endif
#if !TRX_EF_STAMP
#define TRX_EF_ST2(k) do { } while (0)
#endif
#ifndef TRX_EF_PRIO
#define TRX_EF_PRIO 0       // 1: the two blocks of a CU alternate at s_setprio 1 in time slices of 2^TRX_EF_PRIO_BIT cycles (development)
#endif
#ifndef TRX_EF_PRIO_BIT
#define TRX_EF_PRIO_BIT 14
#endif
#ifndef TRX_EF_PARTFIX
#define TRX_EF_PARTFIX 0   // 1: granules that straddle a face in x travel by DMA like full ones and are patched in LDS afterwards, instead of element by element with
                           // ordinary loads (and a wait for everything in flight) - measured alternative: 19.0-19.2 k against 19.2-19.3 k pair-it/s at the rotated
                           // pose (profiles/r05c_eft_item_timeline.txt): the long request phases of boundary tiles are their per-granule tests, not those loads
#endif
#ifndef TRX_EF_SHAPE
#define TRX_EF_SHAPE 0   // voxels of a wave per row: 0 = 16 x by 4 z, 1 = 8 x by 8 z
#endif
#ifndef TRX_EF_PINGPONG
#define TRX_EF_PINGPONG 1   // the ticket queues run the pairs backwards when TRX_FLAG_WALK_DOWN is set (odd iterations of trx_affine_run)
#endif
#ifndef TRX_EF_TICKETS
#define TRX_EF_TICKETS 1   // flat grid of the step kernel behind the z-streaming kernel: the blocks draw their items from one queue per XCD (0: every gridDim-th item)
#endif
#ifndef TRX_EF_CHUNK
#define TRX_EF_CHUNK 0   // 1: flat grid of the step kernel: a block's items are consecutive (one pair, ONE plan for all of them) instead of every gridDim-th one - measured
                         // alternative (profiles/r05c_eft_item_timeline.txt): the plan is made once instead of four times (an item's plan is done after 11.5 k ticks
                         // instead of 19.7 k) and the launch is 12 % SLOWER: an XCD then holds 8 columns of each of 8 pairs instead of whole 32-column slabs of 2
                         // pairs, and a tile step takes 10.4 k ticks instead of 8.8 k
#endif
#ifndef TRX_EF_ISSUE_PRIO
#define TRX_EF_ISSUE_PRIO 0   // s_setprio of a wave while it requests the next tile (the stamps of tools/ebench.hip: 3 500 of a tile step's 8 900 cycles pass there) - development
#endif
#ifndef TRX_EF_ROWSTEP
#define TRX_EF_ROWSTEP 1   // row terms of the coordinates and yn by stepping from the first row of a call instead of one v_readlane per row and term (0: measured alternative)
#endif
#ifndef TRX_EF_V2
#define TRX_EF_V2 1   // round 5, by the price list of profiles/r05a_mfma_coissue_and_op_costs.txt (with TRX_EF_ROWSTEP): the row table in BYTES, so that the four data
                      // addresses of a voxel are plain v_add_u32 (2.6 cycles) instead of v_add_lshl_u32 (5.0), and (x, y) stepped per row by one v_pk_add with a
                      // scalar pair instead of two adds with scalar operands.  (The z-streaming body's lerp + accumulate, also tried here: 22 packed
                      // instructions per voxel against this body's 17.5 - it has no v_mov to save - measured in profiles/r05c_eft_v2.txt.)
#endif
#ifndef TRX_EF_EPS
#define TRX_EF_EPS 0.05f   // slack of every window bound: fp32 rounding of the coordinates + non-uniformity of ATen's coordinate tables
#endif

struct ECfg {
    static constexpr int TX = 16, TY = 16, TZ = 16, Threads = 512, Waves = 8, NH = 2, Rows = 8;
    static constexpr int K = 5;              // granule slots (16 B) a thread fetches per tile
    static constexpr int NY = 32, NZ = 32;   // rows of the plan: at most NY x NZ (any rotation of the tile with a zoom up to ~1.1)
    static constexpr int TP = 40;            // row table: entry (dy - dy0) + TP (dz - dz0).  The pitch is 8 mod 32 banks: the rows a half-wave
                                             // looks up lie in a patch of a few y by a few z, which a pitch of 32 would fold onto one bank each
    static constexpr int GCap = 2304;        // granule slots of one buffer (36 KB)
    static constexpr int BufFloats = GCap * 4, TabInts = TP * NZ;
    static constexpr int ReduceScratch = Waves * 16 * 65 + Waves * 16;
    static constexpr int OrgInts = 64 * 8;   // origins of 64 tiles of the column (8 ints each)
    static constexpr int Alloc = 2 * BufFloats + TabInts + OrgInts;   // 20224 floats = 79.0 KB: two blocks per CU
    static_assert(Alloc >= ReduceScratch && K * Threads >= GCap && NY * NZ == 2 * Threads && TabInts + 2 * Waves <= BufFloats, "geometry");
};

// theta -> slope matrix of the voxel-space map s = b + A q (q = voxel offset inside a tile), its inverse, the extent of a tile's pre-image
struct EfMap {
    float A[3][3], N[3][3], ext_lo[3], ext_hi[3];
    bool ok;
};
__device__ __forceinline__ EfMap ef_map(const float *__restrict__ th, float fD, float fH, float fW)
{
    EfMap m;
    const float A[3][3] = {{th[0], th[1] * fW / fH, th[2] * fW / fD}, {th[4] * fH / fW, th[5], th[6] * fH / fD}, {th[8] * fD / fW, th[9] * fD / fH, th[10]}};
    const float c00 = A[1][1] * A[2][2] - A[1][2] * A[2][1], c01 = A[1][2] * A[2][0] - A[1][0] * A[2][2], c02 = A[1][0] * A[2][1] - A[1][1] * A[2][0];
    const float det = A[0][0] * c00 + A[0][1] * c01 + A[0][2] * c02;
    float amax = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int a = 0; a < 3; a++) { m.A[c][a] = A[c][a]; amax = fmaxf(amax, fabsf(A[c][a])); }
    m.ok = (fabsf(det) > 0.05f) && (amax < 4.0f);   // NaN compares false
    const float r = 1.0f / det;
    m.N[0][0] = c00 * r; m.N[0][1] = (A[0][2] * A[2][1] - A[0][1] * A[2][2]) * r; m.N[0][2] = (A[0][1] * A[1][2] - A[0][2] * A[1][1]) * r;
    m.N[1][0] = c01 * r; m.N[1][1] = (A[0][0] * A[2][2] - A[0][2] * A[2][0]) * r; m.N[1][2] = (A[0][2] * A[1][0] - A[0][0] * A[1][2]) * r;
    m.N[2][0] = c02 * r; m.N[2][1] = (A[0][1] * A[2][0] - A[0][0] * A[2][1]) * r; m.N[2][2] = (A[0][0] * A[1][1] - A[0][1] * A[1][0]) * r;
    const float ex = (float)(ECfg::TX - 1);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        m.ext_lo[c] = m.ext_hi[c] = 0.f;
#pragma unroll
        for (int a = 0; a < 3; a++) { const float e = A[c][a] * ex; m.ext_lo[c] += fminf(e, 0.f); m.ext_hi[c] += fmaxf(e, 0.f); }
    }
    return m;
}
// Rows of the plan: dy = dy0 .. dy0 + ny - 1 relative to R_y = floor(b_y), likewise z; x windows lie inside [xmin, xmax].
struct EfDims {
    int dy0, dz0, ny, nz, xmin, xmax;
    bool ok;
};
__device__ __forceinline__ EfDims ef_dims(const EfMap &m)
{
    EfDims d;
    const float e = TRX_EF_EPS;
    d.dy0 = (int)floorf(m.ext_lo[1] - e); d.ny = (int)floorf(m.ext_hi[1] + e) + 3 - d.dy0;
    d.dz0 = (int)floorf(m.ext_lo[2] - e); d.nz = (int)floorf(m.ext_hi[2] + e) + 3 - d.dz0;
    d.xmin = (int)floorf(m.ext_lo[0] - 2.f * e); d.xmax = (int)floorf(m.ext_hi[0] + 2.f * e) + 2;
    d.ok = m.ok && d.ny <= ECfg::NY && d.nz <= ECfg::NZ && d.ny > 0 && d.nz > 0 && (d.xmax - d.xmin) < 120;
    return d;
}
// x-window of relative row (dy, dz): every cell (x, dy, dz) - relative to R = floor(b) - that the 2 x 2 x 2 neighbourhood of ANY voxel of the
// tile can touch for ANY fractional part of b.  Voxel q touches row dy iff floor(bf_y + v_y) + {0, 1} contains dy for some bf_y in [0, 1),
// v = A q, i.e. v_y in (dy - 2, dy + 1); its x cells are floor(bf_x + v_x) + {0, 1}, inside [floor(v_x), floor(v_x) + 2].  Over the real box
// q = N v in [0, 15]^3 each axis bounds v_x given the (v_y, v_z) rectangle; the three bounds are taken independently (a superset: +10 % of
// granules against the exact union, tools/eft_plan_check.py).  Returns the granule count (0: the row is never touched).
__device__ __forceinline__ int ef_row_window(const EfMap &m, int dy, int dz, int &wlo)
{
    const float e = TRX_EF_EPS, hw = 1.5f + e, yc = (float)dy - 0.5f, zc = (float)dz - 0.5f, ex = (float)(ECfg::TX - 1);
    float lo = m.ext_lo[0] - e, hi = m.ext_hi[0] + e;
    bool feasible = true;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const float gc = m.N[a][1] * yc + m.N[a][2] * zc, gh = (fabsf(m.N[a][1]) + fabsf(m.N[a][2])) * hw;
        const float l = -gc - gh, h = ex - gc + gh, n = m.N[a][0];
        if (fabsf(n) < 1.0e-6f) {
            feasible = feasible && (l <= 1.0e-3f) && (h >= -1.0e-3f);
        } else {
            const float r = 1.0f / n;
            const float a0 = l * r, a1 = h * r;
            lo = fmaxf(lo, fminf(a0, a1)); hi = fminf(hi, fmaxf(a0, a1));
        }
    }
    if (!feasible || !(lo <= hi)) { wlo = 0; return 0; }
    wlo = (int)floorf(lo - e);
    const int whi = (int)floorf(hi + e) + 2;
    return (whi - wlo + 4) >> 2;
}

// Granule count of the plan of this theta by ONE WAVE (16 rows per lane, no barrier): the per-pair test of the step kernels, which hand
// one candidate pair to each of their eight waves.  The result is wave-uniform.
__device__ __forceinline__ int ef_plan_granules_wave(const EfMap &m, const EfDims &d, int lane)
{
    int cnt = 0;
    for (int h = 0; h < (ECfg::NY * ECfg::NZ) / 64; h++) {
        const int r = lane + h * 64, iy = r & (ECfg::NY - 1), iz = r >> 5;
        int wlo;
        if (iy < d.ny && iz < d.nz) cnt += ef_row_window(m, d.dy0 + iy, d.dz0 + iz, wlo);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    return __builtin_amdgcn_readfirstlane(cnt);
}
// theta-only part of the offer (cheap, per lane): the plan's table fits and the map is well conditioned
__device__ __forceinline__ bool ef_candidate(const float *__restrict__ th, float fD, float fH, float fW)
{
    const EfMap m = ef_map(th, fD, fH, fW);
    return ef_dims(m).ok;
}

#ifndef TRX_EF_PATCH
#define TRX_EF_PATCH 0   // 1: an XCD's columns form 8 x 4 patches of the (x, z) tile grid instead of slabs of whole x rows - measured alternative: the kernel alone on a
                         // classic grid -2 % (412 -> 405 us), inside the step's flat grid +1 ... +8 % (the per-pair column rotation balances slabs, not patches)
#endif
typedef float f4 __attribute__((ext_vector_type(4)));
typedef int i2u __attribute__((ext_vector_type(2), aligned(4)));

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
#pragma clang diagnostic ignored "-Wint-to-pointer-cast"
// Inclusive prefix sum over the 64 lanes of a wave by DPP (row shifts, then the two row broadcasts): ~10 vector instructions instead of the six LDS
// round trips of a __shfl_up ladder - the plan's two scans waited 3-4 k ticks on an LDS that the co-resident block's gather keeps busy.
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2, 3
    return v;
}

// What a thread keeps of a plan between the items of one block (round 5): the plan depends on theta alone, so a block whose consecutive items
// belong to the same pair makes it once (`replan` = false: the row table in LDS and these registers are still those of the pair).
struct EfPlanRegs {
    int G, ok;
    int geo[ECfg::K];
    unsigned goff[ECfg::K];
};
template <int MODE>
__device__ __forceinline__ void eft_body(const trx_volumes &vol, const float *__restrict__ theta, const TileGeom &tg, float *__restrict__ partials,
                                         float *lds, const int bx, const int by, const int rows_stride, const int wave_in, EfPlanRegs &pr, const bool replan)
{
    static_assert(MODE == 0 || MODE == 1 || MODE == 4, "step kernels and the moments pass");
    using C = ECfg;
    constexpr int NQ = (MODE == 0) ? 3 : (MODE == 4 ? 1 : 0);
    constexpr int NP = (MODE == 0) ? np_full(3) : (MODE == 4 ? kNpMse : 5);
    constexpr bool kGrad = MODE != 1;
    constexpr int kRows = C::Rows, K = C::K;
    constexpr bool kV2 = (TRX_EF_V2 != 0) && (TRX_EF_ROWSTEP != 0);
    constexpr int kTabShift = kV2 ? 2 : 0;   // the row table holds byte offsets (V2) or dword indices
    const int b = by;
    const int D = vol.D, H = vol.H, W = vol.W;
    const float *__restrict__ th = uni_ptr(theta + (size_t)b * TRX_PSTRIDE);
    const float *__restrict__ mov = uni_ptr(vol.moving + (size_t)b * vol.moving_stride);
    const float *__restrict__ tgt = uni_ptr(vol.target + (size_t)b * vol.target_stride);
    const float *__restrict__ xtab = uni_ptr(vol.xn), *__restrict__ ytab = uni_ptr(vol.yn), *__restrict__ ztab = uni_ptr(vol.zn);
    const int lane = trx_lane_id();
    const int wave = __builtin_amdgcn_readfirstlane(wave_in);
    const int tid = wave * 64 + lane;
#if TRX_EF_SHAPE == 1   // (measured alternative) a wave = 8 x by 8 z voxels of a row instead of 16 x by 4 z: waves 0-3 / 4-7 the two row halves, a wave's quadrant of the (x, z) face by its low two bits
    const int lx = (tid & 7) + 8 * (wave & 1), lz = ((tid >> 3) & 7) + 8 * ((wave >> 1) & 1), lh = wave / (C::Waves / C::NH);
#else
    const int lx = tid & (C::TX - 1), lz = (tid / C::TX) & (C::TZ - 1), lh = wave / (C::Waves / C::NH);
#endif
    const float fW = (float)W, fH = (float)H, fD = (float)D;
    const float hW = 0.5f * fW, hH = 0.5f * fH, hD = 0.5f * fD;
    const float t00 = th[0], t01 = th[1], t02 = th[2], t03 = th[3];
    const float t10 = th[4], t11 = th[5], t12 = th[6], t13 = th[7];
    const float t20 = th[8], t21 = th[9], t22 = th[10], t23 = th[11];
    const float sx = uni(hW * t01), sy = uni(hH * (t11 - 1.0f)), sz = uni(hD * t21);

    // column of this block (same order as tile_body: XCD-contiguous slabs of columns)
    const int ncol = tg.ntx * tg.ntz;
    const int yseg = bx / ncol, cb = bx - yseg * ncol;
    int col = cb;
    if ((ncol & 7) == 0) col = (cb & 7) * (ncol >> 3) + (cb >> 3);
    int colx = col % tg.ntx, colz = col / tg.ntx;
#if TRX_EF_PATCH
    // an XCD's columns as a compact patch of the (x, z) tile grid instead of a slab of whole x rows: the tiles its blocks work on at the
    // same time share more of their footprints' faces (and of the 128-byte lines both touch) inside one L2
    if ((tg.ntx & 7) == 0 && (tg.ntz & 3) == 0 && (ncol & 7) == 0 && ((ncol >> 3) & 31) == 0) {
        const int per = ncol >> 3, k = cb & 7, i = cb >> 3;                    // XCD k, its i-th column
        const int npx = tg.ntx >> 3, patches = per >> 5;                       // 8 x 4 patches: `patches` of them per XCD
        const int pi = k * patches + (i >> 5), ii = i & 31;
        colx = (pi % npx) * 8 + (ii & 7); colz = (pi / npx) * 4 + (ii >> 3);
    }
#endif
    const int X0 = colx * C::TX, Z0 = colz * C::TZ;
    const int nx = min(C::TX, W - X0), nz = min(C::TZ, D - Z0);
    const bool act = (lx < nx) && (lz < nz);
    const int x = X0 + (act ? lx : 0), z = Z0 + (act ? lz : 0);
    const float xn = xtab[x], zn = ztab[z];
    const float base_x = unnorm<3>(xn, fW) + hW * fmaf(t00 - 1.0f, xn, fmaf(t02, zn, t03));
    const float base_y = hH * fmaf(t10, xn, fmaf(t12, zn, t13));
    const float base_z = unnorm<3>(zn, fD) + hD * fmaf(t20, xn, fmaf(t22 - 1.0f, zn, t23));
    const float cxn = xtab[X0], czn = ztab[Z0];
    const float corner_x = uni(unnorm<3>(cxn, fW) + hW * fmaf(t00 - 1.0f, cxn, fmaf(t02, czn, t03)));
    const float corner_y = uni(hH * fmaf(t10, cxn, fmaf(t12, czn, t13)));
    const float corner_z = uni(unnorm<3>(czn, fD) + hD * fmaf(t20, cxn, fmaf(t22 - 1.0f, czn, t23)));
    const int ty_begin = yseg * tg.tiles_per_seg, ty_end = min((yseg + 1) * tg.tiles_per_seg, tg.nty);

    // ---------------- the plan (position independent): rows, packing, row table, this thread's granules ----------------
    const EfMap mp = ef_map(th, fD, fH, fW);
    const EfDims dm = ef_dims(mp);
    int *ilds = reinterpret_cast<int *>(lds);
    int *tab = ilds + 2 * C::BufFloats;
    int *desc = ilds + C::BufFloats;          // (prologue scratch in buffer 1) granule slot -> packed (iz, iy, x - xmin)
    int *wtot = ilds;                         // (prologue scratch in buffer 0) 16 chunk totals
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
#if TRX_EF_STAMP
    unsigned long long ef_stamp[6] = {0, 0, 0, 0, 0, 0}, ef_issue[2] = {0, 0};   // (ef_issue: of a tile step's request phase, the origin of the next tile / yn + eight target rows; the rest is the DMA pieces)
    const unsigned long long ef_t0 = __builtin_amdgcn_s_memtime();
#endif
#if TRX_EF_PRIO
    const int ef_second = (int)((blockIdx.y * gridDim.x + blockIdx.x) * 2 >= gridDim.x * gridDim.y);
#endif
    int w4_s, hw4_s;   // row / plane pitch of the volume in bytes, pinned in SGPRs (v_mad_u32_u24 operands)
    asm("s_mov_b32 %0, %1" : "=s"(w4_s) : "s"(W * 4));
    asm("s_mov_b32 %0, %1" : "=s"(hw4_s) : "s"(H * W * 4));
    int *org = ilds + 2 * C::BufFloats + C::TabInts;
    auto lane_origins = [&](int ty_first) {   // (wave 0 writes; the caller's barrier publishes)
        if (wave != 0) return;
        const int ty = min(ty_first + lane, tg.nty - 1);
        const float yn0 = ytab[ty * C::TY], yid0 = unnorm<3>(yn0, fH);
        const float cx = fmaf(sx, yn0, corner_x), cy = yid0 + fmaf(sy, yn0, corner_y), cz = fmaf(sz, yn0, corner_z);
        const bool sane = (fabsf(cx) < 1.0e6f) && (fabsf(cy) < 1.0e6f) && (fabsf(cz) < 1.0e6f);
        const int rx = sane ? (int)floorf(cx) : -100000, ry = sane ? (int)floorf(cy) : -100000, rz = sane ? (int)floorf(cz) : -100000;
        const int x0 = rx + dm.xmin, y0 = ry + dm.dy0, z0 = rz + dm.dz0;
        const bool interior = (x0 >= 0) && (rx + dm.xmax + 3 < W) && (y0 >= 0) && (y0 + dm.ny <= H) && (z0 >= 0) && (z0 + dm.nz <= D);
        // the plan's bounding box misses the volume: every sample of the tile is zero padding (nothing is staged or gathered)
        const bool outside = (rx + dm.xmax < 0) || (x0 >= W) || (y0 + dm.ny <= 0) || (y0 >= H) || (z0 + dm.nz <= 0) || (z0 >= D);
        int4 v;
        v.x = rx; v.y = ry; v.z = rz;
        v.w = outside ? 0 : (int)((((long long)z0 * H + y0) * W + x0) * 4);   // byte offset of the plan's corner (only dereferenced where inside the volume)
        *reinterpret_cast<int4 *>(org + lane * 8) = v;
        org[lane * 8 + 4] = (interior ? 1 : 0) | (outside ? 2 : 0);
    };
    // Round 5: the origins of the column's tiles are requested FIRST (wave 0 reads the y table: a global load whose latency the other seven waves used
    // to sit out at a barrier behind the plan) and published by the barriers of the plan, or by the one in front of the walk.
    if (dm.ok) lane_origins(ty_begin);
    if (replan) {   // (block-uniform)
    int cnt_r[2], wlo_r[2], pre_r[2];
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int r = tid + h * C::Threads, iy = r & (C::NY - 1), iz = r >> 5;
        cnt_r[h] = 0; wlo_r[h] = 0;
        if (dm.ok && iy < dm.ny && iz < dm.nz) cnt_r[h] = ef_row_window(mp, dm.dy0 + iy, dm.dz0 + iz, wlo_r[h]);
        const int p = wave_scan_incl(cnt_r[h]);   // inclusive scan inside the wave
        pre_r[h] = p;
        if (lane == 63) wtot[h * C::Waves + wave] = p;
    }
    TRX_EF_ST2(0);
    __syncthreads();
    TRX_EF_ST2(1);
    int G = 0;
    {
        int before[2] = {0, 0};
#pragma unroll
        for (int c = 0; c < 2 * C::Waves; c++) {
            const int t = wtot[c];
            if (c < wave) before[0] += t;
            if (c < C::Waves + wave) before[1] += t;
            G += t;
        }
        pre_r[0] += before[0] - cnt_r[0];   // exclusive: first granule slot of the row
        pre_r[1] += before[1] - cnt_r[1];
    }
    G = __builtin_amdgcn_readfirstlane(G);
    const bool plan_ok = dm.ok && G <= C::GCap && G > 0;
    __syncthreads();   // wtot is read; buffer 0 may be written from here on
    TRX_EF_ST2(2);
    if (plan_ok) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int r = tid + h * C::Threads, iy = r & (C::NY - 1), iz = r >> 5;
            // E = dword index of (x = 0) of this row inside a buffer; rows nobody touches point at granule 0 (never read)
            tab[iz * C::TP + iy] = (pre_r[h] * 4 - wlo_r[h]) * (1 << kTabShift);
            for (int k = 0; k < cnt_r[h]; k++) desc[pre_r[h] + k] = (iz << 20) | (iy << 12) | ((wlo_r[h] + 4 * k - dm.xmin) << 2);
        }
    }
    __syncthreads();
    TRX_EF_ST2(3);
    // this thread's granules: slot g = tid + 512 k; geo = (iz << 20) | (iy << 12) | ((x - xmin) << 2): row of the plan and x offset (bytes) from
    // its corner (one register per granule; the byte offset inside the volume is re-formed per tile: iz (H W 4) + iy (W 4) + x 4)
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int g = tid + k * C::Threads;
        pr.geo[k] = (plan_ok && g < G) ? desc[g] : 0;
    }
    // (the DMA staging needs no vector registers for the data, so the byte offsets are kept per granule instead of being re-formed per tile)
    auto goff_of = [&](int pk) -> unsigned {
        unsigned t0, t1, t2;
        asm("v_and_b32 %0, 0xffc, %1" : "=v"(t0) : "v"(pk));
        asm("v_bfe_u32 %0, %1, 12, 8" : "=v"(t1) : "v"(pk));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t2) : "v"(t1), "s"(w4_s), "v"(t0));
        asm("v_lshrrev_b32 %0, 20, %1" : "=v"(t1) : "v"(pk));
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t0) : "v"(t1), "s"(hw4_s), "v"(t2));
        return t0;
    };
#pragma unroll
    for (int k = 0; k < K; k++) pr.goff[k] = goff_of(pr.geo[k]);
    pr.G = G; pr.ok = plan_ok ? 1 : 0;
    __syncthreads();   // desc (buffer 1) is consumed
    TRX_EF_ST2(4);
    }
    const int G = __builtin_amdgcn_readfirstlane(pr.G);
    const bool plan_ok = __builtin_amdgcn_readfirstlane(pr.ok) != 0;
    const int (&geo)[K] = pr.geo;
    const unsigned (&goff)[K] = pr.goff;
#if TRX_EF_STAMP
    ef_stamp[3] = __builtin_amdgcn_s_memtime() - ef_t0;
#endif

    F1Acc acc;
#pragma unroll
    for (int q = 0; q < 3; q++)
#pragma unroll
        for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
    acc.M01 = acc.M23 = (f2)(0.f);
    acc.M4 = 0.f;
    const int j0 = lh * kRows;
    const unsigned toffb = (unsigned)((z * H + j0) * W + x) * 4u;
    const unsigned lane7b = (unsigned)(lane & (kRows - 1)) * 4u;

    if (plan_ok) {
        typedef const __attribute__((address_space(3))) f2u *lds_f2;
        typedef const __attribute__((address_space(3))) i2u *lds_i2;
        // ---- per tile: integer origin R = floor(image of the tile's corner voxel), where the plan lies against the volume, byte offset of the
        // plan's corner.  Computed for 64 tiles at a time, ONE TILE PER LANE, and fetched with v_readlane (a per-tile scalar load of the row
        // table in front of everything it feeds was ~1 us of exposed latency per tile).
        struct TileOrg { int rx, ry, rz; bool interior, outside; const char *base; int dP; };
        auto tile_org = [&](int ty) -> TileOrg {
            const int g = (ty - ty_begin) & 63;
            const int4 v = *reinterpret_cast<const int4 *>(org + g * 8);
            const int fl = org[g * 8 + 4];
            TileOrg o;
            o.rx = __builtin_amdgcn_readfirstlane(v.x); o.ry = __builtin_amdgcn_readfirstlane(v.y); o.rz = __builtin_amdgcn_readfirstlane(v.z);
            o.dP = __builtin_amdgcn_readfirstlane(v.w);
            const int f = __builtin_amdgcn_readfirstlane(fl);
            o.interior = f & 1; o.outside = (f >> 1) & 1;
            o.base = reinterpret_cast<const char *>(mov) + (o.interior ? (long long)o.dP : 0ll);
            return o;
        };
        // ---- the granules of a tile -> buffer `buf`, by LDS-DMA (global_load_lds_dwordx4: no staging registers).  Granule slot g = tid + 512 k
        // sits at byte 16 g of the buffer: the 64 lanes of a wave write 1 KB of consecutive LDS per k, which is exactly what the DMA does
        // (LDS address = M0 + 16 lane), while every lane brings its own global address (any 4-byte alignment).  All K pieces of a tile are
        // issued at once, a whole tile ahead of their use.  Boundary tiles: granules outside the volume are not fetched (exec mask) but
        // zero-filled with ds_write here and now - the target buffer is idle; one that straddles a face in x is patched element by element
        // (rare, behind a wave-uniform test).
        unsigned fixbits = 0;   // (TRX_EF_PARTFIX) per granule slot k of this thread, four bits: the elements of the tile in flight to zero once it has landed
        [[maybe_unused]] const unsigned vol_last16 = (unsigned)D * (unsigned)H * (unsigned)W * 4u - 16u;
        auto fix_tile = [&](int buf) {   // (after the wait for the tile, before the barrier that hands it to the gather)
            if (__builtin_amdgcn_ballot_w64(fixbits != 0) == 0) return;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const unsigned m = (fixbits >> (4 * k)) & 15u;
                float *p = lds + buf * C::BufFloats + (tid + k * C::Threads) * 4;
                if (m & 1u) p[0] = 0.f;
                if (m & 2u) p[1] = 0.f;
                if (m & 4u) p[2] = 0.f;
                if (m & 8u) p[3] = 0.f;
            }
            fixbits = 0;
        };
        auto issue_tile = [&](const TileOrg &o, int buf) {
            const char *bs = uni_ptr(o.base);
#pragma unroll
            for (int k = 0; k < K; k++) {
                int pk = geo[k], tt = tid;
                asm volatile("" : "+v"(pk), "+v"(tt));   // (opaque: what is derived from them is re-formed per tile, not hoisted into registers that live across the walk)
                const int g = tt + k * C::Threads;
                unsigned off = goff[k];
                bool fetch = g < G;
                if (!o.interior) {
                    const int gz = o.rz + dm.dz0 + (pk >> 20), gy = o.ry + dm.dy0 + ((pk >> 12) & 0xff), gx = o.rx + dm.xmin + ((pk & 0xfff) >> 2);
                    const bool rowin = ((unsigned)gz < (unsigned)D) && ((unsigned)gy < (unsigned)H);
                    const bool full = rowin && (gx >= 0) && (gx + 3 < W);
                    const bool part = rowin && !full && (gx + 3 >= 0) && (gx < W);
                    off = (unsigned)((int)off + o.dP);
#if TRX_EF_PARTFIX
                    // A granule that straddles a face in x and whose 16 bytes lie inside this pair's volume (all but the left overhang of the first row and the
                    // right one of the last) is fetched by the DMA like a full one - it brings the neighbouring row's elements along - and the elements
                    // outside the row are zeroed in LDS once the tile has landed (fix_tile).  Element by element with ordinary loads, as below, every such
                    // granule cost the wave a wait for EVERYTHING it had in flight: the dearest tenth of the items spent 7.1 k ticks per tile requesting
                    // the next one against 4.1 k on average (profiles/r05c_eft_item_timeline.txt).
                    const bool dma_part = fetch && part && off <= vol_last16;
                    if (dma_part) {
                        const int lo = max(0, -gx), hi = min(4, W - gx);   // elements [lo, hi) are inside the row
                        fixbits |= (unsigned)(~(((1 << hi) - 1) & ~((1 << lo) - 1)) & 15) << (4 * k);
                    }
#else
                    const bool dma_part = false;
#endif
                    if (fetch && !full && !dma_part) {
                        f4 v = (f4)(0.f);
                        if (part) {
                            const float *row = mov + ((size_t)gz * H + gy) * W;
                            v.x = ((unsigned)(gx + 0) < (unsigned)W) ? row[gx + 0] : 0.f;
                            v.y = ((unsigned)(gx + 1) < (unsigned)W) ? row[gx + 1] : 0.f;
                            v.z = ((unsigned)(gx + 2) < (unsigned)W) ? row[gx + 2] : 0.f;
                            v.w = ((unsigned)(gx + 3) < (unsigned)W) ? row[gx + 3] : 0.f;
                        }
                        *reinterpret_cast<f4 *>(lds + buf * C::BufFloats + g * 4) = v;
                    }
                    fetch = fetch && (full || dma_part);
                }
                if (TRX_EF_DBG & 1) continue;
                const unsigned long long mk = __builtin_amdgcn_ballot_w64(fetch);
                const unsigned ldsa = lds0 + (unsigned)(buf * C::BufFloats * 4) + (unsigned)((wave * 64 + k * C::Threads) * 16);
                unsigned long long sv;
                unsigned m0s;
                asm volatile("s_mov_b64 %[sv], exec\n\t"
                             "s_mov_b32 %[m0s], m0\n\t"
                             "s_mov_b32 m0, %[l0]\n\t"
                             "s_mov_b64 exec, %[k0]\n\t"
                             "s_cbranch_execz 1f\n\t"
                             "global_load_lds_dwordx4 %[o0], %[b0]" TRX_EF_DMA_POLICY "\n\t"
                             "1:\n\t"
                             "s_mov_b64 exec, %[sv]\n\t"
                             "s_mov_b32 m0, %[m0s]"
                             : [sv] "=&s"(sv), [m0s] "=&s"(m0s)
                             : [l0] "s"(ldsa), [b0] "s"(bs), [o0] "v"(off), [k0] "s"(mk)
                             : "memory");
            }
        };
        // ---- targets: ONE register per row, refilled in place - once row j of tile t is accumulated, tv[j] receives row j of tile t + 1 -
        // and the row constants of a tile (lanes 0..7 of every wave hold yn of rows j0 .. j0 + 7)
        auto issue_yn = [&](int ty, float &yn_l) {
            const unsigned yoff = (unsigned)min(ty * C::TY + j0 + (int)(lane7b >> 2), H - 1) * 4u;   // (rows past the volume repeat the last one: never accumulated)
            asm volatile("global_load_dword %0, %1, %2" : "=v"(yn_l) : "v"(yoff), "s"(ytab) : "memory");
        };
        auto issue_target = [&](int ty, int j, float &tvj) {
            const int Y0 = ty * C::TY;
            const float *trow = tgt + (size_t)min(Y0 + j, H - 1 - j0) * W;   // (a row past the volume: any valid address; never accumulated)
            if (TRX_EF_DBG & 2) { tvj = 1.f; return; }
            asm volatile("global_load_dword %0, %1, %2" TRX_TGT_POLICY : "=v"(tvj) : "v"(toffb), "s"(trow) : "memory");
        };
#if TRX_EF_ROWSTEP
        // per-row steps of the row terms (wave-uniform): yn advances by dyn per row, the un-normalised y by H / 2 * dyn (= 1 up to rounding)
        const float dyn_s = uni(H > 1 ? ytab[1] - ytab[0] : 0.f);
        const float dpx_s = uni(sx * dyn_s), dpy_s = uni((hH + sy) * dyn_s), dpz_s = uni(sz * dyn_s);
        const unsigned long long dxy2 = sgpr_pair(dpx_s, dpy_s);   // (V2) per-row step of (x, y)
#endif
        // ---- gather of rows [ja, jb) of tile `ty` from buffer `buf`
        auto gather_rows = [&](int ty, const TileOrg &o, int buf, float yn_l, float (&tv)[kRows], int ja, int jb, bool more) {
            if (TRX_EF_DBG & 4) {
                if (more) for (int j = ja; j < jb; j++) issue_target(ty + 1, j, tv[j]);
                return;
            }
            const int Y0 = ty * C::TY, ny = min(C::TY, H - Y0);
            // row terms of the coordinates, one row per lane (lanes 0..7), broadcast below: i = (per-thread base) + (row term)
            const float px_l = sx * yn_l, py_l = unnorm<3>(yn_l, fH) + sy * yn_l, pz_l = sz * yn_l;
            // coordinates relative to the plan's row origin: x to R_x, y / z to R + (dy0, dz0)
            const float bxt = base_x - (float)o.rx, byt = base_y - (float)(o.ry + dm.dy0), bzt = base_z - (float)(o.rz + dm.dz0);
            const int tab_s = (int)lds0 + 2 * C::BufFloats * 4;
            int tp_s, bufdw_s;   // table pitch and the buffer's dword index inside the LDS array, pinned in SGPRs
            asm("s_mov_b32 %0, %1" : "=s"(tp_s) : "i"(C::TP));
            asm("s_mov_b32 %0, %1" : "=s"(bufdw_s) : "s"(kV2 ? (int)lds0 + buf * C::BufFloats * 4 : (int)(lds0 >> 2) + buf * C::BufFloats));   // (V2: in bytes)
            struct S1 { int e00, e01, e10, e11, xi; float fx, fy, fz; };
            struct S2 { f2 r00, r01, r10, r11; float fx, fy, fz; };
#if TRX_EF_ROWSTEP
            // Row terms by STEPPING (round 5): the row term of row ja comes from its lane (three v_readlane per call of four rows), rows ja + 1 ..
            // add the per-row step - yn is an arithmetic progression in the row index up to one fp32 ulp (ATen's linspace table), i.e. up to
            // ~1e-5 voxels over a tile, below the fp32 resolution of the coordinates themselves; this kernel only runs rotated poses, where no
            // sample sits on the lattice by construction.  Per voxel: three adds instead of three v_readlane (4.5 cycles each) + three adds.
            float ixr = bxt + lane_bcast(px_l, ja), iyr = byt + lane_bcast(py_l, ja), izr = bzt + lane_bcast(pz_l, ja);
            float ynr = lane_bcast(yn_l, ja);
            f2 xyr = {ixr, iyr};   // (V2)
            auto stage1 = [&](int j) -> S1 {
                float ix, iy, iz = izr;
                if constexpr (kV2) {
                    ix = xyr.x; iy = xyr.y;
                    asm("v_pk_add_f32 %0, %1, %2" : "=v"(xyr) : "v"(xyr), "s"(dxy2));
                } else {
                    ix = ixr; iy = iyr;
                    ixr += dpx_s; iyr += dpy_s;
                }
                izr += dpz_s;
#else
            auto stage1 = [&](int j) -> S1 {
                const float ix = bxt + lane_bcast(px_l, j), iy = byt + lane_bcast(py_l, j), iz = bzt + lane_bcast(pz_l, j);
#endif
                int t, ta;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(t) : "v"(floor_to_int(iz)), "s"(tp_s), "v"(floor_to_int(iy)));
                asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(ta) : "v"(t), "s"(tab_s));
                S1 s;
                const i2u ea = *(lds_i2)(unsigned)ta, eb = *(lds_i2)(unsigned)(ta + C::TP * 4);
                s.e00 = ea.x; s.e01 = ea.y; s.e10 = eb.x; s.e11 = eb.y;
                if constexpr (kV2) asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(s.xi) : "v"(floor_to_int(ix)), "s"(bufdw_s));   // byte address of x inside this buffer, before the row's E
                else asm("v_add_u32 %0, %1, %2" : "=v"(s.xi) : "s"(bufdw_s), "v"(floor_to_int(ix)));   // dword index of x inside this buffer, before the row's E
                s.fx = __builtin_amdgcn_fractf(ix); s.fy = __builtin_amdgcn_fractf(iy); s.fz = __builtin_amdgcn_fractf(iz);
                return s;
            };
            auto stage2 = [&](const S1 &s) -> S2 {
                int a00, a01, a10, a11;
                if constexpr (kV2) {
                    asm("v_add_u32 %0, %1, %2" : "=v"(a00) : "v"(s.xi), "v"(s.e00));
                    asm("v_add_u32 %0, %1, %2" : "=v"(a01) : "v"(s.xi), "v"(s.e01));
                    asm("v_add_u32 %0, %1, %2" : "=v"(a10) : "v"(s.xi), "v"(s.e10));
                    asm("v_add_u32 %0, %1, %2" : "=v"(a11) : "v"(s.xi), "v"(s.e11));
                } else {
                    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(a00) : "v"(s.xi), "v"(s.e00));
                    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(a01) : "v"(s.xi), "v"(s.e01));
                    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(a10) : "v"(s.xi), "v"(s.e10));
                    asm("v_add_lshl_u32 %0, %1, %2, 2" : "=v"(a11) : "v"(s.xi), "v"(s.e11));
                }
                S2 f;
                f.r00 = *(lds_f2)(unsigned)a00; f.r01 = *(lds_f2)(unsigned)a01;
                f.r10 = *(lds_f2)(unsigned)a10; f.r11 = *(lds_f2)(unsigned)a11;
                f.fx = s.fx; f.fy = s.fy; f.fz = s.fz;
                return f;
            };
            const int je = min(jb, ny - j0);   // rows of this half that exist (uniform)
            // two rows in flight: the data reads of row j + 1 are issued before the arithmetic of row j (their table reads just before them);
            // two named register sets taken in turns (a conditional `next = ...` made the compiler copy 11 registers per row)
            auto consume = [&](const S2 &f, int j) {
                const Samp3 sm = lerp3_pairs<kGrad>(f.r00, f.r01, f.r10, f.r11, f.fx, f.fy, f.fz);
#if TRX_EF_ROWSTEP
                if (j < je) f1_accumulate_pk<MODE>(sm, tv[j], ynr, acc);   // (uniform: rows of the last, partial tile)
                ynr += dyn_s;
#else
                if (j < je) f1_accumulate_pk<MODE>(sm, tv[j], lane_bcast(yn_l, j), acc);   // (uniform: rows of the last, partial tile)
#endif
                if (more) issue_target(ty + 1, j, tv[j]);
            };
            static_assert(kRows / 2 == 4, "four rows per call");
#if TRX_EF_STAGES == 4   // (measured alternative) the table reads of all four rows of a call first
            S1 ta = stage1(ja), tb = stage1(ja + 1), tc = stage1(ja + 2), td = stage1(ja + 3);
            S2 fa = stage2(ta);
            S2 fb = stage2(tb);
            consume(fa, ja);
            fa = stage2(tc);
            consume(fb, ja + 1);
            fb = stage2(td);
            consume(fa, ja + 2);
            consume(fb, ja + 3);
#elif TRX_EF_STAGES == 3
            S1 ta = stage1(ja);
            S1 tb = stage1(ja + 1);
            S2 fa = stage2(ta);
            ta = stage1(ja + 2);
            S2 fb = stage2(tb);
            consume(fa, ja);
            tb = stage1(ja + 3);
            fa = stage2(ta);
            consume(fb, ja + 1);
            fb = stage2(tb);
            consume(fa, ja + 2);
            consume(fb, ja + 3);
#else
            S2 fa = stage2(stage1(ja));
            S2 fb = stage2(stage1(ja + 1));
            consume(fa, ja);
            fa = stage2(stage1(ja + 2));
            consume(fb, ja + 1);
            fb = stage2(stage1(ja + 3));
            consume(fa, ja + 2);
            consume(fb, ja + 3);
#endif
        };

        // ---------------- the column walk: tile t gathered from buffer t & 1 while tile t + 1 lands in the other one ----------------
        // Tiles whose plan misses the volume altogether (warped = 0, gradient 0: only the target's moments count) form a prefix and a suffix
        // of the column - the plan's box moves along a straight line, and it meets the volume's box in one interval of tiles - and are
        // taken out of the pipelined walk: [ty_begin, t0) and [t1, ty_end) run the plain loop below, [t0, t1) the walk.
        __syncthreads();
        TRX_EF_ST2(5);
        int t0 = ty_begin, t1 = ty_end;
        if (ty_end - ty_begin <= 64) {   // (every lane reads its tile's flags: one LDS read and a ballot instead of a tile_org per candidate)
            const int fl = org[min(lane, ty_end - ty_begin - 1) * 8 + 4];
            const unsigned long long valid = (ty_end - ty_begin) == 64 ? ~0ull : ((1ull << (ty_end - ty_begin)) - 1ull);
            const unsigned long long in = ~__builtin_amdgcn_ballot_w64((fl >> 1) & 1) & valid;
            if (in == 0) t0 = t1 = ty_end;
            else { t0 = ty_begin + __builtin_ctzll(in); t1 = ty_begin + 64 - __builtin_clzll(in); }
        } else {
            while (t0 < ty_end && tile_org(t0).outside) t0++;
            while (t1 > t0 && tile_org(t1 - 1).outside) t1--;
        }
        t0 = __builtin_amdgcn_readfirstlane(t0); t1 = __builtin_amdgcn_readfirstlane(t1);
        auto target_only = [&](int ta, int tb) {   // (the rows of a tile are requested together: eight loads in flight, not one)
            for (int ty = ta; ty < tb; ty++) {
                const int Y0 = ty * C::TY, nrow = min(C::TY, H - Y0) - j0;
                float yv[kRows];
#pragma unroll
                for (int j = 0; j < kRows; j++) yv[j] = tgt[(size_t)(z * H + min(Y0 + j0 + j, H - 1)) * W + x];   // (rows past the volume repeat the last one: not accumulated)
#pragma unroll
                for (int j = 0; j < kRows; j++) {
                    if (j < nrow) {
                        if constexpr (MODE == 4) acc.M4 = fmaf(yv[j], yv[j], acc.M4);
                        else { acc.M01.x += yv[j]; acc.M23.x = fmaf(yv[j], yv[j], acc.M23.x); }
                    }
                }
            }
        };
        // Vector-memory operations of a wave per tile, in program order:  YN' T'0 .. T'7 D0 .. D4  (' = of the next tile; T = the targets, into
        // the OTHER of two register sets, D = the DMA pieces): everything a tile needs is requested at the START of the tile before it, so the
        // one wait per tile finds it landed.  (Refilling the targets in place, row by row, left the last target load a few hundred cycles
        // old at the wait: the waves of a block were parked there half of their time - SQ_WAIT_ANY 49 %.)
        float tvA[kRows], tvB[kRows], ynA = 0.f, ynB = 0.f;
#pragma unroll
        for (int j = 0; j < kRows; j++) tvA[j] = tvB[j] = 0.f;
        TileOrg cur = tile_org(min(t0, ty_end - 1));
        if (t0 < t1) {
            issue_yn(t0, ynA);
#pragma unroll
            for (int j = 0; j < kRows; j++) issue_target(t0, j, tvA[j]);
            issue_tile(cur, 0);
            TRX_EF_ST2(7);
        }
        target_only(ty_begin, t0);   // (the tiles in front of the volume, while the first tile of the walk is on its way)
        TRX_EF_ST2(6);
        if (t0 < t1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fix_tile(0);
#pragma unroll
            for (int j = 0; j < kRows; j++) asm volatile("" : "+v"(tvA[j]));
            asm volatile("" : "+v"(ynA));
        }
        __syncthreads();
#if TRX_EF_STAMP
        const unsigned long long ef_first = __builtin_amdgcn_s_memtime() - ef_t0;
#endif
        auto tile_step = [&](int ty, int par, float (&use)[kRows], float &yn_use, float (&load)[kRows], float &yn_load) {
#if TRX_EF_PRIO
            // fair sharing of a CU between its two blocks (see TRX_ZS_PRIO in affine_zstream.h): they take turns at the higher priority in time slices
            if (((__builtin_amdgcn_s_memtime() >> TRX_EF_PRIO_BIT) + ef_second) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
#if TRX_EF_STAMP
            const unsigned long long es0 = __builtin_amdgcn_s_memtime();
#endif
            const bool more = ty + 1 < t1;
            TileOrg nxt = cur;
            if (more) {
                if (((ty + 1 - ty_begin) & 63) == 0) {   // (columns of more than 64 tiles: the next chunk of origins)
                    __syncthreads();
                    lane_origins(ty + 1);
                    __syncthreads();
                }
                nxt = tile_org(ty + 1);
#if TRX_EF_ISSUE_PRIO
                __builtin_amdgcn_s_setprio(TRX_EF_ISSUE_PRIO);
#endif
#if TRX_EF_STAMP
                const unsigned long long ei0 = __builtin_amdgcn_s_memtime();
#endif
                issue_yn(ty + 1, yn_load);
#pragma unroll
                for (int j = 0; j < kRows; j++) issue_target(ty + 1, j, load[j]);
#if TRX_EF_STAMP
                const unsigned long long ei1 = __builtin_amdgcn_s_memtime();
                ef_issue[0] += ei0 - es0; ef_issue[1] += ei1 - ei0;
#endif
                issue_tile(nxt, par ^ 1);
#if TRX_EF_ISSUE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#if TRX_EF_STAMP
            const unsigned long long es1 = __builtin_amdgcn_s_memtime();
#endif
            gather_rows(ty, cur, par, yn_use, use, 0, kRows / 2, false);
            gather_rows(ty, cur, par, yn_use, use, kRows / 2, kRows, false);
#if TRX_EF_STAMP
            const unsigned long long es2 = __builtin_amdgcn_s_memtime();
#endif
            if (more) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < kRows; j++) asm volatile("" : "+v"(load[j]));
                asm volatile("" : "+v"(yn_load));
                fix_tile(par ^ 1);
            }
            cur = nxt;
            __syncthreads();
#if TRX_EF_STAMP
            const unsigned long long es3 = __builtin_amdgcn_s_memtime();
            ef_stamp[0] += es1 - es0; ef_stamp[1] += es2 - es1; ef_stamp[2] += es3 - es2; ef_stamp[5] += 1;
#endif
        };
        {
            int ty = t0;
            while (ty < t1) {
                tile_step(ty, 0, tvA, ynA, tvB, ynB);
                ty++;
                if (ty >= t1) break;
                tile_step(ty, 1, tvB, ynB, tvA, ynA);
                ty++;
            }
        }
#if TRX_EF_STAMP
        const unsigned long long ef_walk = __builtin_amdgcn_s_memtime() - ef_t0;
#endif
        target_only(t1, ty_end);
#if TRX_EF_STAMP
        ef_stamp[4] = __builtin_amdgcn_s_memtime() - ef_t0;   // (tail done)
        if (trx_lane_id() == 0) {
            unsigned long long *o = trx_ef_stamps + ((size_t)((by * rows_stride + bx) & 1023) * 8 + wave) * 8;
            o[6] = ef_first; o[7] = ef_walk;
        }
#endif
#if TRX_EF_PRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    } else {
        // the plan does not fit (dual_choice tests the same numbers, so this is a safety net): every voxel gathers from global memory
        for (int ty = ty_begin; ty < ty_end; ty++) {
            const int Y0 = ty * C::TY, ny = min(C::TY, H - Y0);
            for (int j = j0; j < min(j0 + kRows, ny); j++) {
                const float yn = ytab[Y0 + j];
                const float ix = fmaf(sx, yn, base_x), iy = unnorm<3>(yn, fH) + fmaf(sy, yn, base_y), iz = fmaf(sz, yn, base_z);
                const Samp3 sm = sample3_padded(mov, D, H, W, ix, iy, iz);
                const float yv = tgt[(size_t)(z * H + Y0 + j) * W + x];
                f1_accumulate_pk<MODE>(sm, yv, yn, acc);
            }
        }
        __syncthreads();
    }
    if (!act) {
#pragma unroll
        for (int q = 0; q < 3; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) acc.AB[q][c] = (f2)(0.f);
        acc.M01 = acc.M23 = (f2)(0.f);
        acc.M4 = 0.f;
    }
    float vals[NP];
    int o = 0;
    if constexpr (MODE == 4) {
        vals[0] = acc.M4;
        o = 1;
    } else {
        vals[0] = acc.M01.x; vals[1] = acc.M01.y; vals[2] = acc.M23.x; vals[3] = acc.M23.y; vals[4] = acc.M4;
        o = 5;
    }
    {
        int tt = tid;   // (re-formed from the thread id: not kept live across the walk)
        asm volatile("" : "+v"(tt));
#if TRX_EF_SHAPE == 1
        const int lx2 = (tt & 7) + 8 * (wave & 1), lz2 = ((tt >> 3) & 7) + 8 * ((wave >> 1) & 1);
#else
        const int lx2 = tt & (C::TX - 1), lz2 = (tt / C::TX) & (C::TZ - 1);
#endif
        const int xx = X0 + ((lx2 < nx && lz2 < nz) ? lx2 : 0), zz = Z0 + ((lx2 < nx && lz2 < nz) ? lz2 : 0);
        const float xn_e = xtab[xx], zn_e = ztab[zz];
#pragma unroll
        for (int q = 0; q < NQ; q++)
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float a = acc.AB[q][c].x;
                vals[o++] = xn_e * a; vals[o++] = acc.AB[q][c].y; vals[o++] = zn_e * a; vals[o++] = a;
            }
    }
    block_reduce_store_nw<NP, C::Waves>(vals, partials + ((size_t)by * rows_stride + bx) * NP, lds, wave);
#if TRX_EF_STAMP
    if (trx_lane_id() == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = trx_ef_stamps + ((size_t)((by * rows_stride + bx) & 1023) * 8 + wave) * 8;
        for (int k = 0; k < 4; k++) o[k] = ef_stamp[k];
        o[0] |= ef_issue[0] << 32; o[1] |= ef_issue[1] << 32;   // (sums of an item fit 32 bits)
        o[4] = (ef_stamp[4] & 0xffffffffull) | ((__builtin_amdgcn_s_memtime() - ef_t0) << 32);   // tail done | all done
        o[5] = ef_stamp[5] | ((unsigned long long)(hwid & 0xffff) << 16) | ((unsigned long long)(xcc & 0xf) << 32);
    }
#endif
}

// stand-alone kernel (tools/ebench.hip; the library runs the body inside affine_tile_dual_kernel)
template <int MODE>
__global__ __launch_bounds__(ECfg::Threads, 4) void affine_eft_kernel(trx_volumes vol, const float *__restrict__ theta, TileGeom tg, float *__restrict__ partials)
{
    __shared__ __attribute__((aligned(16))) float lds[ECfg::Alloc];
    if ((int)blockIdx.x >= tg.blocks_per_pair) return;
    EfPlanRegs pr;
    eft_body<MODE>(vol, theta, tg, partials, lds, blockIdx.x, blockIdx.y, gridDim.x, trx_wave_index(), pr, true);
}
#pragma clang diagnostic pop
