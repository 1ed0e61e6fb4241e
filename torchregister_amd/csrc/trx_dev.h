// Two kinds of macros of the tile kernels live here:
//  * compile-time ALTERNATIVES and tuning constants (first block): their defaults ARE the product; the other values are the measured
//    alternatives DESIGN.md quotes (tools/kbench.hip, tools/lib_variants.sh build them with -D...);
//  * development INSTRUMENTATION (second block: TRX_DEV builds of tools/kbench.hip, -DTRX_TIMING=1 / -DTRX_LDS_PAD=n): in the product
//    library every one of those expands to nothing, so the kernels in affine.hip read without #if blocks.
#pragma once

// ---- compile-time alternatives of the tile kernels: the defaults ARE the product; the others are the measured alternatives DESIGN.md
// quotes (tools/kbench.hip, tools/lib_variants.sh build them with -D...)
#ifndef TRX_TILE_CFG
#define TRX_TILE_CFG 0
#endif
#ifndef TRX_GEOMA_BD
#define TRX_GEOMA_BD 14   // (13 = 52.6 KB: three blocks per CU fit the LDS; measured with TRX_TILE_MIN_WAVES=6)
#endif
#ifndef TRX_GEOMR_BIG
#define TRX_GEOMR_BIG 1
#endif
#ifndef TRX_GEOM_MODEL
#define TRX_GEOM_MODEL 1   // y-split of the tile columns from the occupancy model in tile_geom (0: the round-1 rules - measured alternative)
#endif
#ifndef TRX_DBG_SKIP
#define TRX_DBG_SKIP 0   // development ablation (tools/kbench.hip): 1 = no gather/compute, 2 = no box staging, 3 = no target loads
#endif
#ifndef TRX_SWP_BARRIER
#define TRX_SWP_BARRIER 0
#endif
#ifndef TRX_DMA_SPREAD
#define TRX_DMA_SPREAD 1   // cfg 1: issue the next tile's DMA pieces between the rows of the gather (0: all at once before it)
#endif
#ifndef TRX_TGT_POLICY
#define TRX_TGT_POLICY ""   // cache policy suffix of the target loads (development)
#endif
#ifndef TRX_ZS_TGT_POLICY
#define TRX_ZS_TGT_POLICY " nt"   // the z-streaming body's target loads: read once per launch, kept out of the moving planes' way (+1 % on the headline; the same hint on the exact-footprint body costs 6 %, on the tile kernels nothing)
#endif
#ifndef TRX_ZS_RING_POLICY
#define TRX_ZS_RING_POLICY ""    // cache policy suffix of the z-streaming body's ring DMA (development: " nt", " sc1")
#endif
#ifndef TRX_EF_DMA_POLICY
#define TRX_EF_DMA_POLICY ""     // ... of the exact-footprint kernel's granule DMA
#endif
#ifndef TRX_BOX_POLICY
#define TRX_BOX_POLICY ""   // cache policy suffix of the box DMA (development: " nt", " sc1")
#endif
#ifndef TRX_DUAL_DEFAULT
#define TRX_DUAL_DEFAULT 1   // rigid steps, loss-only, warp and warp-backward launches pick GeomA / GeomR per pair: 1 in one kernel, 2 as two launches; 0: never
#endif
#ifndef TRX_DMA_EXECZ_SKIP
#define TRX_DMA_EXECZ_SKIP 1   // branch over a DMA piece none of whose lanes fetch (an LDS-DMA with exec = 0 still costs its issue)
#endif
#ifndef TRX_DEEP_TILE
#define TRX_DEEP_TILE 1   // the step kernels carry GeomD (deep tile) as a third per-pair choice (0: GeomA / GeomR only - measured alternative)
#endif
#ifndef TRX_DEEP_SMALL
#define TRX_DEEP_SMALL 1   // GeomD also for small batches whose deep tiling fills the block slots (0: only from 1024 blocks - measured alternative)
#endif
#ifndef TRX_ROT_DEEP_TILE
#define TRX_ROT_DEEP_TILE 1   // the step kernels carry GeomRD as a fourth per-pair choice (0: never - measured alternative)
#endif
#ifndef TRX_EFT_BODY
#define TRX_EFT_BODY 1   // the step kernels carry the exact-footprint body for rotated pairs (0: GeomR as before - measured alternative)
#endif
#ifndef TRX_EFT_MERGED
#define TRX_EFT_MERGED 0   // 1: behind the z-streaming kernel the exact-footprint body rides in the tile kernel as a fifth body (one kernel and one launch boundary less per
                           // step) - measured alternative: the headline does not move (30.0-30.4 k against 30.1-30.3 k: the empty launch hides behind the dispatch of the next),
                           // the rotated poses lose 5.5 % (17.2 against 18.1 k: 43 spilled registers in the merged kernel); profiles/r05a_eft_merged.txt
#endif
#ifndef TRX_CARRY
#define TRX_CARRY 1   // trx_affine_run: launch-bound 3-D steps (the two-body GeomA / GeomR kernel on the classic grid) as ONE launch per iteration - the finalise of iteration
                      // k in the prologue of iteration k + 1's kernel (CarryKArgs in affine.hip); 0: a step kernel and a finalise kernel per iteration (measured alternative)
#endif
#ifndef TRX_CARRY_BATCH
#define TRX_CARRY_BATCH 16   // carry prologue: partial rows a lane has in flight per batch (32: measured alternative, +-0: profiles/r06b_carry.txt)
#endif
#ifndef TRX_PERSISTENT_BLOCKS
#define TRX_PERSISTENT_BLOCKS 512   // block slots for the 512-thread step kernels if the device cannot be queried (MI355X: 2 x 256 CUs); persistent_blocks() asks the device
#endif
#ifndef TRX_FLAT_GRID
#define TRX_FLAT_GRID 1             // 0: big batches launch (largest geometry) x (pairs) blocks like the small ones (measured alternative)
#endif
#ifndef TRX_ZS_MIN_BLOCKS
#define TRX_ZS_MIN_BLOCKS 200   // the z-streaming body is offered to launches of at least this many of its blocks (measured, profiles/r03f_zstream_small_batches.txt:
                                // 216 blocks - 2 x 192^3 - gain 20 %, 128 blocks - 4 x 128^3 - lose a factor of two) ...
#endif
#ifndef TRX_ZS_MIN_PLANES
#define TRX_ZS_MIN_PLANES 32    // ... of at least this many planes each (a block pays ~7 planes of pipeline fill; zs_geom never cuts segments shorter)
#endif
#ifndef TRX_SWP
#define TRX_SWP 1   // software pipeline of the gather: LDS reads of row j+1 issued before the arithmetic of row j (0: at use)
#endif
#ifndef TRX_STAGE_PRIO
#define TRX_STAGE_PRIO 3
#endif
#ifndef TRX_TILE_MIN_WAVES
#define TRX_TILE_MIN_WAVES 4   // waves per SIMD the register allocator must allow (16 waves per CU)
#endif
#ifndef TRX_DUAL_MIN_WAVES
#define TRX_DUAL_MIN_WAVES 4   // the dual kernel's LDS (GeomR's 78.6 KB box) allows two blocks per CU whatever the registers
#endif

// ---- instrumentation

#ifndef TRX_TIMING
#define TRX_TIMING 0      // 1: per-block staging / barrier / gather cycle counts of the tile kernel's fast loop into trx_timing[]
#endif
#ifdef TRX_LDS_PAD        // inflate the primary tile kernel's LDS footprint (floats) to force one block per CU
#define TRX_DEV_LDS_PAD (TRX_LDS_PAD)
#else
#define TRX_DEV_LDS_PAD 0
#endif

#if TRX_TIMING
namespace trx {
__device__ unsigned long long trx_timing[4 * 8192];
}
#define TRX_TM_INIT() unsigned long long tm_acc[4] = {0, 0, 0, 0} /* wave 0: stage-wait, barrier-1 wait, gather, barrier-2 wait */
#define TRX_TM_STAMP(name) const unsigned long long name = __builtin_amdgcn_s_memtime()
#define TRX_TM_TILE_DONE()                                                                                              \
    do {                                                                                                                \
        const unsigned long long tm4 = __builtin_amdgcn_s_memtime();                                                   \
        tm_acc[0] += tm1 - tm0; tm_acc[1] += tm2 - tm1; tm_acc[2] += tm3 - tm2; tm_acc[3] += tm4 - tm3;                 \
    } while (0)
#define TRX_TM_STORE()                                                                                                  \
    do {                                                                                                                \
        if (tid == 0) {                                                                                                 \
            const size_t bi = (size_t)by * gridDim.x + bx;                                                              \
            if (bi < 8192) for (int i = 0; i < 4; i++) trx::trx_timing[bi * 4 + i] = tm_acc[i];                        \
        }                                                                                                               \
    } while (0)
#else
#define TRX_TM_INIT() ((void)0)
#define TRX_TM_STAMP(name) ((void)0)
#define TRX_TM_TILE_DONE() ((void)0)
#define TRX_TM_STORE() ((void)0)
#endif

#ifndef TRX_DUAL_PINGPONG
#define TRX_DUAL_PINGPONG 1   // the tile kernel behind the z-streaming kernel takes its work list backwards on odd iterations of a run (TRX_FLAG_WALK_DOWN)
#endif
