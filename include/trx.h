/*
 * trx.h — C ABI of the MI355X-native TorchRegister hot path (libtrx.so).
 *
 * The reference (AgamChopra/TorchRegister v0.2.3, ref: = src/TorchRegister/) is pure Python and
 * has NO plugin / operator / FFI interface; its hot path is a chain of ATen ops inside three
 * Python loops.  These entry points are what a binding for that path would call instead; each
 * one names the reference code it replaces.  All pointers are DEVICE pointers unless marked
 * [host]; `stream` is a hipStream_t passed as void*.  No entry point allocates, synchronises
 * or calls back into the host: everything is enqueued on `stream` and is graph-capturable.
 * Return value: 0 (TRX_OK) or a negative trx_status; nothing throws.
 * Thread-compatible: no global state (the library reads no environment variable and keeps no settings: every choice a
 * caller can make travels in the argument structs), one host thread per GPU/stream may call concurrently.
 *
 * Data layout (fp32, contiguous): volumes [B][D][H][W] (2-D: D = 1, ndim = 2), pair p of a
 * batch at base + p * stride elements (stride 0 = one volume shared by all pairs).
 * theta: row-major [ndim][ndim+1] per pair, row 0 -> x (W axis), row 1 -> y (H), row 2 -> z (D),
 * exactly F.affine_grid's convention; per-pair small vectors are padded to TRX_PSTRIDE floats.
 * flow: [B][ndim][D][H][W], channel i displaces along spatial dim i in voxel units
 * (ref:utils.py:343-351).
 */
#ifndef TRX_H
#define TRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history (trx_version() returns TRX_VERSION; the first two digits follow the reference package version whose API the host layer mirrors):
 *   200  round 2: flags in trx_volumes (no environment variables), trx_nmi_from_pdfs, trx_theta_chain, the flow loop's device-side early stop
 *   220  round 3: TRX_FLAG_ZSTREAM / NO_ZSTREAM, TRX_FLAG_NEAREST, trx_flow_lncc_run, trx_peer_* (first form)
 *   230  round 4: TRX_FLAG_SAVE_LAST (slab updates store flow_last only on request), TRX_FLAG_EFT / TRX_FLAG_NO_EFT, trx_peer_alloc / export /
 *        import on raw HIP IPC handles, sticky peer time-outs
 *   231  round 5: trx_affine_workspace_bytes grows by three ints (the z-streaming kernel's note to the kernels behind it), the rows_used array the
 *        step kernels leave in the workspace carries the kernel body in bits 24-27, TRX_FLAG_ZS_FUSED.  No entry point changed its signature.
 *   232  round 5: trx_affine_workspace_bytes reserves eight more ints (work tickets of the exact-footprint kernel, zeroed by the z-streaming kernel in
 *        front of it on every launch: the workspace still needs no initialisation by the caller).  No entry point changed its signature.
 *   240  round 6: TRX_FLAG_ONE_KERNEL (a step of a chip-filling launch next to the identity is ONE streaming launch + the finalise), trx_affine_near_identity;
 *        trx_affine_run folds the finalise of an iteration into the next iteration's kernel for launch-bound 3-D steps (TRX_FLAG_NO_CARRY keeps two launches);
 *        trx_affine_workspace_bytes grows by a second partial / note buffer and two carry buffers (3-D).  No entry point changed its signature. */
#define TRX_VERSION 240
#define TRX_PSTRIDE 12  /* floats per pair in theta / param / adam / best_theta arrays */

typedef enum {
    TRX_OK = 0,
    TRX_ERR_ARG = -1,       /* null pointer / non-positive size / bad enum / B > 65535 / a volume of >= 2^31 voxels */
    TRX_ERR_NDIM = -2,      /* ndim not 2 or 3 (or D != 1 with ndim 2) */
    TRX_ERR_WORKSPACE = -3, /* workspace too small: see trx_*_workspace_bytes */
    TRX_ERR_HIP = -4,       /* a HIP launch failed (hipGetLastError) */
    TRX_ERR_CAPACITY = -5   /* trx_*_run: `iters` exceeds losses_capacity (the per-pair device counters cannot be read without a
                               sync, so callers that split a run over several calls keep the running total themselves) */
} trx_status;

/* trx_volumes.flags: per-call path selection (0 = the library picks; the others exist so that every path can be tested
 * against the others through the same entry points - results never depend on the path beyond fp32 rounding). */
#define TRX_FLAG_GATHER_PATH 1u    /* affine entry points: the un-tiled row-walking kernel instead of the LDS-tiled ones */
#define TRX_FLAG_SINGLE_GEOM 2u    /* affine entry points: one tile geometry for every pair (no per-pair choice among GeomD / GeomA / GeomRD / GeomR) */
#define TRX_FLAG_TWO_PASS_FLOW 4u  /* trx_flow_run: keep the moments pass of every iteration (no fusion into the previous update) */
#define TRX_FLAG_DEEP_TILE 8u      /* affine steps: offer the deep tiles (GeomD, GeomRD) to every pair they fit, whatever the batch and volume size
                                      (by default only where their larger tiles still fill the chip) */
#define TRX_FLAG_SAVE_LAST 256u      /* trx_flow_slab_update[_fused]: also store the flow this update starts from into st->flow_last (+12 B/voxel written);
                                      callers set it on the LAST iteration of a run - the iteration that meets stop_crit stores it by itself */
#define TRX_FLAG_NEAREST 128u        /* trx_flow_warp: nearest-neighbour sampling (SpatialTransformer(mode='nearest'), ref:utils.py:339-365) instead of bi/trilinear */
#define TRX_FLAG_NO_ZSTREAM 32u     /* affine steps: never use the z-streaming body (pairs next to the identity run GeomD / GeomA like the others) */
#define TRX_FLAG_ZSTREAM 64u        /* affine steps: offer the z-streaming body whatever the batch size (by default only to launches that fill the chip).
                                     * The body re-checks its window per block from the block's own corners and would publish NaN rows for a pair whose
                                     * pre-image left it; the offer rule (theta and sizes only) is that test's worst case with 0.3 voxels to spare, and
                                     * tests/test_gpu_zstream.py::test_zstream_offer_boundary_never_trips bisects to the rule's edge in 60 directions of
                                     * theta space on three shapes without reaching it: a NaN loss from this path would be a bug, not an input property */
#define TRX_FLAG_NO_EFT 512u        /* affine steps: never use the exact-footprint body (rotated pairs run GeomR as before) */
#define TRX_FLAG_EFT 1024u          /* affine steps: offer the exact-footprint body whatever the batch size (by default only to launches that fill the chip) */
#define TRX_FLAG_ZS_FUSED 2048u      /* affine steps on launches that fill the chip: keep the z-streaming body inside the tile kernel (the round 3-4 form) instead of
                                      * running it as a kernel of its own in front (measured alternative; tests compare the two) */
#define TRX_FLAG_WALK_DOWN 8192u      /* affine steps: the z-streaming kernel walks its columns from the last plane to the first (same sums up to rounding); the exact-footprint
                                      * and tile kernels behind it take the pairs in the opposite order (same sums bit for bit).  trx_affine_run
                                      * toggles this bit on every second iteration (unless TRX_FLAG_NO_PINGPONG): the planes (pairs) a pass read last are the ones the next pass
                                      * starts with, and the 256 MiB Infinity Cache still holds them.  Callers that step one iteration per call may alternate it themselves */
#define TRX_FLAG_NO_PINGPONG 16384u   /* trx_affine_run: every iteration walks in the direction the caller's flags say (measured alternative) */
#define TRX_FLAG_NO_ZS_FLAT 4096u     /* affine steps: the z-streaming kernel never uses its flat 64 x 16 tile (pairs beyond the 64 x 32 tile's window run the tile kernels) */
#define TRX_FLAG_ONE_KERNEL 32768u     /* affine steps of launches that fill the chip: the caller EXPECTS every pair to stay next to the identity (inside the z-streaming kernel's
                                      * windows: |theta - I| up to ~0.03 on its 64 x 32 tile, rotations to ~0.15 rad / zooms to 1.15 on its flat tile), so the step launches that
                                      * kernel ALONE - no exact-footprint kernel, no tile kernel behind it (two launches that find nothing to do: 3.2-3.5 us per step,
                                      * profiles/r06a_one_kernel_ab.txt - worth it up to ~4 x 256^3 voxels per launch).  Always correct: a pair that leaves the windows is run by the same kernel
                                      * on GeomR's body, at about twice its usual cost while it stays outside - a hint about speed, never about results (fp32 rounding between
                                      * bodies as with every other path flag).  trx_affine_near_identity() evaluates the expectation for thetas the caller holds on the host;
                                      * torchregister_amd.AffineSolver sets the flag by itself from the initial thetas and from the bodies the previous run() call ended on */
#define TRX_FLAG_NO_CARRY 65536u      /* trx_affine_run: every iteration is a step kernel plus a finalise kernel, also where the run would fold the finalise into the next
                                      * iteration's kernel (launch-bound 3-D registrations: one pair up to ~128^3) - measured alternative; results differ by fp32 rounding at most
                                      * (the folded finalise sums a pair's partial rows in another fixed order) */
#define TRX_FLAG_NO_ROT_DEEP_TILE 16u /* affine steps: never use GeomRD (the 16 x 16 x 16 tile in GeomR's box) - rotated pairs all run GeomR */

/* A batch of B independent (moving, target) pairs. */
typedef struct {
    const float *moving;   /* [B][D][H][W] */
    const float *target;   /* [B][D][H][W] (may be NULL for warp-only calls) */
    size_t moving_stride;  /* elements between pairs */
    size_t target_stride;
    int ndim, B, D, H, W;
    /* Optional base-coordinate tables of affine_grid(align_corners=False): xn[W], yn[H], zn[D] =
     * fp32 linspace(-1,1,S)*(S-1)/S as ATen builds them.  NULL -> closed form (2i+1)/S-1. */
    const float *xn, *yn, *zn;
    unsigned flags;        /* TRX_FLAG_* (0 for normal use) */
} trx_volumes;

/* L = w_mse*MSE + w_ncc*ncc_alpha*(1-NCC) + w_ssd*ssd_alpha*SSD
 * replaces nn.MSELoss call sites (ref:warpings.py:37,39,124,126,179), NCCLoss.forward
 * (ref:utils.py:197-205, EPSILON=1e-10), SSDLoss.forward (ref:utils.py:218-221) and the weighted
 * sum (ref:warpings.py:78-79,144-145,213-214).  Argument order (target, warped) as in the loops. */
typedef struct {
    float w_mse, w_ncc, ncc_alpha, w_ssd, ssd_alpha;
} trx_loss_cfg;

typedef enum { TRX_OPT_SGD = 0, TRX_OPT_ADAM = 1 } trx_opt_kind;
/* SGD replaces torch.optim.SGD(params, lr) (ref:warpings.py:58,131,192); Adam is an extension
 * (torch.optim.Adam defaults beta=(0.9,0.999), eps=1e-8, bias-corrected). */
typedef struct {
    int kind;
    float lr, beta1, beta2, eps;
} trx_opt_cfg;

typedef enum { TRX_PARAM_AFFINE = 0, TRX_PARAM_RIGID = 1 } trx_param_mode;

/* Per-pair optimisation state, all device memory, all [B][TRX_PSTRIDE] unless noted.
 * AFFINE: param IS theta (ref:warpings.py:42-55,70-74: the regressor MLP is dead, theta == its
 *         bias).  RIGID: param is the pose vector of Theta (ref:utils.py:287-310,
 *         6 floats in 3-D, 3 in 2-D) and theta = Theta(param). */
typedef struct {
    int mode;           /* trx_param_mode */
    float *param;       /* in/out: optimised parameters */
    float *theta;       /* in/out: theta of the NEXT forward (caller initialises consistently) */
    float *adam_m;      /* in/out, may be NULL for SGD */
    float *adam_v;
    float *best_theta;  /* out: theta of the first strict loss minimum (ref:warpings.py:85-93) */
    float *best_loss;   /* [B] out */
    int *best_idx;      /* [B] out */
    float *losses;      /* [B][losses_capacity] out: L_t per iteration (ref:warpings.py:83) */
    int losses_capacity;
    int *step;          /* [B] in/out: iteration counter t (device-resident; 0 before the run) */
    float *grad;        /* optional out [B][TRX_PSTRIDE]: dL/dparam of the last step (may be NULL) */
} trx_affine_state;

int trx_version(void);
const char *trx_status_string(int status);

/* Bytes of scratch the affine entry points need for this batch geometry. */
size_t trx_affine_workspace_bytes(const trx_volumes *vol /*[host]*/);

/* Diagnostics: byte offset, inside the affine workspace, of int rows_used[B] - the number of partial rows the last 3-D step's streaming
 * launch wrote for each pair.  The step kernels pick a kernel body per pair from theta (tile geometry / z-streaming) and each body has its
 * own row count, so this tells a test or a profiler which body ran; the step's own reduction reads the same array. */
size_t trx_affine_workspace_rows_offset(const trx_volumes *vol /*[host]*/);

/* [host] 1 if a step of this batch at the given thetas (HOST memory, [B][TRX_PSTRIDE]) would run entirely on the z-streaming kernel's tiles - the
 * test that kernel evaluates per pair on the device - i.e. if TRX_FLAG_ONE_KERNEL costs nothing at these poses; 0 otherwise (also for 2-D and
 * for batches the z-streaming kernel is not offered).  No device work. */
int trx_affine_near_identity(const trx_volumes *vol /*[host]*/, const float *theta_host);

/* ONE optimiser iteration for all B pairs: fused forward warp + loss + analytic backward
 * (single pass over moving/target), then loss/gradient/optimiser/best-tracking on device.
 * Replaces one trip of the loop bodies ref:warpings.py:67-93 (affine) / :138-159 (rigid):
 * get_affine_warp (ref:warpings.py:18-26) -> criterions -> error.backward() -> optimizer.step()
 * -> losses_train.append(error.item()) -> best tracking. */
int trx_affine_step(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                    const trx_affine_state *st, void *workspace, size_t workspace_bytes, void *stream);

/* Pass 1 of trx_affine_step alone (the streaming kernel: per-block partial sums into `workspace`,
 * no finalise, no state change).  Exposed so that the dominant kernel can be timed in isolation
 * (bench.py roofline leg) and for callers that want to overlap their own finalise. */
int trx_affine_accumulate(const trx_volumes *vol, const float *theta, void *workspace, size_t workspace_bytes, void *stream);

/* `iters` iterations enqueued back to back (no host sync in between; replaces the whole loop). */
int trx_affine_run(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                   const trx_affine_state *st, int iters, void *workspace, size_t workspace_bytes, void *stream);

/* Loss only (forward), theta untouched: terms[B][4] = {total, MSE, NCC-loss, SSD-loss}. */
int trx_affine_loss(const trx_volumes *vol, const trx_loss_cfg *loss, const float *theta, float *terms,
                    void *workspace, size_t workspace_bytes, void *stream);

/* get_affine_warp forward (ref:warpings.py:18-26; used by Register.__call__, ref:torchregister.py:123-128):
 * out[B][C][D][H][W] = warp(moving[B][C][D][H][W], theta[B]); vol->moving_stride is the stride
 * between BATCH items (C*D*H*W for a dense tensor); channels share theta. */
int trx_affine_warp(const trx_volumes *vol, const float *theta, int channels, float *out, void *stream);

/* Generic backward of the warp wrt theta: dtheta[B][TRX_PSTRIDE] = sum_p,c grad_out * d warp/d theta
 * (grid_sampler backward + affine_grid backward of the autograd chain behind error.backward(),
 * ref:warpings.py:80,146) — lets any torch loss drive the HIP warp. */
int trx_affine_warp_backward(const trx_volumes *vol, const float *theta, int channels, const float *grad_out,
                             float *dtheta, void *workspace, size_t workspace_bytes, void *stream);

/* The warp and its theta-backward on a sub-lattice of the output grid: out[B][nz][ny][nx] = warp(moving, theta) at the output voxels
 * (iz[kz], iy[ky], ix[kx]) (device int32 tables; 2-D: nz = 1, iz ignored) - what the NMI loss's
 * F.interpolate(warped, size, mode="nearest") keeps of a full-volume warp (ref:utils.py:236-252: 100^3 of a 256^3 volume) when the
 * tables hold that call's source indices - and dtheta[B][TRX_PSTRIDE] = sum_k grad_out[B][k] * d out_k / d theta (replaces the
 * nearest-interpolate backward + grid_sampler backward + affine_grid backward of the same chain).  Single channel.
 * workspace (backward): at least max(trx_affine_workspace_bytes(vol), B * 1024 * 12 * sizeof(float)) bytes. */
int trx_affine_warp_lattice(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny, const int *ix, int nx,
                            float *out, void *stream);
int trx_affine_warp_lattice_backward(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny, const int *ix,
                                     int nx, const float *grad_out, float *dtheta, void *workspace, size_t workspace_bytes, void *stream);

/* Theta (ref:utils.py:287-310: pose vector -> affine matrix, 6 floats in 3-D / 3 in 2-D) and its vector-Jacobian product, for
 * callers that assemble dL/dtheta themselves: theta_out[B][TRX_PSTRIDE] = Theta(pose[B][TRX_PSTRIDE]) and / or
 * dpose_out[B][TRX_PSTRIDE] = J(pose)^T dtheta[B][TRX_PSTRIDE] (fp64 inside, as in trx_affine_step's rigid branch). */
int trx_theta_chain(const float *pose, const float *dtheta, int ndim, int B, float *theta_out, float *dpose_out, void *stream);

/* ------------------------------------------------------------------ dense flow (SpatialTransformer) */
typedef struct {
    float *flow;        /* [B][ndim][D][H][W] in/out */
    float *flow_tmp;    /* same size; required only when smooth_weight != 0 (double buffer) */
    float *adam_m;      /* same size, Adam only */
    float *adam_v;
    float *losses;      /* [B][losses_capacity] */
    int losses_capacity;
    int *step;          /* [B] */
    float smooth_weight; /* extension: lambda * mean squared forward differences of the flow */
    /* Early stop of ref:warpings.py:231-233 (`if losses_train[-1] <= self.stop_crit: break`), decided on the device, per pair, with
     * no host sync: once the recorded loss of iteration k is <= stop_crit the update of iteration k is still applied (the reference
     * steps before it tests) and every later iteration of this and of later calls is a no-op for that pair - step[b] stays k + 1 =
     * the number of recorded losses.  stopped == NULL disables the test. */
    float stop_crit;
    int *stopped;        /* [B] in/out, 0 before the run; non-zero once pair b has stopped */
    /* optional [B][ndim][D][H][W]: the flow of the LAST FORWARD (the one the last recorded loss was computed from, i.e. before
     * that iteration's update) - what the reference's flow_register.flow holds after optimize() (ref:warpings.py:194-196,211). */
    float *flow_last;
} trx_flow_state;

size_t trx_flow_workspace_bytes(const trx_volumes *vol);

/* SpatialTransformer.forward (ref:utils.py:350-365), C channels sharing the flow
 * (flow_register.deform, ref:warpings.py:238-242; Register.__call__, ref:torchregister.py:123-126). */
int trx_flow_warp(const trx_volumes *vol, const float *flow, int channels, float *out, void *stream);

/* One iteration of direct flow-field optimisation (pass A moments, pass B gradient + optimiser):
 * the warp + loss + backward + step span of ref:warpings.py:208-220 with the flow itself as parameter. */
int trx_flow_step(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                  const trx_flow_state *st, void *workspace, size_t workspace_bytes, void *stream);
int trx_flow_run(const trx_volumes *vol, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                 const trx_flow_state *st, int iters, void *workspace, size_t workspace_bytes, void *stream);

/* Loss + dL/dflow for a given flow (the autograd.Function backward for U-Net-generated flows):
 * terms[B][4]; dflow may be NULL (loss only). */
int trx_flow_loss_grad(const trx_volumes *vol, const trx_loss_cfg *loss, const float *flow, float *terms,
                       float *dflow, void *workspace, size_t workspace_bytes, void *stream);

/* Z-slab partition of ONE 3-D volume over ranks (BASELINE config 5): `vol` describes the rank's slab — target,
 * flow and optimiser state hold planes [z_offset, z_offset + vol->D) — while vol->moving is the WHOLE moving
 * volume [B][D_full][H][W] (replicated, constant: no halo of it is exchanged).  One iteration =
 *   (if smooth_weight != 0) exchange one flow plane with each Z neighbour (xGMI P2P): halo_lo / halo_hi =
 *        the neighbour's flow plane just below / above the slab, [ndim][H][W], NULL at the ends of the volume;
 *   trx_flow_slab_moments: pass A -> this rank's 8 raw fp64 sums {Sy,Sw,Syy,Sww,Syw, smooth_z,smooth_y,smooth_x};
 *   the caller all-reduces `moments` over ranks (RCCL; 64 bytes);
 *   trx_flow_slab_update: loss of the whole volume into losses[t], pass B on the slab.  With the regulariser the
 *        new flow is written to st->flow_tmp (it reads neighbours of the old flow): the caller swaps the two.
 * Same span of the reference as trx_flow_step (ref:warpings.py:208-220). */
int trx_flow_slab_moments(const trx_volumes *vol, int z_offset, int D_full, const float *flow, int smooth, const float *halo_hi, double *moments,
                          void *workspace, size_t workspace_bytes, void *stream);
int trx_flow_slab_update(const trx_volumes *vol, int z_offset, int D_full, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                         const trx_flow_state *st, const double *global_moments, const float *halo_lo, const float *halo_hi,
                         void *workspace, size_t workspace_bytes, void *stream);
/* The cross-slab part of trx_flow_slab_moments' smoothness sums on its own: sum over the slab's top plane of (halo_hi - flow)^2, added to
 * moments[b][5].  trx_flow_slab_moments(..., halo_hi = NULL, ...) followed by this equals trx_flow_slab_moments with the halo - and lets
 * the first run while the neighbour's plane is still travelling over xGMI (SlabFlowSolver.run puts the exchange on a side stream). */
int trx_flow_slab_boundary_smooth(const trx_volumes *vol, const float *flow, const float *halo_hi, double *moments, void *stream);

/* Peer-mapped transport of the slab partition (not in the reference, which is single-device): the halo planes and
 * the 64-byte sum of moments as direct writes into a PEER rank's mailbox - device memory of another GPU of the node (or of another
 * process on the same GPU), mapped by the caller through HIP IPC / peer access - instead of torch.distributed.batch_isend_irecv and
 * all_reduce (RCCL).  Flags are 32-bit iteration numbers that only increase; waits poll with system-scope loads in a one-thread
 * kernel, give up after timeout_us and OR a bit into *status (1: trx_peer_wait, 2: trx_peer_gather) instead of hanging the device.
 *   trx_peer_signal : *flag = value, ordered after everything enqueued on `stream` before it (e.g. the copy of a plane into the
 *                     peer's halo slot);
 *   trx_peer_wait   : returns (on the stream) once *flag >= value;
 *   trx_peer_publish: sums[0..8) -> slot_ptrs[r][0..8) for r < n, then *flag_ptrs[r] = value (slot_ptrs / flag_ptrs: DEVICE arrays of n
 *                     device pointers, one per peer: this rank's slot and flag inside that peer's mailbox);
 *   trx_peer_gather : waits until flags[r] >= value for all r < n (this rank's own mailbox), then out[k] = sum over r in rank order of
 *                     slots[r * 8 + k] - the same fp64 additions in the same order on every rank.
 * torchregister_amd.SlabPeers wraps them (mailbox layout, IPC exchange of the handles, double buffering by iteration parity). */
int trx_peer_signal(unsigned *flag, unsigned value, void *stream);
int trx_peer_wait(const unsigned *flag, unsigned value, unsigned timeout_us, int *status, void *stream);
int trx_peer_publish(const double *sums, const void *slot_ptrs, const void *flag_ptrs, int n, unsigned value, void *stream);
int trx_peer_gather(const double *slots, const unsigned *flags, int n, unsigned value, unsigned timeout_us, double *out, int *status, void *stream);
/* Mailbox memory.  A flag that a RUNNING kernel polls while another device (or process) writes it must live in memory whose remote
 * writes become visible without a kernel boundary: fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained)), not
 * the coarse-grained memory of an ordinary hipMalloc / caching allocator.
 *   trx_peer_alloc  : `bytes` of zeroed fine-grained memory on the CURRENT device;          trx_peer_free releases it;
 *   trx_peer_export : the 64-byte HIP IPC handle of such an allocation (hipIpcGetMemHandle) - to be sent to the peers by any channel;
 *   trx_peer_import : maps a peer's handle into this process (hipIpcOpenMemHandle with hipIpcMemLazyEnablePeerAccess: enables peer
 *                     access between the current device and the allocation's device); trx_peer_close unmaps it.
 * They return TRX_ERR_HIP when the runtime refuses (IPC unavailable, device not visible, no peer access): the caller then keeps
 * torch.distributed (SlabPeers.try_exchange decides that on all ranks together). */
#define TRX_PEER_HANDLE_BYTES 64
int trx_peer_alloc(size_t bytes, void **ptr);
int trx_peer_free(void *ptr);
int trx_peer_export(void *ptr, void *handle);
int trx_peer_import(const void *handle, void **ptr);
int trx_peer_close(void *ptr);
/* Without the smoothness term (3-D): the update that also leaves the slab's block partials of the UPDATED flow in the workspace, and the
 * reduction of those partials to the 8 sums - together they replace trx_flow_slab_moments from the second iteration on (one pass over
 * the slab per iteration instead of two; same numbers). */
int trx_flow_slab_update_fused(const trx_volumes *vol, int z_offset, int D_full, const trx_loss_cfg *loss, const trx_opt_cfg *opt,
                               const trx_flow_state *st, const double *global_moments, void *workspace, size_t workspace_bytes, void *stream);
int trx_flow_slab_moments_ready(const trx_volumes *vol, int z_offset, int D_full, double *moments, void *workspace, size_t workspace_bytes,
                                void *stream);

/* Generic backward of the flow warp: dflow[B][ndim][...] = sum_c grad_out[B][c][...] * d warp/d flow. */
int trx_flow_warp_backward(const trx_volumes *vol, const float *flow, int channels, const float *grad_out,
                           float *dflow, void *stream);

/* ---- local-window NCC (extension; not in the reference: no such criterion exists in src/TorchRegister/utils.py, whose
 * NCCLoss (utils.py:182-205) is global.  Definition = the box-window NCC of VoxelMorph-style registration; CPU restatement:
 * oracle/compose.py::local_ncc_loss).  target / warped: [B][D][H][W] fp32 (D = 1 for ndim 2), contiguous.
 *   loss[b] = alpha * (1 - mean_q cc_b(q)),  cc = c^2 / (a b + eps) over the window^ndim box around q (zero padding),
 *   grad[b] = d loss[b] / d warped[b]   (same shape as warped).  Either output may be NULL.
 * window: 3, 5, 7 or 9.  Workspace: trx_lncc_workspace_bytes (12 B per voxel of intermediate fields + block partials). */
size_t trx_lncc_workspace_bytes(int ndim, int B, int D, int H, int W);
int trx_lncc_loss_grad(const float *target, const float *warped, int ndim, int B, int D, int H, int W, int window, float alpha,
                       float eps, float *loss, float *grad, void *workspace, size_t workspace_bytes, void *stream);

/* Direct flow field + LOCAL-window NCC (+ smoothness regulariser st->smooth_weight) as one device-side loop: `iters` iterations of
 *   warp at the current flow -> window sums, loss, dL/dwarped (the kernels behind trx_lncc_loss_grad) -> loss curve / early stop / optimiser
 *   scalars -> dL/dflow through the trilinear derivative + smoothness gradient + SGD / Adam in place
 * with no autograd, no torch optimiser and no host sync (3-D only; 2-D callers compose trx_flow_warp / trx_lncc_loss_grad /
 * trx_flow_warp_backward).  Recorded loss of pair b: lncc_alpha * (1 - mean cc) + smooth_weight / ndim * sum_d mean (forward difference)^2,
 * both of the flow the iteration STARTS from.  State, early stop and flow_last as for trx_flow_run.  vol->target: dense [B][D][H][W].
 * Extension (the reference has neither a local NCC nor a direct flow mode): arbiter = oracle/compose.py under torch autograd. */
size_t trx_flow_lncc_workspace_bytes(const trx_volumes *vol);
int trx_flow_lncc_run(const trx_volumes *vol, int window, float lncc_alpha, float lncc_eps, const trx_opt_cfg *opt,
                      const trx_flow_state *st, int iters, void *workspace, size_t workspace_bytes, void *stream);

/* ---- Parzen-window PDF of the reference's NMI loss (ref:utils.py:18-37 K_gauss / PDF_xis / PDF).
 * signals [N][S], xis [N][bins] (bins <= 1024), h > 0:
 *   pdf[n][k] = (1/h) * mean_i K((signals[n][i] - xis[n][k]) / h),  K(u) = exp(-u*u/2) / (2 pi)   (the reference's constant)
 * trx_kde_pdf_backward: grad_signals[n][i] = sum_k grad_pdf[n][k] * d pdf[n][k] / d signals[n][i]  (xis carry no gradient:
 * the reference builds them from .item() values, ref:utils.py:40-48).  No [N][S][bins] tensor is ever formed. */
size_t trx_kde_workspace_bytes(int N, long S, int bins);
int trx_kde_pdf(const float *signals, const float *xis, int N, long S, int bins, float h, float *pdf, void *workspace,
                size_t workspace_bytes, void *stream);
int trx_kde_pdf_backward(const float *signals, const float *xis, const float *grad_pdf, int N, long S, int bins, float h,
                         float *grad_signals, void *stream);
/* The same two functions as a series in (s - x)^2 / (2 h^2) (thirteen terms, 2e-14): the S x bins exponentials become 25 fp64 power
 * sums of the samples and a polynomial per bin.  VALID ONLY when |signals[n][i] - xis[n][k]| <= h for every pair of the call (the NMI
 * loss's bandwidth 3 on intensities normalised to [0, 1]); the caller checks that and passes `center`, the middle of the value range
 * of signals and xis.  One workspace size serves both. */
size_t trx_kde_series_workspace_bytes(int N, long S, int bins);
int trx_kde_pdf_series(const float *signals, const float *xis, int N, long S, int bins, float h, double center, float *pdf,
                       void *workspace, size_t workspace_bytes, void *stream);
int trx_kde_pdf_series_backward(const float *signals, const float *xis, const float *grad_pdf, int N, long S, int bins, float h,
                                double center, float *grad_signals, void *workspace, size_t workspace_bytes, void *stream);
/* The PDF of the SAME signals on another sample line without a second pass over them: `sums` = the workspace an earlier
 * trx_kde_pdf_series(signals, ..., N, S, ..., center, ...) call left behind (the power sums live at its start; same N, S, center).  The NMI
 * loss's pooled sample line moves with the warped image every iteration while the target's samples do not. */
int trx_kde_pdf_series_cached(const void *sums, const float *xis, int xis_stride /* floats between rows of xis, >= bins */, int N, long S, int bins,
                              float h, double center, float *pdf, void *stream);

/* The 256-bin algebra of the NMI loss behind the three PDFs (ref:utils.py:53-79 NMI, :224-259 NMILoss.forward) in one kernel, value and
 * gradient: h1 / h2 / hj [N][bins] = the Parzen "histograms" of target, warped and of the pooled samples (get_pdf, ref:utils.py:40-51).
 *   p = h / sum(h),  E = sum p log2(p + 1e-10)  (the reference's sign convention),  MI = E1 + E2 - Ej,  NMI = 2 MI / (E1 + E2),
 *   loss = alpha * mean_n |NMI_n - 1|  =  sum_n loss_terms[n].
 * Outputs (each may be NULL): nmi [N], mi [N], loss_terms [N], grad_h* [N][bins] = d loss / d h*. */
int trx_nmi_from_pdfs(const float *h1, const float *h2, const float *hj, int N, int bins, float alpha, float *nmi, float *mi,
                      float *loss_terms, float *grad_h1, float *grad_h2, float *grad_hj, void *stream);

/* The same algebra for the loop that evaluates the warped image's PDFs as ONE 2 x bins call (its own line | the pooled line): pdf_w
 * [N][2 bins], pdf_t [N][bins] = the target's PDF on the pooled line; the pooled histogram is 0.5 (pdf_w[:, bins:] + pdf_t) (fp32, like
 * the torch composition) and grad_w [N][2 bins] = [d loss / d h2 | 0.5 d loss / d hj] is what trx_kde_pdf_series_backward takes. */
int trx_nmi_from_pdfs_pooled(const float *h1, const float *pdf_w, const float *pdf_t, int N, int bins, float alpha, float *nmi, float *mi,
                             float *loss_terms, float *grad_w, void *stream);

/* trx_affine_warp_lattice plus the NMI loss's two sample lines in the same call (ref:utils.py:40-48 get_pdf: linspace(max, min, bins)
 * of the samples): out as trx_affine_warp_lattice; xis[B * patches][2 bins] = per pair the line between the extrema of ITS lattice
 * values, then the line between the extrema of those and of minmax_target[B][2] = (min, max) of the target's samples;
 * minmax_warped[B][2] (nullable) receives the warped extrema.  Replaces torch.aminmax + 2 lerp + maximum + minimum + cat per iteration.
 * workspace: B * 8192 * 2 * sizeof(float) bytes. */
int trx_nmi_lattice_lines(const trx_volumes *vol, const float *theta, const int *iz, int nz, const int *iy, int ny, const int *ix, int nx,
                          float *out, const float *minmax_target, int patches, int bins, float *xis, float *minmax_warped, void *workspace,
                          size_t workspace_bytes, void *stream);

/* Tail of one iteration of a loop that assembles dL/dtheta itself (ref:warpings.py:80-93 / :146-159 after error.backward()), one pair:
 * *hist_loss_t = sum(loss_terms[0 .. n_terms)) + *loss_b (nullable); hist_theta_t[TRX_PSTRIDE] = theta (of this forward);
 * g = grad_a + grad_b (nullable); pose == NULL: theta -= lr g (affine SGD), else pose -= lr J(pose)^T g and theta = Theta(pose)
 * (rigid, as trx_theta_chain); param_copy[TRX_PSTRIDE] (nullable) = the new theta. */
int trx_nmi_loop_update(int ndim, float *theta, float *pose, const float *grad_a, const float *grad_b, float lr, const float *loss_terms,
                        int n_terms, const float *loss_b, float *hist_loss_t, float *hist_theta_t, float *param_copy, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TRX_H */
