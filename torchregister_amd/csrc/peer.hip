// Peer-mapped transport of the Z-slab partition (BASELINE config 5): the two exchanges a slab iteration needs - one
// boundary plane of the flow per face, and the sum of 8 fp64 moments over all ranks - as direct writes into memory the PEER owns
// (its mailbox, mapped here through HIP IPC or peer access; over xGMI on a multi-GPU node) instead of RCCL calls:
//   * the sender copies its plane into the neighbour's halo slot (hipMemcpyAsync on its stream) and then raises a 32-bit flag in the
//     neighbour's mailbox to the iteration number (trx_peer_signal: a release store at system scope);
//   * the sums: every rank writes its 8 doubles into slot [rank] of EVERY peer's mailbox and raises flag [rank] there
//     (trx_peer_publish: one small kernel, N x 64 bytes); each rank then waits until all N flags of its own mailbox have reached the
//     iteration number and adds the N slots in rank order (trx_peer_gather) - the same order on every rank, so the whole-volume sums,
//     and with them the loss curve and the early stop, are bit-identical everywhere, which a ring all-reduce does not promise.
// Waiting is a one-thread kernel that polls with system-scope acquire loads and s_sleep; it gives up after `timeout_us` and sets
// *status (the host checks it after the run) - a missing peer must not hang the GPU.  The failure is STICKY: once *status is non-zero
// every later wait / gather returns at once (an absent peer costs one time-out, not one per remaining iteration) and the gather writes
// NaN sums, so the update that follows poisons the flow and the loss curve shows the failure even if the caller never checks.  Flags only ever increase (iteration numbers),
// so nothing is reset; the caller double-buffers slots and halo planes by the parity of the iteration (a rank can be one iteration
// ahead of a peer, never two: its next publish needs the peer's previous one).
#include <cstring>
#include "trx_common.h"

namespace trx {

__device__ __forceinline__ unsigned peer_load(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM); }

// true when *flag reached `value` before the timeout (wall clock: 100 MHz)
__device__ __forceinline__ bool peer_spin(const unsigned *flag, unsigned value, unsigned timeout_us)
{
    const unsigned long long t0 = wall_clock64(), limit = (unsigned long long)timeout_us * 100ull;
    while ((int)(peer_load(flag) - value) < 0) {   // (wrap-safe comparison of iteration numbers)
        __builtin_amdgcn_s_sleep(32);
        if (wall_clock64() - t0 > limit) return false;
    }
    return true;
}

__global__ void peer_signal_kernel(unsigned *flag, unsigned value)
{
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void peer_wait_kernel(const unsigned *flag, unsigned value, unsigned timeout_us, int *status)
{
    if (__hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // an earlier wait already gave up
    if (!peer_spin(flag, value, timeout_us)) atomicOr(status, 1);
    __threadfence_system();
}

// thread (r, k): sums[k] -> slot_ptrs[r][k]; then flag_ptrs[r] = value
__global__ __launch_bounds__(TRX_BLOCK) void peer_publish_kernel(const double *__restrict__ sums, double *const *__restrict__ slot_ptrs,
                                                                 unsigned *const *__restrict__ flag_ptrs, int n, unsigned value)
{
    const int tid = threadIdx.x, r = tid >> 3, k = tid & 7;
    for (int rr = r; rr < n; rr += TRX_BLOCK / 8)
        __hip_atomic_store(slot_ptrs[rr] + k, sums[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    for (int rr = tid; rr < n; rr += TRX_BLOCK) __hip_atomic_store(flag_ptrs[rr], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// waits for flags[0..n) >= value, then out[k] = slots[0][k] + slots[1][k] + ... in rank order
__global__ __launch_bounds__(64) void peer_gather_kernel(const double *slots, const unsigned *flags, int n, unsigned value, unsigned timeout_us,
                                                         double *__restrict__ out, int *status)
{
    __shared__ int bad;
    const int tid = threadIdx.x;
    if (tid == 0) bad = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ? 2 : 0;   // 2: an earlier wait already gave up
    __syncthreads();
    if (bad == 0)
        for (int r = tid; r < n; r += 64)
            if (!peer_spin(flags + r, value, timeout_us)) bad = 1;
    __syncthreads();
    if (bad) {
        if (tid == 0 && bad == 1) atomicOr(status, 2);
        if (tid < 8) out[tid] = __builtin_nan("");   // the update that consumes these sums must not look like a step
        return;
    }
    __threadfence_system();
    if (tid < 8) {
        double s = 0.0;
        for (int r = 0; r < n; r++) s += __hip_atomic_load(slots + (size_t)r * 8 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        out[tid] = s;
    }
}

}  // namespace trx

using namespace trx;

extern "C" int trx_peer_signal(unsigned *flag, unsigned value, void *stream)
{
    if (!flag) return TRX_ERR_ARG;
    hipLaunchKernelGGL(peer_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_peer_wait(const unsigned *flag, unsigned value, unsigned timeout_us, int *status, void *stream)
{
    if (!flag || !status) return TRX_ERR_ARG;
    hipLaunchKernelGGL(peer_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value, timeout_us, status);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_peer_publish(const double *sums, const void *slot_ptrs, const void *flag_ptrs, int n, unsigned value, void *stream)
{
    if (!sums || !slot_ptrs || !flag_ptrs || n < 1 || n > 4096) return TRX_ERR_ARG;
    hipLaunchKernelGGL(peer_publish_kernel, dim3(1), dim3(TRX_BLOCK), 0, (hipStream_t)stream, sums, (double *const *)slot_ptrs, (unsigned *const *)flag_ptrs, n, value);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

extern "C" int trx_peer_gather(const double *slots, const unsigned *flags, int n, unsigned value, unsigned timeout_us, double *out, int *status, void *stream)
{
    if (!slots || !flags || !out || !status || n < 1 || n > 4096) return TRX_ERR_ARG;
    hipLaunchKernelGGL(peer_gather_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slots, flags, n, value, timeout_us, out, status);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// ---- mailbox memory: fine-grained (remote writes visible to a running kernel), shared through HIP IPC handles
extern "C" int trx_peer_alloc(size_t bytes, void **ptr)
{
    if (!ptr || bytes == 0) return TRX_ERR_ARG;
    *ptr = nullptr;
    if (hipExtMallocWithFlags(ptr, bytes, hipDeviceMallocFinegrained) != hipSuccess) { (void)hipGetLastError(); return TRX_ERR_HIP; }
    if (hipMemset(*ptr, 0, bytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { (void)hipGetLastError(); (void)hipFree(*ptr); *ptr = nullptr; return TRX_ERR_HIP; }
    return TRX_OK;
}

extern "C" int trx_peer_free(void *ptr)
{
    if (!ptr) return TRX_ERR_ARG;
    if (hipFree(ptr) != hipSuccess) { (void)hipGetLastError(); return TRX_ERR_HIP; }
    return TRX_OK;
}

extern "C" int trx_peer_export(void *ptr, void *handle)
{
    static_assert(sizeof(hipIpcMemHandle_t) == TRX_PEER_HANDLE_BYTES, "handle size of the header");
    if (!ptr || !handle) return TRX_ERR_ARG;
    hipIpcMemHandle_t h;
    if (hipIpcGetMemHandle(&h, ptr) != hipSuccess) { (void)hipGetLastError(); return TRX_ERR_HIP; }
    memcpy(handle, &h, sizeof(h));
    return TRX_OK;
}

extern "C" int trx_peer_import(const void *handle, void **ptr)
{
    if (!handle || !ptr) return TRX_ERR_ARG;
    *ptr = nullptr;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof(h));
    if (hipIpcOpenMemHandle(ptr, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) { (void)hipGetLastError(); *ptr = nullptr; return TRX_ERR_HIP; }
    return TRX_OK;
}

extern "C" int trx_peer_close(void *ptr)
{
    if (!ptr) return TRX_ERR_ARG;
    if (hipIpcCloseMemHandle(ptr) != hipSuccess) { (void)hipGetLastError(); return TRX_ERR_HIP; }
    return TRX_OK;
}
