// What would ONE persistent launch per K iterations cost the launch-bound configurations (BASELINE cfg 1 / cfg 2: two kernels per iteration,
// F1 partial sums -> finalise)?  Skeleton of such an iteration with no registration work in it:
//   every block publishes a 41-float partial row -> grid barrier -> block 0 reduces the rows in fp64 and writes 12 floats (theta) ->
//   grid barrier -> every block reads theta
// against the same skeleton as two launches per iteration (the product's structure).  Barrier: one monotonic device-scope counter per phase,
// lane-0 release fence before the arrive, relaxed sc1-load poll, acquire fence after (cdna_hip_programming.md Guideline 16; spins bounded).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o build/grid_barrier && build/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, unsigned *timeout)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22)) { *timeout = 1; ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    return ok;
}

// rows [nb][41] -> 12 floats: 8 groups of 64 lanes, 8 independent loads in flight per thread, fp64, fixed order (the shape of the product's finalise)
__device__ __forceinline__ void reduce_rows(const float *partials, int nb, float *theta)
{
    __shared__ double red[8][64];
    const int tid = threadIdx.x, k = tid & 63, grp = (tid >> 6) & 7;
    double s = 0.0;
    if (k < 41 && tid < 512)
        for (int r0 = grp; r0 < nb; r0 += 64) {
            float a[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { const int r = r0 + 8 * i; a[i] = r < nb ? __builtin_nontemporal_load(partials + r * 41 + k) : 0.f; }
#pragma unroll
            for (int i = 0; i < 8; i++) s += (double)a[i];
        }
    if (tid < 512) red[grp][k] = s;
    __syncthreads();
    if (tid < 12) { double t = 0.0; for (int g = 0; g < 8; g++) t += red[g][tid]; theta[tid] = (float)t; }
}

__global__ __launch_bounds__(512) void persistent_kernel(float *partials, float *theta, unsigned *counters, unsigned *timeout, int iters, float *sink)
{
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    float acc = 0.f;
    for (int it = 0; it < iters; it++) {
        if (tid < 41) partials[b * 41 + tid] = (float)(b + it) * 1e-3f + tid;
        if (!grid_barrier(counters, (unsigned)(2 * it + 1) * nb, timeout)) return;
        if (b == 0) reduce_rows(partials, nb, theta);
        if (!grid_barrier(counters, (unsigned)(2 * it + 2) * nb, timeout)) return;
        acc += __builtin_nontemporal_load(theta + (tid % 12));
    }
    if (tid == 0) sink[b] = acc;
}

__global__ __launch_bounds__(512) void part_kernel(float *partials, const float *theta, int it, float *sink)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid < 41) partials[b * 41 + tid] = (float)(b + it) * 1e-3f + tid + theta[tid % 12] * 0.f;
}
__global__ __launch_bounds__(512) void fin_kernel(const float *partials, float *theta, int nb)
{
    reduce_rows(partials, nb, theta);
}

int main()
{
    float *partials, *theta, *sink;
    unsigned *counters, *timeout;
    CK(hipMalloc(&partials, 1024 * 41 * 4)); CK(hipMalloc(&theta, 64)); CK(hipMalloc(&sink, 1024 * 4));
    CK(hipMalloc(&counters, 256)); CK(hipMalloc(&timeout, 4));
    CK(hipMemset(theta, 0, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    for (int nb : {256, 512}) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipMemset(counters, 0, 256)); CK(hipMemset(timeout, 0, 4));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(persistent_kernel, dim3(nb), dim3(512), 0, 0, partials, theta, counters, timeout, iters, sink);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned to; CK(hipMemcpy(&to, timeout, 4, hipMemcpyDeviceToHost));
            if (to) { printf("persistent, %d blocks: barrier TIMEOUT\n", nb); break; }
            if (ms < best) best = ms;
        }
        printf("persistent launch, %3d blocks of 512 threads: %6.2f us per iteration (2 grid barriers + publish + fp64 reduction of %d rows by block 0)\n", nb, best * 1e3f / iters, nb);
        best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int it = 0; it < iters; it++) {
                hipLaunchKernelGGL(part_kernel, dim3(nb), dim3(512), 0, 0, partials, theta, it, sink);
                hipLaunchKernelGGL(fin_kernel, dim3(1), dim3(512), 0, 0, partials, theta, nb);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("two launches per iteration, %3d blocks:           %6.2f us per iteration (the same publish + reduction as kernels)\n", nb, best * 1e3f / iters);
    }
    return 0;
}
