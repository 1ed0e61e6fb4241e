// Micro-benchmark (development, round 5): does the 256 MiB Infinity Cache keep the TAIL of a streaming pass, so that a pass walking the
// same buffer in the OPPOSITE direction starts on cache hits?  A registration re-reads the same 1 GiB of volumes every iteration; walked
// in the same order each time, an LRU-like memory-side cache of 256 MiB hits nothing.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mall_pingpong.hip -o build/mall_pingpong && build/mall_pingpong [MiB]
// Persistent grid of 512 blocks x 512 threads; block b reads chunks b, b + 512, ... of 64 KiB in time order (ascending or descending), so
// the launch sweeps the buffer front to back (or back to front) like the z-streaming kernel sweeps its volumes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT>
__global__ __launch_bounds__(512) void sweep(const v4f *__restrict__ buf, size_t nchunks, int descending, float *out)
{
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const size_t per_block = (nchunks + gridDim.x - 1) / gridDim.x;
    for (size_t i = 0; i < per_block; i++) {
        const size_t k = descending ? (per_block - 1 - i) : i;
        const size_t c = k * gridDim.x + blockIdx.x;
        if (c >= nchunks) continue;
        const v4f *p = buf + c * 4096 + threadIdx.x;   // 64 KiB chunk = 4096 float4, 8 per thread
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const v4f v = NT ? __builtin_nontemporal_load(p + j * 512) : p[j * 512];
            acc += v;
        }
    }
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 1.2345f) out[0] = s;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    for (int mib : {128, 512, 1024, 2048}) {
        if (argc > 1 && atoi(argv[1]) != mib) continue;
        const size_t bytes = (size_t)mib << 20, nchunks = bytes / 65536;
        v4f *buf; float *out;
        CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 4));
        CK(hipMemset(buf, 0, bytes));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int reps = argc > 2 ? atoi(argv[2]) : 100;
        for (int mode = 0; mode < 4; mode++) {
            if (argc > 3 && atoi(argv[3]) != mode) continue;   // bit 0: alternate directions (else every pass ascending); bit 1: non-temporal loads
            auto launch = [&](int i) {
                if (mode & 2) hipLaunchKernelGGL(sweep<1>, dim3(512), dim3(512), 0, 0, buf, nchunks, (mode & 1) ? (i & 1) : 0, out);
                else hipLaunchKernelGGL(sweep<0>, dim3(512), dim3(512), 0, 0, buf, nchunks, (mode & 1) ? (i & 1) : 0, out);
            };
            for (int i = 0; i < 20; i++) launch(i);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; i++) launch(i);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%5d MiB  %-22s %8.1f us per pass  %6.2f TB/s\n", mib, (mode & 1) ? ((mode & 2) ? "alternating, nt loads" : "alternating direction") : ((mode & 2) ? "same direction, nt" : "same direction"), ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12);
        }
        CK(hipFree(buf)); CK(hipFree(out));
    }
    return 0;
}
