// Micro-benchmark (development; VERDICT r3 #4, north_star "MFMA only if the local-NCC window sums are cast as small dense contractions"):
// the x / y box passes of ONE plane of the local-window NCC kernels (csrc/lncc.hip: plane_window_sums - a 32 x 16 output tile, halo 4,
// NF fields), (a) as the product does them - sliding sums on the VALU - and (b) as banded 0 / 1 contractions on the matrix pipe:
//     T[y][xo] = sum_k X[y][k] Bx[k][xo]   (x pass),      Out[yo][xo] = sum_k By[yo][k] T[k][xo]   (y pass)
// with v_mfma_f32_16x16x4_f32 (fp32 in, fp32 accumulate: exact products with 0 / 1, no operand splitting; 32 cycles per instruction and
// SIMD).  Same LDS tile in, same per-thread outputs (thread (ox, oy): two y-adjacent outputs of every field) out, so (b) could replace (a)
// inside the kernels as it stands.  Timed in isolation: a block re-fills its tile from registers and runs the passes `planes` times.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/lncc_mfma.hip -o build/lncc_mfma && build/lncc_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

constexpr int kLO = 2, kLX = 32, kLY = 16, kRows = kLY + 8, kCols = kLX + 8;     // as csrc/lncc.hip (two-row build)
constexpr int kCells = (kRows * kCols + 255) / 256;
typedef float f4v __attribute__((ext_vector_type(4)));

// ---- (a) the product's passes (copied from plane_window_sums)
template <int R, int NF>
__device__ __forceinline__ void passes_valu(float (*raw)[kRows][kCols], float (*xs)[kLX][kRows + 1], float (&P)[kLO][NF])
{
    const int tid = threadIdx.x;
    if (tid < kRows * (kLX / 4)) {
        const int row = tid >> 3, q = tid & 7;
        if (row >= 4 - R && row < kLY + 4 + R) {
#pragma unroll
            for (int f = 0; f < NF; f++) {
                float v[12];
                const float4 *src = reinterpret_cast<const float4 *>(&raw[f][row][4 * q]);
                const float4 a = src[0], b = src[1], c = src[2];
                v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
                v[8] = c.x; v[9] = c.y; v[10] = c.z; v[11] = c.w;
                float s = 0.f;
#pragma unroll
                for (int k = 4 - R; k <= 4 + R; k++) s += v[k];
                xs[f][4 * q][row] = s;
#pragma unroll
                for (int i = 1; i < 4; i++) {
                    s += v[4 + R + i] - v[3 - R + i];
                    xs[f][4 * q + i][row] = s;
                }
            }
        }
    }
    __syncthreads();
    const int ox = tid & (kLX - 1), oy = kLO * (tid >> 5);
#pragma unroll
    for (int f = 0; f < NF; f++) {
        float v[2 * R + kLO];
#pragma unroll
        for (int k = 0; k < 2 * R + kLO; k++) v[k] = xs[f][ox][oy + 4 - R + k];
#pragma unroll
        for (int o = 0; o < kLO; o++) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k <= 2 * R; k++) s += v[o + k];
            P[o][f] = s;
        }
    }
}

// ---- (b) the same sums on the matrix pipe.  `raw` rows are read with pitch kCols (40 floats: 8 banks per row - the A fragments of 16 rows
// collide 4-way; a pitch of 41 is what a kernel built around this would use: TPITCH); ts[f][row][xo]: x-pass result; out aliases raw.
template <int R, int NF, int RP>
__device__ __forceinline__ void passes_mfma(float *raw, float *ts, float (&P)[kLO][NF])
{
    constexpr int KS = (16 + 2 * R + 3) / 4;   // k-steps of 4: 16 outputs see 16 + 2R inputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    float band[KS];   // Bx[k][j] of step s (this lane: k = lk, j = li) = By[i][k] with i = li: input 4 s + lk belongs to output li's window
#pragma unroll
    for (int s = 0; s < KS; s++) { const int d = 4 * s + lk - li; band[s] = (d >= 0 && d <= 2 * R) ? 1.f : 0.f; }
    constexpr int TP = kLX + 1;   // pitch of ts rows
    {   // x pass: wave (rb, cb) -> T rows 16 rb .., outputs 16 cb ..
        const int rb = wave >> 1, cb = wave & 1;
#pragma unroll
        for (int f = 0; f < NF; f++) {
            f4v acc = {0.f, 0.f, 0.f, 0.f};
            const float *src = raw + (f * kRows + min(16 * rb + li, kRows - 1)) * RP + 16 * cb + 4 - R + lk;
#pragma unroll
            for (int s = 0; s < KS; s++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(src[4 * s], band[s], acc, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int row = 16 * rb + 4 * lk + v;
                if (row < kRows) ts[(f * kRows + row) * TP + 16 * cb + li] = acc[v];
            }
        }
    }
    __syncthreads();
    // y pass: tiles (f, cb), dealt to the four waves; Out[yo][xo] = sum over input rows yo + 4 - R .. yo + 4 + R
    for (int t = wave; t < NF * 2; t += 4) {
        const int f = t >> 1, cb = t & 1;
        f4v acc = {0.f, 0.f, 0.f, 0.f};
        const float *src = ts + (f * kRows + 4 - R + lk) * TP + 16 * cb + li;
#pragma unroll
        for (int s = 0; s < KS; s++) {
            const int row = 4 - R + 4 * s + lk;
            const float b = row < kRows ? src[4 * s * TP] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(band[s], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) raw[(f * kLY + 4 * lk + v) * (kLX + 1) + 16 * cb + li] = acc[v];   // out[f][yo][xo] (raw is free by now)
    }
    __syncthreads();
    const int ox = tid & (kLX - 1), oy = kLO * (tid >> 5);
#pragma unroll
    for (int f = 0; f < NF; f++)
#pragma unroll
        for (int o = 0; o < kLO; o++) P[o][f] = raw[(f * kLY + oy + o) * (kLX + 1) + ox];
}

template <int R, int NF, int MODE, int RP>
__global__ __launch_bounds__(256, 3) void k(float *out, int planes, float seed)
{
    __shared__ __attribute__((aligned(16))) float raw[NF * kRows * (RP > kCols ? RP : kCols)];
    __shared__ float xs[NF * kLX * (kRows + 1)];
    const int tid = threadIdx.x;
    float Z[kLO][NF];
#pragma unroll
    for (int o = 0; o < kLO; o++)
#pragma unroll
        for (int f = 0; f < NF; f++) Z[o][f] = 0.f;
    for (int p = 0; p < planes; p++) {
        __syncthreads();
        // the tile of this plane (as the kernels' expand step: values from registers, 5 products of two inputs)
#pragma unroll
        for (int c = 0; c < kCells; c++) {
            const int rc = tid + c * 256;
            if (rc < kRows * kCols) {
                const int row = rc / kCols, col = rc - row * kCols;
                const float i = seed * (float)((rc * 7 + p * 13 + blockIdx.x) % 97) * 0.01f, j = 0.5f + seed * (float)((rc * 11 + p * 5) % 89) * 0.01f;
                const float vals[5] = {i, j, i * i, j * j, i * j};
#pragma unroll
                for (int f = 0; f < NF; f++) raw[(f * kRows + row) * RP + col] = vals[f % 5];
            }
        }
        __syncthreads();
        float P[kLO][NF];
        if constexpr (MODE == 0) passes_valu<R, NF>(reinterpret_cast<float(*)[kRows][kCols]>(raw), reinterpret_cast<float(*)[kLX][kRows + 1]>(xs), P);
        else passes_mfma<R, NF, RP>(raw, xs, P);
#pragma unroll
        for (int o = 0; o < kLO; o++)
#pragma unroll
            for (int f = 0; f < NF; f++) Z[o][f] += P[o][f];
    }
#pragma unroll
    for (int o = 0; o < kLO; o++)
#pragma unroll
        for (int f = 0; f < NF; f++) out[((size_t)blockIdx.x * 256 + tid) * (kLO * NF) + o * NF + f] = Z[o][f];
}

template <int R, int NF, int MODE, int RP>
static float run(float *out, int blocks, int planes, std::vector<float> *res)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<R, NF, MODE, RP>), dim3(blocks), dim3(256), 0, 0, out, planes, 1.0f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL((k<R, NF, MODE, RP>), dim3(blocks), dim3(256), 0, 0, out, planes, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (res) { res->resize((size_t)256 * kLO * NF); CK(hipMemcpy(res->data(), out, res->size() * 4, hipMemcpyDeviceToHost)); }
    return ms * 1000.f / 10;
}

template <int R, int NF>
static void compare(float *out, int blocks, int planes)
{
    std::vector<float> a, b, c;
    const float tv = run<R, NF, 0, kCols>(out, blocks, planes, &a);
    const float tm = run<R, NF, 1, kCols>(out, blocks, planes, &b);
    const float tp = run<R, NF, 1, kCols + 1>(out, blocks, planes, &c);
    double worst = 0, worst2 = 0;
    for (size_t i = 0; i < a.size(); i++) { worst = std::max(worst, (double)fabsf(a[i] - b[i]) / std::max(1.0, (double)fabsf(a[i]))); worst2 = std::max(worst2, (double)fabsf(a[i] - c[i]) / std::max(1.0, (double)fabsf(a[i]))); }
    const double per = 1e3 / ((double)blocks / 768.0 * planes);   // ns per plane and block slot (768 slots: three 256-thread blocks per CU)
    printf("window %d, %d fields: VALU passes %8.1f us (%6.1f ns per plane and block slot) | MFMA f32 %8.1f us (%6.1f ns) x%.2f | MFMA f32, tile pitch 41 %8.1f us (%6.1f ns) x%.2f | max rel diff %.1e / %.1e\n",
           2 * R + 1, NF, tv, tv * per, tm, tm * per, tm / tv, tp, tp * per, tp / tv, worst, worst2);
}

int main()
{
    const int blocks = 768 * 4, planes = 64;
    float *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * kLO * 5 * 4));
    compare<4, 5>(out, blocks, planes);   // fields kernel, w = 9
    compare<2, 5>(out, blocks, planes);   // fields kernel, w = 5
    compare<4, 3>(out, blocks, planes);   // gradient kernel, w = 9
    compare<2, 3>(out, blocks, planes);   // gradient kernel, w = 5
    return 0;
}
