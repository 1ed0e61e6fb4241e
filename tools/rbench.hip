// Development harness for the packed-box rotated-tile body (tools/experiments/affine_rot.h; shelved, not shipped): the 41 sums against the tile kernels' at rotated poses, and hipEvent timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DTRX_DEV tools/rbench.hip -o build/rbench
//   build/rbench [B] [S] [reps] [ax ay az]...      rotation angles (rad) about x, y, z; several triples = several poses
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../torchregister_amd/csrc/affine.hip"
namespace trx {
#include "experiments/affine_rot.h"   // the shelved packed-box rotated-tile body (round 3): a measured alternative, never part of the library
#include "experiments/affine_rot2.h"  // its double-buffered 1024-thread variant
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename F>
static float time_it(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < (reps >= 50 ? 100 : 3); i++) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / reps;
}

int main(int argc, char **argv)
{
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int B = argc > 1 ? atoi(argv[1]) : 8, S = argc > 2 ? atoi(argv[2]) : 256;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    int D = S, H = S, W = S;
    if (getenv("SB_DHW")) sscanf(getenv("SB_DHW"), "%d,%d,%d", &D, &H, &W);
    const size_t nvox = (size_t)D * H * W, n = nvox * B;
    std::vector<float> h(n);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xffffff) / 16777216.f; };
    for (size_t i = 0; i < n; i++) h[i] = rnd();
    float *mov, *tgt, *theta, *partials;
    CK(hipMalloc(&mov, n * 4)); CK(hipMalloc(&tgt, n * 4));
    CK(hipMemcpy(mov, h.data(), n * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n; i++) h[i] = rnd();
    CK(hipMemcpy(tgt, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> th(B * 12);
    CK(hipMalloc(&theta, B * 12 * 4));
    float *tab;
    CK(hipMalloc(&tab, (W + H + D) * 4));
    hipLaunchKernelGGL(trx::fill_tables_kernel, dim3((std::max(W, std::max(H, D)) + 255) / 256), dim3(256), 0, 0, tab, W, H, D);
    trx_volumes vol = {mov, tgt, nvox, nvox, 3, B, D, H, W, tab, tab + W, tab + W + H, 0};
    const size_t prow = 8192;
    CK(hipMalloc(&partials, (size_t)B * prow * 41 * 4));
    const double alg = 8.0 * nvox;
    auto rep = [&](const char *name, float us) { printf("%-36s %9.1f us/launch  %7.2f us/pair  %6.2f TB/s alg\n", name, us, us / B, alg * B / us / 1e6); };
    auto sums = [&](int rows) {
        std::vector<float> hp((size_t)B * rows * 41);
        CK(hipMemcpy(hp.data(), partials, hp.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> out((size_t)B * 41, 0.0);
        for (int b = 0; b < B; b++) for (int r = 0; r < rows; r++) for (int k = 0; k < 41; k++) out[b * 41 + k] += hp[((size_t)b * rows + r) * 41 + k];
        return out;
    };
    const trx::TileGeom ta = trx::tile_geom<trx::GeomA>(vol), tr = trx::tile_geom<trx::GeomR>(vol), td = trx::tile_geom<trx::GeomD>(vol), trd = trx::tile_geom<trx::GeomRD>(vol);
    const int gx = std::max(std::max(ta.blocks_per_pair, tr.blocks_per_pair), std::max(td.blocks_per_pair, trd.blocks_per_pair));
    const trx::ZGeom none = trx::ZGeom{};
    int *ru; CK(hipMalloc(&ru, B * 4));
    printf("B=%d %dx%dx%d  16^3 tiles: %d blocks/pair, %d tiles per block\n", B, D, H, W, trd.blocks_per_pair, trd.tiles_per_seg);
    for (int a = 4; a + 2 < argc || a == 4; a += 3) {
        const double ax = a + 2 < argc ? atof(argv[a]) : 0.5, ay = a + 2 < argc ? atof(argv[a + 1]) : 0.4, az = a + 2 < argc ? atof(argv[a + 2]) : 0.3;
        const double Rx[9] = {1, 0, 0, 0, cos(ax), -sin(ax), 0, sin(ax), cos(ax)}, Ry[9] = {cos(ay), 0, sin(ay), 0, 1, 0, -sin(ay), 0, cos(ay)},
                     Rz[9] = {cos(az), -sin(az), 0, sin(az), cos(az), 0, 0, 0, 1};
        double T[9], Rm[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i * 3 + j] = 0; for (int k = 0; k < 3; k++) T[i * 3 + j] += Rz[i * 3 + k] * Ry[k * 3 + j]; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm[i * 3 + j] = 0; for (int k = 0; k < 3; k++) Rm[i * 3 + j] += T[i * 3 + k] * Rx[k * 3 + j]; }
        const double sc[3] = {1.05, 0.95, 1.02};
        for (int b = 0; b < B; b++) for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) th[b * 12 + i * 4 + j] = (float)(Rm[i * 3 + j] * sc[j] * (1.0 + 0.003 * b)); th[b * 12 + i * 4 + 3] = 0.02f * (i + 1) - 0.01f * b; }
        CK(hipMemcpy(theta, th.data(), B * 12 * 4, hipMemcpyHostToDevice));
        printf("-- R(%.2f, %.2f, %.2f) diag(1.05, .95, 1.02)\n", ax, ay, az);
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 1, td, trd, none, ru, 0);
        CK(hipDeviceSynchronize());
        std::vector<int> hru(B); CK(hipMemcpy(hru.data(), ru, B * 4, hipMemcpyDeviceToHost));
        const std::vector<double> ref = sums(gx);
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        hipLaunchKernelGGL((trx::affine_rot_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(512), 0, 0, vol, theta, trd, partials, trd.blocks_per_pair);
        CK(hipDeviceSynchronize());
        const std::vector<double> got = sums(trd.blocks_per_pair);
        double worst = 0; int wk = -1, wb = -1; bool nan = false;
        for (int b = 0; b < B; b++) {
            double scale = 0;
            for (int k = 5; k < 41; k++) scale = std::max(scale, fabs(ref[b * 41 + k]));
            for (int k = 0; k < 41; k++) {
                if (!(got[b * 41 + k] == got[b * 41 + k])) nan = true;
                const double e = fabs(got[b * 41 + k] - ref[b * 41 + k]) / (k < 5 ? std::max(1.0, fabs(ref[b * 41 + k])) : scale);
                if (e > worst) { worst = e; wk = k; wb = b; }
            }
        }
        printf("packed box vs tile kernels: worst relative difference of the 41 sums %.3e (sum %d, pair %d)%s   Sy %.3f / %.3f  Sw %.3f / %.3f  Syw %.3f / %.3f\n", worst, wk, wb,
               nan ? "  NaN" : "", got[0], ref[0], got[1], ref[1], got[4], ref[4]);
        if (getenv("SB_DUMP")) for (int k = 0; k < 41; k++) printf("   sum %2d: %.6e  %.6e\n", k, got[k], ref[k]);
        rep("tile kernels (rows used below)", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, none, ru, 0); }, reps));
        printf("   rows used by the tile kernels: %d\n", hru[0]);
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        hipLaunchKernelGGL((trx::affine_rot2_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(1024), 0, 0, vol, theta, trd, partials, trd.blocks_per_pair);
        CK(hipDeviceSynchronize());
        {
            const std::vector<double> g2 = sums(trd.blocks_per_pair);
            double w2 = 0; bool nan2 = false;
            for (int b = 0; b < B; b++) {
                double scale = 0;
                for (int k = 5; k < 41; k++) scale = std::max(scale, fabs(ref[b * 41 + k]));
                for (int k = 0; k < 41; k++) {
                    if (!(g2[b * 41 + k] == g2[b * 41 + k])) nan2 = true;
                    w2 = std::max(w2, fabs(g2[b * 41 + k] - ref[b * 41 + k]) / (k < 5 ? std::max(1.0, fabs(ref[b * 41 + k])) : scale));
                }
            }
            printf("double-buffered vs tile kernels: worst relative difference %.3e%s   Sy %.3f Sw %.3f\n", w2, nan2 ? "  NaN" : "", g2[0], g2[1]);
        }
        rep("packed-box, double-buffered (1024 thr)", time_it([&] { hipLaunchKernelGGL((trx::affine_rot2_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(1024), 0, 0, vol, theta, trd, partials, trd.blocks_per_pair); }, reps));
        rep("packed-box 16^3 tiles", time_it([&] { hipLaunchKernelGGL((trx::affine_rot_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(512), 0, 0, vol, theta, trd, partials, trd.blocks_per_pair); }, reps));
    }
    return 0;
}
