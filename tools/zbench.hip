// Development harness for the z-streaming F1 kernel (not shipped): correctness against the tile kernel's 41 sums and hipEvent timing.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize -DTRX_DEV tools/zbench.hip -o build/zbench
//   build/zbench [B] [S] [eps]      eps: size of the deviation of theta from the identity (0 = identity)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../torchregister_amd/csrc/affine.hip"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <typename F>
static float time_it(F f, int reps)
{
    
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < (reps >= 50 ? 150 : 5); i++) f();
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
    const float eps = argc > 3 ? (float)atof(argv[3]) : 0.004f;
    const int reps = argc > 4 ? atoi(argv[4]) : 100;
    const bool zonly = argc == 6;   // profiling runs: only the first z-streaming kernel
    const size_t nvox = (size_t)S * S * S;
    const size_t pad = getenv("ZB_PAD") ? (size_t)atol(getenv("ZB_PAD")) : 0;   // floats between the pairs' volumes (round 6: is the power-of-two pair stride a DRAM channel conflict?)
    const size_t pstride = nvox + pad, n = pstride * B;
    std::vector<float> h(n);
    srand(1);
    // smooth-ish random volumes: random values low-pass filtered along x so that gradients are not pure noise
    for (size_t i = 0; i < n; i++) h[i] = (float)rand() / RAND_MAX;
    float *mov, *tgt, *theta, *partials;
    CK(hipMalloc(&mov, n * 4)); CK(hipMalloc(&tgt, n * 4 + (1 << 22)));
    CK(hipMemcpy(mov, h.data(), n * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n; i++) h[i] = (float)rand() / RAND_MAX;
    CK(hipMemcpy(tgt, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> th(B * 12);
    const float id[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    for (int b = 0; b < B; b++)
        for (int i = 0; i < 12; i++) th[b * 12 + i] = id[i] + eps * (2.f * rand() / RAND_MAX - 1.f);
    if (const char *pose = getenv("ZB_POSE")) {   // fixed poses of the convergence basin instead of random ones
        double M[12];
        if (!strcmp(pose, "inv")) { const double m[12] = {1.0413, 0.1074, -0.0204, -0.0484, -0.1074, 1.0199, 0.0021, 0.0359, 0.0032, -0.0300, 0.9803, -0.0209}; memcpy(M, m, sizeof M); }
        else if (!strcmp(pose, "rz0.1")) { const double m[12] = {cos(0.1), -sin(0.1), 0, 0.01, sin(0.1), cos(0.1), 0, -0.02, 0, 0, 1, 0.015}; memcpy(M, m, sizeof M); }
        else if (!strcmp(pose, "rz0.2")) { const double m[12] = {cos(0.2), -sin(0.2), 0, 0.01, sin(0.2), cos(0.2), 0, -0.02, 0, 0, 1, 0.015}; memcpy(M, m, sizeof M); }
        else if (!strcmp(pose, "rz0.3")) { const double m[12] = {cos(0.3), -sin(0.3), 0, 0.01, sin(0.3), cos(0.3), 0, -0.02, 0, 0, 1, 0.015}; memcpy(M, m, sizeof M); }
        else if (!strcmp(pose, "rz0.15")) { const double m[12] = {cos(0.15), -sin(0.15), 0, 0.01, sin(0.15), cos(0.15), 0, -0.02, 0, 0, 1, 0.015}; memcpy(M, m, sizeof M); }
        else { const double m[12] = {0.9975, -0.0474, 0.0524, 0.01, 0.0499, 0.9963, -0.0474, -0.02, -0.05, 0.0499, 0.9975, 0.015}; memcpy(M, m, sizeof M); }   // R(0.05, 0.05, 0.05)
        for (int b = 0; b < B; b++) for (int i = 0; i < 12; i++) th[b * 12 + i] = (float)M[i] + 1e-4f * (2.f * rand() / RAND_MAX - 1.f);
    }
    CK(hipMalloc(&theta, B * 12 * 4));
    CK(hipMemcpy(theta, th.data(), B * 12 * 4, hipMemcpyHostToDevice));
    float *tab;
    CK(hipMalloc(&tab, 3 * S * 4));
    hipLaunchKernelGGL(trx::fill_tables_kernel, dim3((S + 255) / 256), dim3(256), 0, 0, tab, S, S, S);
    trx_volumes vol = {mov, tgt + (getenv("ZB_TOFF") ? atol(getenv("ZB_TOFF")) : 0), pstride, pstride, 3, B, S, S, S, tab, tab + S, tab + 2 * S, 0};
    const size_t prow = 8192;
    CK(hipMalloc(&partials, (size_t)B * prow * 41 * 4));
    const double alg = 8.0 * nvox;
    auto rep = [&](const char *name, float us) { printf("%-34s %9.1f us/launch  %7.2f us/pair  %6.2f TB/s alg  frac %.3f\n", name, us, us / B, alg * B / us / 1e6, alg * B / us / 1e6 / 8.0); };
    auto sums = [&](int rows) {
        std::vector<float> hp((size_t)B * rows * 41);
        CK(hipMemcpy(hp.data(), partials, hp.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> out((size_t)B * 41, 0.0);
        for (int b = 0; b < B; b++) for (int r = 0; r < rows; r++) for (int k = 0; k < 41; k++) out[b * 41 + k] += hp[((size_t)b * rows + r) * 41 + k];
        return out;
    };
    const trx::TileGeom tgm = trx::tile_geom(vol);
    dim3 tgrid(tgm.blocks_per_pair, B);
    CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
    hipLaunchKernelGGL((trx::affine_tile_kernel<0>), tgrid, dim3(trx::kTileThreads), 0, 0, vol, theta, tgm, 1, partials);
    CK(hipDeviceSynchronize());
    const std::vector<double> ref = sums(tgm.blocks_per_pair);
    auto check = [&](const char *name, int rows) {
        const std::vector<double> got = sums(rows);
        double worst = 0; int wk = -1;
        bool nan = false;
        for (int b = 0; b < B; b++) {
            double scale = 0;
            for (int k = 5; k < 41; k++) scale = std::max(scale, fabs(ref[b * 41 + k]));
            for (int k = 0; k < 41; k++) {
                if (!(got[b * 41 + k] == got[b * 41 + k])) nan = true;
                const double e = fabs(got[b * 41 + k] - ref[b * 41 + k]) / (k < 5 ? std::max(1.0, fabs(ref[b * 41 + k])) : scale);
                if (e > worst) { worst = e; wk = k; }
            }
        }
        printf("%s vs tile: worst relative difference of the 41 sums %.3e (sum %d)%s   Sw %.6f / %.6f\n", name, worst, wk, nan ? "  NaN: window does not fit" : "", got[1], ref[1]);
    };
    auto run_zs = [&](auto cfg, const char *name) {
        using C = decltype(cfg);
        if (!trx::zs_shape_ok<C>(vol)) { printf("%s: shape not supported\n", name); return; }
        const trx::ZGeom zg = trx::zs_geom<C>(vol);
        printf("%s geom: %d x %d columns, %d z segments of %d planes, %d blocks/pair, LDS %d B\n", name, zg.ntx, zg.nty, zg.nzseg, zg.planes_per_seg, zg.blocks_per_pair, C::Alloc * 4);
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        hipLaunchKernelGGL((trx::affine_zstream_kernel<0, C>), dim3(zg.blocks_per_pair, B), dim3(C::Threads), 0, 0, vol, theta, zg, partials, zg.blocks_per_pair);
        CK(hipDeviceSynchronize());
        check(name, zg.blocks_per_pair);
        rep(name, time_it([&] { hipLaunchKernelGGL((trx::affine_zstream_kernel<0, C>), dim3(zg.blocks_per_pair, B), dim3(C::Threads), 0, 0, vol, theta, zg, partials, zg.blocks_per_pair); }, reps));
#if TRX_ZS_STAMP
        {   // phases of a step, per wave (s_memtime ticks = shader cycles), and the blocks' start / end in real time (100 MHz)
            std::vector<unsigned long long> st(512 * 8 * 8);
            CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(trx::trx_zs_stamps), st.size() * 8));
            double sum[6] = {0, 0, 0, 0, 0, 0};
            unsigned long long r0min = ~0ull, r0max = 0, r1min = ~0ull, r1max = 0;
            const int nb = std::min(512, zg.blocks_per_pair * B);
            for (int b = 0; b < nb; b++)
                for (int w = 0; w < 8; w++) {
                    const unsigned long long *o = &st[((size_t)b * 8 + w) * 8];
                    for (int k = 0; k < 5; k++) sum[k] += (double)o[k];
                    sum[5] += (double)(o[5] & 0xffff);
                    r0min = std::min(r0min, o[6]); r0max = std::max(r0max, o[6]); r1min = std::min(r1min, o[7]); r1max = std::max(r1max, o[7]);
                }
            const double nw = nb * 8.0, steps = sum[5] / nw;
            printf("   stamps: %.0f steps per wave; per step (cycles): wait %.0f  barrier %.0f  issue %.0f  gather %.0f  = %.0f;  wave lifetime %.0f cycles\n", steps, sum[0] / sum[5], sum[1] / sum[5],
                   sum[2] / sum[5], sum[3] / sum[5], (sum[0] + sum[1] + sum[2] + sum[3]) / sum[5], sum[4] / nw);
            if (getenv("ZS_DUMP")) {   // per block: CU (xcc, se, sh, cu), start and end in us after the first start, wave 0's phase sums
                for (int b = 0; b < nb; b++) {
                    const unsigned long long *o = &st[(size_t)b * 8 * 8];
                    const unsigned hw = (unsigned)(o[5] >> 16) & 0xffff, xcc = (unsigned)(o[5] >> 32) & 0xf;
                    printf("   blk %3d xcc %u se %u sh %u cu %2u  start %7.2f end %7.2f  wait %6.0f bar %6.0f issue %6.0f gather %6.0f\n", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15,
                           (o[6] - r0min) / 100.0, (o[7] - r0min) / 100.0, (double)o[0] / 128, (double)o[1] / 128, (double)o[2] / 128, (double)o[3] / 128);
                }
            }
            printf("   blocks: first start -> last start %.2f us, first end -> last end %.2f us, first start -> last end %.2f us\n", (r0max - r0min) / 100.0, (r1max - r1min) / 100.0, (r1max - r0min) / 100.0);
        }
#endif
    };
    printf("B=%d S=%d eps=%g\n", B, S, eps);
    run_zs(trx::ZS64{}, "zstream 64x32");
    run_zs(trx::ZSF{}, "zstream 64x16 flat");
#ifdef ZB_WIDE   // round 6 experiment: tiles that are WIDE in x - longer contiguous runs per DMA row and per target row (DRAM page locality), same voxels per plane and rows per thread
    run_zs(trx::ZCfg<128, 16, 6, 136, 24>{}, "zstream 128x16 (ring 6 x 136 x 24)");
    run_zs(trx::ZCfg<256, 8, 5, 264, 14>{}, "zstream 256x8 (ring 5 x 264 x 14)");
    run_zs(trx::ZS64{}, "zstream 64x32 (again)");
    return 0;
#endif
    if (zonly) return 0;
    rep("tile MODE0 (GeomP)", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_kernel<0>), tgrid, dim3(trx::kTileThreads), 0, 0, vol, theta, tgm, 1, partials); }, reps));
    {
        const trx::TileGeom ta = trx::tile_geom<trx::GeomA>(vol), tr = trx::tile_geom<trx::GeomR>(vol), td = trx::tile_geom<trx::GeomD>(vol), trd = trx::tile_geom<trx::GeomRD>(vol);
        int gx = std::max(std::max(ta.blocks_per_pair, tr.blocks_per_pair), std::max(td.blocks_per_pair, trd.blocks_per_pair));
        rep("dual MODE0 (GeomD/A/RD/R per pair)", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd); }, reps));
    }
    {   // the fused step kernel with the z-streaming body offered (what trx_affine_step launches)
        const trx::TileGeom ta = trx::tile_geom<trx::GeomA>(vol), tr = trx::tile_geom<trx::GeomR>(vol), td = trx::tile_geom<trx::GeomD>(vol), trd = trx::tile_geom<trx::GeomRD>(vol);
        const trx::ZGeom zg = trx::zs_geom<trx::ZS64>(vol);
        int gx = std::max(std::max(ta.blocks_per_pair, tr.blocks_per_pair), std::max(td.blocks_per_pair, trd.blocks_per_pair));
        gx = std::max(gx, zg.blocks_per_pair);
        int *ru; CK(hipMalloc(&ru, B * 4));
        printf("fused grid: %d blocks/pair (A %d R %d D %d RD %d ZS %d)\n", gx, ta.blocks_per_pair, tr.blocks_per_pair, td.blocks_per_pair, trd.blocks_per_pair, zg.blocks_per_pair);
        rep("fused dual MODE0, classic grid", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, 0); }, reps));
        rep("fused dual MODE0, flat grid 512", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, gx); }, reps));
        const trx::ZGeom none = trx::ZGeom{};
        rep("fused dual, no ZS, classic grid", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, none, ru, 0); }, reps));
        rep("fused dual, no ZS, flat grid 512", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, none, ru, gx); }, reps));
    }
    if (argc > 6) {   // rotated poses through the fused kernel at several grid sizes (looping): R(0.4,0.4,0.4), Ry(0.3), Rz(0.15)
        const trx::TileGeom ta = trx::tile_geom<trx::GeomA>(vol), tr = trx::tile_geom<trx::GeomR>(vol), td = trx::tile_geom<trx::GeomD>(vol), trd = trx::tile_geom<trx::GeomRD>(vol);
        const trx::ZGeom zg = trx::zs_geom<trx::ZS64>(vol);
        int gx = std::max(std::max(ta.blocks_per_pair, tr.blocks_per_pair), std::max(td.blocks_per_pair, trd.blocks_per_pair));
        int *ru; CK(hipMalloc(&ru, B * 4));
        const double angs[4][3] = {{0.4, 0.4, 0.4}, {0.0, 0.3, 0.0}, {0.0, 0.0, 0.15}, {0.0, 0.0, 0.0}};
        for (int c = 0; c < 4; c++) {
            const double ax = angs[c][0], ay = angs[c][1], az = angs[c][2];
            const double Rx[9] = {1, 0, 0, 0, cos(ax), -sin(ax), 0, sin(ax), cos(ax)}, Ry[9] = {cos(ay), 0, sin(ay), 0, 1, 0, -sin(ay), 0, cos(ay)},
                         Rz[9] = {cos(az), -sin(az), 0, sin(az), cos(az), 0, 0, 0, 1};
            double T[9], Rm[9];
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i * 3 + j] = 0; for (int k = 0; k < 3; k++) T[i * 3 + j] += Rz[i * 3 + k] * Ry[k * 3 + j]; }
            for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm[i * 3 + j] = 0; for (int k = 0; k < 3; k++) Rm[i * 3 + j] += T[i * 3 + k] * Rx[k * 3 + j]; }
            for (int b = 0; b < B; b++) for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) th[b * 12 + i * 4 + j] = (float)Rm[i * 3 + j]; th[b * 12 + i * 4 + 3] = 0.01f * (i + 1); }
            CK(hipMemcpy(theta, th.data(), B * 12 * 4, hipMemcpyHostToDevice));
            printf("-- R(%.2f, %.2f, %.2f)\n", ax, ay, az);
            rep("fused, classic grid (max x pairs)", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, 0); }, reps));
            for (int g : {256, 512, 768, 1024}) {
                char nm[64]; snprintf(nm, sizeof nm, "fused, flat grid of %d blocks", g);
                rep(nm, time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(g, 1), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, zg, ru, gx); }, reps));
            }
            std::vector<int> hru(B); CK(hipMemcpy(hru.data(), ru, B * 4, hipMemcpyDeviceToHost));
            printf("   rows used %d\n", hru[0]);
        }
        return 0;
    }
    run_zs(trx::ZS64{}, "zstream 64x32 (again)");
    return 0;
}
