// Development harness for the exact-footprint F1 body (not shipped): its 41 sums against the tile kernels' (GeomR / GeomRD through the fused
// step kernel) at a rotated pose, and hipEvent timing of both.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast -fno-slp-vectorize tools/ebench.hip -o build/ebench
//   build/ebench [B] [S] [ax ay az] [scale] [reps] [only]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../torchregister_amd/csrc/affine.hip"

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
    const double ax = argc > 3 ? atof(argv[3]) : 0.5, ay = argc > 4 ? atof(argv[4]) : 0.4, az = argc > 5 ? atof(argv[5]) : 0.3;
    const double scale = argc > 6 ? atof(argv[6]) : 1.0;
    const int reps = argc > 7 ? atoi(argv[7]) : 50;
    const bool only = argc > 8;   // profiling runs: only the exact-footprint kernel
    const size_t nvox = (size_t)S * S * S, n = nvox * B;
    std::vector<float> h(n);
    srand(1);
    for (size_t i = 0; i < n; i++) h[i] = (float)rand() / RAND_MAX;
    float *mov, *tgt, *theta, *partials;
    CK(hipMalloc(&mov, n * 4)); CK(hipMalloc(&tgt, n * 4));
    CK(hipMemcpy(mov, h.data(), n * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < n; i++) h[i] = (float)rand() / RAND_MAX;
    CK(hipMemcpy(tgt, h.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> th(B * 12);
    {
        const double Rx[9] = {1, 0, 0, 0, cos(ax), -sin(ax), 0, sin(ax), cos(ax)}, Ry[9] = {cos(ay), 0, sin(ay), 0, 1, 0, -sin(ay), 0, cos(ay)},
                     Rz[9] = {cos(az), -sin(az), 0, sin(az), cos(az), 0, 0, 0, 1};
        const double sc[3] = {1.05 * scale, 0.95 * scale, 1.02 * scale};
        double T[9], Rm[9];
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { T[i * 3 + j] = 0; for (int k = 0; k < 3; k++) T[i * 3 + j] += Rz[i * 3 + k] * Ry[k * 3 + j]; }
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { Rm[i * 3 + j] = 0; for (int k = 0; k < 3; k++) Rm[i * 3 + j] += T[i * 3 + k] * Rx[k * 3 + j]; }
        for (int b = 0; b < B; b++) for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) th[b * 12 + i * 4 + j] = (float)(Rm[i * 3 + j] * sc[j]); th[b * 12 + i * 4 + 3] = 0.01f * (i + 1) + 0.003f * b; }
    }
    CK(hipMalloc(&theta, B * 12 * 4));
    CK(hipMemcpy(theta, th.data(), B * 12 * 4, hipMemcpyHostToDevice));
    float *tab;
    CK(hipMalloc(&tab, 3 * S * 4));
    hipLaunchKernelGGL(trx::fill_tables_kernel, dim3((S + 255) / 256), dim3(256), 0, 0, tab, S, S, S);
    trx_volumes vol = {mov, tgt, nvox, nvox, 3, B, S, S, S, tab, tab + S, tab + 2 * S, 0};
    const size_t prow = 8192;
    CK(hipMalloc(&partials, (size_t)B * prow * 41 * 4));
    const double alg = 8.0 * nvox;
    auto rep = [&](const char *name, float us) { printf("%-40s %9.1f us/launch  %7.2f us/pair  %6.2f TB/s alg  frac %.3f\n", name, us, us / B, alg * B / us / 1e6, alg * B / us / 1e6 / 8.0); };
    auto sums = [&](int rows) {
        std::vector<float> hp((size_t)B * rows * 41);
        CK(hipMemcpy(hp.data(), partials, hp.size() * 4, hipMemcpyDeviceToHost));
        std::vector<double> out((size_t)B * 41, 0.0);
        for (int b = 0; b < B; b++) for (int r = 0; r < rows; r++) for (int k = 0; k < 41; k++) out[b * 41 + k] += hp[((size_t)b * rows + r) * 41 + k];
        return out;
    };
    printf("B=%d S=%d R(%.2f, %.2f, %.2f) diag(1.05, .95, 1.02) x %.2f\n", B, S, ax, ay, az, scale);
    const trx::TileGeom ta = trx::tile_geom<trx::GeomA>(vol), tr = trx::tile_geom<trx::GeomR>(vol), td = trx::tile_geom<trx::GeomD>(vol), trd = trx::tile_geom<trx::GeomRD>(vol);
    const int gx = std::max(std::max(ta.blocks_per_pair, tr.blocks_per_pair), std::max(td.blocks_per_pair, trd.blocks_per_pair));
    int *ru; CK(hipMalloc(&ru, (B + 11) * 4));
    std::vector<double> ref;
    if (!only) {
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 1, td, trd, trx::ZGeom{}, ru, 0);
        CK(hipDeviceSynchronize());
        ref = sums(gx);
        std::vector<int> hru(B); CK(hipMemcpy(hru.data(), ru, B * 4, hipMemcpyDeviceToHost));
        printf("tile kernels: %d rows used of %d\n", hru[0], gx);
    }
    CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
    hipLaunchKernelGGL((trx::affine_eft_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(512), 0, 0, vol, theta, trd, partials);
    CK(hipDeviceSynchronize());
    if (!only) {
        const std::vector<double> got = sums(trd.blocks_per_pair);
        double worst = 0; int wk = -1, wb = -1;
        bool nan = false;
        for (int b = 0; b < B; b++) {
            double scl = 0;
            for (int k = 5; k < 41; k++) scl = std::max(scl, fabs(ref[b * 41 + k]));
            for (int k = 0; k < 41; k++) {
                if (!(got[b * 41 + k] == got[b * 41 + k])) nan = true;
                const double e = fabs(got[b * 41 + k] - ref[b * 41 + k]) / (k < 5 ? std::max(1.0, fabs(ref[b * 41 + k])) : scl);
                if (e > worst) { worst = e; wk = k; wb = b; }
            }
        }
        printf("exact-footprint vs tile kernels: worst relative difference of the 41 sums %.3e (sum %d, pair %d)%s   Sw %.4f / %.4f  Syw %.4f / %.4f  S[5] %.5f / %.5f\n", worst, wk, wb,
               nan ? "  NaN" : "", got[1], ref[1], got[4], ref[4], got[5], ref[5]);
    }
    rep("exact-footprint 16^3 tiles", time_it([&] { hipLaunchKernelGGL((trx::affine_eft_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(512), 0, 0, vol, theta, trd, partials); }, reps));
#if TRX_EF_STAMP
    {   // the flat step kernel once more, then its stamps: phases of a tile step per wave (cycles), blocks' start / end (100 MHz), blocks per CU
        // (as behind affine_zs_step_kernel: nothing taken in front - rows_used[B] = B pairs left, empty mask - and the eight work tickets zeroed per launch)
        { const int note[3] = {B, 0, 0}; CK(hipMemcpy(ru + B, note, sizeof(note), hipMemcpyHostToDevice)); }
        auto ef_flat = [&] { CK(hipMemsetAsync(ru + B + 3, 0, 8 * sizeof(int), 0)); hipLaunchKernelGGL((trx::affine_eft_step_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, trd, partials, ru, gx, 1, 1, -1); };
        rep("EF step kernel, flat grid (stamped)", time_it(ef_flat, reps));
        std::vector<unsigned long long> st(1024 * 8 * 8);
        CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(trx::trx_ef_stamps), st.size() * 8));
        double sum[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, issue_t = 0;
        int nb = 0;
        for (int b = 0; b < 1024; b++) {
            if (st[(size_t)b * 64 + 4] == 0) continue;
            nb++;
            for (int w = 0; w < 8; w++) {
                const unsigned long long *o = &st[((size_t)b * 8 + w) * 8];
                for (int k = 0; k < 4; k++) sum[k] += (double)(o[k] & 0xffffffffull);
                sum[9] += (double)(o[0] >> 32); issue_t += (double)(o[1] >> 32);
                sum[4] += (double)(o[4] & 0xffffffffull); sum[8] += (double)(o[4] >> 32);
                sum[5] += (double)(o[5] & 0xffff);
                sum[6] += (double)o[6]; sum[7] += (double)o[7];
            }
        }
        const double nw = nb * 8.0;
        printf("   stamps of %d items (s_memtime ticks, mean over their waves): %.1f tiles in the walk; per tile: issue %.0f  gather %.0f  wait+barrier %.0f = %.0f\n", nb, sum[5] / nw,
               sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5], (sum[0] + sum[1] + sum[2]) / sum[5]);
        {
            std::vector<unsigned long long> s2(1024 * 8 * 8);
            CK(hipMemcpyFromSymbol(s2.data(), HIP_SYMBOL(trx::trx_ef_stamps2), s2.size() * 8));
            double m[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b < 1024; b++) { if (st[(size_t)b * 64 + 4] == 0) continue; for (int w = 0; w < 8; w++) for (int k = 0; k < 8; k++) m[k] += (double)s2[((size_t)b * 8 + w) * 8 + k]; }
            printf("   the prologue: windows + wave scans %.0f, barrier %.0f, totals + barrier %.0f, table + descriptors + barrier %.0f, granule registers + barrier %.0f | origins + barrier %.0f, head tiles %.0f, first tile requested %.0f\n",
                   m[0] / nw, m[1] / nw, m[2] / nw, m[3] / nw, m[4] / nw, m[5] / nw, m[6] / nw, m[7] / nw);
        }
        printf("   of the request phase: origin of the next tile %.0f, yn + eight target rows %.0f, the DMA pieces (and, on boundary tiles, their zero fills) %.0f\n", sum[9] / sum[5], issue_t / sum[5], (sum[0] - sum[9] - issue_t) / sum[5]);
        {   // spread of the items' costs (the kernel ends with its slowest block: four items each)
            std::vector<double> tot, tl;
            for (int b = 0; b < 1024; b++) { if (st[(size_t)b * 64 + 4] == 0) continue; double t = 0, n = 0; for (int w = 0; w < 8; w++) { t += (double)(st[((size_t)b * 8 + w) * 8 + 4] >> 32); n += (double)(st[((size_t)b * 8 + w) * 8 + 5] & 0xffff); } tot.push_back(t / 8); tl.push_back(n / 8); }
            std::vector<double> srt = tot; std::sort(srt.begin(), srt.end());
            printf("   items by cost: min %.0f  p10 %.0f  median %.0f  p90 %.0f  max %.0f ticks;", srt[0], srt[srt.size() / 10], srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt.back());
            double c[17] = {0}, k[17] = {0};
            for (size_t i = 0; i < tot.size(); i++) { const int n = (int)(tl[i] + 0.5); c[n] += tot[i]; k[n] += 1; }
            {   // the dearest tenth of the items: where their tile steps differ from everybody's
                std::vector<int> idx;
                for (int b = 0, i = 0; b < 1024; b++) { if (st[(size_t)b * 64 + 4] == 0) continue; if (tot[i] >= srt[srt.size() * 9 / 10]) idx.push_back(b); i++; }
                double a[4] = {0, 0, 0, 0};
                for (int b : idx) for (int w = 0; w < 8; w++) { const unsigned long long *o = &st[((size_t)b * 8 + w) * 8]; for (int k = 0; k < 3; k++) a[k] += (double)(o[k] & 0xffffffffull); a[3] += (double)(o[5] & 0xffff); }
                printf(" the dearest tenth: %.1f tiles, per tile request %.0f gather %.0f wait %.0f;", a[3] / (idx.size() * 8.0), a[0] / a[3], a[1] / a[3], a[2] / a[3]);
            }
            printf(" by tiles in the walk:");
            for (int n = 0; n <= 16; n++) if (k[n] > 0) printf(" %d: %.0f x %.0f k", n, k[n], c[n] / k[n] / 1000);
            printf("\n");
        }
        {   // the blocks of the flat launch against the 100 MHz clock
            std::vector<unsigned long long> bl(1024 * 8);
            CK(hipMemcpyFromSymbol(bl.data(), HIP_SYMBOL(trx::trx_ef_blocks), bl.size() * 8));
            unsigned long long s0 = ~0ull;
            for (int b = 0; b < 512; b++) s0 = std::min(s0, bl[b * 8]);
            std::vector<double> en, st0, first;
            for (int b = 0; b < 512; b++) { unsigned long long e = 0; for (int k = 1; k < 8; k++) e = std::max(e, bl[b * 8 + k]); en.push_back((e - s0) / 100.0); st0.push_back((bl[b * 8] - s0) / 100.0); first.push_back((bl[b * 8 + 1] - bl[b * 8]) / 100.0); }
            std::vector<double> se = en; std::sort(se.begin(), se.end());
            std::vector<double> sf = first; std::sort(sf.begin(), sf.end());
            std::sort(st0.begin(), st0.end());
            printf("   blocks of the flat launch (us since the first one started): last start %.1f; end of a block's LAST item: min %.1f  p10 %.1f  median %.1f  p90 %.1f  max %.1f; its first item done after: min %.1f median %.1f max %.1f\n",
                   st0.back(), se[0], se[51], se[256], se[460], se.back(), sf[0], sf[256], sf.back());
        }
        printf("   an item: plan done at %.0f, first tile landed at %.0f, walk done at %.0f, tail (tiles outside the volume) done at %.0f, sums stored at %.0f\n", sum[3] / nw, sum[6] / nw, sum[7] / nw, sum[4] / nw, sum[8] / nw);
    }
#endif
    if (only) return 0;
    rep("tile kernels, classic grid", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, trx::ZGeom{}, ru, 0); }, reps));
    rep("tile kernels, flat grid 512", time_it([&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, trx::ZGeom{}, ru, gx); }, reps));
    {   // what trx_affine_step launches: the exact-footprint step kernel (takes the rotated pairs, marks them rows_used < 0) + the fused kernel behind it
        const int er = trd.blocks_per_pair;
        auto ef_flat = [&] { hipLaunchKernelGGL((trx::affine_eft_step_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, trd, partials, ru, gx, 1, 1, 0); };
        { const int note[3] = {B, 0, 0}; CK(hipMemcpy(ru + B, note, sizeof(note), hipMemcpyHostToDevice)); }
        auto ef_flat_tickets = [&] { CK(hipMemsetAsync(ru + B + 3, 0, 8 * sizeof(int), 0)); hipLaunchKernelGGL((trx::affine_eft_step_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, trd, partials, ru, gx, 1, 1, -1); };
        rep("   EF step kernel alone, flat, items drawn (+ a memset)", time_it(ef_flat_tickets, reps));
        auto ef_classic = [&] { hipLaunchKernelGGL((trx::affine_eft_step_kernel<0>), dim3(er, B), dim3(512), 0, 0, vol, theta, trd, partials, ru, -gx, 1, 1, 0); };
        auto fused_flat = [&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(512, 1), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, trx::ZGeom{}, ru, gx, 1); };
        auto fused_classic = [&] { hipLaunchKernelGGL((trx::affine_tile_dual_kernel<0>), dim3(gx, B), dim3(512), 0, 0, vol, theta, ta, tr, 1, partials, 0, td, trd, trx::ZGeom{}, ru, 0, 1); };
        rep("   EF step kernel alone, flat", time_it(ef_flat, reps));
        rep("   EF step kernel alone, classic", time_it(ef_classic, reps));
        rep("   fused kernel alone (skips every pair), flat", time_it(fused_flat, reps));
        rep("step launches (EF kernel + fused), flat", time_it([&] { ef_flat(); fused_flat(); }, reps));
        CK(hipMemset(partials, 0, (size_t)B * prow * 41 * 4));
        ef_flat(); fused_flat();
        CK(hipDeviceSynchronize());
        std::vector<int> hru(B); CK(hipMemcpy(hru.data(), ru, B * 4, hipMemcpyDeviceToHost));
        const std::vector<double> got = sums(gx);
        double worst = 0;
        for (int b = 0; b < B; b++) for (int k = 0; k < 5; k++) worst = std::max(worst, fabs(got[b * 41 + k] - ref[b * 41 + k]) / std::max(1.0, fabs(ref[b * 41 + k])));
        printf("   rows_used[0] = %d; moments vs tile kernels %.2e\n", hru[0], worst);
        rep("step launches (EF kernel + fused), classic", time_it([&] { ef_classic(); fused_classic(); }, reps));
    }
    rep("exact-footprint 16^3 tiles (again)", time_it([&] { hipLaunchKernelGGL((trx::affine_eft_kernel<0>), dim3(trd.blocks_per_pair, B), dim3(512), 0, 0, vol, theta, trd, partials); }, reps));
    return 0;
}
