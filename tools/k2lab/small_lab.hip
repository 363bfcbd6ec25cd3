// tools/k2lab/small_lab.hip -- the one-kernel chain (kernels_small.hip built with SDRHIP_SMALL_PROBE): per-phase shader cycles,
// and microseconds per launch from a C caller on one stream and on two / four streams in turn.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -Isdr_amd/csrc tools/k2lab/small_lab.hip -o tools/k2lab/small_lab
//   tools/k2lab/small_lab [log2 samples = 20] [seam = 8192] [tile outputs = 0 (auto)]
#define SDRHIP_SMALL_PROBE 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>
#include <vector>
#include "../../sdr_amd/csrc/kernels_small.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace sdrhip;

int main(int argc, char** argv)
{
    const int64_t n = (int64_t)1 << (argc > 1 ? atoi(argv[1]) : 20);      // input samples
    const int64_t seam = argc > 2 ? atoll(argv[2]) : 8192;
    const int tile = argc > 3 ? atoi(argv[3]) : 0;
    const int64_t halo = 4400;
    std::vector<uint8_t> hu((size_t)2 * (n + halo));
    uint64_t s = 12345;
    for (auto& v : hu) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (uint8_t)(s >> 56); }
    std::vector<float> dt(128), g(3 * 64), rp(191), fh(64), fp(128);
    for (size_t i = 0; i < 127; i++) dt[i] = (0.01f * (float)((i * 29) % 31) - 0.15f) / 128.0f;
    dt[127] = 0.0f;
    for (size_t i = 0; i < g.size(); i++) g[i] = 0.01f * (float)((i * 37) % 23) - 0.1f;
    for (size_t i = 0; i < rp.size(); i++) rp[i] = 0.01f * (float)((i * 11) % 19) - 0.08f;
    for (size_t i = 0; i < fh.size(); i++) fh[i] = 0.02f * (float)((i * 7) % 13) - 0.1f;
    for (size_t i = 0; i < 64; i++) { fp[i] = fh[i]; fp[127 - i] = fh[i]; }
    const int64_t K = (n + halo - 128) / 8 + 1, M = (K * 3 - 192) / 10 + 1;
    const int64_t nq = (n / 8) * 3 / 10;           // outputs whose receptive field starts inside the shard, roughly
    if (nq + 127 > M) { printf("too small\n"); return 1; }
    uint8_t* du; float *audio[4], *dtaps, *groups, *rplain, *fhalf, *fplain;
    CK(hipMalloc(&du, hu.size()));
    CK(hipMemcpy(du, hu.data(), hu.size(), hipMemcpyHostToDevice));
    for (int i = 0; i < 4; i++) CK(hipMalloc(&audio[i], (size_t)nq * 4));
    CK(hipMalloc(&dtaps, 512)); CK(hipMalloc(&groups, g.size() * 4)); CK(hipMalloc(&rplain, 1024)); CK(hipMalloc(&fhalf, 256)); CK(hipMalloc(&fplain, 512));
    CK(hipMemcpy(dtaps, dt.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(groups, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rplain, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(fhalf, fh.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(fplain, fp.data(), 512, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_small_probe, sizeof(SmallProbe)));
    hipStream_t st[4];
    for (auto& x : st) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const int inc[3] = {4, 3, 3};
    auto run = [&](int i) { return launch_fm_chain_small(st[i], du, 0, n + halo, audio[i], 0, nq, 8, 128, dtaps, true, groups, 64, 64, inc, 3, 3, 10, 192,
                                                          rplain, 191, fhalf, 64, fplain, 0.2f, seam, tile); };
    const int A = tile > 0 ? tile : fm_chain_small_tile_outputs(nq);
    printf("samples %lld, audio outputs %lld, tile %d outputs, %lld workgroups\n", (long long)n, (long long)nq, A, (long long)((nq + A - 1) / A));
    for (int ns : {1, 2, 4}) {
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(g_small_probe, 0, sizeof(SmallProbe)));
            for (int i = 0; i < 200; i++) if (!run(i % ns)) { printf("not applicable\n"); return 1; }
            CK(hipDeviceSynchronize());
            CK(hipMemset(g_small_probe, 0, sizeof(SmallProbe)));
            const int reps = 2000;
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < reps; i++) run(i % ns);
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
            SmallProbe h; CK(hipMemcpy(&h, g_small_probe, sizeof h, hipMemcpyDeviceToHost));
            const double m = (double)h.n;
            const double tot = (double)(h.cyc[0] + h.cyc[1] + h.cyc[2] + h.cyc[3] + h.cyc[4]);
            printf("%d stream(s): %.2f us per launch = %.1f Gsample/s | per WG cycles: load %.0f mac %.0f demod %.0f resample %.0f filter %.0f, in-kernel %.2f us, clock %.0f MHz\n",
                   ns, us, n / us / 1e3, h.cyc[0] / m, h.cyc[1] / m, h.cyc[2] / m, h.cyc[3] / m, h.cyc[4] / m, h.rt / m / 100.0, tot / (double)h.rt * 100.0);
        }
    }
    return 0;
}
