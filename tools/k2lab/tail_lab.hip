// tools/k2lab/tail_lab.hip -- per-phase cycle counts of the fused tail kernel (kernels_tail.hip built with SDRHIP_TAIL_PROBE).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -Iinclude -Isdr_amd/csrc tools/k2lab/tail_lab.hip -o tools/k2lab/tail_lab
#define SDRHIP_TAIL_PROBE 1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../sdr_amd/csrc/kernels_tail.hip"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
using namespace sdrhip;

int main(int argc, char** argv)
{
    const int64_t nd = (int64_t)1 << (argc > 1 ? atoi(argv[1]) : 26);      // decimator outputs
    const int64_t seam = argc > 2 ? atoll(argv[2]) : 8192;
    std::vector<float> hd((size_t)2 * nd);
    uint64_t s = 12345;
    for (auto& v : hd) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = (float)((double)(s >> 11) / 9007199254740992.0 * 2.0 - 1.0) * 0.3f; }
    float *d, *audio, *groups, *rplain, *fhalf, *fplain;
    std::vector<float> g(3 * 64), rp(191), fh(64), fp(128);
    for (size_t i = 0; i < g.size(); i++) g[i] = 0.01f * (float)((i * 37) % 23) - 0.1f;
    for (size_t i = 0; i < rp.size(); i++) rp[i] = 0.01f * (float)((i * 11) % 19) - 0.08f;
    for (size_t i = 0; i < fh.size(); i++) fh[i] = 0.02f * (float)((i * 7) % 13) - 0.1f;
    for (size_t i = 0; i < 64; i++) { fp[i] = fh[i]; fp[127 - i] = fh[i]; }
    const int64_t nq = nd * 3 / 10 - 400;
    CK(hipMalloc(&d, hd.size() * 4)); CK(hipMalloc(&audio, (size_t)nq * 4));
    CK(hipMalloc(&groups, g.size() * 4)); CK(hipMalloc(&rplain, rp.size() * 4)); CK(hipMalloc(&fhalf, 256)); CK(hipMalloc(&fplain, 512));
    CK(hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(groups, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(rplain, rp.data(), rp.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(fhalf, fh.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(fplain, fp.data(), 512, hipMemcpyHostToDevice));
    CK(hipMalloc(&g_tail_probe, sizeof(TailProbe)));
    const int inc[3] = {4, 3, 3};
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto run = [&] { return launch_fm_tail_fused(0, d, 0, nd, 0, nd, audio, 0, nq, groups, 64, 64, inc, 3, 3, 10, 192, rplain, 191, fhalf, 64, fplain, 0.2f, seam); };
    for (int rep = 0; rep < 3; rep++) {
        CK(hipMemset(g_tail_probe, 0, sizeof(TailProbe)));
        run();
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < 5; i++) if (!run()) { printf("not applicable\n"); return 1; }
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        TailProbe h; CK(hipMemcpy(&h, g_tail_probe, sizeof h, hipMemcpyDeviceToHost));
        const double n = (double)h.n;
        printf("tail fused: %.3f ms per run (%lld decimator outputs, %lld audio), %.1f G y/s | per WG: demod %.0f resample %.0f filter %.0f cycles, %.2f us, clock %.0f MHz\n",
               ms / 5, (long long)nd, (long long)nq, nd / (ms / 5) / 1e6, h.cyc[0] / n, h.cyc[1] / n, h.cyc[2] / n, h.rt / n / 100.0,
               (h.cyc[0] + h.cyc[1] + h.cyc[2]) / (double)h.rt * 100.0);
    }
    return 0;
}
