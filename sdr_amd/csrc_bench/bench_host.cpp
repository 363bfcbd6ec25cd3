// bench_host.cpp -- measurement utilities (NOT part of the FM path): timing loops over the host-block operators written
// against the PUBLIC C ABI only, so that bench.py reports what a compiled caller (the Haskell pipeline, examples/fm_replay.c)
// pays per push -- a Python loop adds 10-20 us of interpreter and ctypes overhead to every call, which at the reference's
// block size (8192 samples) is as much as the work itself.
#include <string.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "common.hpp"

using namespace sdrhip;

extern "C" {

// Push `pushes` batches of `n_samples` u8 IQ samples through a fresh sdrhip_fm_stream on `chain` and pop every audio
// block; zero_copy: the source writes into the operator's pinned staging buffer (as the RTL-SDR read would), else
// sdrhip_fm_stream_push copies from the caller's buffer.  *samples_per_s = n_samples * pushes / wall time (flush included).
int sdrhip_bench_fm_stream(sdrhip_fm_chain* chain, int n_samples, int pushes, int zero_copy, int coalesce_samples, double* samples_per_s,
                           long long* audio_blocks)
{
    SDRHIP_REQUIRE(chain != nullptr && n_samples > 0 && pushes > 0 && samples_per_s != nullptr, "sdrhip_bench_fm_stream");
    sdrhip_fm_stream* st = nullptr;
    int rc = sdrhip_fm_stream_create(&st, chain, n_samples, 8192);
    if (rc != SDRHIP_OK) return rc;
    if (coalesce_samples == 1) rc = sdrhip_fm_stream_set_adaptive(st, 0);        // every push on its own
    else if (coalesce_samples < 0) rc = sdrhip_fm_stream_set_adaptive(st, -coalesce_samples);
    else if (coalesce_samples > 0) rc = sdrhip_fm_stream_set_coalesce(st, coalesce_samples);
    if (rc != SDRHIP_OK) { sdrhip_fm_stream_destroy(st); return rc; }
    std::vector<uint8_t> src((size_t)2 * n_samples);
    uint32_t s = 12345u;
    for (auto& b : src) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
    std::vector<float> out(8192);
    long long blocks = 0;
    auto one = [&](int i) -> int {
        int ready;
        if (zero_copy) {
            uint8_t* dst = sdrhip_fm_stream_input_buffer(st);
            if (!dst) return SDRHIP_ERR_STATE;
            // the "radio" writes the lent buffer itself (DMA): the timed figure carries no source-side copy.  Two pushes in a row
            // are filled now and then, so that BOTH staging slots hold real samples (consecutive pushes alternate slots)
            if ((i & 63) < 2) memcpy(dst, src.data(), src.size());
            ready = sdrhip_fm_stream_push(st, dst, n_samples);
        } else {
            ready = sdrhip_fm_stream_push(st, src.data(), n_samples);
        }
        if (ready < 0) return ready;
        while (ready-- > 0) {
            if (sdrhip_fm_stream_pop(st, out.data(), 8192) > 0) blocks++;
        }
        return SDRHIP_OK;
    };
    const int warm = pushes / 10 + 4;
    for (int i = 0; i < warm; i++)
        if ((rc = one(i)) != SDRHIP_OK) { sdrhip_fm_stream_destroy(st); return rc; }
    blocks = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < pushes; i++)
        if ((rc = one(i)) != SDRHIP_OK) { sdrhip_fm_stream_destroy(st); return rc; }
    int ready = sdrhip_fm_stream_flush(st);
    while (ready-- > 0)
        if (sdrhip_fm_stream_pop(st, out.data(), 8192) > 0) blocks++;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *samples_per_s = (double)n_samples * pushes / dt;
    if (audio_blocks) *audio_blocks = blocks;
    sdrhip_fm_stream_destroy(st);
    return SDRHIP_OK;
}

// Push-to-audio LATENCY of sdrhip_fm_stream (round 4): `pushes` pushes of `n_samples` u8 IQ samples; audio leaves the operator
// in blocks of 256 samples (every push of 8192 samples completes at least one), and every popped block is charged to the push
// that made it computable (sdrhip_fm_chain_ready): latency = host time of the pop - host time at which that push was called.
// pace_us > 0: one push every pace_us microseconds (a 1.28 MS/s source delivers 8192 samples every 6400 us), polling for the
// audio in between, as a real-time consumer would; 0: back to back, collecting what is ready at every push (results then lag
// behind by the submissions in flight).  adaptive_off: every push its own submission.  out[0..4] = p50, p99, max, mean latency
// and mean duration of the push call itself, all in microseconds.
int sdrhip_bench_fm_stream_latency(sdrhip_fm_chain* chain, int n_samples, int pushes, double pace_us, int adaptive_off, double* out)
{
    SDRHIP_REQUIRE(chain != nullptr && n_samples > 0 && pushes > 8 && out != nullptr, "sdrhip_bench_fm_stream_latency");
    sdrhip_fm_stream* st = nullptr;
    constexpr int kOut = 256;
    int rc = sdrhip_fm_stream_create(&st, chain, n_samples, kOut);
    if (rc != SDRHIP_OK) return rc;
    if (adaptive_off && (rc = sdrhip_fm_stream_set_adaptive(st, 0)) != SDRHIP_OK) { sdrhip_fm_stream_destroy(st); return rc; }
    std::vector<uint8_t> src((size_t)2 * n_samples);
    uint32_t sd = 99u;
    for (auto& b : src) { sd = sd * 1664525u + 1013904223u; b = (uint8_t)(sd >> 24); }
    using clk = std::chrono::steady_clock;
    const int warm = 32;
    std::vector<clk::time_point> t_push((size_t)warm + pushes);
    std::vector<int64_t> blocks_after((size_t)warm + pushes);       // audio blocks computable once push i is in
    for (int i = 0; i < warm + pushes; i++) blocks_after[i] = sdrhip_fm_chain_ready(chain, (int64_t)(i + 1) * n_samples) / kOut;
    std::vector<double> lat;
    lat.reserve((size_t)pushes * 2);
    std::vector<float> blk(kOut);
    int64_t popped = 0;
    size_t owner = 0;                                               // first push whose blocks_after exceeds `popped`
    double push_call_us = 0.0;
    auto collect = [&](int ready) {
        const auto now = clk::now();
        while (ready-- > 0) {
            if (sdrhip_fm_stream_pop(st, blk.data(), kOut) <= 0) break;
            while (owner < blocks_after.size() && blocks_after[owner] <= popped) owner++;
            if (owner < blocks_after.size() && owner >= (size_t)warm) lat.push_back(std::chrono::duration<double, std::micro>(now - t_push[owner]).count());
            popped++;
        }
    };
    const auto t_start = clk::now();
    for (int i = 0; i < warm + pushes; i++) {
        if (pace_us > 0) {
            const auto due = t_start + std::chrono::duration_cast<clk::duration>(std::chrono::duration<double, std::micro>(pace_us * i));
            // a real-time consumer polls for audio while it waits for the next block of the source
            while (clk::now() < due) {
                const int r = sdrhip_fm_stream_poll(st);
                if (r < 0) { sdrhip_fm_stream_destroy(st); return r; }
                collect(r);
            }
        }
        t_push[i] = clk::now();
        const int r = sdrhip_fm_stream_push(st, src.data(), n_samples);
        if (r < 0) { sdrhip_fm_stream_destroy(st); return r; }
        if (i >= warm) push_call_us += std::chrono::duration<double, std::micro>(clk::now() - t_push[i]).count();
        collect(r);
    }
    {
        const int r = sdrhip_fm_stream_flush(st);
        if (r < 0) { sdrhip_fm_stream_destroy(st); return r; }
        collect(r);
    }
    sdrhip_fm_stream_destroy(st);
    if (lat.empty()) { set_error("sdrhip_bench_fm_stream_latency: no audio block came back"); return SDRHIP_ERR_STATE; }
    std::sort(lat.begin(), lat.end());
    double mean = 0.0;
    for (double v : lat) mean += v;
    out[0] = lat[lat.size() / 2];
    out[1] = lat[(size_t)((lat.size() - 1) * 0.99)];
    out[2] = lat.back();
    out[3] = mean / (double)lat.size();
    out[4] = push_call_us / pushes;
    return SDRHIP_OK;
}

// The same for one Pipe (firDecimator / firFilter / firResampler): `n` elements per push (floats, or complex pairs for a
// complex stage), zero-copy staging or not; *elements_per_s = n * pushes / wall time.
int sdrhip_bench_pipe(sdrhip_pipe* p, int n, int floats_per_element, int block_size_out, int pushes, int zero_copy, double* elements_per_s)
{
    SDRHIP_REQUIRE(p != nullptr && n > 0 && pushes > 0 && elements_per_s != nullptr && floats_per_element >= 1, "sdrhip_bench_pipe");
    std::vector<float> src((size_t)n * floats_per_element);
    uint32_t s = 777u;
    for (auto& v : src) { s = s * 1664525u + 1013904223u; v = (float)(int32_t)s * (1.0f / 2147483648.0f); }
    std::vector<float> out((size_t)block_size_out * 2);
    auto one = [&](int i) -> int {
        int ready;
        if (zero_copy) {
            float* dst = sdrhip_pipe_input_buffer(p, n);
            if (!dst) return SDRHIP_ERR_STATE;
            if ((i & 63) < 2) memcpy(dst, src.data(), src.size() * sizeof(float));      // both staging slots, as above
            ready = sdrhip_pipe_push(p, dst, n);
        } else {
            ready = sdrhip_pipe_push(p, src.data(), n);
        }
        if (ready < 0) return ready;
        while (ready-- > 0) (void)sdrhip_pipe_pop(p, out.data(), (int)out.size());
        return SDRHIP_OK;
    };
    int rc;
    for (int i = 0; i < pushes / 10 + 4; i++)
        if ((rc = one(i)) != SDRHIP_OK) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < pushes; i++)
        if ((rc = one(i)) != SDRHIP_OK) return rc;
    int ready = sdrhip_pipe_flush(p);
    while (ready-- > 0) (void)sdrhip_pipe_pop(p, out.data(), (int)out.size());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *elements_per_s = (double)n * pushes / dt;
    return SDRHIP_OK;
}

// The FM receiver as the reference composes it (examples/fm/fm.hs:34-41) out of four Level-1 Pipes: firDecimator -> fmDemod ->
// firResampler -> firFilter, every stage re-blocking to `block` elements, fed `pushes` cfloat blocks of `block` samples
// (the convert is the caller's P.map); every output block of a stage is popped and pushed into the next one by this loop, as
// the Pipes library would.  *samples_per_s = source samples per wall second; *audio_blocks = blocks that left the filter.
int sdrhip_bench_fm_pipes(const sdrhip_decimator* dec, const sdrhip_resampler* res, const sdrhip_filter* fil, int block, int pushes,
                          double* samples_per_s, long long* audio_blocks)
{
    SDRHIP_REQUIRE(dec && res && fil && block > 0 && pushes > 0 && samples_per_s, "sdrhip_bench_fm_pipes");
    sdrhip_pipe *pd = nullptr, *pm = nullptr, *pr = nullptr, *pf = nullptr;
    int rc = sdrhip_pipe_fir_decimator(&pd, dec, block);
    if (rc == SDRHIP_OK) rc = sdrhip_pipe_fm_demod(&pm);
    if (rc == SDRHIP_OK) rc = sdrhip_pipe_fir_resampler(&pr, res, block);
    if (rc == SDRHIP_OK) rc = sdrhip_pipe_fir_filter(&pf, fil, block);
    auto cleanup = [&]() {
        for (sdrhip_pipe* p : {pd, pm, pr, pf})
            if (p) sdrhip_pipe_destroy(p);
    };
    if (rc != SDRHIP_OK) { cleanup(); return rc; }
    std::vector<float> src((size_t)2 * block), a((size_t)2 * block), b((size_t)2 * block);
    uint32_t s = 4242u;
    for (auto& v : src) { s = s * 1664525u + 1013904223u; v = (float)(int32_t)s * (1.0f / 2147483648.0f); }
    long long blocks = 0;
    // drain stage `from` downwards: every ready block goes into the next stage, whose ready blocks go on
    auto cascade = [&](int ready_dec) -> int {
        int r;
        while (ready_dec-- > 0) {
            int n1 = sdrhip_pipe_pop(pd, a.data(), block);
            if (n1 <= 0) continue;
            int ready_dem = sdrhip_pipe_push(pm, a.data(), n1);
            if (ready_dem < 0) return ready_dem;
            while (ready_dem-- > 0) {
                int n2 = sdrhip_pipe_pop(pm, b.data(), 2 * block);
                if (n2 <= 0) continue;
                int ready_res = sdrhip_pipe_push(pr, b.data(), n2);
                if (ready_res < 0) return ready_res;
                while (ready_res-- > 0) {
                    int n3 = sdrhip_pipe_pop(pr, a.data(), block);
                    if (n3 <= 0) continue;
                    r = sdrhip_pipe_push(pf, a.data(), n3);
                    if (r < 0) return r;
                    while (r-- > 0)
                        if (sdrhip_pipe_pop(pf, b.data(), block) > 0) blocks++;
                }
            }
        }
        return SDRHIP_OK;
    };
    auto one = [&]() -> int {
        int ready = sdrhip_pipe_push(pd, src.data(), block);
        if (ready < 0) return ready;
        return cascade(ready);
    };
    for (int i = 0; i < pushes / 10 + 64; i++)
        if ((rc = one()) != SDRHIP_OK) { cleanup(); return rc; }
    blocks = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < pushes; i++)
        if ((rc = one()) != SDRHIP_OK) { cleanup(); return rc; }
    // drain: flush each stage in turn and hand what it releases downstream
    if ((rc = sdrhip_pipe_flush(pd)) < 0 || (rc = cascade(rc)) != SDRHIP_OK) { cleanup(); return rc < 0 ? rc : SDRHIP_ERR_STATE; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    *samples_per_s = (double)block * pushes / dt;
    if (audio_blocks) *audio_blocks = blocks;
    cleanup();
    return SDRHIP_OK;
}

}  // extern "C"
