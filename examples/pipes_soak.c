/* pipes_soak.c -- the host-block operators of libsdr_hip.so driven the awkward ways, in plain C, for sanitizer runs
 * (AddressSanitizer / UndefinedBehaviorSanitizer builds: `SDRHIP_ASAN=1 python -m sdr_amd.build`; tests/test_gpu_sanitizers.py).
 * The reference's Pipes (hs_sources/SDR/Filter.hs:536-727) see blocks of any length the source hands them; the library's
 * versions add pinned rings, leased streams, lent staging buffers, coalesced and adaptive submission, checkpoints and helper
 * threads for large copies -- 3 k lines of pointer-heavy host code.  This program walks those paths:
 *   1. firDecimator / firResampler / firFilter / fmDemod / dcBlockingFilter Pipes fed RAGGED blocks (lengths from a small LCG,
 *      never shorter than the filter), pushed from caller memory and from the lent staging buffer in turn, with coalescing,
 *      with adaptive submission, with neither; popped at odd times; saved and restored half way into a fresh Pipe;
 *   2. the whole-receiver operator (sdrhip_fm_stream) with pushes of 1, 7, 64 and 4096 source blocks from caller memory (the
 *      4096-block push goes through the threaded copy) and from its own staging buffer, flush, checkpoint, restore, destroy;
 *   3. every handle destroyed, some of them while results are still pending.
 * Values are not checked here (tests/test_gpu_pipes.py does, against the restated Pipes); a wrong total count, an error return or
 * any sanitizer report fails the run.  Exit code 0 and "pipes_soak: ok" = clean. */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "sdr_hip.h"

#define BLOCK 8192
#define PI 3.14159265358979323846

static void check(int rc, const char *what)
{
    if (rc < 0) { fprintf(stderr, "pipes_soak: %s: %s\n", what, sdrhip_last_error()); exit(1); }
}

static unsigned lcg_state = 12345u;
static unsigned lcg(void) { lcg_state = lcg_state * 1664525u + 1013904223u; return lcg_state >> 8; }

static void lowpass(float *h, int n, double cutoff)
{
    for (int i = 0; i < n; i++) {
        const double t = i - (n - 1) / 2.0;
        const double s = t == 0.0 ? 2.0 * cutoff : sin(2.0 * PI * cutoff * t) / (PI * t);
        h[i] = (float)(s * (0.54 - 0.46 * cos(2.0 * PI * i / (n - 1))));
    }
}

/* feed `total` elements in ragged blocks; mode 0 plain, 1 coalesce 3, 2 adaptive 8; every other block through the lent buffer */
static long long soak_pipe(sdrhip_pipe *p, int fpe, int min_len, int max_len, long long total, int mode, int block_out, int save_half_way,
                           sdrhip_pipe *fresh, int can_lend /* filter / decimator / resampler pipes lend their staging buffer */)
{
    float *src = (float *)malloc((size_t)max_len * fpe * sizeof(float));
    float *out = (float *)malloc((size_t)block_out * fpe * sizeof(float));
    for (int i = 0; i < max_len * fpe; i++) src[i] = (float)((int)(lcg() & 0xffff) - 32768) / 32768.0f;
    if (mode == 1) check(sdrhip_pipe_set_coalesce(p, 3), "set_coalesce");
    if (mode == 2) check(sdrhip_pipe_set_adaptive(p, 8), "set_adaptive");
    if (mode == 0) check(sdrhip_pipe_set_adaptive(p, 0), "set_adaptive off");
    long long fed = 0, popped = 0;
    int k = 0, saved = 0;
    while (fed < total) {
        int n = min_len + (int)(lcg() % (unsigned)(max_len - min_len + 1));
        if (mode == 1 && (k % 5) != 4) n = max_len;                        /* coalescing needs equal-sized pushes; every fifth ends the run */
        int ready;
        if ((k & 1) && can_lend) {
            float *lent = sdrhip_pipe_input_buffer(p, n);
            if (!lent) check(-1, "sdrhip_pipe_input_buffer");
            memcpy(lent, src, (size_t)n * fpe * sizeof(float));
            ready = sdrhip_pipe_push(p, lent, n);
        } else {
            ready = sdrhip_pipe_push(p, src + (lcg() % 3) * fpe, n - 3 > min_len ? n - 3 : n);   /* unaligned caller memory */
            if (n - 3 > min_len) n -= 3;
        }
        check(ready, "sdrhip_pipe_push");
        fed += n;
        k++;
        if ((k % 3) == 0) ready = sdrhip_pipe_poll(p);
        check(ready, "sdrhip_pipe_poll");
        while (ready > 0 && (lcg() & 3) != 0) {                            /* leave some blocks pending now and then */
            int got = sdrhip_pipe_pop(p, out, block_out);
            check(got, "sdrhip_pipe_pop");
            if (got == 0) break;
            popped += got;
            ready--;
        }
        if (save_half_way && !saved && fed >= total / 2 && fresh) {
            size_t need = sdrhip_pipe_state_bytes(p), used = 0;
            if (need == 0) check(-1, "sdrhip_pipe_state_bytes");
            void *buf = malloc(need);                                       /* exactly `need`: one byte less must be refused */
            if (need > 1 && sdrhip_pipe_save(p, buf, need - 1, &used) >= 0) { fprintf(stderr, "pipes_soak: save into a short buffer succeeded\n"); exit(1); }
            check(sdrhip_pipe_save(p, buf, need, &used), "sdrhip_pipe_save");
            check(sdrhip_pipe_restore(fresh, buf, used), "sdrhip_pipe_restore");
            free(buf);
            sdrhip_pipe *t = p; p = fresh; fresh = t;                       /* go on with the restored one; the old one is destroyed with results pending */
            saved = 1;
            if (mode == 1) check(sdrhip_pipe_set_coalesce(p, 3), "set_coalesce");
            if (mode == 2) check(sdrhip_pipe_set_adaptive(p, 8), "set_adaptive");
        }
    }
    int ready = sdrhip_pipe_flush(p);
    check(ready, "sdrhip_pipe_flush");
    for (;;) {
        int got = sdrhip_pipe_pop(p, out, block_out);
        check(got, "sdrhip_pipe_pop");
        if (got == 0) break;
        popped += got;
    }
    free(src);
    free(out);
    sdrhip_pipe_destroy(p);
    if (fresh) sdrhip_pipe_destroy(fresh);
    return popped;
}

int main(int argc, char **argv)
{
    const long long scale = argc > 1 ? atoll(argv[1]) : 1;
    float h127[127], h191[191], half64[64], full128[128];
    lowpass(h127, 127, 1.0 / 16);
    lowpass(h191, 191, 1.0 / 10);
    lowpass(full128, 128, 0.3);
    memcpy(half64, full128, sizeof half64);
    sdrhip_decimator *dec = NULL;
    sdrhip_resampler *res = NULL;
    sdrhip_filter *fil = NULL;
    check(sdrhip_decimator_create(&dec, SDRHIP_ORDER_AVX, 1, 8, h127, 127), "sdrhip_decimator_create");
    check(sdrhip_resampler_create(&res, SDRHIP_ORDER_AVX, 0, 3, 10, h191, 191), "sdrhip_resampler_create");
    check(sdrhip_filter_sym_create(&fil, SDRHIP_ORDER_AVX, half64, 64), "sdrhip_filter_sym_create");
    long long outs = 0;
    for (int mode = 0; mode < 3; mode++) {
        sdrhip_pipe *p = NULL, *q = NULL;
        check(sdrhip_pipe_fir_decimator(&p, dec, BLOCK), "pipe decimator");
        check(sdrhip_pipe_fir_decimator(&q, dec, BLOCK), "pipe decimator");
        outs += soak_pipe(p, 2, 128, 3 * BLOCK, 400000 * scale, mode, BLOCK, 1, q, 1);
        check(sdrhip_pipe_fir_resampler(&p, res, BLOCK), "pipe resampler");
        check(sdrhip_pipe_fir_resampler(&q, res, BLOCK), "pipe resampler");
        outs += soak_pipe(p, 1, 192, 65536, 900000 * scale, mode, BLOCK, 1, q, 1);
        check(sdrhip_pipe_fir_filter(&p, fil, 1000), "pipe filter");                 /* an output block that divides nothing */
        outs += soak_pipe(p, 1, 128, 20000, 300000 * scale, mode, 1000, 0, NULL, 1);
    }
    {
        sdrhip_pipe *p = NULL, *q = NULL;
        check(sdrhip_pipe_fm_demod(&p), "pipe fm_demod");
        check(sdrhip_pipe_fm_demod(&q), "pipe fm_demod");
        outs += soak_pipe(p, 2, 1, 30000, 200000 * scale, 0, 30000, 1, q, 0);
        check(sdrhip_pipe_dc_blocker(&p), "pipe dc_blocker");
        outs += soak_pipe(p, 1, 1, 30000, 200000 * scale, 0, 30000, 0, NULL, 0);
    }
    /* the whole-receiver operator */
    sdrhip_fm_chain *chain = NULL;
    check(sdrhip_fm_chain_create(&chain, SDRHIP_ORDER_AVX, 8, h127, 127, 3, 10, h191, 191, half64, 64, 0.2f, BLOCK), "sdrhip_fm_chain_create");
    static const int bpps[4] = {1, 7, 64, 4096};
    float *audio = (float *)malloc(BLOCK * sizeof(float));
    for (int c = 0; c < 4; c++) {
        const int bpp = bpps[c];
        sdrhip_fm_stream *st = NULL, *st2 = NULL;
        check(sdrhip_fm_stream_create(&st, chain, bpp * BLOCK, BLOCK), "sdrhip_fm_stream_create");
        check(sdrhip_fm_stream_create(&st2, chain, bpp * BLOCK, BLOCK), "sdrhip_fm_stream_create");
        uint8_t *mine = (uint8_t *)malloc((size_t)bpp * 2 * BLOCK + 5);
        for (size_t i = 0; i < (size_t)bpp * 2 * BLOCK + 5; i++) mine[i] = (uint8_t)(lcg() & 0xff);
        const int pushes = bpp >= 4096 ? 3 : (bpp >= 64 ? 6 : 40);
        for (int k = 0; k < pushes; k++) {
            int n = bpp > 1 ? (1 + (int)(lcg() % (unsigned)bpp)) * BLOCK : BLOCK;    /* ragged: whole source blocks, any number of them */
            if (bpp >= 4096) n = bpp * BLOCK;
            int ready;
            if (k & 1) {
                uint8_t *lent = sdrhip_fm_stream_input_buffer(st);
                if (!lent) check(-1, "sdrhip_fm_stream_input_buffer");
                memcpy(lent, mine, (size_t)n * 2);
                ready = sdrhip_fm_stream_push(st, lent, n);
            } else {
                ready = sdrhip_fm_stream_push(st, mine + (k % 5), n);               /* unaligned caller memory */
            }
            check(ready, "sdrhip_fm_stream_push");
            while (ready-- > 0 && (lcg() & 1)) { check(sdrhip_fm_stream_pop(st, audio, BLOCK), "sdrhip_fm_stream_pop"); outs += BLOCK; }
            if (k == pushes / 2) {
                size_t need = sdrhip_fm_stream_state_bytes(st), used = 0;
                if (need == 0) check(-1, "sdrhip_fm_stream_state_bytes");
                void *buf = malloc(need);
                check(sdrhip_fm_stream_save(st, buf, need, &used), "sdrhip_fm_stream_save");
                check(sdrhip_fm_stream_restore(st2, buf, used), "sdrhip_fm_stream_restore");
                free(buf);
                sdrhip_fm_stream *t = st; st = st2; st2 = t;
            }
        }
        check(sdrhip_fm_stream_flush(st), "sdrhip_fm_stream_flush");
        while (sdrhip_fm_stream_pop(st, audio, BLOCK) > 0) outs += BLOCK;
        sdrhip_fm_stream_destroy(st);
        sdrhip_fm_stream_destroy(st2);          /* results still pending in this one */
        free(mine);
    }
    free(audio);
    sdrhip_fm_chain_destroy(chain);
    sdrhip_decimator_destroy(dec);
    sdrhip_resampler_destroy(res);
    sdrhip_filter_destroy(fil);
    if (outs <= 0) { fprintf(stderr, "pipes_soak: no output at all\n"); return 1; }
    printf("pipes_soak: ok (%lld output elements)\n", outs);
#ifdef SDRHIP_FAST_EXIT
    /* sanitizer builds: ROCm's AddressSanitizer runtime trips over its own device allocator when the HIP runtime's static
     * destructors free memory after the HSA runtime has unloaded (a CHECK in sanitizer_allocator_device.h at process exit, in
     * code that is not ours): leave without running them */
    fflush(NULL);
    _exit(0);
#endif
    return 0;
}
