/* fm_replay.c -- the FM receiver of the reference's examples/fm/fm.hs:30-41 fed from a file instead of a radio
 * ("file replay source feeding pinned buffers"), in plain C over the C ABI of libsdr_hip.so.
 *
 *     fm_replay <iq_u8_file> <audio_f32_file> [blocks_per_push]
 *
 * Reads interleaved unsigned 8-bit IQ (what rtl_sdr writes, what RTLSDRStream.hs:48-67 yields) in source blocks of
 * 8192 samples, `blocks_per_push` of them at a time, straight into the stream operator's pinned staging buffer
 * (fread is the "source that can write where it is told"), and writes the audio blocks (8192 floats each, 48 kHz for
 * a 1.28 MHz capture) as raw little-endian f32.  Taps arrive as three raw f32 files next to the capture
 * (<iq>.decim.f32, <iq>.resamp.f32, <iq>.audio_half.f32) so the example carries no filter design of its own.
 * Output is bit-identical to the reference pipeline's (tests/test_gpu_examples.py). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sdr_hip.h"

#define SOURCE_BLOCK 8192 /* fm.hs:17 `samples` */

static float *read_floats(const char *base, const char *suffix, int *n)
{
    char path[4096];
    snprintf(path, sizeof path, "%s%s", base, suffix);
    FILE *f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "fm_replay: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    float *v = (float *)malloc((size_t)bytes);
    if (fread(v, 1, (size_t)bytes, f) != (size_t)bytes) { fprintf(stderr, "fm_replay: short read on %s\n", path); exit(2); }
    fclose(f);
    *n = (int)(bytes / 4);
    return v;
}

static void check(int rc, const char *what)
{
    if (rc < 0) { fprintf(stderr, "fm_replay: %s: %s\n", what, sdrhip_last_error()); exit(1); }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: fm_replay <iq_u8_file> <audio_f32_file> [blocks_per_push]\n"); return 2; }
    const int bpp = argc > 3 ? atoi(argv[3]) : 64;
    if (bpp < 1) { fprintf(stderr, "fm_replay: blocks_per_push must be >= 1\n"); return 2; }
    int n_decim, n_resamp, n_audio;
    float *decim = read_floats(argv[1], ".decim.f32", &n_decim);
    float *resamp = read_floats(argv[1], ".resamp.f32", &n_resamp);
    float *audio_half = read_floats(argv[1], ".audio_half.f32", &n_audio);

    sdrhip_fm_chain *chain = NULL;
    check(sdrhip_fm_chain_create(&chain, SDRHIP_ORDER_AVX, 8, decim, n_decim, 3, 10, resamp, n_resamp, audio_half, n_audio,
                                 0.2f, SOURCE_BLOCK), "sdrhip_fm_chain_create");
    sdrhip_fm_stream *st = NULL;
    check(sdrhip_fm_stream_create(&st, chain, bpp * SOURCE_BLOCK, SOURCE_BLOCK), "sdrhip_fm_stream_create");

    FILE *in = fopen(argv[1], "rb"), *out = fopen(argv[2], "wb");
    if (!in || !out) { fprintf(stderr, "fm_replay: cannot open input/output\n"); return 2; }
    float *block = (float *)malloc(SOURCE_BLOCK * sizeof(float));
    long long samples = 0, audio = 0;
    for (;;) {
        uint8_t *buf = sdrhip_fm_stream_input_buffer(st);   /* pinned: the upload needs no intermediate copy */
        if (!buf) check(-1, "sdrhip_fm_stream_input_buffer");
        size_t got = fread(buf, 2 * SOURCE_BLOCK, (size_t)bpp, in);   /* whole source blocks only, like the Pipe source */
        int ready = 0;
        if (got > 0) {
            ready = sdrhip_fm_stream_push(st, buf, (int)got * SOURCE_BLOCK);
            check(ready, "sdrhip_fm_stream_push");
            samples += (long long)got * SOURCE_BLOCK;
        }
        if (got < (size_t)bpp) {
            ready = sdrhip_fm_stream_flush(st);
            check(ready, "sdrhip_fm_stream_flush");
        }
        for (int i = 0; i < ready; i++) {
            check(sdrhip_fm_stream_pop(st, block, SOURCE_BLOCK), "sdrhip_fm_stream_pop");
            fwrite(block, sizeof(float), SOURCE_BLOCK, out);
            audio += SOURCE_BLOCK;
        }
        if (got < (size_t)bpp) break;
    }
    fprintf(stderr, "fm_replay: %lld IQ samples in, %lld audio samples out\n", samples, audio);
    fclose(in);
    fclose(out);
    free(block);
    sdrhip_fm_stream_destroy(st);
    sdrhip_fm_chain_destroy(chain);
    free(decim);
    free(resamp);
    free(audio_half);
    return 0;
}
