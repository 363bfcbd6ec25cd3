/* fm_replay.c -- the FM receiver of the reference's examples/fm/fm.hs:30-41 fed from a file instead of a radio
 * ("file replay source feeding pinned buffers"), in plain C over the C ABI of libsdr_hip.so.
 *
 *     fm_replay <iq_u8_file | udp:PORT> <audio_f32_file> [blocks_per_push] [taps_prefix]
 *
 * Reads interleaved unsigned 8-bit IQ (what rtl_sdr writes, what RTLSDRStream.hs:48-67 yields) in source blocks of
 * 8192 samples, `blocks_per_push` of them at a time, straight into the stream operator's pinned staging buffer
 * (fread is the "source that can write where it is told"), and writes the audio blocks (8192 floats each, 48 kHz for
 * a 1.28 MHz capture) as raw little-endian f32.  With `udp:PORT` the samples come from UDP datagrams on 127.0.0.1:PORT
 * instead (the reference's udpSource, NetworkStream.hs:28-35), reassembled into source blocks; a zero-length datagram
 * ends the stream.  Taps arrive as three raw f32 files (<taps_prefix>.decim.f32, .resamp.f32, .audio_half.f32; the
 * prefix defaults to the capture's path) so the example carries no filter design of its own.
 * Output is bit-identical to the reference pipeline's (tests/test_gpu_examples.py). */
#define _POSIX_C_SOURCE 200809L
#include <arpa/inet.h>
#include <netinet/in.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <unistd.h>

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

/* The sample source: a file, or a UDP socket whose datagrams are concatenated.  fill() blocks until `bytes` bytes are in
 * `dst` or the source ends; returns the bytes delivered. */
typedef struct { FILE *f; int sock; } source;

static size_t fill(source *src, uint8_t *dst, size_t bytes)
{
    if (src->f) return fread(dst, 1, bytes, src->f);
    size_t have = 0;
    while (have < bytes) {
        ssize_t n = recv(src->sock, dst + have, bytes - have, 0);   /* datagrams never straddle a request: see main() */
        if (n <= 0) break;                                           /* empty datagram = end of stream; error/timeout too */
        have += (size_t)n;
    }
    return have;
}

static void check(int rc, const char *what)
{
    if (rc < 0) { fprintf(stderr, "fm_replay: %s: %s\n", what, sdrhip_last_error()); exit(1); }
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: fm_replay <iq_u8_file | udp:PORT> <audio_f32_file> [blocks_per_push] [taps_prefix]\n"); return 2; }
    const int bpp = argc > 3 ? atoi(argv[3]) : 64;
    if (bpp < 1) { fprintf(stderr, "fm_replay: blocks_per_push must be >= 1\n"); return 2; }
    int n_decim, n_resamp, n_audio;
    const int is_udp = strncmp(argv[1], "udp:", 4) == 0;
    const char *prefix = argc > 4 ? argv[4] : argv[1];
    if (is_udp && argc <= 4) { fprintf(stderr, "fm_replay: udp source needs a taps_prefix\n"); return 2; }
    float *decim = read_floats(prefix, ".decim.f32", &n_decim);
    float *resamp = read_floats(prefix, ".resamp.f32", &n_resamp);
    float *audio_half = read_floats(prefix, ".audio_half.f32", &n_audio);

    sdrhip_fm_chain *chain = NULL;
    check(sdrhip_fm_chain_create(&chain, SDRHIP_ORDER_AVX, 8, decim, n_decim, 3, 10, resamp, n_resamp, audio_half, n_audio,
                                 0.2f, SOURCE_BLOCK), "sdrhip_fm_chain_create");
    sdrhip_fm_stream *st = NULL;
    check(sdrhip_fm_stream_create(&st, chain, bpp * SOURCE_BLOCK, SOURCE_BLOCK), "sdrhip_fm_stream_create");

    source src = {NULL, -1};
    if (is_udp) {
        /* datagram sizes must divide the source block (the sender of tests/test_gpu_examples.py uses 4096 bytes), so a
         * recv() never has to split one */
        struct sockaddr_in addr;
        memset(&addr, 0, sizeof addr);
        addr.sin_family = AF_INET;
        addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
        addr.sin_port = htons((unsigned short)atoi(argv[1] + 4));
        src.sock = socket(AF_INET, SOCK_DGRAM, 0);
        int rcvbuf = 64 << 20;
        setsockopt(src.sock, SOL_SOCKET, SO_RCVBUF, &rcvbuf, sizeof rcvbuf);
        struct timeval tv = {20, 0};                                  /* a silent sender ends the stream */
        setsockopt(src.sock, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof tv);
        if (src.sock < 0 || bind(src.sock, (struct sockaddr *)&addr, sizeof addr) != 0) { perror("fm_replay: bind"); return 2; }
        {   /* warm the device up (module load, first allocations) on a throw-away stream, so that the first real push does
             * not stall long enough for the socket buffer to overflow */
            sdrhip_fm_stream *warm = NULL;
            check(sdrhip_fm_stream_create(&warm, chain, bpp * SOURCE_BLOCK, SOURCE_BLOCK), "sdrhip_fm_stream_create");
            uint8_t *wb = sdrhip_fm_stream_input_buffer(warm);
            if (!wb) check(-1, "sdrhip_fm_stream_input_buffer");
            memset(wb, 128, (size_t)bpp * 2 * SOURCE_BLOCK);
            for (int i = 0; i < 3; i++) check(sdrhip_fm_stream_push(warm, wb, bpp * SOURCE_BLOCK), "warm-up push");
            check(sdrhip_fm_stream_flush(warm), "warm-up flush");
            sdrhip_fm_stream_destroy(warm);
        }
        fprintf(stderr, "fm_replay: listening on 127.0.0.1:%d\n", atoi(argv[1] + 4));
        fflush(stderr);
    } else {
        src.f = fopen(argv[1], "rb");
    }
    FILE *out = fopen(argv[2], "wb");
    if ((!src.f && src.sock < 0) || !out) { fprintf(stderr, "fm_replay: cannot open input/output\n"); return 2; }
    float *block = (float *)malloc(SOURCE_BLOCK * sizeof(float));
    long long samples = 0, audio = 0;
    for (;;) {
        uint8_t *buf = sdrhip_fm_stream_input_buffer(st);   /* pinned: the upload needs no intermediate copy */
        if (!buf) check(-1, "sdrhip_fm_stream_input_buffer");
        size_t got = fill(&src, buf, (size_t)bpp * 2 * SOURCE_BLOCK) / (2 * SOURCE_BLOCK);   /* whole source blocks only */
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
    if (src.f) fclose(src.f);
    if (src.sock >= 0) close(src.sock);
    fclose(out);
    free(block);
    sdrhip_fm_stream_destroy(st);
    sdrhip_fm_chain_destroy(chain);
    free(decim);
    free(resamp);
    free(audio_half);
#ifdef SDRHIP_FAST_EXIT
    /* sanitizer builds only (examples/pipes_soak.c explains): skip the HIP runtime's static destructors */
    fflush(NULL);
    _exit(0);
#endif
    return 0;
}
