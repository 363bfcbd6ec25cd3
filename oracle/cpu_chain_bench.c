/* oracle/cpu_chain_bench.c -- the CPU baseline of bench.py as a COMPILED caller.  TEST / MEASUREMENT INFRASTRUCTURE ONLY:
 * nothing in the product links or runs this.
 *
 * The FM receiver of examples/fm/fm.hs:34-41 block by block, exactly as the reference runs it on one pipeline thread:
 *     u8 IQ blocks of 8192 samples -> interleavedIQUnsignedByteToFloatFast -> firDecimator (/8, 128 taps) -> fmDemod
 *     -> firResampler 3/10 (191 taps) -> firFilter (sym, 64 half-taps) -> * 0.2,   every Pipe with blockSizeOut = 8192.
 * The within-buffer kernels are the REFERENCE'S OWN C (oracle/_ref/libsdr_ref.so = c_sources compiled unmodified:
 * convertCAVX, decimateAVXRC, resampleAVXRR, filterAVXSymmetricRR, scaleAVX) when that library is present ("reference"),
 * else the restatement's twins ("port").  What is Haskell in the reference -- the Pipes' state machines
 * (Filter.hs:532-727), the cross-buffer kernels (FilterInternal.hs:397-423) and fmDemod (Demod.hs:21-46) -- comes from the
 * restatement (libsdr_oracle.so) and from the state machines below, which follow oracle/pipes_model.py statement by statement
 * (tests/test_cpu_chain_bench.py compares the audio this program writes with that model's).
 *
 *   cpu_chain_bench <taps.bin> <seconds> <threads>          -> one JSON line: samples per second of input, all threads
 *   cpu_chain_bench <taps.bin> --dump <in.u8> <out.f32>     -> run the file's blocks once, write the audio blocks
 *   cpu_chain_bench <taps.bin> --stages <seconds>           -> each kernel alone on one block + their harmonic sum per input sample
 * taps.bin: int32 n_decim, n_resamp, n_half, then the three float32 arrays (written by bench.py / the test).
 *
 * Build: oracle/Makefile (gcc -O2, -ldl -lpthread -lm). */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#define BLOCK 8192

/* ---- kernels (resolved at start-up) ---- */
typedef void (*convert_fn)(int, uint8_t *, float *);
typedef void (*decim_fn)(int, int, int, float *, float *, float *);
typedef int (*resamp_fn)(int, int, int, int, int *, float **, float *, float *);
typedef void (*filt_fn)(int, int, float *, float *, float *);
typedef void (*scale_fn)(int, float, float *, float *);
typedef void (*o_decim_rc_fn)(int, int, int, int, const float *, const float *, float *);
typedef int (*o_resamp_rr_fn)(int, int, int, int, int, const int *, float *const *, const float *, float *);
typedef void (*o_filt_sym_fn)(int, int, int, const float *, const float *, float *);
typedef void (*o_convert_fn)(int, const uint8_t *, float *);
typedef void (*o_scale_fn)(int, float, const float *, float *);
typedef void (*cross_dec_fn)(int, int, const float *, int, const float *, int, const float *, float *);
typedef int (*cross_res_fn)(int, int, int, const float *, int, int, const float *, int, const float *, float *);
typedef void (*demod_fn)(int, float, float, const float *, float *);
typedef int (*prep_fn)(int, int, int, const float *, int, int *, int *, int *, int *, int *, float *);

static convert_fn r_convert;
static decim_fn r_decim, r_decim_scalar_c, r_decim_scalar_r;
static resamp_fn r_resamp;
static filt_fn r_filt;
static scale_fn r_scale;
static o_decim_rc_fn o_decim;
static o_resamp_rr_fn o_resamp;
static o_filt_sym_fn o_filt;
static o_convert_fn o_convert;
static o_scale_fn o_scale;
static cross_dec_fn x_dec_c, x_dec_r;
static cross_res_fn x_res_r;
static demod_fn x_demod;
static prep_fn x_prep;
static int have_ref;

static void *must_sym(void *h, const char *name)
{
    void *p = dlsym(h, name);
    if (!p) { fprintf(stderr, "cpu_chain_bench: symbol %s missing\n", name); exit(2); }
    return p;
}

static void load_libs(const char *argv0)
{
    char dir[4096], path[4400];
    strncpy(dir, argv0, sizeof dir - 1);
    dir[sizeof dir - 1] = 0;
    char *slash = strrchr(dir, '/');
    if (slash) *slash = 0; else strcpy(dir, ".");
    snprintf(path, sizeof path, "%s/libsdr_oracle.so", dir);
    void *ho = dlopen(path, RTLD_NOW);
    if (!ho) { fprintf(stderr, "cpu_chain_bench: %s\n", dlerror()); exit(2); }
    o_decim = (o_decim_rc_fn)must_sym(ho, "orc_decimate_rc");
    o_resamp = (o_resamp_rr_fn)must_sym(ho, "orc_resample_rr");
    o_filt = (o_filt_sym_fn)must_sym(ho, "orc_filter_sym_rr");
    o_convert = (o_convert_fn)must_sym(ho, "orc_convert_u8");
    o_scale = (o_scale_fn)must_sym(ho, "orc_scale");
    x_dec_c = (cross_dec_fn)must_sym(ho, "orc_decimate_cross_c");
    x_dec_r = (cross_dec_fn)must_sym(ho, "orc_decimate_cross_r");
    x_res_r = (cross_res_fn)must_sym(ho, "orc_resample_cross_r");
    x_demod = (demod_fn)must_sym(ho, "orc_fm_demod");
    x_prep = (prep_fn)must_sym(ho, "orc_prepare_coeffs");
    snprintf(path, sizeof path, "%s/_ref/libsdr_ref.so", dir);
    void *hr = dlopen(path, RTLD_NOW);
    have_ref = hr != NULL;
    if (hr) {
        r_convert = (convert_fn)must_sym(hr, "convertCAVX");
        r_decim = (decim_fn)must_sym(hr, "decimateAVXRC");
        r_decim_scalar_c = (decim_fn)must_sym(hr, "decimateRC");      /* scalar C: the sequential order of the Haskell cross kernels */
        r_decim_scalar_r = (decim_fn)must_sym(hr, "decimateRR");
        r_resamp = (resamp_fn)must_sym(hr, "resampleAVXRR");
        r_filt = (filt_fn)must_sym(hr, "filterAVXSymmetricRR");
        r_scale = (scale_fn)must_sym(hr, "scaleAVX");
    }
}

/* ---- taps ---- */
static int n_decim, n_resamp, n_half;
static float *t_decim, *t_resamp, *t_half;

static void read_taps(const char *path)
{
    FILE *f = fopen(path, "rb");
    int32_t hdr[3];
    if (!f || fread(hdr, 4, 3, f) != 3) { fprintf(stderr, "cpu_chain_bench: cannot read %s\n", path); exit(2); }
    n_decim = hdr[0]; n_resamp = hdr[1]; n_half = hdr[2];
    t_decim = malloc(4 * (size_t)n_decim); t_resamp = malloc(4 * (size_t)n_resamp); t_half = malloc(4 * (size_t)n_half);
    if (fread(t_decim, 4, n_decim, f) != (size_t)n_decim || fread(t_resamp, 4, n_resamp, f) != (size_t)n_resamp ||
        fread(t_half, 4, n_half, f) != (size_t)n_half) { fprintf(stderr, "cpu_chain_bench: short taps file\n"); exit(2); }
    fclose(f);
}

/* ---- one receiver (one pipeline thread's state) ---- */
struct OutBuf { int w, block, offset; float *buf; };      /* Buffer + advanceOutBuf, Filter.hs:504-523 */

struct Fir {                    /* firDecimator / firFilter, Filter.hs:532-611 */
    int w, D, L, cplx, sym;
    float *one_taps; int n_one;         /* as passed to the C kernel (duplicated for RC, half for sym) */
    float *cross_taps;                  /* L plain taps */
    float *last; int nlast;             /* remainder of the previous buffer (< L elements) */
    float *stitch;                      /* drop i last ++ next, as far as the crossover's windows reach (< 2 L elements) */
    struct OutBuf out;
};

struct Res {                    /* firResampler, Filter.hs:679-727 */
    int I, D, L, ntaps, num_coeffs, num_groups;
    int *increments; float **rows; float *groups;
    float *last; int nlast;
    int group, filter_offset;
    struct OutBuf out;
};

struct Rx {
    struct Fir dec, flt;
    struct Res res;
    float *iq, *y, *audio;
    float last_re, last_im;
    double checksum;
    long blocks_out;
    FILE *dump;
};

static int quot_up(int q, int d) { return (q + d - 1) / d; }

static void rx_audio(struct Rx *rx, float *blk, int n);
static void rx_filter_in(struct Rx *rx, float *blk, int n);
static void rx_resamp_in(struct Rx *rx, float *blk, int n);
static void rx_demod_in(struct Rx *rx, float *blk, int n);

/* `n` elements were just written at out->offset; a full buffer goes to `next` (only exactly-full blocks are ever yielded) */
static void out_advance(struct Rx *rx, struct OutBuf *o, int n, void (*next)(struct Rx *, float *, int))
{
    if (n == o->block - o->offset) {
        next(rx, o->buf, o->block);
        o->offset = 0;
    } else {
        o->offset += n;
    }
}

static void fir_one(struct Fir *f, int count, float *in, float *out)
{
    if (f->cplx) {
        if (have_ref) r_decim(count, f->D, f->n_one, f->one_taps, in, out);
        else o_decim(4, count, f->D, f->n_one, f->one_taps, in, out);
    } else {
        if (have_ref) r_filt(count, f->n_one, f->one_taps, in, out);
        else o_filt(8, count, f->n_one, f->one_taps, in, out);
    }
}

/* one input buffer through firDecimator / firFilter (the push form of Filter.hs:536-611) */
static void fir_push(struct Rx *rx, struct Fir *f, float *blk, int n, void (*next)(struct Rx *, float *, int))
{
    const int w = f->w, D = f->D, L = f->L;
    float *buf_in = blk;
    int len = n;
    if (f->nlast > 0) {
        /* crossover: outputs whose window straddles the two buffers, sequential order */
        float *last = f->last;
        int nlast = f->nlast;
        for (;;) {
            int space = f->out.block - f->out.offset;
            int count = quot_up(nlast, D);
            if (count > space) count = space;
            float *dst = f->out.buf + (size_t)f->out.offset * w;
            if (have_ref) {
                /* decimateCrossHighLevel / filterCrossHighLevel (FilterInternal.hs:397-408) = a sequential sum over
                 * `drop i last ++ next`: the reference's scalar C kernel on the concatenated window computes exactly that
                 * (tests/test_oracle_vs_ref.py::test_sequential_order_is_scalar_c), at compiled speed */
                const int need = (count - 1) * D + L;                       /* elements the windows touch */
                memcpy(f->stitch, last, (size_t)nlast * w * 4);
                memcpy(f->stitch + (size_t)nlast * w, blk, (size_t)(need - nlast) * w * 4);
                if (f->cplx) r_decim_scalar_c(count, D, L, f->cross_taps, f->stitch, dst);
                else r_decim_scalar_r(count, D, L, f->cross_taps, f->stitch, dst);
            } else if (f->cplx) x_dec_c(D, L, f->cross_taps, count, last, nlast, blk, dst);
            else x_dec_r(D, L, f->cross_taps, count, last, nlast, blk, dst);
            out_advance(rx, &f->out, count, next);
            if (nlast <= count * D) {
                buf_in = blk + (size_t)(count * D - nlast) * w;
                len = n - (count * D - nlast);
                break;
            }
            last += (size_t)count * D * w;
            nlast -= count * D;
        }
        f->nlast = 0;
    }
    while (len >= L) {
        int space = f->out.block - f->out.offset;
        int count = (len - L) / D + 1;
        if (count > space) count = space;
        fir_one(f, count, buf_in, f->out.buf + (size_t)f->out.offset * w);
        out_advance(rx, &f->out, count, next);
        buf_in += (size_t)count * D * w;
        len -= count * D;
    }
    memcpy(f->last, buf_in, (size_t)len * w * 4);
    f->nlast = len;
}

static void res_push(struct Rx *rx, struct Res *r, float *blk, int n, void (*next)(struct Rx *, float *, int))
{
    const int I = r->I, D = r->D, L = r->L;
    float *buf_in = blk;
    int len = n;
    if (r->nlast > 0) {
        float *last = r->last;
        int nlast = r->nlast;
        for (;;) {
            int space = r->out.block - r->out.offset;
            int count = quot_up(nlast * I + r->filter_offset, D);          /* outputsComputable, Filter.hs:712-716 */
            if (count > space) count = space;
            float *dst = r->out.buf + r->out.offset;
            int end_off = x_res_r(I, D, r->ntaps, t_resamp, r->filter_offset, count, last, nlast, blk, dst);
            r->group = (r->group + count) % I;                              /* Filter.hs:419-421 */
            out_advance(rx, &r->out, count, next);
            int used = quot_up(count * D - r->filter_offset, I);
            r->filter_offset = end_off;
            if (used >= nlast) {
                buf_in = blk + (used - nlast);
                len = n - (used - nlast);
                break;
            }
            last += used;
            nlast -= used;
        }
        r->nlast = 0;
    }
    while (len * I >= L - r->filter_offset) {
        int space = r->out.block - r->out.offset;
        int count = (len * I - L + r->filter_offset) / D + 1;
        if (count > space) count = space;
        float *dst = r->out.buf + r->out.offset;
        int g;
        if (have_ref) g = r_resamp(count, r->num_coeffs, r->group, r->num_groups, r->increments, r->rows, buf_in, dst);
        else g = o_resamp(8, count, r->num_coeffs, r->group, r->num_groups, r->increments, r->rows, buf_in, dst);
        int end_off = I - 1 - ((I + g * D - 1) % I);                        /* func1, Filter.hs:423 */
        r->group = g;
        out_advance(rx, &r->out, count, next);
        int used = quot_up(count * D - r->filter_offset, I);
        buf_in += used;
        len -= used;
        r->filter_offset = end_off;
    }
    memcpy(r->last, buf_in, (size_t)len * 4);      /* len == 0: `simple next` (Filter.hs:707-709) -- nothing carried */
    r->nlast = len;
}

static void rx_audio(struct Rx *rx, float *blk, int n)
{
    if (have_ref) r_scale(n, 0.2f, blk, rx->audio);            /* fm.hs:40 `P.map (VG.map (* 0.2))` */
    else o_scale(n, 0.2f, blk, rx->audio);
    rx->checksum += rx->audio[0] + rx->audio[n - 1];
    rx->blocks_out++;
    if (rx->dump) fwrite(rx->audio, 4, n, rx->dump);
}
static void rx_filter_in(struct Rx *rx, float *blk, int n) { fir_push(rx, &rx->flt, blk, n, rx_audio); }
static void rx_resamp_in(struct Rx *rx, float *blk, int n) { res_push(rx, &rx->res, blk, n, rx_filter_in); }
static void rx_demod_in(struct Rx *rx, float *blk, int n)
{
    x_demod(n, rx->last_re, rx->last_im, blk, rx->y);           /* fmDemod, Demod.hs:40-46 */
    rx->last_re = blk[2 * (n - 1)];
    rx->last_im = blk[2 * (n - 1) + 1];
    rx_resamp_in(rx, rx->y, n);
}

static void rx_push(struct Rx *rx, uint8_t *u8, int nsamples)
{
    if (have_ref) r_convert(2 * nsamples, u8, rx->iq);          /* interleavedIQUnsignedByteToFloatFast, Util.hs:137-138 */
    else o_convert(2 * nsamples, u8, rx->iq);
    fir_push(rx, &rx->dec, rx->iq, nsamples, rx_demod_in);
}

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

static void outbuf_init(struct OutBuf *o, int w, int block)
{
    o->w = w; o->block = block; o->offset = 0;
    o->buf = aligned_alloc(64, (size_t)block * w * 4 + 256);
}

static void rx_init(struct Rx *rx)
{
    memset(rx, 0, sizeof *rx);
    /* mkDecimatorC (Filter.hs:322-331): pad to a multiple of 4, duplicate for the RC kernel */
    struct Fir *d = &rx->dec;
    d->w = 2; d->D = 8; d->cplx = 1; d->L = round_up(n_decim, 4);
    d->cross_taps = calloc((size_t)d->L, 4);
    memcpy(d->cross_taps, t_decim, (size_t)n_decim * 4);
    d->n_one = 2 * d->L;
    d->one_taps = aligned_alloc(64, (size_t)d->n_one * 4 + 64);
    for (int i = 0; i < d->L; i++) d->one_taps[2 * i] = d->one_taps[2 * i + 1] = d->cross_taps[i];
    d->last = malloc((size_t)d->L * 2 * 4 + 64);
    d->stitch = malloc((size_t)d->L * 2 * 2 * 4 + 256);
    outbuf_init(&d->out, 2, BLOCK);
    /* mkFilterSymR (Filter.hs:234-245): half taps to the kernel, coeffs ++ reverse coeffs to the cross kernel */
    struct Fir *f = &rx->flt;
    f->w = 1; f->D = 1; f->sym = 1; f->L = 2 * n_half;
    f->n_one = n_half;
    f->one_taps = aligned_alloc(64, (size_t)n_half * 4 + 64);
    memcpy(f->one_taps, t_half, (size_t)n_half * 4);
    f->cross_taps = malloc((size_t)f->L * 4);
    for (int i = 0; i < n_half; i++) { f->cross_taps[i] = t_half[i]; f->cross_taps[f->L - 1 - i] = t_half[i]; }
    f->last = malloc((size_t)f->L * 4 + 64);
    f->stitch = malloc((size_t)f->L * 2 * 4 + 256);
    outbuf_init(&f->out, 1, BLOCK);
    /* mkResampler (Filter.hs:408-425) */
    struct Res *r = &rx->res;
    r->I = 3; r->D = 10; r->ntaps = n_resamp; r->L = round_up(n_resamp, 3 * 8);
    int cap = round_up((n_resamp + 2) / 3 + 1, 8) + 8, pl = 0;
    r->groups = calloc((size_t)3 * cap, 4);
    r->increments = calloc(3, sizeof(int));
    int offs[3];
    x_prep(8, 3, 10, t_resamp, n_resamp, &r->num_coeffs, &r->num_groups, &pl, r->increments, offs, r->groups);
    r->rows = malloc(sizeof(float *) * (size_t)r->num_groups);
    for (int g = 0; g < r->num_groups; g++) r->rows[g] = r->groups + (size_t)g * pl;
    r->last = malloc((size_t)(r->L / 3 + 8) * 4);
    outbuf_init(&r->out, 1, BLOCK);
    rx->iq = aligned_alloc(64, (size_t)BLOCK * 8 + 256);
    rx->y = aligned_alloc(64, (size_t)BLOCK * 4 + 256);
    rx->audio = aligned_alloc(64, (size_t)BLOCK * 4 + 256);
}

/* ---- timing ---- */
static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

struct Worker { pthread_t th; double seconds; double sps; double checksum; int id; };
static pthread_barrier_t start_bar;

static void *worker(void *arg)
{
    struct Worker *wk = arg;
    struct Rx rx;
    rx_init(&rx);
    const int nblk = 64;
    uint8_t *u8 = malloc((size_t)nblk * BLOCK * 2);
    uint64_t s = 0x9E3779B97F4A7C15ull * (uint64_t)(wk->id + 1);
    for (size_t i = 0; i < (size_t)nblk * BLOCK * 2; i++) {          /* splitmix64: uniform random bytes */
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        u8[i] = (uint8_t)((z ^ (z >> 31)) >> 24);
    }
    for (int b = 0; b < nblk; b++) rx_push(&rx, u8 + (size_t)b * BLOCK * 2, BLOCK);      /* warm-up pass */
    pthread_barrier_wait(&start_bar);
    const double t0 = now_s();
    long samples = 0;
    double t1;
    do {
        for (int b = 0; b < nblk; b++) rx_push(&rx, u8 + (size_t)b * BLOCK * 2, BLOCK);
        samples += (long)nblk * BLOCK;
        t1 = now_s();
    } while (t1 - t0 < wk->seconds);
    wk->sps = samples / (t1 - t0);
    wk->checksum = rx.checksum;
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: %s <taps.bin> <seconds> <threads> | %s <taps.bin> --dump <in.u8> <out.f32>\n", argv[0], argv[0]);
        return 2;
    }
    load_libs(argv[0]);
    read_taps(argv[1]);
    if (strcmp(argv[2], "--dump") == 0) {
        if (argc < 5) return 2;
        struct Rx rx;
        rx_init(&rx);
        FILE *fi = fopen(argv[3], "rb");
        rx.dump = fopen(argv[4], "wb");
        if (!fi || !rx.dump) { fprintf(stderr, "cpu_chain_bench: cannot open the dump files\n"); return 2; }
        uint8_t *u8 = malloc((size_t)BLOCK * 2);
        while (fread(u8, 2, BLOCK, fi) == BLOCK) rx_push(&rx, u8, BLOCK);
        fclose(fi);
        fclose(rx.dump);
        printf("{\"kind\": \"%s\", \"audio_blocks\": %ld}\n", have_ref ? "reference" : "port", rx.blocks_out);
        return 0;
    }
    if (strcmp(argv[2], "--stages") == 0) {
        /* each kernel alone on one block, compiled caller: input elements per second, and what they add up to per input sample */
        const double secs = atof(argv[3]);
        struct Rx rx;
        rx_init(&rx);
        uint8_t *u8 = malloc((size_t)BLOCK * 2);
        for (int i = 0; i < 2 * BLOCK; i++) u8[i] = (uint8_t)(i * 37 + (i >> 5));
        float *a = aligned_alloc(64, (size_t)BLOCK * 8 + 256), *b = aligned_alloc(64, (size_t)BLOCK * 8 + 256);
        rx_push(&rx, u8, BLOCK);
        memcpy(a, rx.iq, (size_t)BLOCK * 8);
        double rate[5];
        const char *name[5] = {"convert", "decimate", "fm_demod", "resample", "filter"};
        for (int st = 0; st < 5; st++) {
            long n = 0;
            const double t0 = now_s();
            double t1;
            do {
                for (int rep = 0; rep < 16; rep++) {
                    switch (st) {
                    case 0: if (have_ref) r_convert(2 * BLOCK, u8, b); else o_convert(2 * BLOCK, u8, b); break;
                    case 1: fir_one(&rx.dec, (BLOCK - rx.dec.L) / 8 + 1, a, b); break;
                    case 2: x_demod(BLOCK, 0.0f, 0.0f, a, b); break;
                    case 3: {
                        const int count = (BLOCK * 3 - rx.res.L) / 10 + 1;
                        if (have_ref) r_resamp(count, rx.res.num_coeffs, 0, rx.res.num_groups, rx.res.increments, rx.res.rows, a, b);
                        else o_resamp(8, count, rx.res.num_coeffs, 0, rx.res.num_groups, rx.res.increments, rx.res.rows, a, b);
                        break;
                    }
                    default: fir_one(&rx.flt, BLOCK - rx.flt.L + 1, a, b); break;
                    }
                }
                n += 16L * BLOCK;
                t1 = now_s();
            } while (t1 - t0 < secs);
            rate[st] = n / (t1 - t0);
        }
        /* per input IQ sample: convert 1, decimate 1, fmDemod 1/8, resample 1/8, filter 3/80 elements */
        const double per = 1.0 / rate[0] + 1.0 / rate[1] + 0.125 / rate[2] + 0.125 / rate[3] + 0.0375 / rate[4];
        printf("{\"kind\": \"%s\"", have_ref ? "reference" : "port");
        for (int st = 0; st < 5; st++) printf(", \"%s_elements_per_s\": %.1f", name[st], rate[st]);
        printf(", \"harmonic_sum_sps\": %.1f}\n", 1.0 / per);
        return 0;
    }
    const double seconds = atof(argv[2]);
    int threads = atoi(argv[3]);
    if (threads < 1) threads = 1;
    struct Worker *wk = calloc((size_t)threads, sizeof *wk);
    pthread_barrier_init(&start_bar, NULL, (unsigned)threads);
    for (int i = 0; i < threads; i++) {
        wk[i].seconds = seconds;
        wk[i].id = i;
        pthread_create(&wk[i].th, NULL, worker, &wk[i]);
    }
    double total = 0, mn = 1e30, chk = 0;
    for (int i = 0; i < threads; i++) {
        pthread_join(wk[i].th, NULL);
        total += wk[i].sps;
        if (wk[i].sps < mn) mn = wk[i].sps;
        chk += wk[i].checksum;
    }
    printf("{\"kind\": \"%s\", \"threads\": %d, \"seconds\": %.2f, \"sps_total\": %.1f, \"sps_slowest_thread\": %.1f, \"checksum\": %.6g}\n",
           have_ref ? "reference" : "port", threads, seconds, total, mn, chk);
    return 0;
}
