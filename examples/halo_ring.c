/* halo_ring.c -- the ntaps-1 halo exchange of the sharded FM chain, one process per GPU, in plain C over the C ABI of
 * libsdr_hip.so; what a sharded Haskell host would do through the FFI, minus the radio.
 *
 *     halo_ring <nranks> <rank> <id_file> [device] [shard_samples] [steps]
 *
 * Rank 0 creates the RCCL rendezvous id and writes it to <id_file>; the other ranks wait for the file.  Every rank
 * fills its shard of u8 IQ with a pattern that depends on (rank, sample index), runs `steps` exchanges on its compute
 * stream and checks after each that the halo region holds exactly the head of its RIGHT neighbour's shard (rank 0's
 * head for the last rank: the head of the next super-block).  With nranks = 1 the ring closes on itself.
 * Exit code 0 and "halo_ring rank r/N: OK (<transport>)" on success. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "sdr_hip.h"

static void check(int rc, const char *what)
{
    if (rc < 0) {
        fprintf(stderr, "halo_ring: %s failed (%d): %s\n", what, rc, sdrhip_last_error());
        exit(1);
    }
}

static uint8_t pattern(int rank, int64_t byte_index, int step) { return (uint8_t)(rank * 53 + byte_index * 7 + (byte_index >> 8) + step * 11); }

/* a small symmetric low-pass so the example needs no taps files: the halo size only depends on the tap COUNTS */
static void taps(float *t, int n) { for (int i = 0; i < n; i++) t[i] = 1.0f / (float)(1 + (i < n - 1 - i ? i : n - 1 - i)); }

int main(int argc, char **argv)
{
    if (argc < 4) {
        fprintf(stderr, "usage: halo_ring <nranks> <rank> <id_file> [device] [shard_samples] [steps]\n");
        return 2;
    }
    const int nranks = atoi(argv[1]), rank = atoi(argv[2]);
    const char *id_file = argv[3];
    const int device = argc > 4 ? atoi(argv[4]) : rank;
    const int64_t shard = argc > 5 ? atoll(argv[5]) : (1 << 20);
    const int steps = argc > 6 ? atoi(argv[6]) : 3;
    check(sdrhip_set_device(device), "sdrhip_set_device");

    unsigned char id[SDRHIP_COMM_ID_BYTES];
    if (rank == 0) {
        check(sdrhip_comm_get_unique_id(id), "sdrhip_comm_get_unique_id");
        char tmp[4096];
        snprintf(tmp, sizeof tmp, "%s.tmp", id_file);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "halo_ring: cannot write %s\n", tmp); return 1; }
        fclose(f);
        if (rename(tmp, id_file) != 0) { perror("rename"); return 1; }
    } else {
        FILE *f = NULL;
        for (int i = 0; i < 6000 && !(f = fopen(id_file, "rb")); i++) {
            struct timespec ts = {0, 10 * 1000 * 1000};
            nanosleep(&ts, NULL);
        }
        if (!f || fread(id, 1, sizeof id, f) != sizeof id) { fprintf(stderr, "halo_ring: rank %d never saw %s\n", rank, id_file); return 1; }
        fclose(f);
    }

    /* the FM chain of BASELINE.json configs[4]: decimate by 8 (127 taps), resample 3/10 (191 taps), 128-tap symmetric filter */
    float td[127], tr[191], ta[64];
    taps(td, 127); taps(tr, 191); taps(ta, 64);
    sdrhip_fm_chain *chain = NULL;
    check(sdrhip_fm_chain_create(&chain, SDRHIP_ORDER_AVX, 8, td, 127, 3, 10, tr, 191, ta, 64, 0.2f, 8192), "sdrhip_fm_chain_create");
    const int64_t halo = sdrhip_fm_chain_halo_samples(chain);
    if (halo <= 0 || halo > shard) { fprintf(stderr, "halo_ring: halo %lld vs shard %lld\n", (long long)halo, (long long)shard); return 1; }

    sdrhip_comm *comm = NULL;
    check(sdrhip_comm_init_rank(&comm, nranks, rank, id), "sdrhip_comm_init_rank");
    void *stream = NULL;
    check(sdrhip_stream_create(&stream), "sdrhip_stream_create");
    const size_t nbytes = (size_t)(2 * (shard + halo));
    uint8_t *h = (uint8_t *)malloc(nbytes), *back = (uint8_t *)malloc((size_t)(2 * halo));
    void *d = NULL;
    check(sdrhip_malloc(&d, nbytes), "sdrhip_malloc");
    const int right = (rank + 1) % nranks;
    for (int step = 0; step < steps; step++) {
        for (int64_t i = 0; i < 2 * shard; i++) h[i] = pattern(rank, i, step);
        memset(h + 2 * shard, 0xEE, (size_t)(2 * halo));
        check(sdrhip_memcpy_h2d(d, h, nbytes, stream), "sdrhip_memcpy_h2d");
        check(sdrhip_fm_chain_halo_exchange(chain, comm, stream, (uint8_t *)d, shard), "sdrhip_fm_chain_halo_exchange");
        check(sdrhip_memcpy_d2h(back, (uint8_t *)d + 2 * shard, (size_t)(2 * halo), stream), "sdrhip_memcpy_d2h");
        check(sdrhip_stream_sync(stream), "sdrhip_stream_sync");
        for (int64_t i = 0; i < 2 * halo; i++) {
            if (back[i] != pattern(right, i, step)) {
                fprintf(stderr, "halo_ring rank %d step %d: halo byte %lld is %u, expected %u\n", rank, step, (long long)i, back[i],
                        pattern(right, i, step));
                return 1;
            }
        }
    }
    printf("halo_ring rank %d/%d: OK (%s, %lld-sample halo, %d steps)\n", rank, nranks, sdrhip_comm_transport(comm), (long long)halo, steps);
    sdrhip_comm_destroy(comm);
    sdrhip_free(d);
    sdrhip_stream_destroy(stream);
    sdrhip_fm_chain_destroy(chain);
    free(h);
    free(back);
    return 0;
}
