// abi_records.cpp -- the record seam of the reference on HOST vectors: the closures a Haskell constructor puts into the
// reference's own records (hs_sources/SDR/Filter.hs:116-144)
//     Filter    { filterOne,   filterCross   }      Decimator { decimateOne, decimateCross }
//     Resampler { resampleOne, resampleCross }
// so that the reference's UNCHANGED Pipes (firFilter / firDecimator / firResampler, Filter.hs:532-727) drive the device:
//   *One   = what the C SIMD kernel computes on one buffer            (FilterInternal.hs:66-71, 172-177, 335-342)
//   *Cross = what the pure-Haskell sequential kernel computes on `drop i last ++ next` (FilterInternal.hs:397-423)
// Host pointers in and out, synchronous, the descriptor's prepared taps stay resident on the device.  Buffers up to
// 512 KiB are read and written in place over PCIe (pinned staging, no copy engine): a Cross call moves a few hundred
// bytes.  Every call leases its scratch context (stream, pinned staging, device buffers) from the pool the drop-in symbols
// use (scratch_pool.hpp), so the closures of several pipelines do not queue behind one another.
#include <string.h>


#include "descriptors.hpp"
#include "scratch_pool.hpp"

using namespace sdrhip;

namespace {

using RecScratch = ScratchCtx;          // leased per call (scratch_pool.hpp): concurrent callers do not queue
constexpr size_t kRecDirectBytes = 512 << 10;

// stage `bytes` of host input (already assembled in sc.hin) and give back the device-visible input / output pointers
int stage(RecScratch& sc, size_t in_bytes, size_t out_bytes, const float** d_in, float** d_out, bool* direct)
{
    int rc;
    if ((rc = sc.hout.ensure(out_bytes + 64)) != SDRHIP_OK) return rc;
    *direct = in_bytes <= kRecDirectBytes && out_bytes <= kRecDirectBytes;
    if (*direct) {
        *d_in = (const float*)sc.hin.dev;
        *d_out = (float*)sc.hout.dev;
        return SDRHIP_OK;
    }
    if ((rc = sc.in.ensure(in_bytes + 64)) != SDRHIP_OK) return rc;
    if ((rc = sc.out.ensure(out_bytes + 64)) != SDRHIP_OK) return rc;
    SDRHIP_CHECK_HIP(hipMemcpyAsync(sc.in.p, sc.hin.p, in_bytes, hipMemcpyHostToDevice, sc.stream));
    *d_in = (const float*)sc.in.p;
    *d_out = (float*)sc.out.p;
    return SDRHIP_OK;
}

int finish(RecScratch& sc, bool direct, float* out, size_t out_bytes)
{
    if (!direct) SDRHIP_CHECK_HIP(hipMemcpyAsync(sc.hout.p, sc.out.p, out_bytes, hipMemcpyDeviceToHost, sc.stream));
    SDRHIP_CHECK_HIP(hipStreamSynchronize(sc.stream));
    memcpy(out, sc.hout.p, out_bytes);
    return SDRHIP_OK;
}

// hin <- a ++ b, zero-filled up to `total` elements of esz floats
int concat(RecScratch& sc, const float* a, int64_t na, const float* b, int64_t nb, int64_t total, int esz)
{
    int rc = sc.hin.ensure((size_t)total * esz * 4 + 64);
    if (rc != SDRHIP_OK) return rc;
    float* h = (float*)sc.hin.p;
    if (na > total) na = total;
    if (na > 0) memcpy(h, a, (size_t)na * esz * 4);
    if (nb > total - na) nb = total - na;
    if (nb > 0) memcpy(h + na * esz, b, (size_t)nb * esz * 4);
    if (na + nb < total) memset(h + (na + nb) * esz, 0, (size_t)(total - na - nb) * esz * 4);
    return SDRHIP_OK;
}

int fir_one(const FirDesc* d, int num, const float* in, float* out, const char* who)
{
    SDRHIP_REQUIRE(d != nullptr && num >= 0, who);
    if (num == 0) return SDRHIP_OK;
    SDRHIP_REQUIRE(in != nullptr && out != nullptr, who);
    const int esz = d->cplx ? 2 : 1;
    const int64_t need = (int64_t)(num - 1) * d->factor + d->Lp;      // the elements the C kernel reads
    ScratchLease lease;
    if (!lease.c) return SDRHIP_ERR_HIP;
    RecScratch& sc = *lease.c;
    int rc = concat(sc, in, need, nullptr, 0, need, esz);
    if (rc != SDRHIP_OK) return rc;
    const float* d_in;
    float* d_out;
    bool direct;
    const size_t ob = (size_t)num * esz * 4;
    if ((rc = stage(sc, (size_t)need * esz * 4, ob, &d_in, &d_out, &direct)) != SDRHIP_OK) return rc;
    if ((rc = fir_run(d, sc.stream, d_in, false, 0, d_out, 0, num, 0)) != SDRHIP_OK) return rc;
    return finish(sc, direct, out, ob);
}

int fir_cross(const FirDesc* d, int num, const float* last, int n_last, const float* next, int n_next, float* out, const char* who)
{
    SDRHIP_REQUIRE(d != nullptr && num >= 0 && n_last >= 0 && n_next >= 0, who);
    if (num == 0) return SDRHIP_OK;
    SDRHIP_REQUIRE(last != nullptr && next != nullptr && out != nullptr, who);
    const int esz = d->cplx ? 2 : 1;
    // output i = sum_j coeffs[j] * (drop (i*factor) last ++ next)[j]: every window must lie inside last ++ next
    const int64_t need = (int64_t)(num - 1) * d->factor + d->Lp;
    if ((int64_t)n_last + n_next < need) {
        set_error("%s: %d outputs need %lld elements of last ++ next, %d + %d given", who, num, (long long)need, n_last, n_next);
        return SDRHIP_ERR_ARG;
    }
    ScratchLease lease;
    if (!lease.c) return SDRHIP_ERR_HIP;
    RecScratch& sc = *lease.c;
    int rc = concat(sc, last, n_last, next, n_next, need, esz);
    if (rc != SDRHIP_OK) return rc;
    const float* d_in;
    float* d_out;
    bool direct;
    const size_t ob = (size_t)num * esz * 4;
    if ((rc = stage(sc, (size_t)need * esz * 4, ob, &d_in, &d_out, &direct)) != SDRHIP_OK) return rc;
    if ((rc = fir_run(d, sc.stream, d_in, false, 0, d_out, 0, num, -1)) != SDRHIP_OK) return rc;     // seam_block < 0: every output Cross
    return finish(sc, direct, out, ob);
}

}  // namespace

extern "C" {

int sdrhip_filter_one(const sdrhip_filter* f, int num, const float* in, float* out) { return fir_one(f, num, in, out, "sdrhip_filter_one"); }
int sdrhip_filter_cross(const sdrhip_filter* f, int num, const float* last, int n_last, const float* next, int n_next, float* out)
{
    return fir_cross(f, num, last, n_last, next, n_next, out, "sdrhip_filter_cross");
}
int sdrhip_decimator_one(const sdrhip_decimator* d, int num, const float* in, float* out) { return fir_one(d, num, in, out, "sdrhip_decimator_one"); }
int sdrhip_decimator_cross(const sdrhip_decimator* d, int num, const float* last, int n_last, const float* next, int n_next, float* out)
{
    return fir_cross(d, num, last, n_last, next, n_next, out, "sdrhip_decimator_cross");
}

// resampleOne's C call (mkResampler, FilterInternal.hs:335-342): `num` outputs starting in polyphase group `group` at in[0];
// returns the group the next output would use (what resampleAVXRR returns), or a negative error.
int sdrhip_resampler_one(const sdrhip_resampler* r, int group, int num, const float* in, int n_in, float* out)
{
    SDRHIP_REQUIRE(r != nullptr && num >= 0 && group >= 0 && group < r->num_groups && n_in >= 0, "sdrhip_resampler_one");
    // an output index whose phase is `group`: the stream API addresses outputs by global index
    int64_t m0 = -1;
    for (int64_t m = 0; m < 4 * (int64_t)r->I + 4; m++)
        if (r->group(m) == group) { m0 = m; break; }
    SDRHIP_REQUIRE(m0 >= 0, "sdrhip_resampler_one: no output has that group");
    if (num == 0) return r->group(m0);
    SDRHIP_REQUIRE(in != nullptr && out != nullptr, "sdrhip_resampler_one");
    const int esz = r->cplx ? 2 : 1;
    const int64_t base = r->in_offset(m0);
    const int64_t need = r->in_offset(m0 + num - 1) - base + r->nloop;      // the SIMD loop walks nloop taps (zero padded)
    ScratchLease lease;
    if (!lease.c) return SDRHIP_ERR_HIP;
    RecScratch& sc = *lease.c;
    int rc = concat(sc, in, n_in, nullptr, 0, need, esz);                   // past the caller's vector the taps are zero: zero fill
    if (rc != SDRHIP_OK) return rc;
    const float* d_in;
    float* d_out;
    bool direct;
    const size_t ob = (size_t)num * esz * 4;
    if ((rc = stage(sc, (size_t)need * esz * 4, ob, &d_in, &d_out, &direct)) != SDRHIP_OK) return rc;
    if ((rc = resamp_run(r, sc.stream, d_in, base, d_out, m0, m0 + num, 0, 0)) != SDRHIP_OK) return rc;
    if ((rc = finish(sc, direct, out, ob)) != SDRHIP_OK) return rc;
    return r->group(m0 + num);
}

// resampleCrossHighLevel (FilterInternal.hs:410-423): `num` outputs over last ++ next starting with filter offset
// `filter_offset` at element 0 of `last`; returns the filter offset after the last output, or a negative error.
int sdrhip_resampler_cross(const sdrhip_resampler* r, int filter_offset, int num, const float* last, int n_last, const float* next, int n_next,
                           float* out)
{
    SDRHIP_REQUIRE(r != nullptr && num >= 0 && filter_offset >= 0 && filter_offset < r->I && n_last >= 0 && n_next >= 0, "sdrhip_resampler_cross");
    int64_t m0 = -1;
    for (int64_t m = 0; m < 4 * (int64_t)r->I + 4; m++)
        if (r->filter_offset(m) == filter_offset) { m0 = m; break; }
    SDRHIP_REQUIRE(m0 >= 0, "sdrhip_resampler_cross: no output has that filter offset");
    if (num == 0) return filter_offset;
    SDRHIP_REQUIRE(last != nullptr && next != nullptr && out != nullptr, "sdrhip_resampler_cross");
    const int esz = r->cplx ? 2 : 1;
    const int64_t base = r->in_offset(m0);
    // the sequential kernel strides the UNPADDED taps: ceil((ntaps - fo) / I) elements per output
    const int fo_last = r->filter_offset(m0 + num - 1);
    const int64_t need = r->in_offset(m0 + num - 1) - base + (r->ntaps - fo_last + r->I - 1) / r->I;
    if ((int64_t)n_last + n_next < need) {
        set_error("sdrhip_resampler_cross: %d outputs need %lld elements of last ++ next, %d + %d given", num, (long long)need, n_last, n_next);
        return SDRHIP_ERR_ARG;
    }
    ScratchLease lease;
    if (!lease.c) return SDRHIP_ERR_HIP;
    RecScratch& sc = *lease.c;
    int rc = concat(sc, last, n_last, next, n_next, need + r->nloop, esz);
    if (rc != SDRHIP_OK) return rc;
    const float* d_in;
    float* d_out;
    bool direct;
    const size_t ob = (size_t)num * esz * 4;
    if ((rc = stage(sc, (size_t)(need + r->nloop) * esz * 4, ob, &d_in, &d_out, &direct)) != SDRHIP_OK) return rc;
    if ((rc = resamp_run(r, sc.stream, d_in, base, d_out, m0, m0 + num, -1, 0)) != SDRHIP_OK) return rc;   // every output sequential
    if ((rc = finish(sc, direct, out, ob)) != SDRHIP_OK) return rc;
    return r->filter_offset(m0 + num);
}

}  // extern "C"
