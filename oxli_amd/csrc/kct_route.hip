// kct_route.hip -- the multi-GPU "early" route: entries go to the GPU that OWNS their slice of the key space
// while they are being counted, so that every owner sees its k-mers at the input's FULL coverage.
//
//   every rank      K1 (partition_windows_kernel) over its own records, bins grouped by owner GPU:
//                     compact entries (k <= 21): the bin is the top 10 bits of the 42-bit mix42 value; owner r holds the bins
//                                                [ceil(1024 r / world), ceil(1024 (r + 1) / world))
//                     64-bit entries (MurmurHash3 or mix64 values): bin = owner * 2^pl_bits + local super-bin,
//                                                owner = floor(hi32(value) * world / 2^32)  (the late route's owner rule)
//                   pack the regions of every owner's bins into one send buffer (pack_regions_kernel)
//   exchange        three all-to-alls through the caller's callback (RCCL over xGMI in oxli_amd/distributed.py): the per-region
//                   entry counts, the entries (4 or 8 bytes each), and the few entries that overflowed K1's LDS ring
//   every owner     K1b (repartition_kernel, reading the received regions through an offset table) spreads each of its
//                   super-bins over that super-bin's blocks; K2 counts them in LDS -- into the compact / 64-bit shadow table
//                   (dedupe-first modes) or the real table (hashing mode); overflow entries take the direct insert.
//
// Reference semantics: independent records (README.md:96-98), per-key sums (add(), lib.rs:778-837).  The ranks' tables end up a
// DISJOINT partition of the key space (by k-mer slice in the dedupe-first modes, by hash slice in hashing mode), so len /
// sum_counts of the global table are sums over ranks, exactly as after the late route (merge_across_ranks).
#include "kct_internal.h"
#include "path_policy.h"

#include <numeric>

namespace kct {

// region rho = bin * nwg + wg of K1's output ([wg][P bins][region_cap]) -> dst + off[rho]; cnt[rho] entries (whole 64-byte lines)
template <class T>
__global__ __launch_bounds__(kBlock) void pack_regions_kernel(const T *__restrict__ scratch, u32 region_cap, const u32 *__restrict__ cnt,
                                                              const u64 *__restrict__ off, u32 nregions, u32 nwg, u32 P, T *__restrict__ dst) {
    constexpr u32 kVec = 16 / sizeof(T);
    for (u32 rho = blockIdx.x; rho < nregions; rho += gridDim.x) {
        const u32 bin = rho / nwg, wg = rho - bin * nwg, n = cnt[rho];
        const uint4 *src = reinterpret_cast<const uint4 *>(scratch + ((u64)wg * P + bin) * region_cap);
        uint4 *out = reinterpret_cast<uint4 *>(dst + off[rho]);
        for (u32 i = threadIdx.x; i < n / kVec; i += kBlock) out[i] = src[i];
    }
}

// owner of a 64-bit value as K1's overflow regions carry them
template <int MODE>
__device__ __forceinline__ u32 owner_of_value(u64 v, u32 world) {
    if constexpr (MODE == 2) return (((u32)(v >> 32) & 1023u) * world) >> 10;  // compact: bit 63 | bin << 32 | entry
    else return __umulhi((u32)(v >> 32), world);
}

// K1's overflow regions ([nwg][ovf_cap] values, counts[wg]) bucketed by owner: count pass (dst == nullptr), then scatter pass
// (cursor[r] preset to owner r's start)
template <int MODE>
__global__ __launch_bounds__(kBlock) void bucket_overflow_kernel(const u64 *__restrict__ regions, const u32 *__restrict__ counts, u32 nwg, u32 ovf_cap,
                                                                 u32 world, u64 *cursor, u64 *dst) {
    for (u32 wg = blockIdx.x; wg < nwg; wg += gridDim.x) {
        const u32 n = counts[wg];
        const u64 *src = regions + (u64)wg * ovf_cap;
        for (u32 i = threadIdx.x; i < n; i += kBlock) {
            const u64 v = src[i];
            if (v == 0) continue;
            const u64 pos = atomicAdd(&cursor[owner_of_value<MODE>(v, world)], 1ULL);
            if (dst) dst[pos] = v;
        }
    }
}

}  // namespace kct

namespace kcth {

namespace {

int ceil_log2(u64 v) { int b = 0; while ((1ULL << b) < v) ++b; return b; }

struct Exchange {
    kct_table *t;
    unsigned world, rank;
    kct_alloc_fn alloc;
    kct_exchange_fn xfn;
    void *user;
    std::vector<void *> own;  // loopback (world == 1, no callbacks): buffers of this call
    ~Exchange() { for (void *p : own) (void)hipFree(p); }
    kct_status get(u64 bytes, void **p) {
        bytes = std::max<u64>(bytes, 256);
        if (alloc) {
            *p = alloc(user, bytes);
            if (!*p) { set_err("the exchange's allocator returned no buffer of %llu bytes", (unsigned long long)bytes); return KCT_ERR_NOMEM; }
            return KCT_OK;
        }
        HIP_TRY(hipMalloc(p, bytes));
        own.push_back(*p);
        return KCT_OK;
    }
    // all-to-all of send_elems[r] elements for rank r (contiguous, in rank order); waits for the table's stream first
    kct_status run(const void *d_send, const std::vector<u64> &send_elems, unsigned elem_bytes, void **d_recv, std::vector<u64> &recv_elems) {
        HIP_TRY(hipStreamSynchronize(t->stream));
        recv_elems.assign(world, 0);
        if (!xfn) {  // loopback: one rank
            *d_recv = const_cast<void *>(d_send);
            recv_elems[0] = send_elems[0];
            return KCT_OK;
        }
        *d_recv = nullptr;
        const int rc = xfn(user, d_send, send_elems.data(), elem_bytes, d_recv, recv_elems.data());
        if (rc != 0 || (!*d_recv && std::accumulate(recv_elems.begin(), recv_elems.end(), (u64)0) != 0)) {
            set_err("the exchange callback failed (%d)", rc);
            return KCT_ERR_ARG;
        }
        return KCT_OK;
    }
};

}  // namespace

kct_status consume_routed(kct_table *t, const unsigned char *d_stream, u64 nbytes, unsigned world, unsigned rank, int mode, kct_alloc_fn alloc,
                          kct_exchange_fn xfn, void *user, u64 *n_out, u64 stats[8]) {
    *n_out = 0;
    const int k = t->k, nwg = t->num_cus;
    if (world < 1 || world > 64 || rank >= world || mode < 0 || mode > 2 || (world > 1 && (!xfn || !alloc))) { set_err("bad world / rank / mode / callbacks"); return KCT_ERR_ARG; }
    if ((mode == 2 && k > 21) || (mode == 1 && k > 32) || k > 64 || k < 1) { set_err("mode %d does not take k = %d", mode, k); return KCT_ERR_ARG; }
    if (t->block_bits != kct::kBlockBitsMax) { set_err("the early route needs a table of at least 2^19 slots"); return KCT_ERR_ARG; }
    const int bbits = log2_u64(t->cap >> t->block_bits);
    // ---- who owns which of K1's bins; the owner-side fan-out (sub_bits) ------------------------------------------------------
    std::vector<unsigned> lo(world + 1);
    int sub_bits, pl_bits = 0;
    if (mode == 2) {
        for (unsigned r = 0; r <= world; ++r) lo[r] = (r * 1024u + world - 1) / world;
        const unsigned nb = lo[rank + 1] - lo[rank];
        sub_bits = std::max(6, ceil_log2(((1ULL << bbits) + nb - 1) / nb));  // shadow slots ~ table slots, >= 64 blocks per super-bin
        if (sub_bits > 10) { set_err("table too large for the early route at this world size"); return KCT_ERR_ARG; }
    } else {
        int pl_max = 0;
        while ((2u << pl_max) * world <= 1024u) ++pl_max;
        pl_bits = std::min(pl_max, bbits - 6);
        if (pl_bits < 0) { set_err("the early route needs a table of at least 2^19 slots"); return KCT_ERR_ARG; }
        sub_bits = bbits - pl_bits;
        if (sub_bits > 10) { set_err("table too large for the early route at this world size"); return KCT_ERR_ARG; }
        for (unsigned r = 0; r <= world; ++r) lo[r] = r << pl_bits;
    }
    const unsigned nb_me = lo[rank + 1] - lo[rank], bins_used = lo[world];
    const unsigned esz = mode == 2 ? 4 : 8;
    // K1's ring is split over 2^pbits bins: as few as hold the bins in use, so that a bin is as deep as it can be
    const int pbits = mode == 2 ? 10 : std::max(1, ceil_log2(bins_used));
    const u64 P = 1ULL << pbits;
    Exchange ex{t, world, rank, alloc, xfn, user, {}};
    // the shadow the entries will be counted into (dedupe-first modes) -- first of all: making it may convert what an older
    // shadow holds, which uses the scratch buffers below
    if (mode == 2) {
        bool ok = true;
        KCT_TRY(ensure_shadow32(t, 10 + sub_bits, &ok, nb_me, lo[rank]));
        if (!ok) { set_err("no room for the compact shadow table"); return KCT_ERR_NOMEM; }
        if (t->s32_dirty && t->s32_windows >= (1ULL << 38)) KCT_TRY(flush_shadow(t));  // (u32 counts: see aggregate_blocks32_kernel; a pass brings < 2^38)
    } else if (mode == 1) {
        bool ok = true;
        KCT_TRY(ensure_shadow(t, t->cap, &ok));
        if (!ok) { set_err("no room for the shadow table"); return KCT_ERR_NOMEM; }
    }

    // ---- K1 over this rank's records --------------------------------------------------------------------------------------
    const u64 npos = nbytes >= (u64)k ? nbytes - k + 1 : 0;
    const u64 ntiles = (npos + kct::kPartTile - 1) / kct::kPartTile;
    const u64 tiles_per_wg = (ntiles + nwg - 1) / nwg;
    unsigned int region_cap = region_capacity((double)(tiles_per_wg * kct::kPartTile) / (double)bins_used);
    region_cap = (region_cap + 15u) & ~15u;
    // (an abandoned pass cannot fall back to the direct kernel here: room for EVERY entry of a workgroup, up to the 2^20 the ring's
    // position arithmetic allows -- only a workgroup with more than a million overflowing entries makes the call fail)
    const unsigned int ovf_cap = (unsigned int)std::min<u64>(1ULL << 20, std::max<u64>(4096, tiles_per_wg * kct::kPartTile));
    KCT_TRY(t->d_scratch.reserve((u64)nwg * P * region_cap * esz));
    KCT_TRY(t->d_regions.reserve((u64)nwg * P * 4));
    KCT_TRY(t->d_irr.reserve((u64)nwg * ovf_cap * 8 + (u64)nwg * 4));
    KCT_TRY(zero_counters(t));
    du64 *d_overflow = t->d_counters + kNumCounters + 6;
    unsigned int *d_ovf_count = (unsigned int *)((du64 *)t->d_irr.p + (u64)nwg * ovf_cap);
    kct::PartitionArgs pa;
    pa.mask = t->cap - 1; pa.block_bits = kct::kBlockBitsMax + sub_bits; pa.pbits = pbits;
    pa.scratch = (du64 *)t->d_scratch.p; pa.region_cap = region_cap; pa.region_count = (unsigned int *)t->d_regions.p;
    pa.ovf = (du64 *)t->d_irr.p; pa.ovf_cap = ovf_cap; pa.ovf_count = d_ovf_count; pa.overflow = d_overflow;
    pa.ablate = 0;
    if (mode != 2) { pa.world = world; pa.pl_bits = pl_bits; }
    launch_partition(t, mode, d_stream, std::min<u64>(nbytes, npos + k - 1), ntiles, pa);
    HIP_TRY(hipGetLastError());
    // the region sizes, the overflow sizes and the abandon flag
    const u64 nreg = (u64)bins_used * nwg;
    std::vector<unsigned int> h_cnt(nreg), h_ovf(nwg);
    HIP_TRY(hipMemcpyAsync(h_cnt.data(), t->d_regions.p, nreg * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(h_ovf.data(), d_ovf_count, (u64)nwg * 4, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipMemcpyAsync(t->h_counters, d_overflow, 8, hipMemcpyDeviceToHost, t->stream));
    HIP_TRY(hipStreamSynchronize(t->stream));
    const bool abandoned = t->h_counters[0] != 0;  // an overflow region overflowed (hopelessly skewed input): this rank sends nothing
    if (abandoned) { std::fill(h_cnt.begin(), h_cnt.end(), 0u); std::fill(h_ovf.begin(), h_ovf.end(), 0u); HIP_TRY(hipMemsetAsync(t->d_regions.p, 0, nreg * 4, t->stream)); }

    // ---- pack: every owner's regions, contiguous, in (bin, workgroup) order -------------------------------------------------
    std::vector<u64> h_off(nreg + 1);
    h_off[0] = 0;
    for (u64 i = 0; i < nreg; ++i) h_off[i + 1] = h_off[i] + h_cnt[i];
    const u64 total_send = h_off[nreg];
    std::vector<u64> send_cnt(world), send_ent(world), recv_cnt, recv_ent, recv_ovf;
    for (unsigned r = 0; r < world; ++r) {
        send_cnt[r] = (u64)(lo[r + 1] - lo[r]) * nwg;
        send_ent[r] = h_off[(u64)lo[r + 1] * nwg] - h_off[(u64)lo[r] * nwg];
    }
    void *d_send_cnt = nullptr, *d_send = nullptr;
    KCT_TRY(ex.get(nreg * 4, &d_send_cnt));
    KCT_TRY(ex.get(total_send * esz, &d_send));
    HIP_TRY(hipMemcpyAsync(d_send_cnt, t->d_regions.p, nreg * 4, hipMemcpyDeviceToDevice, t->stream));
    KCT_TRY(t->d_aux.reserve((nreg + 1) * 8));
    HIP_TRY(hipMemcpyAsync(t->d_aux.p, h_off.data(), (nreg + 1) * 8, hipMemcpyHostToDevice, t->stream));
    if (total_send) {
        ProfScope ps(t, "pack_regions_kernel");
        const unsigned grid = (unsigned)std::min<u64>(nreg, 1u << 16);
        if (mode == 2) hipLaunchKernelGGL(kct::pack_regions_kernel<unsigned int>, dim3(grid), dim3(kct::kBlock), 0, t->stream, (const unsigned int *)t->d_scratch.p,
                                          region_cap, (const unsigned int *)t->d_regions.p, (const du64 *)t->d_aux.p, (unsigned)nreg, (unsigned)nwg, (unsigned)P, (unsigned int *)d_send);
        else hipLaunchKernelGGL(kct::pack_regions_kernel<du64>, dim3(grid), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_scratch.p, region_cap,
                                (const unsigned int *)t->d_regions.p, (const du64 *)t->d_aux.p, (unsigned)nreg, (unsigned)nwg, (unsigned)P, (du64 *)d_send);
        HIP_TRY(hipGetLastError());
    }
    // ---- K1's overflow entries, bucketed by owner ----------------------------------------------------------------------------
    u64 ovf_total = 0;
    for (unsigned v : h_ovf) ovf_total += v;
    std::vector<u64> send_ovf(world, 0);
    void *d_send_ovf = nullptr;
    KCT_TRY(ex.get(ovf_total * 8, &d_send_ovf));
    KCT_TRY(t->d_aux2.reserve((u64)world * 8 + 64));
    du64 *d_cursor = (du64 *)t->d_aux2.p;  // one cursor per owner
    if (ovf_total) {
        HIP_TRY(hipMemsetAsync(d_cursor, 0, (u64)world * 8, t->stream));
        auto bucket = [&](du64 *dst) {
            const unsigned grid = (unsigned)std::min<int>(nwg, 1024);
            if (mode == 2) hipLaunchKernelGGL(kct::bucket_overflow_kernel<2>, dim3(grid), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p, (const unsigned int *)d_ovf_count, (unsigned)nwg, ovf_cap, world, d_cursor, dst);
            else hipLaunchKernelGGL(kct::bucket_overflow_kernel<0>, dim3(grid), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->d_irr.p, (const unsigned int *)d_ovf_count, (unsigned)nwg, ovf_cap, world, d_cursor, dst);
        };
        bucket(nullptr);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(send_ovf.data(), d_cursor, (u64)world * 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        std::vector<u64> start(world, 0);
        for (unsigned r = 1; r < world; ++r) start[r] = start[r - 1] + send_ovf[r - 1];
        HIP_TRY(hipMemcpyAsync(d_cursor, start.data(), (u64)world * 8, hipMemcpyHostToDevice, t->stream));
        bucket((du64 *)d_send_ovf);
        HIP_TRY(hipGetLastError());
    }
    // ---- the three all-to-alls --------------------------------------------------------------------------------------------------
    void *d_recv_cnt = nullptr, *d_recv = nullptr, *d_recv_ovf = nullptr;
    const double t0 = now_ms();
    KCT_TRY(ex.run(d_send_cnt, send_cnt, 4, &d_recv_cnt, recv_cnt));
    KCT_TRY(ex.run(d_send, send_ent, esz, &d_recv, recv_ent));
    KCT_TRY(ex.run(d_send_ovf, send_ovf, 8, &d_recv_ovf, recv_ovf));
    const double exchange_ms = now_ms() - t0;
    for (unsigned p = 0; p < world; ++p)
        if (recv_cnt[p] != (u64)nb_me * nwg) { set_err("rank %u sent %llu region counts, expected %llu: the ranks' tables differ in geometry", p, (unsigned long long)recv_cnt[p], (unsigned long long)nb_me * nwg); return KCT_ERR_ARG; }
    // ---- owner side: the received regions as K1b's input ([super-bin][segment = (peer, workgroup)]) ---------------------------
    const u64 nseg = (u64)world * nwg, nidx = (u64)nb_me * nseg;
    std::vector<unsigned int> h_rc(nidx), h_c2(nidx);
    std::vector<u64> h_o2(nidx);
    HIP_TRY(hipMemcpy(h_rc.data(), d_recv_cnt, nidx * 4, hipMemcpyDeviceToHost));
    u64 total_recv = 0;
    for (unsigned p = 0; p < world; ++p) {
        u64 run = 0;
        for (unsigned s = 0; s < nb_me; ++s)
            for (int w = 0; w < nwg; ++w) {
                const unsigned int c = h_rc[((u64)p * nb_me + s) * nwg + w];
                const u64 i = (u64)s * nseg + (u64)p * nwg + w;
                h_c2[i] = c; h_o2[i] = total_recv + run;
                run += c;
            }
        if (run != recv_ent[p]) { set_err("rank %u announced %llu entries and sent %llu", p, (unsigned long long)run, (unsigned long long)recv_ent[p]); return KCT_ERR_ARG; }
        total_recv += run;
    }
    u64 recv_ovf_total = 0;
    for (u64 v : recv_ovf) recv_ovf_total += v;
    KCT_TRY(t->d_regions.reserve(std::max<u64>((u64)nwg * P * 4, nidx * 4)));
    KCT_TRY(t->d_aux.reserve(nidx * 8 + 64));
    HIP_TRY(hipMemcpyAsync(t->d_regions.p, h_c2.data(), nidx * 4, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(t->d_aux.p, h_o2.data(), nidx * 8, hipMemcpyHostToDevice, t->stream));
    du64 *d_ovf_total = (du64 *)((char *)t->d_aux.p + nidx * 8);
    HIP_TRY(hipMemcpyAsync(d_ovf_total, &recv_ovf_total, 8, hipMemcpyHostToDevice, t->stream));

    // ---- K1b: every super-bin of mine over its 2^sub_bits blocks ---------------------------------------------------------------
    // (a writer's sixteen waves take one input segment each: more than nseg / 16 writers would find nothing to read)
    const u64 B = (u64)nb_me << sub_bits, W = std::max<u64>(1, std::min<u64>((u64)nwg / nb_me, nseg / 16));
    unsigned int out_cap = region_capacity((double)total_recv / (double)B / (double)W);
    out_cap = (out_cap + 63u) & ~63u;
    // K1b's overflow regions: room for EVERY entry of the busiest workgroup (an abandoned pass cannot fall back to the direct kernel
    // here), up to the 2^20 the ring's position arithmetic allows.  Workgroup (s, w) reads segments 16 w + wave, + 16 W, ...
    u64 busiest = 0;
    for (unsigned sb = 0; sb < nb_me; ++sb)
        for (u64 w = 0; w < W; ++w) {
            u64 sum = 0;
            for (u64 seg0 = w * 16; seg0 < nseg; seg0 += W * 16)
                for (u64 seg = seg0; seg < std::min<u64>(seg0 + 16, nseg); ++seg) sum += h_c2[(u64)sb * nseg + seg];
            busiest = std::max(busiest, sum);
        }
    const unsigned int ovf2_cap = (unsigned int)std::min<u64>(1ULL << 20, std::max<u64>(4096, busiest));
    KCT_TRY(t->d_scratch2.reserve(B * W * out_cap * esz));
    KCT_TRY(t->d_regions2.reserve(B * W * 4));
    KCT_TRY(t->d_irr2.reserve((u64)nb_me * W * ovf2_cap * 8 + (u64)nb_me * W * 4));
    unsigned int *d_ovf2_count = (unsigned int *)((du64 *)t->d_irr2.p + (u64)nb_me * W * ovf2_cap);
    kct::RepartitionArgs ra;
    ra.mask = (mode == 2 ? compact_slots(t) : t->cap) - 1; ra.block_bits = kct::kBlockBitsMax; ra.sub_bits = sub_bits;
    ra.in = d_recv; ra.in_cap = 0; ra.in_count = (const unsigned int *)t->d_regions.p; ra.in_off = (const du64 *)t->d_aux.p;
    ra.nseg = (int)nseg; ra.nbins = (int)nb_me; ra.writers = (int)W;
    ra.out = t->d_scratch2.p; ra.out_cap = out_cap; ra.out_count = (unsigned int *)t->d_regions2.p;
    ra.ovf = (du64 *)t->d_irr2.p; ra.ovf_cap = ovf2_cap; ra.ovf_count = d_ovf2_count; ra.overflow = d_overflow; ra.ovf_n = nullptr;
    ra.bin0 = lo[rank];
    ra.min_lines = repartition_min_lines(t, mode == 2 ? kct::kRingEntries * 2 : kct::kRingEntries, sub_bits, (int)esz);
    HIP_TRY(hipMemsetAsync(d_overflow, 0, 8, t->stream));  // (K1's flag has been read; K1b raises it again if ITS overflow regions overflow)
    launch_repartition(t, mode, (unsigned)(nb_me * W), ra, false);
    HIP_TRY(hipGetLastError());

    // ---- K2: one workgroup per block, counts in LDS ------------------------------------------------------------------------------
    kct::FailedBlocks fb;
    KCT_TRY(failed_blocks(t, B, &fb));
    const void *k2_scratch = t->d_scratch2.p;
    const unsigned int *k2_counts = (const unsigned int *)t->d_regions2.p;
    if (mode == 2) {
        kct::Aggregate32Args aa;
        aa.words = t->shadow32; aa.block_bits = kct::kBlockBitsMax; aa.sbits = 10 + sub_bits;
        aa.scratch = (const unsigned int *)k2_scratch; aa.seg_stride = out_cap; aa.block_stride = W * out_cap;
        aa.region_count = k2_counts; aa.nregions = (int)W;
        aa.fresh = t->s32_empty ? 1 : 0; aa.overflow = d_overflow; aa.ablate = 0; aa.failed = fb; aa.counters = t->d_counters; aa.nblocks = (unsigned int)B;
        launch_aggregate32(t, (unsigned)std::min<u64>(B, 2 * (u64)nwg), aa);
    } else {
        kct::AggregateArgs aa;
        aa.words = mode == 1 ? t->shadow : t->slots; aa.block_bits = kct::kBlockBitsMax; aa.pbits = bbits;
        aa.scratch = (const du64 *)k2_scratch; aa.seg_stride = out_cap; aa.block_stride = W * out_cap;
        aa.region_count = k2_counts; aa.nregions = (int)W;
        aa.fresh = (mode == 1 ? t->shadow_empty : t->lazy_empty) ? 1 : 0; aa.overflow = d_overflow; aa.ablate = 0; aa.nblocks = (unsigned int)B;
        aa.failed = fb; aa.counters = t->d_counters;
        launch_aggregate64(t, (unsigned)std::min<u64>(B, (u64)nwg), aa, mode == 1);
    }
    HIP_TRY(hipGetLastError());
    u64 c[4], unused;
    KCT_TRY(read_counters(t, c, &unused));
    if (t->h_counters[kNumCounters + 6] != 0) { set_err("the received entries are too skewed for the early route (K1b's overflow regions overflowed)"); return KCT_ERR_ARG; }
    const u64 nfailed = t->h_counters[kNumCounters + 7], failed_entries = t->h_counters[kNumCounters + 3];
    const u64 counted = c[kct::CTR_COUNTED], new_in_k2 = mode == 2 ? c[kct::CTR_NEW_BY_ZERO] : c[kct::CTR_NEWKEYS];
    if (mode == 0) { t->lazy_empty = false; t->n_keys += new_in_k2; }
    else if (mode == 1) { t->shadow_empty = false; t->shadow_dirty = true; t->shadow_keys += new_in_k2; }
    else { t->s32_empty = false; t->s32_dirty = true; t->s32_keys += new_in_k2; t->s32_windows += total_recv; }

    // ---- what did not pass a ring: K1b's overflow regions and the overflow entries received -> the real table, direct insert ------
    u64 ovf2_total = 0;
    KCT_TRY(overflow_total(t, d_ovf2_count, (size_t)(nb_me * W), nullptr, 0, &ovf2_total));
    KCT_TRY(materialize(t));
    const u64 spill_cap = std::max<u64>(ovf2_total + recv_ovf_total, 1);
    KCT_TRY(t->d_spill.reserve(spill_cap * 16));
    HIP_TRY(hipMemsetAsync(t->d_counters, 0, (kNumCounters + 1) * sizeof(u64), t->stream));  // tallies and the spill cursor
    kct::TableView mv = view(t, spill_cap);
    launch_merge_overflow(t, mode, (const du64 *)t->d_irr2.p, d_ovf2_count, (int)(nb_me * W), ovf2_cap, nullptr, mv, nullptr);
    if (recv_ovf_total) launch_merge_overflow(t, mode, (const du64 *)d_recv_ovf, nullptr, (int)((recv_ovf_total + 65535) / 65536), 65536u, nullptr, mv, d_ovf_total);
    HIP_TRY(hipGetLastError());
    u64 c2[4], spilled;
    KCT_TRY(read_counters(t, c2, &spilled));
    *n_out += counted + c2[kct::CTR_TOTAL_ADDED];
    t->n_keys += c2[kct::CTR_NEWKEYS];
    if (spilled) {
        KCT_TRY(t->d_aux2.reserve(spilled * 16));
        HIP_TRY(hipMemcpyAsync(t->d_aux2.p, t->d_spill.p, spilled * 16, hipMemcpyDeviceToDevice, t->stream));
        KCT_TRY(replay_spill(t, spilled, n_out));
    }
    if (nfailed) {  // blocks K2 had to abandon: their entries are (hashed and) counted into the real table
        u64 tl[4] = {0, 0, 0, 0};
        KCT_TRY(recount_failed(t, mode, k2_scratch, out_cap, W * out_cap, k2_counts, (int)W, nfailed, failed_entries, 10 + sub_bits, tl));
        *n_out += tl[kct::CTR_TOTAL_ADDED];
    }
    if (mode == 1 && t->shadow && (double)t->shadow_keys > kMaxLoad * (double)t->shadow_cap) { KCT_TRY(flush_shadow(t)); KCT_TRY(grow_to(t, t->cap * 2)); }
    if (mode == 2 && (double)t->s32_keys > 0.8 * (double)compact_slots(t)) KCT_TRY(flush_shadow(t));
    t->windows_since_read += total_recv;
    HIP_TRY(hipStreamSynchronize(t->stream));  // (the exchange buffers may be released when this call returns)
    if (stats) {
        stats[0] = total_send - send_ent[rank]; stats[1] = total_recv - recv_ent[rank];  // entries that crossed to / from OTHER ranks
        stats[2] = esz; stats[3] = ovf_total; stats[4] = recv_ovf_total; stats[5] = (u64)(exchange_ms * 1000.0); stats[6] = nfailed; stats[7] = abandoned ? 1 : 0;
    }
    KCT_DBG(t, "routed pass (mode %d, rank %u of %u): npos=%llu sent=%llu recv=%llu entries of %u B, overflow sent=%llu recv=%llu, counted=%llu merged=%llu new=%llu abandoned blocks=%llu exchange=%.3f ms\n",
            mode, rank, world, (unsigned long long)npos, (unsigned long long)total_send, (unsigned long long)total_recv, esz, (unsigned long long)ovf_total,
            (unsigned long long)recv_ovf_total, (unsigned long long)counted, (unsigned long long)c2[kct::CTR_TOTAL_ADDED], (unsigned long long)new_in_k2,
            (unsigned long long)nfailed, exchange_ms);
    if (abandoned) { set_err("this rank's input is too skewed for the early route (its share was NOT counted)"); return KCT_ERR_ARG; }
    return KCT_OK;
}

}  // namespace kcth

using namespace kcth;

extern "C" kct_status kct_consume_device_routed(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes, uint32_t world, uint32_t rank,
                                                int mode, kct_alloc_fn alloc, kct_exchange_fn exchange, void *user, uint64_t *n_owned, uint64_t *stats8) {
    KCT_TRY(use_consume(t));
    if (!n_owned || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    u64 n = 0;
    const kct_status st = consume_routed(t, (const unsigned char *)d_stream, nbytes, world, rank, mode, alloc, exchange, user, &n, stats8);
    *n_owned = n;
    if (st == KCT_OK) t->consumed += consumed_bytes;
    return st;
}
