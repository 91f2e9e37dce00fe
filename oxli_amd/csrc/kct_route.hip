// kct_route.hip -- the multi-GPU "early" route: every k-mer is counted by the GPU that OWNS it, at the input's full
// coverage, and what crosses xGMI is SUPER-K-MERS -- about one byte per window at k = 21, 0.9 at k = 51 -- not one entry per
// window.
//
//   every rank      split_superkmers_kernel over its own records: owner(k-mer) = sk_owner(minimiser, world), maximal runs of good
//                   windows with one owner as 2-bit bases + one start bit per window (superkmer_kernels.h); gather_units_kernel makes one
//                   contiguous part per owner
//   exchange        through the caller's kct_exchange_ops (RCCL over xGMI: csrc/kct_rccl.cpp; torch.distributed: oxli_amd/distributed.py):
//                   a small host-side all-to-all of sizes (the per-stream directory rides along, and every rank's status: a failure
//                   anywhere ends the call on EVERY rank before the next payload moves), then ONE all-to-all of the parts
//   every owner     run_directory_kernel (one RunGroup per 64 windows), then the table's ORDINARY bulk path over the received windows
//                   (consume_stream_runs: K1's RUNS instantiations read the runs; compact / 64-bit dedupe-first or hashing, one or two
//                   levels, growth, probe -- whatever path_policy.h picks for a table of the owner's size: nothing about it is shared
//                   between ranks, so no geometry has to agree)
//
// The call is cut into passes and pipelined: while pass p's parts are on the wire the GPU splits pass p + 1, and while the owner side of
// pass p runs, pass p + 1's parts are on the wire.
//
// Reference semantics: independent records (README.md:96-98), per-key sums (add(), lib.rs:778-837).  The ranks' tables end up a DISJOINT
// partition of the key space (by minimiser), so len / sum_counts of the global table are sums over ranks, exactly as after the late
// route (merge_across_ranks).
#include "kct_internal.h"
#include "path_policy.h"

#include <numeric>

namespace kcth {

namespace {

// expected bases per window: a run of n windows is n + k - 1 bases, and a minimiser over W = k - m + 1 m-mers changes every (W + 1) / 2
// windows on random sequence
double bases_per_window(int k) {
    const int m = std::min(k, 8), w = k - m + 1;
    return 1.0 + 2.0 * (double)(k - 1) / (double)(w + 1);
}

struct Slab {  // a send or receive buffer of the exchange (the caller's allocator, or the table's own for the loop-back)
    void *p = nullptr;
    u64 cap = 0;
};

struct Route {
    kct_table *t;
    unsigned world, rank;
    const kct_exchange_ops *ops;
    int nwg;
    double wait_ms = 0;

    // kct_debug_inject_fault: is the armed fault this point (of this pass)?  One shot.
    bool fault(int point, u64 pass = 0) {
        if (t->fault_point != point) return false;
        if ((point == KCT_FAULT_SPLIT || point == KCT_FAULT_START || point == KCT_FAULT_WAIT) && t->fault_pass != pass) return false;
        t->fault_point = KCT_FAULT_NONE;
        return true;
    }
    kct_status slab(Slab &s, DevBuf &own, u64 bytes) {
        bytes = std::max<u64>(bytes, 256) + 64;   // (K1 reads up to two words past a window's last base)
        if (fault(KCT_FAULT_ALLOC)) { set_err("injected fault: no buffer of %llu bytes", (unsigned long long)bytes); return KCT_ERR_NOMEM; }
        if (bytes <= s.cap) return KCT_OK;
        if (!ops) {
            KCT_TRY(own.reserve(bytes + bytes / 8));
            s.p = own.p; s.cap = own.cap;
            return KCT_OK;
        }
        HIP_TRY(hipStreamSynchronize(t->stream));   // (whatever still reads the old buffer)
        if (s.p && ops->release) ops->release(ops->user, s.p);
        s.p = ops->alloc(ops->user, bytes + bytes / 8);
        s.cap = s.p ? bytes + bytes / 8 : 0;
        if (!s.p) { set_err("the exchange's allocator returned no buffer of %llu bytes", (unsigned long long)(bytes + bytes / 8)); return KCT_ERR_NOMEM; }
        return KCT_OK;
    }
    // all-to-all of nvals u64 per peer, host memory; slot 0 of every message is the sender's status
    kct_status sizes(std::vector<u64> &send, unsigned nvals, std::vector<u64> &recv) {
        recv.assign((size_t)world * nvals, 0);
        if (!ops) { recv = send; return KCT_OK; }
        const double t0 = now_ms();
        const int rc = ops->exchange_sizes(ops->user, send.data(), nvals, recv.data());
        wait_ms += now_ms() - t0;
        if (rc != 0) { set_err("the exchange's size all-to-all failed (%d)", rc); return KCT_ERR_HIP; }
        return KCT_OK;
    }
    // every rank learns whether ANY rank has failed; `mine` is passed through for the rank that did
    kct_status agree(kct_status mine) {
        if (!ops) return mine;
        std::vector<u64> s(world, (u64)mine), r;
        const kct_status st = sizes(s, 1, r);
        if (st != KCT_OK) return mine != KCT_OK ? mine : st;
        if (mine != KCT_OK) return mine;
        for (unsigned p = 0; p < world; ++p)
            if (r[p] != 0) { set_err("the early route failed on rank %u (status %llu): this rank's table is short of what that rank was to send", p, (unsigned long long)r[p]); return KCT_ERR_HIP; }
        return KCT_OK;
    }
};

struct PassOut {  // what the split of one pass left for the exchange
    std::vector<u64> part_off, part_bytes;   // per owner, in the send slab
    std::vector<u64> dir;                    // [owner][stream]: windows | base units << 32
    u64 windows_out = 0, runs = 0, bytes = 0;
    int retries = 0;
};

// this rank's records [d_stream, d_stream + nbytes) -> one part per owner in `send`
kct_status split_pass(Route &r, const unsigned char *d_stream, u64 nbytes, Slab &send, DevBuf &own_send, PassOut &out) {
    kct_table *t = r.t;
    const int k = t->k, nwg = r.nwg;
    const unsigned world = r.world;
    const u64 nstreams = (u64)nwg * world;
    out.part_off.assign(world, 0); out.part_bytes.assign(world, 0); out.dir.assign(nstreams, 0);
    out.windows_out = out.runs = out.bytes = 0;
    const u64 npos = nbytes >= (u64)k ? nbytes - k + 1 : 0;
    if (!npos) return KCT_OK;
    const u64 ntiles = (npos + kct::kSkTile - 1) / kct::kSkTile, tiles_per_wg = (ntiles + nwg - 1) / nwg;
    const double per_stream = (double)(tiles_per_wg * kct::kSkTile) / world;
    u64 cap_units = (u64)(per_stream * bases_per_window(k) * 1.3 / 64.0) + 16, cap_sunits = (u64)(per_stream * 1.3 / 128.0) + 16;
    std::vector<unsigned int> meta(4 * nstreams);
    for (;;) {
        KCT_TRY(t->d_sk_bases.reserve(nstreams * cap_units * 16));
        KCT_TRY(t->d_sk_starts.reserve(nstreams * cap_sunits * 16));
        KCT_TRY(t->d_sk_meta.reserve(4 * nstreams * 4 + 64));
        du64 *d_overflow = (du64 *)((char *)t->d_sk_meta.p + 4 * nstreams * 4);
        HIP_TRY(hipMemsetAsync(d_overflow, 0, 8, t->stream));
        kct::SplitArgs sa;
        sa.world = world;
        sa.bases_out = (uint4 *)t->d_sk_bases.p; sa.cap_units = (unsigned int)cap_units;
        sa.starts_out = (uint4 *)t->d_sk_starts.p; sa.cap_sunits = (unsigned int)cap_sunits;
        unsigned int *m = (unsigned int *)t->d_sk_meta.p;
        sa.nwin = m; sa.nunits = m + nstreams; sa.nsunits = m + 2 * nstreams; sa.nruns = m + 3 * nstreams;
        sa.overflow = d_overflow;
        sa.ablate = (unsigned int)t->ablate;
        sa.pcodes = nullptr; sa.pvalid = nullptr;
        launch_split(t, d_stream, std::min<u64>(nbytes, npos + k - 1), ntiles, sa);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(meta.data(), t->d_sk_meta.p, 4 * nstreams * 4, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_overflow, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->h_counters[0] == 0) break;
        // a region was too small (input far from random: many short runs, or one owner takes most): the counts say what is needed
        u64 need_u = 0, need_s = 0;
        for (u64 i = 0; i < nstreams; ++i) { need_u = std::max<u64>(need_u, meta[nstreams + i]); need_s = std::max<u64>(need_s, meta[2 * nstreams + i]); }
        if (++out.retries > 3 || need_u >= (1ULL << 31) || need_s >= (1ULL << 31)) { set_err("the super-k-mer regions overflowed %d times", out.retries); return KCT_ERR_NOMEM; }
        cap_units = std::max(cap_units, need_u + need_u / 16 + 16);
        cap_sunits = std::max(cap_sunits, need_s + need_s / 16 + 16);
        KCT_DBG(t, "split: regions too small, again with %llu / %llu units per stream\n", (unsigned long long)cap_units, (unsigned long long)cap_sunits);
    }
    // ---- one part per owner: [base units of its nwg streams][start units of its nwg streams] -------------------------------------------
    const unsigned int *nwin = meta.data(), *nun = meta.data() + nstreams, *nsu = meta.data() + 2 * nstreams, *nrun = meta.data() + 3 * nstreams;
    std::vector<u64> src_off(2 * nstreams), dst_off(2 * nstreams);
    std::vector<unsigned int> cnt(2 * nstreams);
    u64 total_units = 0;
    for (unsigned o = 0; o < world; ++o) {
        out.part_off[o] = total_units * 16;
        for (int half = 0; half < 2; ++half)
            for (int w = 0; w < nwg; ++w) {
                const u64 i = (u64)w * world + o, j = (u64)half * nstreams + (u64)o * nwg + w;
                src_off[j] = i * (half ? cap_sunits : cap_units);
                dst_off[j] = total_units;
                cnt[j] = half ? nsu[i] : nun[i];
                total_units += cnt[j];
                if (!half) { out.dir[(u64)o * nwg + w] = (u64)nwin[i] | ((u64)nun[i] << 32); out.windows_out += nwin[i]; out.runs += nrun[i]; }
            }
        out.part_bytes[o] = total_units * 16 - out.part_off[o];
    }
    out.bytes = total_units * 16;
    KCT_TRY(r.slab(send, own_send, out.bytes));
    const u64 list_bytes = 2 * nstreams * (8 + 8 + 4);
    KCT_TRY(t->d_sk_lists.reserve(list_bytes));
    du64 *d_src = (du64 *)t->d_sk_lists.p, *d_dst = d_src + 2 * nstreams;
    unsigned int *d_cnt = (unsigned int *)(d_dst + 2 * nstreams);
    HIP_TRY(hipMemcpyAsync(d_src, src_off.data(), 2 * nstreams * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(d_dst, dst_off.data(), 2 * nstreams * 8, hipMemcpyHostToDevice, t->stream));
    HIP_TRY(hipMemcpyAsync(d_cnt, cnt.data(), 2 * nstreams * 4, hipMemcpyHostToDevice, t->stream));
    launch_gather_units(t, t->d_sk_bases.p, d_src, d_dst, d_cnt, (unsigned)nstreams, send.p);
    launch_gather_units(t, t->d_sk_starts.p, d_src + nstreams, d_dst + nstreams, d_cnt + nstreams, (unsigned)nstreams, send.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(t->stream));   // (the host vectors above; and the exchange may read the slab from another stream)
    return KCT_OK;
}

// What this rank has received and not yet counted: the passes' parts back to back in ONE buffer (about a byte per window, so a whole
// call usually fits), counted together -- one pass of the table's bulk path over all of it instead of one per exchange: K2 and the
// conversion visit every table block once per pass whatever the pass brings.
struct Inbox {
    Slab buf;
    static constexpr u64 kFront = 64;   // (a stream never begins in the buffer's first 2 (k - 1) bits: partition_args.h RunGroup)
    u64 used = kFront;                  // bytes of buf in use
    std::vector<kct::RunStream> streams;
    u64 groups = 0, windows = 0;
};

// the parts of one pass, received at buf + base (recv_off / recv_bytes per peer, directories dir[peer][stream]) -> the inbox's streams
kct_status inbox_add(Route &r, Inbox &in, u64 base, const std::vector<u64> &recv_off, const std::vector<u64> &recv_bytes, const std::vector<u64> &dir) {
    const int nwg = r.nwg;
    for (unsigned p = 0; p < r.world; ++p) {
        u64 units = 0, sunits = 0;
        for (int w = 0; w < nwg; ++w) { const u64 d = dir[(u64)p * nwg + w]; units += d >> 32; sunits += ((d & 0xFFFFFFFFULL) + 127) / 128; }
        if ((units + sunits) * 16 != recv_bytes[p]) { set_err("rank %u announced %llu bytes of super-k-mers and its directory adds up to %llu", p, (unsigned long long)recv_bytes[p], (unsigned long long)((units + sunits) * 16)); return KCT_ERR_ARG; }
        u64 boff = base + recv_off[p], soff = boff + units * 16;
        for (int w = 0; w < nwg; ++w) {
            const u64 d = dir[(u64)p * nwg + w], nwin = d & 0xFFFFFFFFULL;
            if (nwin) {
                kct::RunStream s;
                s.bit0 = boff * 8; s.word0 = soff >> 3; s.nwin = (unsigned int)nwin; s.group0 = in.groups;
                in.streams.push_back(s);
                in.groups += (nwin + 63) >> 6;
                in.windows += nwin;
            }
            boff += (d >> 32) * 16; soff += ((nwin + 127) / 128) * 16;
        }
    }
    return KCT_OK;
}

// everything in the inbox -> counted into this rank's table (`more` = windows later passes of the same call will still bring)
kct_status inbox_count(Route &r, Inbox &in, u64 more, u64 *n_out, bool keep_room = false) {  // keep_room: a pass is arriving behind what is counted
    kct_table *t = r.t;
    *n_out = 0;
    kct_status st = KCT_OK;
    if (in.groups) {
        KCT_TRY(t->d_sk_dir.reserve(in.streams.size() * sizeof(kct::RunStream) + in.groups * sizeof(kct::RunGroup) + 64));
        kct::RunGroup *d_groups = (kct::RunGroup *)t->d_sk_dir.p;
        kct::RunStream *d_streams = (kct::RunStream *)(d_groups + in.groups);
        HIP_TRY(hipMemcpyAsync(d_streams, in.streams.data(), in.streams.size() * sizeof(kct::RunStream), hipMemcpyHostToDevice, t->stream));
        launch_run_directory(t, d_streams, (unsigned)in.streams.size(), (const du64 *)in.buf.p, d_groups);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(t->stream));   // (`streams` is host memory)
        kct::RunsInput ri;
        ri.bases = (const unsigned int *)in.buf.p; ri.groups = d_groups;
        t->more_windows = more;
        st = consume_stream_runs(t, ri, in.groups, n_out);
        t->more_windows = 0;
    }
    in.streams.clear(); in.groups = in.windows = 0;
    if (!keep_room) in.used = Inbox::kFront;
    return st;
}

}  // namespace

kct_status consume_routed(kct_table *t, const unsigned char *d_stream, u64 nbytes, unsigned world, unsigned rank, const kct_exchange_ops *ops, u64 max_windows,
                          u64 *n_out, u64 stats[16]) {
    *n_out = 0;
    const int k = t->k;
    if (world < 1 || world > kct::kSkMaxWorld || rank >= world) { set_err("bad world / rank"); return KCT_ERR_ARG; }
    if (ops && (!ops->alloc || !ops->exchange_sizes || !ops->start || !ops->wait)) { set_err("kct_exchange_ops needs alloc, exchange_sizes, start and wait"); return KCT_ERR_ARG; }
    const bool solo = !ops && world > 1;   // no exchange: of this rank's OWN records, count the k-mers it owns (what its peers own is dropped)
    if (k > 64) { set_err("the early route takes k <= 64"); return KCT_ERR_ARG; }
    Route r{t, world, rank, ops, split_streams(t)};
    // ---- passes: as many as the tightest rank needs (HBM: regions, two send and two receive slabs, the owner's scratch), at least
    // four (eight with more than four ranks) for a long stream so that the wire hides behind the kernels
    const u64 windows = nbytes >= (u64)k ? nbytes - k + 1 : 0;
    // A failure of THIS rank from here on is kept in `status` and travels in its messages: the rank stays in step with its peers, every
    // rank learns of it in the same collective and every rank ends the call there -- no return between the first collective and the last.
    kct_status status = KCT_OK;
    if (r.fault(KCT_FAULT_MEMINFO)) { set_err("injected fault: the HBM query before the first collective"); status = KCT_ERR_HIP; max_windows = 1ULL << 24; }
    if (!max_windows) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { set_err("hipMemGetInfo failed at the start of the early route"); status = KCT_ERR_HIP; free_b = 0; (void)hipGetLastError(); }
        const double held = (double)(t->d_scratch.cap + t->d_scratch2.cap + t->d_sk_bases.cap + t->d_sk_starts.cap);
        const double per_window = 6.0 * bases_per_window(k) / 4.0 + 24.0;
        max_windows = (u64)std::max(1.0 * (1 << 24), 0.6 * ((double)free_b + held) / per_window);
        // (what no pass can hide is the first cut and the last transfer: with many peers the wire is short and eight passes expose half as much)
        if (windows >= (1ULL << 28)) max_windows = std::min<u64>(max_windows, std::max<u64>(1ULL << 26, windows / (world > 4 ? 8 : 4)));
    }
    max_windows = std::max<u64>(1 << 16, max_windows & ~0xFFFFULL);
    std::vector<u64> sz(world * 3), rz;
    // (the first round also carries every rank's stream count: the per-pass size messages and the stream directories are laid out by it,
    // so ranks with different CU counts -- mixed SKUs, partition modes, CU masks -- must not go on; and every rank's status so far)
    for (unsigned p = 0; p < world; ++p) { sz[3 * p] = (u64)r.nwg; sz[3 * p + 1] = windows ? (windows + max_windows - 1) / max_windows : 1; sz[3 * p + 2] = (u64)status; }
    KCT_TRY(r.sizes(sz, 3, rz));   // (the first collective: if IT fails there is nothing to agree through)
    u64 passes = 1;
    for (unsigned p = 0; p < world; ++p) passes = std::max(passes, rz[3 * p + 1]);
    // every rank reads the same rz columns, so these three verdicts fall alike everywhere
    for (unsigned p = 0; p < world; ++p)
        if (rz[3 * p + 2] != 0) {
            if (status == KCT_OK) set_err("the early route failed on rank %u (status %llu) before its first pass", p, (unsigned long long)rz[3 * p + 2]);
            return status != KCT_OK ? status : KCT_ERR_HIP;
        }
    if (passes > (1u << 20)) { set_err("a rank asked for %llu passes", (unsigned long long)passes); return KCT_ERR_ARG; }
    for (unsigned p = 0; p < world; ++p)
        if (rz[3 * p] != rz[0]) { set_err("ranks disagree on the number of super-k-mer streams (%llu on rank 0, %llu on rank %u): the early route needs GPUs with the same CU count", (unsigned long long)rz[0], (unsigned long long)rz[3 * p], p); return KCT_ERR_ARG; }
    const u64 step = windows ? (((windows + passes - 1) / passes) + 0xFFFF) & ~0xFFFFULL : 0;  // window starts per pass (16-byte aligned cuts)
    auto pass_range = [&](u64 p, u64 *off, u64 *len) {
        *off = std::min(p * step, nbytes);
        *len = p + 1 < passes ? std::min(nbytes - *off, step + k - 1) : nbytes - *off;
        if (*off >= nbytes || *len < (u64)k) *len = 0;
    };
    Slab send[2];
    Inbox inbox;
    PassOut po[2];
    const unsigned nvals = 2 + (unsigned)r.nwg;
    std::vector<u64> msg((size_t)world * nvals), got;
    std::vector<u64> recv_off[2], recv_bytes[2], dirs[2];
    u64 recv_base[2] = {0, 0};
    for (int b = 0; b < 2; ++b) { po[b].part_off.assign(world, 0); po[b].part_bytes.assign(world, 0); po[b].dir.assign((size_t)r.nwg * world, 0); }
    u64 st_windows_out = 0, st_windows_in = 0, st_bytes_out = 0, st_bytes_in = 0, st_runs = 0, st_retries = 0, st_counts = 0;
    double split_ms = 0, owner_ms = 0;
    // From here on the ranks are inside a protocol of per-pass collectives: NO early return.  A local failure is kept in `status` -- the
    // rank stays in step, announces it in its next size message (or the final agree()) and every rank ends the call with an error; the
    // exchange's slabs are released on every way out.
    // the inbox: room for the whole call where HBM allows (what arrives is about what leaves), at most 30 GiB (the directory's offsets)
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { set_err("hipMemGetInfo failed inside the early route"); status = KCT_ERR_HIP; free_b = 0; }
    const u64 inbox_target = std::min<u64>({(u64)((double)windows * (bases_per_window(k) / 4.0 + 0.125) * (solo ? 1.5 / world : 1.25)) + (8ULL << 20),
                                            (u64)(0.3 * (double)free_b), 30ULL << 30});
    const u64 batches = ops && world <= 2 ? 3 : ops && world <= 4 ? 2 : 1;
    const u64 batch_windows = std::max<u64>(1, windows / batches);   // (what arrives is about what this rank cuts)
    u64 counted_upto = 0;                                            // passes already counted
    // count what the inbox holds (a local failure is kept in `status`, see cut)
    auto count_inbox = [&](u64 more, bool keep_room = false) {
        const double t0 = now_ms();
        u64 n = 0;
        if (status == KCT_OK && r.fault(KCT_FAULT_COUNT)) { set_err("injected fault: the owner-side count"); status = KCT_ERR_HIP; }
        if (status == KCT_OK) { status = inbox_count(r, inbox, more, &n, keep_room); ++st_counts; }
        else { inbox.streams.clear(); inbox.groups = inbox.windows = 0; if (!keep_room) inbox.used = Inbox::kFront; }
        owner_ms += now_ms() - t0;
        *n_out += n;
    };
    // cut(p): this rank's records of pass p -> one part per owner in send[p & 1].  A local failure is kept in `status`: the rank then
    // stays in step with its peers (every collective below is still made) and announces the failure in its next size message.
    auto cut = [&](u64 p) {
        const int b = (int)(p & 1);
        u64 off, len;
        pass_range(p, &off, &len);
        const double t0 = now_ms();
        if (status == KCT_OK && r.fault(KCT_FAULT_SPLIT, p)) { set_err("injected fault: the split of pass %llu", (unsigned long long)p); status = KCT_ERR_HIP; }
        if (status == KCT_OK) status = split_pass(r, d_stream + off, len, send[b], b ? t->d_sk_recv : t->d_sk_send, po[b]);
        split_ms += now_ms() - t0;
        if (status != KCT_OK) { po[b].part_off.assign(world, 0); po[b].part_bytes.assign(world, 0); po[b].dir.assign((size_t)r.nwg * world, 0); }
    };
    // post(p): sizes and directories of pass p to every peer (with this rank's status), room in the inbox for what will arrive -- what
    // it holds is counted first if it has to be -- then the payload starts.  Returns non-OK on EVERY rank alike (the job is over), OK
    // with the transfer under way otherwise.
    auto post = [&](u64 p) -> kct_status {
        const int b = (int)(p & 1);
        for (unsigned q = 0; q < world; ++q) {
            msg[(size_t)q * nvals] = (u64)status;
            msg[(size_t)q * nvals + 1] = po[b].part_bytes[q];
            for (int w = 0; w < r.nwg; ++w) msg[(size_t)q * nvals + 2 + w] = po[b].dir[(size_t)q * r.nwg + w];
        }
        kct_status st = r.sizes(msg, nvals, got);
        recv_off[b].assign(world, 0); recv_bytes[b].assign(world, 0); dirs[b].assign((size_t)world * r.nwg, 0);
        u64 total = 0;
        if (st == KCT_OK)
            for (unsigned q = 0; q < world; ++q) {
                if (solo && q != rank) continue;   // (no exchange: only what this rank cut for itself arrives)
                if (got[(size_t)q * nvals] != 0 && status == KCT_OK && st == KCT_OK) {
                    set_err("the early route failed on rank %u (status %llu) before pass %llu was exchanged", q, (unsigned long long)got[(size_t)q * nvals], (unsigned long long)p);
                    st = KCT_ERR_HIP;
                }
                const unsigned from = solo ? rank : q;   // (the loop-back's "received" message is the one this rank wrote for that peer)
                recv_off[b][q] = total; recv_bytes[b][q] = got[(size_t)from * nvals + 1];
                total += recv_bytes[b][q];
                for (int w = 0; w < r.nwg; ++w) dirs[b][(size_t)q * r.nwg + w] = got[(size_t)from * nvals + 2 + w];
            }
        if (status != KCT_OK) st = status;
        if (st == KCT_OK && inbox.used + total + 64 > inbox.buf.cap) {
            if (inbox.groups) count_inbox((passes - p) * (inbox.windows / std::max<u64>(1, p - counted_upto))), counted_upto = p;   // (every earlier pass has been waited for)
            inbox.used = Inbox::kFront;   // (nothing is in flight: the whole buffer is free again)
            if (status == KCT_OK && Inbox::kFront + total + 64 > inbox.buf.cap) status = r.slab(inbox.buf, t->d_sk_inbox, std::max(total, inbox_target) + Inbox::kFront);
            st = status;
        }
        recv_base[b] = inbox.used;
        st = r.agree(st);   // (a rank without room for what it is to receive must not leave its peers inside the collective)
        if (st != KCT_OK) return st;
        inbox.used += total;
        char *dst = (char *)inbox.buf.p + recv_base[b];
        if (!ops) {   // loop-back: what this rank cut for itself is what it receives (no peers: a failure simply ends the job)
            for (unsigned q = 0; q < world; ++q)
                if (recv_bytes[b][q] && hipMemcpyAsync(dst + recv_off[b][q], (const char *)send[b].p + po[b].part_off[solo ? rank : q], recv_bytes[b][q], hipMemcpyDeviceToDevice, t->stream) != hipSuccess) {
                    set_err("the loop-back copy of pass %llu failed", (unsigned long long)p);
                    status = KCT_ERR_HIP;
                    return status;
                }
            return KCT_OK;
        }
        // A start() that reports a failure on THIS rank only: the rank does not walk away -- it stays in step, announces the failure in the
        // next size message or the final agree(), and the call then ends on every rank.  That holds for a start() which has made its
        // part of the collective before it reports (a failed copy behind it, a bookkeeping error: what the injected fault stands for); an
        // implementation that fails BEFORE joining a blocking collective leaves its peers inside that collective, which no protocol
        // above it can undo -- such a failure is fatal to the communicator (csrc/kct_rccl.cpp aborts it) and the launcher's hang guard
        // (bench.py --job-timeout) is what ends the peers.
        int start_rc = ops->start(ops->user, send[b].p, po[b].part_off.data(), po[b].part_bytes.data(), dst, recv_off[b].data(), recv_bytes[b].data());
        if (start_rc == 0 && r.fault(KCT_FAULT_START, p)) start_rc = -1;
        if (start_rc != 0) {
            set_err("the exchange failed to start pass %llu", (unsigned long long)p);
            if (status == KCT_OK) status = KCT_ERR_HIP;
        }
        return KCT_OK;
    };
    cut(0);
    kct_status job = post(0);   // the JOB's state: changes on every rank in the same collective
    for (u64 p = 0; p < passes && job == KCT_OK; ++p) {
        const int b = (int)(p & 1);
        // pass p + 1 is cut while pass p is on the wire, and goes on the wire as soon as pass p has arrived
        if (p + 1 < passes) cut(p + 1);
        if (ops) {
            const double t0 = now_ms();
            int wait_rc = ops->wait(ops->user);
            if (wait_rc == 0 && r.fault(KCT_FAULT_WAIT, p)) wait_rc = -1;
            if (wait_rc != 0 && status == KCT_OK) { set_err("the exchange of pass %llu failed", (unsigned long long)p); status = KCT_ERR_HIP; }
            r.wait_ms += now_ms() - t0;
        }
        if (status == KCT_OK) status = inbox_add(r, inbox, recv_base[b], recv_off[b], recv_bytes[b], dirs[b]);
        for (unsigned q = 0; q < world; ++q) {   // what crossed to / from OTHER ranks
            if (q == rank) continue;
            st_bytes_out += po[b].part_bytes[q]; st_bytes_in += solo ? 0 : recv_bytes[b][q];
            for (int w = 0; w < r.nwg; ++w) { st_windows_out += po[b].dir[(size_t)q * r.nwg + w] & 0xFFFFFFFFULL; st_windows_in += solo ? 0 : dirs[b][(size_t)q * r.nwg + w] & 0xFFFFFFFFULL; }
        }
        st_runs += po[b].runs; st_retries += po[b].retries;
        if (p + 1 < passes) job = post(p + 1);
        // Few GPUs = few links in use = a long transfer (xGMI is point to point): the owner then counts what has arrived in two or
        // three batches while the later passes are on the wire, instead of once at the end (every batch is a pass of the table's
        // bulk path: K2 and the conversion visit every block once more -- with 8 GPUs the wire is short and one batch is cheaper).
        if (job == KCT_OK && batches > 1 && p + 1 < passes && inbox.windows >= batch_windows) count_inbox((passes - 1 - p) * (inbox.windows / (p + 1 - counted_upto)), true), counted_upto = p + 1;
    }
    if (job == KCT_OK) count_inbox(0);
    if (status == KCT_OK) status = job;
    if (hipStreamSynchronize(t->stream) != hipSuccess && status == KCT_OK) {   // (an asynchronous kernel fault surfaces here)
        set_err("the table's stream failed at the end of the early route: %s", hipGetErrorString(hipGetLastError()));
        status = KCT_ERR_HIP;
    }
    if (world > 1) status = r.agree(status);   // (the last owner pass may have failed somewhere)
    if (ops && ops->release) {
        for (int b = 0; b < 2; ++b)
            if (send[b].p) ops->release(ops->user, send[b].p);
        if (inbox.buf.p) ops->release(ops->user, inbox.buf.p);
    }
    if (stats) {
        stats[0] = st_windows_out; stats[1] = st_windows_in; stats[2] = st_bytes_out; stats[3] = st_bytes_in; stats[4] = st_runs; stats[5] = passes;
        stats[6] = (u64)(split_ms * 1000.0); stats[7] = (u64)(r.wait_ms * 1000.0); stats[8] = (u64)(owner_ms * 1000.0); stats[9] = st_retries;
        stats[10] = windows; stats[11] = st_counts; stats[12] = 0; stats[13] = 0; stats[14] = 0; stats[15] = 0;
    }
    KCT_DBG(t, "routed call (rank %u of %u): %llu window starts in %llu passes; %llu runs; sent %llu B / %llu windows, received %llu B / %llu windows; split %.3f ms, waiting %.3f ms, owner %.3f ms\n",
            rank, world, (unsigned long long)windows, (unsigned long long)passes, (unsigned long long)st_runs, (unsigned long long)st_bytes_out, (unsigned long long)st_windows_out,
            (unsigned long long)st_bytes_in, (unsigned long long)st_windows_in, split_ms, r.wait_ms, owner_ms);
    return status;
}

}  // namespace kcth

using namespace kcth;

extern "C" kct_status kct_consume_device_routed(kct_table *t, const void *d_stream, size_t nbytes, uint64_t consumed_bytes, uint32_t world, uint32_t rank,
                                                const kct_exchange_ops *ops, uint64_t max_windows, uint64_t *n_owned, uint64_t *stats16) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if (!n_owned || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    u64 n = 0;
    const kct_status st = consume_routed(t, (const unsigned char *)d_stream, nbytes, world, rank, ops, max_windows, &n, stats16);
    *n_owned = n;
    if (st == KCT_OK) t->consumed += consumed_bytes;
    return st;
}

// The sender half alone (tests, tools): this rank's records cut into super-k-mers for `world` owners.  *d_parts (owned by the table, valid
// until its next bulk call) holds owner o's part at part_off[o] .. + part_bytes[o]: the base units of its streams, then their start
// units; dir[o * streams + s] = windows | base units << 32 of stream s (streams = kct_superkmer_streams(t)).
extern "C" kct_status kct_superkmer_split_device(kct_table *t, const void *d_stream, size_t nbytes, uint32_t world, const void **d_parts, uint64_t *part_off,
                                                 uint64_t *part_bytes, uint64_t *dir) {
    KCT_BORROW(t);
    KCT_TRY(use_consume(t));
    if (!d_parts || !part_off || !part_bytes || !dir || (!d_stream && nbytes)) { set_err("null argument"); return KCT_ERR_ARG; }
    if (world < 1 || world > kct::kSkMaxWorld || t->k > 64) { set_err("bad world, or k > 64"); return KCT_ERR_ARG; }
    if (((uintptr_t)d_stream & 15) != 0) { set_err("d_stream must be 16-byte aligned"); return KCT_ERR_ARG; }
    Route r{t, world, 0, nullptr, split_streams(t)};
    Slab send;
    PassOut po;
    KCT_TRY(split_pass(r, (const unsigned char *)d_stream, nbytes, send, t->d_sk_send, po));
    *d_parts = send.p;
    for (unsigned o = 0; o < world; ++o) { part_off[o] = po.part_off[o]; part_bytes[o] = po.part_bytes[o]; }
    for (size_t i = 0; i < po.dir.size(); ++i) dir[i] = po.dir[i];
    return KCT_OK;
}

extern "C" uint32_t kct_superkmer_streams(const kct_table *t) { return t ? (uint32_t)split_streams(t) : 0; }

extern "C" kct_status kct_debug_inject_fault(kct_table *t, int point, uint64_t pass) {
    KCT_BORROW(t);
    if (point < KCT_FAULT_NONE || point > KCT_FAULT_COUNT) { set_err("unknown fault point %d", point); return KCT_ERR_ARG; }
    t->fault_point = point; t->fault_pass = pass;
    return KCT_OK;
}
