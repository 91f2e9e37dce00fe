// kct_analytics.hip -- table analytics on the resident table: count statistics and
// histogram, cuts and removal, table-against-table comparison and set operations.  Everything that touches
// a hash or a count runs on the device; the host only folds in key 0 (which the library keeps beside the
// device table, 0 being its EMPTY sentinel) and moves results.
#include "kct_internal.h"
#include "analytics_kernels.h"

extern "C" int kx_sort_keys_u64(const unsigned long long *keys_in, unsigned long long *keys_out, size_t n, void *tmp, size_t *tmp_bytes,
                                void *stream);  // sort.hip
extern "C" int kx_rle_u64(const unsigned long long *keys_in, size_t n, unsigned long long *unique_out, unsigned long long *runs_out,
                          unsigned long long *nruns_out, void *tmp, size_t *tmp_bytes, void *stream);

using namespace kcth;

namespace {

// device words that hold live keys, or nullptr when the table is (lazily) empty
const du64 *live_words(const kct_table *t) { return (t->n_keys && !t->lazy_empty) ? (const du64 *)t->slots : nullptr; }

kct_status same_device(const kct_table *a, const kct_table *b) {
    if (a->device != b->device) { set_err("the two tables live on different devices (%d and %d)", a->device, b->device); return KCT_ERR_ARG; }
    return KCT_OK;
}

// Rebuilds t from the `m` interleaved pairs in d_pairs (device): the probe layout has no tombstones, so removal
// re-inserts the survivors into the cleared slot array.  consumed and key 0 are left to the caller.
kct_status rebuild_from_pairs(kct_table *t, const du64 *d_pairs, u64 m) {
    HIP_TRY(hipMemsetAsync(t->slots, 0, t->cap * 16, t->stream));
    t->lazy_empty = false;
    t->n_keys = 0;
    u64 tl[4] = {0, 0, 0, 0};
    if (m) KCT_TRY(merge_pairs(t, d_pairs, d_pairs + 1, m, 2, tl));
    HIP_TRY(hipStreamSynchronize(t->stream));
    return KCT_OK;
}

// keep lo <= count <= hi and hash != drop (has_drop); *removed = keys that went away
kct_status retain(kct_table *t, u64 lo, u64 hi, bool has_drop, u64 drop, u64 *removed) {
    u64 gone = 0;
    if (t->zero_present && (t->zero_count < lo || t->zero_count > hi || (has_drop && drop == 0))) {
        t->zero_present = false; t->zero_count = 0;
        ++gone;
    }
    if (live_words(t) && !(has_drop && drop == 0 && lo == 0 && hi == ~0ULL)) {
        KCT_TRY(t->d_aux.reserve(t->n_keys * 16));
        du64 *d_n = t->d_counters + kNumCounters + 4;
        HIP_TRY(hipMemsetAsync(d_n, 0, 8, t->stream));
        {
            ProfScope ps(t, "compact_filtered_kernel");
            hipLaunchKernelGGL(kct::compact_filtered_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots,
                               geom(t), lo, hi, has_drop ? drop : 0ULL, (du64 *)t->d_aux.p, t->n_keys, d_n);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_n, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        const u64 m = t->h_counters[0];
        if (m > t->n_keys) { set_err("table scan found %llu keys, expected at most %llu", (unsigned long long)m, (unsigned long long)t->n_keys); return KCT_ERR_HIP; }
        if (m < t->n_keys) {
            gone += t->n_keys - m;
            KCT_TRY(rebuild_from_pairs(t, (const du64 *)t->d_aux.p, m));
        }
    }
    if (removed) *removed = gone;
    return KCT_OK;
}

}  // namespace

extern "C" {

kct_status kct_count_stats(kct_table *t, uint64_t *min_out, uint64_t *max_out, double *sum_squares_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 lo = ~0ULL, hi = 0;
    double sq = 0.0;
    if (live_words(t)) {
        du64 *d_mm = t->d_counters + kNumCounters + 4;  // [0] min, [1] max, [2] sum of squares (f64 bits)
        u64 init[3] = {~0ULL, 0ULL, 0ULL};
        HIP_TRY(hipMemcpyAsync(d_mm, init, sizeof init, hipMemcpyHostToDevice, t->stream));
        {
            ProfScope ps(t, "count_stats_kernel");
            hipLaunchKernelGGL(kct::count_stats_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t),
                               d_mm, (double *)(d_mm + 2));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_mm, 24, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        lo = t->h_counters[0]; hi = t->h_counters[1];
        memcpy(&sq, &t->h_counters[2], 8);
    }
    if (t->zero_present) {
        lo = std::min<u64>(lo, t->zero_count); hi = std::max<u64>(hi, t->zero_count);
        sq += (double)t->zero_count * (double)t->zero_count;
    }
    if (lo == ~0ULL && !t->zero_present && !live_words(t)) lo = 0;  // empty table: lib.rs:494-496, 507-509
    if (min_out) *min_out = lo;
    if (max_out) *max_out = hi;
    if (sum_squares_out) *sum_squares_out = sq;
    return KCT_OK;
}

kct_status kct_digest(kct_table *t, uint64_t *sum_hc_out, uint64_t *xor_hc_out, uint64_t *sum_sq_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 d[3] = {0, 0, 0};
    if (live_words(t)) {
        du64 *d_d = t->d_counters + kNumCounters + 4;
        HIP_TRY(hipMemsetAsync(d_d, 0, 24, t->stream));
        {
            ProfScope ps(t, "digest_kernel");
            hipLaunchKernelGGL(kct::digest_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t), d_d);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_d, 24, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        for (int i = 0; i < 3; ++i) d[i] = t->h_counters[i];
    }
    if (t->zero_present) d[2] += t->zero_count * t->zero_count;  // (key 0 adds nothing to hash * count)
    if (sum_hc_out) *sum_hc_out = d[0];
    if (xor_hc_out) *xor_hc_out = d[1];
    if (sum_sq_out) *sum_sq_out = d[2];
    return KCT_OK;
}

kct_status kct_histogram(kct_table *t, uint64_t *values_out, uint64_t *freq_out, size_t cap, uint64_t *n_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    if (!n_out || (cap && (!values_out || !freq_out))) { set_err("null argument"); return KCT_ERR_ARG; }
    std::vector<u64> vals, freq;
    if (live_words(t)) {
        const u64 n = t->n_keys;
        if (n >= (1ULL << 32)) { set_err("histogram of more than 2^32 keys is not supported"); return KCT_ERR_ARG; }
        // counts -> sorted counts -> (value, run length): compaction + rocPRIM radix sort + run-length encode
        KCT_TRY(t->d_aux.reserve(n * 16));
        KCT_TRY(t->d_aux2.reserve(n * 16));
        du64 *raw = (du64 *)t->d_aux.p, *sorted = (du64 *)t->d_aux2.p;
        du64 *d_n = t->d_counters + kNumCounters + 4;
        HIP_TRY(hipMemsetAsync(d_n, 0, 8, t->stream));
        {
            ProfScope ps(t, "compact_counts_kernel");
            hipLaunchKernelGGL(kct::compact_counts_kernel, dim3(merge_grid(t->cap)), dim3(kct::kBlock), 0, t->stream, (const du64 *)t->slots, geom(t),
                               raw, n, d_n);
        }
        HIP_TRY(hipGetLastError());
        size_t tmp_sort = 0, tmp_rle = 0;
        if (kx_sort_keys_u64(raw, sorted, n, nullptr, &tmp_sort, t->stream) != 0) { set_err("rocprim size query failed"); return KCT_ERR_HIP; }
        du64 *uniq = raw, *runs = raw + n;  // the unsorted counts are dead once sorted
        if (kx_rle_u64(sorted, n, uniq, runs, d_n + 1, nullptr, &tmp_rle, t->stream) != 0) { set_err("rocprim size query failed"); return KCT_ERR_HIP; }
        KCT_TRY(t->d_sort.reserve(std::max(tmp_sort, tmp_rle) + 16));
        {
            ProfScope ps(t, "radix_sort_keys(counts)");
            if (kx_sort_keys_u64(raw, sorted, n, t->d_sort.p, &tmp_sort, t->stream) != 0) { set_err("rocprim radix sort failed"); return KCT_ERR_HIP; }
        }
        {
            ProfScope ps(t, "run_length_encode(counts)");
            if (kx_rle_u64(sorted, n, uniq, runs, d_n + 1, t->d_sort.p, &tmp_rle, t->stream) != 0) { set_err("rocprim run-length encode failed"); return KCT_ERR_HIP; }
        }
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_n, 16, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->h_counters[0] != n) { set_err("table scan found %llu keys, expected %llu", (unsigned long long)t->h_counters[0], (unsigned long long)n); return KCT_ERR_HIP; }
        const u64 nruns = t->h_counters[1];
        vals.resize(nruns); freq.resize(nruns);
        if (nruns) {
            HIP_TRY(hipMemcpy(vals.data(), uniq, nruns * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(freq.data(), runs, nruns * 8, hipMemcpyDeviceToHost));
        }
    }
    if (t->zero_present) {  // key 0's count joins the (value-sorted) histogram
        const size_t at = (size_t)(std::lower_bound(vals.begin(), vals.end(), t->zero_count) - vals.begin());
        if (at < vals.size() && vals[at] == t->zero_count) ++freq[at];
        else { vals.insert(vals.begin() + at, t->zero_count); freq.insert(freq.begin() + at, 1); }
    }
    *n_out = vals.size();
    const size_t ncopy = std::min(cap, vals.size());
    if (ncopy) { memcpy(values_out, vals.data(), ncopy * 8); memcpy(freq_out, freq.data(), ncopy * 8); }
    return KCT_OK;
}

kct_status kct_retain_counts(kct_table *t, uint64_t min_count, uint64_t max_count, uint64_t *removed_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    return retain(t, min_count, max_count, false, 0, removed_out);
}

kct_status kct_remove_hash(kct_table *t, uint64_t hash, uint64_t *removed_out) {
    KCT_BORROW(t);
    KCT_TRY(use(t));
    u64 gone = 0;
    if (hash == 0) {  // key 0 lives host-side
        if (t->zero_present) { t->zero_present = false; t->zero_count = 0; gone = 1; }
    } else if (live_words(t)) {  // O(probe run), like the reference's HashMap::remove: no scan, no rebuild
        du64 *d_found = t->d_counters + kNumCounters + 4;
        {
            ProfScope ps(t, "remove_hash_kernel");
            hipLaunchKernelGGL(kct::remove_hash_kernel, dim3(1), dim3(1), 0, t->stream, t->slots, geom(t), (u64)hash, d_found);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(t->h_counters, d_found, 8, hipMemcpyDeviceToHost, t->stream));
        HIP_TRY(hipStreamSynchronize(t->stream));
        if (t->h_counters[0]) { gone = 1; t->n_keys -= 1; }
    }
    if (removed_out) *removed_out = gone;
    return KCT_OK;
}

kct_status kct_compare(kct_table *a, kct_table *b, uint64_t *common_out, uint64_t *dot_out) {
    KCT_BORROW(a);
    KCT_BORROW(b);
    KCT_TRY(use(b));
    KCT_TRY(use(a));
    KCT_TRY(same_device(a, b));
    u64 common = 0, dot = 0;
    if (live_words(a) && live_words(b)) {
        kct_table *x = a->cap <= b->cap ? a : b, *y = x == a ? b : a;  // scan the smaller slot array, look up in the other
        du64 *d_out = a->d_counters + kNumCounters + 4;
        HIP_TRY(hipMemsetAsync(d_out, 0, 16, a->stream));
        {
            ProfScope ps(a, "compare_tables_kernel");
            hipLaunchKernelGGL(kct::compare_tables_kernel, dim3(merge_grid(x->cap)), dim3(kct::kBlock), 0, a->stream, (const du64 *)x->slots, geom(x),
                               (const du64 *)y->slots, geom(y), d_out);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(a->h_counters, d_out, 16, hipMemcpyDeviceToHost, a->stream));
        HIP_TRY(hipStreamSynchronize(a->stream));
        common = a->h_counters[0]; dot = a->h_counters[1];
    }
    if (a->zero_present && b->zero_present) { ++common; dot += a->zero_count * b->zero_count; }
    if (common_out) *common_out = common;
    if (dot_out) *dot_out = dot;
    return KCT_OK;
}

kct_status kct_set_op(kct_table *a, kct_table *b, int op, uint64_t *hashes_out, size_t cap, uint64_t *n_out) {
    KCT_BORROW(a);
    KCT_BORROW(b);
    KCT_TRY(use(b));
    KCT_TRY(use(a));
    KCT_TRY(same_device(a, b));
    if (!n_out || (cap && !hashes_out) || op < 0 || op > 3) { set_err("bad argument"); return KCT_ERR_ARG; }
    // op 0 union = a + (b not in a); 1 intersection = a in b; 2 difference = a not in b; 3 symmetric = (a not in b) + (b not in a)
    const u64 worst = (op == 1 || op == 2) ? a->n_keys : a->n_keys + b->n_keys;
    u64 n_dev = 0;
    if (worst) {
        KCT_TRY(a->d_aux.reserve(worst * 8));
        du64 *d_n = a->d_counters + kNumCounters + 4;
        HIP_TRY(hipMemsetAsync(d_n, 0, 8, a->stream));
        auto select = [&](kct_table *x, kct_table *y, int want) -> kct_status {
            if (!live_words(x)) return KCT_OK;
            ProfScope ps(a, "select_keys_kernel");
            hipLaunchKernelGGL(kct::select_keys_kernel, dim3(merge_grid(x->cap)), dim3(kct::kBlock), 0, a->stream, (const du64 *)x->slots, geom(x),
                               live_words(y), geom(y), want, (du64 *)a->d_aux.p, worst, d_n);
            HIP_TRY(hipGetLastError());
            return KCT_OK;
        };
        if (op == 0) { KCT_TRY(select(a, b, 2)); KCT_TRY(select(b, a, 0)); }
        if (op == 1) KCT_TRY(select(a, b, 1));
        if (op == 2) KCT_TRY(select(a, b, 0));
        if (op == 3) { KCT_TRY(select(a, b, 0)); KCT_TRY(select(b, a, 0)); }
        HIP_TRY(hipMemcpyAsync(a->h_counters, d_n, 8, hipMemcpyDeviceToHost, a->stream));
        HIP_TRY(hipStreamSynchronize(a->stream));
        n_dev = a->h_counters[0];
        if (n_dev > worst) { set_err("set operation produced %llu keys, expected at most %llu", (unsigned long long)n_dev, (unsigned long long)worst); return KCT_ERR_HIP; }
    }
    const bool za = a->zero_present, zb = b->zero_present;
    const bool zero_in = op == 0 ? (za || zb) : op == 1 ? (za && zb) : op == 2 ? (za && !zb) : (za != zb);
    *n_out = n_dev + (zero_in ? 1 : 0);
    const size_t ncopy = std::min<size_t>(cap, n_dev);
    if (ncopy) {
        KCT_TRY(a->h_stage.reserve(ncopy * 8));
        HIP_TRY(hipMemcpyAsync(a->h_stage.p, a->d_aux.p, ncopy * 8, hipMemcpyDeviceToHost, a->stream));
        HIP_TRY(hipStreamSynchronize(a->stream));
        parallel_memcpy(hashes_out, a->h_stage.p, ncopy * 8);
    }
    if (zero_in && n_dev < cap) hashes_out[n_dev] = 0;
    return KCT_OK;
}

}  // extern "C"
