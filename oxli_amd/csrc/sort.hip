// sort.hip -- device radix sort of (hash, count) pairs for dump(sortkeys / sortcounts) (lib.rs:330-381), and
// sort + run-length encoding of the counts for histo (lib.rs:464-488).
// rocPRIM's radix sort is a plain library primitive; it lives in its own translation unit so that
// the other sources do not pay its compile time.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

// Sorts n pairs by key (ascending, stable).  Call with tmp == nullptr to get the temporary size.
extern "C" __attribute__((visibility("hidden"))) int kx_sort_pairs_u64(const unsigned long long *keys_in, unsigned long long *keys_out,
                                                                      const unsigned long long *vals_in, unsigned long long *vals_out,
                                                                      size_t n, void *tmp, size_t *tmp_bytes, void *stream) {
    return (int)rocprim::radix_sort_pairs(tmp, *tmp_bytes, keys_in, keys_out, vals_in, vals_out, n, 0, 64, (hipStream_t)stream);
}

// Sorts n keys ascending.  Call with tmp == nullptr to get the temporary size.
extern "C" __attribute__((visibility("hidden"))) int kx_sort_keys_u64(const unsigned long long *keys_in, unsigned long long *keys_out, size_t n,
                                                                     void *tmp, size_t *tmp_bytes, void *stream) {
    return (int)rocprim::radix_sort_keys(tmp, *tmp_bytes, keys_in, keys_out, n, 0, 64, (hipStream_t)stream);
}

// Run-length encodes n sorted keys: unique_out[i] occurs runs_out[i] times, *nruns_out runs in all (device memory).
extern "C" __attribute__((visibility("hidden"))) int kx_rle_u64(const unsigned long long *keys_in, size_t n, unsigned long long *unique_out,
                                                               unsigned long long *runs_out, unsigned long long *nruns_out, void *tmp,
                                                               size_t *tmp_bytes, void *stream) {
    return (int)rocprim::run_length_encode(tmp, *tmp_bytes, keys_in, (unsigned int)n, unique_out, runs_out, nruns_out, (hipStream_t)stream);
}
