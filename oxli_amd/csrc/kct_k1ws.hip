// kct_k1ws.hip -- the wave-specialised K1 (k1ws_kernel.h) and its launcher.  A translation unit of its own: it compiles beside
// kct_consume.hip, which instantiates the barrier-synchronised K1 (k1_kernel.h) that remains for k > 64, super-k-mer input and the k
// this file has no instantiation for.
#include "kct_internal.h"

#include "k1ws_kernel.h"

namespace kcth {

namespace {

template <int KW, int KC, int MODE>
void k1ws(kct_table *t, int flushers, const unsigned char *d_stream, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    const dim3 grid(t->num_cus), block(kct::kPartThreads);
    if (flushers == 2) hipLaunchKernelGGL((kct::partition_windows_ws_kernel<KW, KC, MODE, 2>), grid, block, 0, t->stream, d_stream, chunk_bytes, (int)t->k, ntiles, pa);
    else hipLaunchKernelGGL((kct::partition_windows_ws_kernel<KW, KC, MODE, 4>), grid, block, 0, t->stream, d_stream, chunk_bytes, (int)t->k, ntiles, pa);
}

}  // namespace

// false: no instantiation for this (mode, k) -- the caller launches partition_windows_kernel
bool launch_partition_ws(kct_table *t, int mode, const unsigned char *d_stream, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    const int k = t->k, f = t->tune.k1_flushers;
    if (f <= 0 || k > 64 || pa.runs.groups || pa.region_cap >= (1u << 20)) return false;   // (a flusher's list entry holds 20 bits of region position)
    // Only the compact mode (u32 entries): a bin's stretch of the ring is two 64-byte lines whatever the entry size, and with 8-byte
    // entries that is 16 appends -- while a complete line waits for its flusher (thousands of cycles: the flushers share the SIMDs
    // with the hashing waves) only 8 more fit, against 16 with 4-byte entries: measured, the 64-bit modes overflow their rings until
    // the pass is abandoned (tests/test_gpu_robustness.py [25-partitioned-...] with KCT_K1_FLUSHERS=4).  The instantiations of the
    // other modes stay compiled (k1ws<..., 0 / 1>) for measurements: KCT_K1_FLUSHERS_ALL_MODES=1.
    static const bool all_modes = getenv("KCT_K1_FLUSHERS_ALL_MODES") != nullptr;
    if (mode != 2 && !all_modes) return false;
    if (mode == 2) {
        if (k == 21) k1ws<1, 21, 2>(t, f, d_stream, chunk_bytes, ntiles, pa);
        else if (k <= 21) k1ws<1, 0, 2>(t, f, d_stream, chunk_bytes, ntiles, pa);
        else return false;
    } else if (mode == 1) {
        if (k == 31) k1ws<1, 31, 1>(t, f, d_stream, chunk_bytes, ntiles, pa);
        else return false;
    } else if (mode == 0) {
        if (k == 21) k1ws<1, 21, 0>(t, f, d_stream, chunk_bytes, ntiles, pa);
        else if (k == 51) k1ws<2, 51, 0>(t, f, d_stream, chunk_bytes, ntiles, pa);
        else return false;
    } else return false;
    return true;
}

}  // namespace kcth
