// kct_runs.hip -- the kernels of the multi-GPU early route (kct_route.hip drives them): the sender's super-k-mer split, the owner's run
// directory, and K1's RUNS instantiations (the same partition_windows_kernel reading received super-k-mers instead of a record stream).
// A translation unit of its own so that it compiles beside kct_consume.hip, which instantiates K1 for ordinary input.
#include "kct_internal.h"

#include "k1_kernel.h"
#include "superkmer_kernels.h"

namespace kcth {

namespace {

template <int KW, int KC, int MODE>
void k1_runs(kct_table *t, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    hipLaunchKernelGGL((kct::partition_windows_kernel<KW, KC, MODE, true>), dim3(t->num_cus), dim3(kct::kPartThreads), 0, t->stream, (const unsigned char *)nullptr,
                       chunk_bytes, (int)t->k, ntiles, pa);
}

template <int K>
struct SplitByK {
    static void run(int k, hipStream_t s, int grid, const unsigned char *stream, u64 nbytes, u64 ntiles, const kct::SplitArgs &sa) {
        if (k == K) hipLaunchKernelGGL(kct::split_superkmers_kernel<K>, dim3(grid), dim3(kct::kSkThreads), 0, s, stream, nbytes, ntiles, sa);
        else SplitByK<K - 1>::run(k, s, grid, stream, nbytes, ntiles, sa);
    }
};
template <>
struct SplitByK<0> {
    static void run(int, hipStream_t, int, const unsigned char *, u64, u64, const kct::SplitArgs &) {}
};

}  // namespace

// K1 over super-k-mers.  The popular k get their own instantiation (K1 is instruction-issue bound: a compile-time k is worth ~10 %),
// every other k <= 64 the run-time-k one.
void launch_partition_runs(kct_table *t, int mode, u64 chunk_bytes, u64 ntiles, const kct::PartitionArgs &pa) {
    const int k = t->k;
    ProfScope ps(t, mode == 2 ? "partition_windows_kernel<compact, runs>" : mode == 1 ? "partition_windows_kernel<raw, runs>" : "partition_windows_kernel<runs>");
    if (mode == 2) {
        if (k == 21) k1_runs<1, 21, 2>(t, chunk_bytes, ntiles, pa);
        else k1_runs<1, 0, 2>(t, chunk_bytes, ntiles, pa);
    } else if (mode == 1) {
        if (k == 21) k1_runs<1, 21, 1>(t, chunk_bytes, ntiles, pa);
        else if (k == 31) k1_runs<1, 31, 1>(t, chunk_bytes, ntiles, pa);
        else k1_runs<1, 0, 1>(t, chunk_bytes, ntiles, pa);
    } else {
        if (k == 21) k1_runs<1, 21, 0>(t, chunk_bytes, ntiles, pa);
        else if (k == 31) k1_runs<1, 31, 0>(t, chunk_bytes, ntiles, pa);
        else if (k == 51) k1_runs<2, 51, 0>(t, chunk_bytes, ntiles, pa);
        else if (k <= 32) k1_runs<1, 0, 0>(t, chunk_bytes, ntiles, pa);
        else k1_runs<2, 0, 0>(t, chunk_bytes, ntiles, pa);
    }
}

void launch_split(kct_table *t, const unsigned char *d_stream, u64 nbytes, u64 ntiles, const kct::SplitArgs &sa) {
    ProfScope ps(t, "split_superkmers_kernel");
    SplitByK<64>::run(t->k, t->stream, split_streams(t), d_stream, nbytes, ntiles, sa);
}

void launch_gather_units(kct_table *t, const void *src, const du64 *src_off, const du64 *dst_off, const unsigned int *n, unsigned int count, void *dst) {
    if (!count) return;
    ProfScope ps(t, "gather_units_kernel");
    hipLaunchKernelGGL(kct::gather_units_kernel, dim3(std::min(count, 16384u)), dim3(kct::kBlock), 0, t->stream, (const uint4 *)src, src_off, dst_off, n, count, (uint4 *)dst);
}

void launch_run_directory(kct_table *t, const kct::RunStream *streams, unsigned int nstreams, const du64 *starts, kct::RunGroup *groups) {
    if (!nstreams) return;
    ProfScope ps(t, "run_directory_kernel");
    hipLaunchKernelGGL(kct::run_directory_kernel, dim3(std::min(nstreams, 4096u)), dim3(kct::kBlock), 0, t->stream, streams, nstreams, starts, (int)t->k, groups);
}

void launch_expand_runs(kct_table *t, const kct::RunsInput &in, u64 ngroups, unsigned char *out) {
    if (!ngroups) return;
    ProfScope ps(t, "expand_runs_kernel");
    hipLaunchKernelGGL(kct::expand_runs_kernel, dim3((unsigned)std::min<u64>((ngroups + 3) / 4, 1u << 16)), dim3(kct::kBlock), 0, t->stream, in, ngroups, (int)t->k, out);
}

}  // namespace kcth
