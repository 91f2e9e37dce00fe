// policy_harness.cpp -- drives oxli_amd/csrc/path_policy.h (the pure host logic that chooses a pass's device path and sizes the
// partitioned paths) with FAKE table geometries, on the CPU: built by tests/test_policy_cpu.py with g++, no HIP anywhere.
#include "../oxli_amd/csrc/path_policy.h"

struct FakeTune { int k1b_lines = 0; };
struct FakeTable {   // the members path_policy.h reads, with kct_table's names
    int k = 21;
    uint64_t cap = 1 << 16;
    int block_bits = 13;
    int force_path = 0;
    bool dedupe_off = false, compact_off = false, dedupe_hint = false, auto_sized = false, shadow_dirty = false, s32_dirty = false, dedupe128_off = false;
    uint64_t n_keys = 0, shadow_keys = 0, s32_keys = 0, s128_keys = 0, windows_since_read = 0, call_windows_left = 0;
    FakeTune tune;
};

extern "C" {

struct PolicyIn {
    int k, block_bits, force_path, dedupe_off, compact_off, dedupe_hint, auto_sized, shadow_dirty, s32_dirty;
    uint64_t cap, n_keys, shadow_keys, s32_keys, windows_since_read, call_windows_left;
};

static FakeTable make(const PolicyIn *in) {
    FakeTable t;
    t.k = in->k; t.cap = in->cap; t.block_bits = in->block_bits; t.force_path = in->force_path;
    t.dedupe_off = in->dedupe_off; t.compact_off = in->compact_off; t.dedupe_hint = in->dedupe_hint; t.auto_sized = in->auto_sized;
    t.shadow_dirty = in->shadow_dirty; t.s32_dirty = in->s32_dirty;
    t.n_keys = in->n_keys; t.shadow_keys = in->shadow_keys; t.s32_keys = in->s32_keys;
    t.windows_since_read = in->windows_since_read; t.call_windows_left = in->call_windows_left;
    return t;
}

int policy_choose_path(const PolicyIn *in, uint64_t npos) { const FakeTable t = make(in); return (int)kcth::choose_path(&t, npos); }
int policy_probe_wanted(const PolicyIn *in, uint64_t call_windows) { const FakeTable t = make(in); return kcth::probe_wanted(&t, call_windows) ? 1 : 0; }
int policy_mostly_new_expected(const PolicyIn *in, uint64_t call_windows) { const FakeTable t = make(in); return kcth::mostly_new_expected(&t, call_windows) ? 1 : 0; }
int policy_probe_verdict(const PolicyIn *in, double per_key, uint64_t call_windows) { const FakeTable t = make(in); return kcth::probe_verdict(&t, per_key, call_windows) ? 1 : 0; }
int policy_compact_sbits(const PolicyIn *in) { const FakeTable t = make(in); return kcth::compact_sbits_for(&t); }
double policy_draws_per_distinct(double r) { return kcth::draws_per_distinct(r); }
double policy_per_distinct_two_depths(double n1, double f1, double n2, double f2, double N) { return kcth::per_distinct_two_depths(n1, f1, n2, f2, N); }
unsigned policy_region_capacity(double avg) { return kcth::region_capacity(avg); }
unsigned policy_overflow_capacity(uint64_t per_wg) { return kcth::overflow_capacity(per_wg); }
void policy_levels(int bbits, int nwg, int out[6]) {
    const kcth::Levels L = kcth::levels_for(bbits, nwg);
    out[0] = L.pbits; out[1] = L.sub_bits; out[2] = L.two ? 1 : 0; out[3] = (int)L.P; out[4] = (int)L.W; out[5] = (int)(L.B >> 10);
}
unsigned policy_min_lines(int ring_entries, int sub_bits, int entry_bytes) { FakeTable t; return kcth::repartition_min_lines(&t, ring_entries, sub_bits, entry_bytes); }

}  // extern "C"
