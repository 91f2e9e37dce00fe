"""Prints bench_detail.json's e2e_C2 entry in one line per variant (after `python bench.py --configs e2e_C2 ...`)."""
import json, sys
e = json.load(open(sys.argv[1] if len(sys.argv) > 1 else "bench_detail.json"))["configs"]["e2e_C2"]
print("e2e_C2 %.3g k-mers/s (best variant %.3g)" % (e["kmers_per_s"], e["kmers_per_s_best_variant"]))
for vn, v in e["variants"].items():
    tl = v["timeline"]
    print("  %-10s call %.2f ms (min/max %.2f / %.2f incl. sync), packers %.2f -> %.2f ms, busiest %.2f ms, %.1f GB/s per thread, passes submitted %.2f ms" % (
        vn, v["call_ms"], v["seconds_min_max"][0] * 1e3, v["seconds_min_max"][1] * 1e3, tl["first_packer_start_ms"], tl["last_packer_end_ms"], tl["thread_busy_ms_max"],
        v["packer_GB_per_s_per_thread"], tl["submitted_ms"]))
