"""e2e_C2 (kct_consume_batch from host memory) with the whole process bound to NUMA node 0 / node 1 / unbound: does the bimodal rate follow the
socket the process runs on relative to the GPU's?   python tools/e2e_numa.py"""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
def cpus(s):
    out = set()
    for part in s.strip().split(","):
        a, _, b = part.partition("-")
        out |= set(range(int(a), int(b or a) + 1))
    return out
nodes = {int(p.split("node")[-1].split("/")[0]): cpus(open(p).read()) for p in glob.glob("/sys/devices/system/node/node*/cpulist")}
gpu_nodes = sorted({open(p).read().strip() for p in glob.glob("/sys/class/drm/card*/device/numa_node") if os.path.exists(p.replace("numa_node", "mem_info_vram_total"))})
print("nodes", {k: len(v) for k, v in nodes.items()}, "GPU numa_node candidates", gpu_nodes, flush=True)
code = "import os,sys,runpy; os.sched_setaffinity(0, {cp}); sys.argv=['bench.py','--configs','e2e_C2','--no-cpu-baseline','--no-second-process','--no-headline','--verbose']; runpy.run_path(os.path.join({root!r},'bench.py'), run_name='__main__')"
for rep in range(2):
    for name, cp in [("unbound", set().union(*nodes.values()))] + [(f"node{n}", c) for n, c in sorted(nodes.items())]:
        out = subprocess.run([sys.executable, "-c", code.format(cp=sorted(cp), root=ROOT)], capture_output=True, text=True, cwd=ROOT)
        try:
            d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["configs"]["e2e_C2"]["variants"]
            print(name, {k: (round(v["kmers_per_s"] / 1e10, 3), round(v["call_ms"], 2)) for k, v in d.items()}, flush=True)
        except Exception as e:
            print(name, "failed", e, out.stderr[-300:], flush=True)
