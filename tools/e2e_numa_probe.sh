for i in 1 2 3 4 5 6; do KCT_DEBUG=1 python tools/e2e_diag.py child 2>&1 | grep "batch: source pages\|kmers_per_s" | tail -2 | cut -c1-400; done
