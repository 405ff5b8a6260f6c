#!/bin/bash
# r6_proj_prof.sh D... -- kernel times of the Standardized projections at n = 500k for the given widths (rocprofv3 --stats)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for D in "$@"; do
  rm -rf /tmp/pp_$D
  rocprofv3 --kernel-trace --stats -d /tmp/pp_$D --output-format csv -- python $R/tools/r6_proj_sweep.py $D > /dev/null 2>&1
  f=$(find /tmp/pp_$D -name "*kernel_stats.csv" | head -1)
  echo "== d = $D"
  python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_gram_rows", "k_rmul_rows", "k_colsum_rows", "k_sum_chunks", "invsqrt", "k_ns_", "k_center", "newton", "k_gram", "k_rmul")):
        print("  %-70s calls %5s  avg %9.1f us" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
P
done
