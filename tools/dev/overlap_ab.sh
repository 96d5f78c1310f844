#!/bin/bash
# two-stream adjoint schedule on / off, twice each (same box, back to back)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for ov in 1 0 1 0; do
  echo "## GODE_OVERLAP=$ov"
  GODE_OVERLAP=$ov python $R/bench.py --no-configs --no-cpu-baseline --no-secondary --steps 4 2>/dev/null
done | python $R/tools/dev/overlap_ab.py
