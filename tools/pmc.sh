#!/bin/bash
# PMC passes over tools/kbench.py (separate passes per counter group, as the microarch guide prescribes).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [kbench args]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python $R/tools/kbench.py "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fh:
    for k, cs in sorted(agg.items()):
        if not any(x in k for x in ("spmm", "gemm", "wgrad", "lincomb", "colsum", "reduce", "ratio")):
            continue
        fh.write(k + "\n")
        for c, v in sorted(cs.items()):
            fh.write("    %-32s n=%4d mean=%.4g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
