#!/bin/bash
# SQ counters of the dense kernels at the C5 size (tools/kbench.py --only gemm): where do the wave-cycles go?
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_dense
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout -k 10 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/$n -- python $R/tools/kbench.py --only gemm > $OUT/$n.log 2>&1
  echo "pmc [$grp] rc=$?"
done
python - "$OUT" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("gn_gemm_fwd_kernel<8, 4>", "gn_gemm_bwd_kernel<8, 4>", "gn_gemm_bwd_wgrad_kernel<8, 4>", "wgrad_kernel<8, 4>"):
            if key in k:                       # first match wins: "wgrad_kernel" is a substring of the fused kernel's name
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $OUT/SQ_*/
