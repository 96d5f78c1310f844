#!/bin/bash
# SQ / TCC counters of pgemm_kernel (the edge encoder's three products) and cut_kernel: separate passes per group, as the
# microarch guide prescribes; writes gpurun_out/profiles_r04/r04_pgemm_pmc.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
OUT=$R/gpurun_out/prof_pgemm
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out/profiles_r04
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python $R/tools/dev/pgemm_bench.py --once > $OUT/p$i.log 2>&1
  echo "pass $i [$grp] rc=$?"
done
python - "$OUT" > $R/gpurun_out/profiles_r04/r04_pgemm_pmc.txt <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        for key in ("pgemm_kernel<true, false, 8>", "pgemm_kernel<true, true, 8>", "pgemm_kernel<false, false, 8>", "cut_kernel", "gemm_f32_kernel"):
            if key in k:
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                break
print("# rocprofv3 --pmc <group> -- python tools/dev/pgemm_bench.py --once   (one pass per counter group; 760 x 2667 x 5329;")
print("# pgemm<true,false> = H W2, <true,true> = dA W2^T (two k-slices per tile), <false,false> = H^T dA; per-dispatch means)")
print("# FETCH_SIZE is in KB as rocprofv3 reports it and counts 64 B per 128-B request on gfx950: double it (MI355X_MICROARCH.md)")
for k, cs in sorted(agg.items()):
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-28s mean %.5g (n=%d)" % (c, sum(v) / len(v), len(v)))
    if "SQ_VALU_MFMA_BUSY_CYCLES" in cs and "SQ_BUSY_CYCLES" in cs:
        m, b = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"]), sum(cs["SQ_BUSY_CYCLES"]) / len(cs["SQ_BUSY_CYCLES"])
        print("    MFMA-busy cycles / SQ-busy cycles = %.3f" % (m / b if b else 0))
PY
cat $R/gpurun_out/profiles_r04/r04_pgemm_pmc.txt
rm -rf $OUT/p*/
