# LDS-side counters of the GEMM kernels in the bench step (one SQ pass): bank conflicts vs active LDS cycles, wait buckets
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/lds_pmc -o l -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2> $OUT/lds_pmc.log
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
f = glob.glob("gpurun_out/lds_pmc/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "gemm" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in sorted(acc.items(), key=lambda x: -x[1]["SQ_WAVE_CYCLES"])[:12]:
    w = v["SQ_WAVE_CYCLES"] or 1
    print("%-70s n=%3d conflict/active %.3f | of wave cycles: wait_any %.2f wait_inst %.2f (lds %.2f) active %.2f" % (
        k[-70:], n[k], v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), v["SQ_WAIT_ANY"] / w, v["SQ_WAIT_INST_ANY"] / w, v["SQ_WAIT_INST_LDS"] / w, v["SQ_ACTIVE_INST_ANY"] / w))
PY
find gpurun_out/lds_pmc -name "*.csv" -size +1M -delete
