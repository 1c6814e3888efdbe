#!/bin/bash
# usage: pmc_lab.sh <binary> <args...> : a few PMC passes over a lab binary, per-dispatch averages
BIN=$1; shift
export TMPDIR=/tmp
OUT=/tmp/pmc_$$
for set in "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY" \
           "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL TCC_BUSY"; do
  rm -rf $OUT
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT -o p -- $BIN "$@" > /dev/null 2>&1
  python3 - $OUT <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f[0])):
    a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
print("  ".join("%s=%.4g" % (k, v[1] / v[0]) for k, v in acc.items()))
PY
done
