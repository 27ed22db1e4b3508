#!/bin/bash
# PMC passes of the persistent chain kernel (one counter set per pass; no other trace domains with --pmc)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/${1:-chain_pmc}
rm -rf $out; mkdir -p $out
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 tools/chain_run.py 100000 50 chain > $out/$name.log 2>&1; echo "$name rc=$?"; }
run busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run lds SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/*/*counter_collection.csv") + glob.glob(out + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in agg.items():
    if "chain_kernel" in k or "gemm_glds" in k:
        print(k, {c: v for c, v in d.items()})
PY
find $out -name "*agent_info.csv" -delete
