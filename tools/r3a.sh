#!/bin/bash
set -o pipefail
out=gpurun_out/r3a; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_gpu_chain.py tests/test_gpu_train.py tests/test_gpu_hygiene.py -x -q -k "not config3_full" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $out/pytest.log
for v in "" "OSD_WGRAD_MID_FLUSH=1" "OSD_WGRAD_MID_FLUSH=1 OSD_WGRAD_MID_CAP=128" "OSD_WGRAD_MID_FLUSH=1 OSD_WGRAD_MID_CAP=512"; do
  echo "== $v"; env $v timeout -k 10 120 python bench.py --train-only --train-steps 60 2>&1 | tail -1 | cut -c1-140
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $out/busy -- python3 bench.py --train-only --train-steps 2 > $out/busy.log 2>&1
f=$(find $out/busy -name "*counter_collection.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
seen = collections.Counter()
for r in rows:
    k = r["Kernel_Name"]
    if ("EpiInput" in k or "wgrad_group_kernel" in k or "EpiGnSilu<32, false>" in k):
        seen[(k[:60], r["Counter_Name"])] += 1
        if seen[(k[:60], r["Counter_Name"])] == 25:
            print(k[:60], r["Counter_Name"], r["Counter_Value"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r.get("Grid_Size"), r.get("Workgroup_Size"))
PY
find $out/busy -type f -delete
