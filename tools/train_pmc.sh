#!/bin/bash
# PMC passes over the training step (BASELINE config 2: B = 4096, D = 2000), one counter set per pass, no other trace
# domain beside --pmc.  Summary table -> gpurun_out/train_pmc/summary.md (copied to profiles/rNN_train_pmc.md).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/train_pmc
tag=${1:-r03}
rm -rf $out; mkdir -p $out
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/$name -- python3 bench.py --train-only --train-steps 6 --no-split > $out/$name.log 2>&1; echo "train pmc $name rc=$?"; }
run busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run insts SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/train_pmc_summary.py $out $tag > $out/summary.md; cat $out/summary.md
find $out -type f ! -name "summary.md" ! -name "*.log" -delete
