#!/bin/bash
# PMC passes over the bf16x3 engine (one counter set per pass): split_pmc.sh <tag>  -> gpurun_out/pmc_split/summary.md
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
tag=${1:-r04}
out=gpurun_out/pmc_split
rm -rf $out; mkdir -p $out
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/split_$name -- python3 tools/chain_run.py 65536 4 split > $out/split_$name.log 2>&1; echo "split $name rc=$?"; }
run busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run wait GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run lds SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run mix SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/split_pmc_summary.py $out > $out/summary.md; cat $out/summary.md
find $out -type f ! -name "summary.md" ! -name "*.log" -delete
