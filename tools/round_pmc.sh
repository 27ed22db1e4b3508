#!/bin/bash
# PMC passes of the final build (one counter set per pass; no other trace domains with --pmc):
#   both persistent chain kernels (workspace, LDS-resident) at 100 000 patients x 50 steps, one eager reverse step of the per-layer kernels at 32 768 rows,
#   the bf16x3 engine (csrc/split.hip) at 32 768 rows x 6 steps (one chunk: the launches see 32 768 rows like the per-layer pass),
#   and the squad chain (csrc/chain_squad.h) at 2 976 rows (93 panels: three workgroups per CU) x 100 steps.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_final
rm -rf $out; mkdir -p $out
run() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/chain_$name -- python3 tools/chain_run.py 100000 50 chain > $out/chain_$name.log 2>&1; echo "chain $name rc=$?";
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/panel_$name -- python3 tools/chain_run.py 100000 50 chain panel > $out/panel_$name.log 2>&1; echo "panel $name rc=$?";
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/layer_$name -- python3 tools/prof_step.py 32768 2 > $out/layer_$name.log 2>&1; echo "layer $name rc=$?";
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/split_$name -- python3 tools/chain_run.py 32768 6 split > $out/split_$name.log 2>&1; echo "split $name rc=$?";
        rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $out/squad_$name -- python3 tools/chain_run.py 2976 100 chain squad > $out/squad_$name.log 2>&1; echo "squad $name rc=$?"; }
run busy GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY
run lds SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum
python3 tools/pmc_summary.py $out r04 5000000 > $out/summary.md; cat $out/summary.md
find $out -type f ! -name "summary.md" ! -name "*.log" ! -name "*_traffic.json" -delete
