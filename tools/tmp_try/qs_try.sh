#!/bin/bash
cd "$GRAFT_REPO_ROOT/osteosarcoma_diffusionmodel_amd/csrc" || exit 1
cp ../lib/libosdiff.so /tmp/libosdiff.keep
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT/osteosarcoma_diffusionmodel_amd/csrc"
for r in 2 8; do
  sed "s/constexpr int QS_ROWS = 4;/constexpr int QS_ROWS = $r;/" k_elem.hip > /tmp/k_elem_$r.hip
  cp /tmp/k_elem_$r.hip ./k_elem_try.hip
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -c k_elem_try.hip -o /tmp/k_elem_try.o || exit 1
  rm -f k_elem_try.hip
  objs=$(ls build/*.o | grep -v "/k_elem.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libosdiff.so $objs /tmp/k_elem_try.o -ldl || exit 1
  (cd "$GRAFT_REPO_ROOT" && rm -rf gpurun_out/qs$r && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/qs$r -- python3 bench.py --train-only --train-steps 40 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('QS_ROWS', $r, d['train']['ms_per_step'])"; f=$(find gpurun_out/qs$r -name "*kernel_stats.csv" | head -1); grep -i "q_sample_src" $f | sed 's/.*)",//')
done
cp /tmp/libosdiff.keep ../lib/libosdiff.so
