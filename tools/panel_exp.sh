#!/bin/bash
# timing experiments on the LDS-resident chain's posterior epilogue: builds variants of the library under /tmp and times them
cd "$GRAFT_REPO_ROOT/osteosarcoma_diffusionmodel_amd/csrc" || exit 1
cp ../lib/libosdiff.so /tmp/libosdiff.keep
for e in 0 1 2 4 7; do
  rm -rf build_exp; mkdir build_exp
  for f in chain_panel; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -I. -DOSD_EXP=$e -c $f.hip -o build_exp/$f.o || exit 1; done
  objs=$(ls build/*.o | grep -v chain_panel.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libosdiff.so $objs build_exp/chain_panel.o -ldl || exit 1
  echo "OSD_EXP=$e"; (cd "$GRAFT_REPO_ROOT" && timeout -k 10 200 python tools/panel_try.py 100000 50 2>&1 | grep "^panel" | tail -1)
done
cp /tmp/libosdiff.keep ../lib/libosdiff.so
