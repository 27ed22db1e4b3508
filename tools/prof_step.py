#!/usr/bin/env python3
"""Run a few eager reverse steps (osd_profile_step) at the bench chunk shape -- the target of
`rocprofv3 --pmc ...` counter passes (diagnostic; not part of the product path)."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from bench import CONF, scenario_conditions  # noqa: E402
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(0)
model = BiologyAwareDiffusionModel(50, 1900, 50, 3, CONF).cuda().eval()
eng = model._engine()
cond = scenario_conditions(rows, 0).cuda()
ms, fl, ne = (C.c_float * 64)(), (C.c_double * 64)(), C.c_int()
L.check(L.lib().osd_profile_step(eng.handle, L.ptr(cond), rows, reps, ms, fl, 64, C.byref(ne)))
torch.cuda.synchronize()
print("step ms", sum(ms[i] for i in range(ne.value)))
