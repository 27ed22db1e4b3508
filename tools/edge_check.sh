timeout -k 10 200 python bench.py --gpus 1 --steps 1 --warmup 0 --patients 1000 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
timeout -k 10 200 python bench.py --steps 1 --warmup 0 --patients 20000 --no-graph --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout -k 10 100 python - <<'PY'
import torch
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, SyntheticPatientGenerator
from bench import CONF
m = BiologyAwareDiffusionModel(50,1900,50,3,CONF)
g = SyntheticPatientGenerator(m, dict(CONF, generation={}), "cuda")
out = g.generate(7, scenario={"survival_time": 300, "event_occurred": 1})
print({k: v.shape for k, v in out.items()}, out["mutations"].dtype, set(out["mutations"].ravel().tolist()) <= {0.0, 1.0})
PY
