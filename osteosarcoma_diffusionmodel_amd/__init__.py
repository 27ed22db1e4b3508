"""MI355X-native diffusion hot path of rare-resilience-ai/Osteosarcoma_DiffusionModel.

Host-side mirror of the reference's Python API over the C ABI of libosdiff.so:
    diffusion.py  <- models/diffusion.py   (BiologyAwareDiffusionModel)
    train.py      <- utils/train.py        (Trainer, MixupAugmentation, EarlyStopping, ...)
    generate.py   <- utils/generate.py     (SyntheticPatientGenerator, load_trained_model)
"""
from .diffusion import BiologyAwareDiffusion, BiologyAwareDiffusionModel  # noqa: F401
from .generate import SyntheticPatientGenerator, generate_patients, load_trained_model  # noqa: F401

__all__ = ["BiologyAwareDiffusionModel", "BiologyAwareDiffusion", "SyntheticPatientGenerator",
           "generate_patients", "load_trained_model"]


def _respect_cpu_quota():
    """PyTorch sizes its intra-op thread pool by the CPUs it can see (256 on an MI355X host) even when the container's cgroup
    grants far fewer (cpu.max: 16 cores per GPU on the boxes this was measured on).  One small CPU tensor op per epoch -- an
    index gather of 57 000 rows -- then wakes the whole OpenMP team, the idle workers spin at the region's barrier, the CFS quota
    of the 100 ms period is gone, and EVERY thread of the process, HIP's runtime threads included, is frozen until the next
    period: 60-90 ms GPU stalls on a 100 ms grid (tools/throttle_check.sh: nr_throttled 2 -> 20 during 170 training steps;
    none with a capped pool, and the epoch loop went from 2.0 to 1.03 ms/step).  Cap the pool at the quota once, at import, and
    log the change once (logger `osteosarcoma_diffusionmodel_amd`, WARNING); OSD_KEEP_TORCH_THREADS=1 leaves torch's setting alone."""
    import os
    if os.environ.get("OSD_KEEP_TORCH_THREADS") == "1":
        return
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota == "max":
            return
        cores = max(1, int(int(quota) / int(period)))
    except (OSError, ValueError):
        return
    import torch
    before = torch.get_num_threads()
    if before > cores:
        torch.set_num_threads(cores)
        # a drop-in package changing a process-wide setting of its host says so, once (the import runs once per process)
        import logging
        logging.getLogger(__name__).warning(
            "torch intra-op threads capped %d -> %d (the cgroup's cpu.max quota; OSD_KEEP_TORCH_THREADS=1 leaves torch's setting alone)",
            before, cores)


_respect_cpu_quota()
