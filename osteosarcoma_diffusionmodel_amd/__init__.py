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
