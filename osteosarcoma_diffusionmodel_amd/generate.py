"""Generation pipeline on MI355X -- host-side mirror of the reference's utils/generate.py.

``SyntheticPatientGenerator`` keeps the reference's constructor and methods
(utils/generate.py:19-235); ``generate`` runs the whole T-step reverse chain and the
mutation binarisation on the device through ``BiologyAwareDiffusionModel.sample``.
``generate_patients`` is the north-star convenience wrapper.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import pandas as pd
import torch

logger = logging.getLogger(__name__)


class SyntheticPatientGenerator:
    """Generate synthetic patients using a trained model (utils/generate.py:19)."""

    def __init__(self, model, config: dict, device: str = "cuda"):
        self.model = model.to(device)
        self.model.eval()
        # the reference generates 1000 patients per scenario by default (config.yaml:119): small batches, where input_proj's long
        # K loop over few output tiles is the step's longest launch -- let the per-layer engine split it over workgroups unless
        # the caller chose (model.input_splitk = 0 keeps results bit-independent of the batch split)
        if getattr(self.model, "input_splitk", 0) is None:
            self.model.input_splitk = -1
        self.config = config
        self.device = device
        self.mutation_dim = model.mutation_dim
        self.expression_dim = model.expression_dim
        self.pathway_dim = model.pathway_dim
        self.condition_dim = model.condition_dim

    def create_conditions(self, num_samples: int, scenario: Optional[Dict] = None) -> torch.Tensor:
        """Constant condition rows for a scenario, or randn rows (utils/generate.py:39-94)."""
        if scenario is None:
            return torch.randn(num_samples, self.condition_dim, device=self.device)
        values: List[float] = []
        for name in self.config["model"]["condition_on"]:
            if name == "survival_time":
                values.append((scenario.get("survival_time", 800) - 800) / 500)   # hard-wired (800, 500)
            elif name == "event_occurred":
                values.append(scenario.get("event_occurred", 0))
            elif name == "age":
                values.append(scenario.get("age", 15.0))
            elif name == "metastasis_at_diagnosis":
                values.append(scenario.get("metastasis_at_diagnosis", 0))
            # unknown names are skipped, as in the reference
        if len(values) != self.condition_dim:
            logger.warning(f"Condition mismatch: expected {self.condition_dim}, got {len(values)}")
            if len(values) < self.condition_dim:
                values.extend([0.0] * (self.condition_dim - len(values)))
            else:
                values = values[:self.condition_dim]
        row = torch.tensor([values], dtype=torch.float32, device=self.device)
        return row.repeat(num_samples, 1)

    @torch.no_grad()
    def generate(self, num_samples: int, scenario: Optional[Dict] = None, guidance_scale: float = 1.0,
                 *, seed: Optional[int] = None, row_offset: int = 0, x_T=None, noise=None) -> Dict[str, np.ndarray]:
        """utils/generate.py:96-144.  ``guidance_scale`` is accepted and ignored, as in the reference.
        Keyword-only extras inject the random draws / shard the Philox stream."""
        logger.info(f"Generating {num_samples} synthetic patients...")
        if scenario:
            logger.info(f"Scenario: {scenario}")
        conditions = self.create_conditions(num_samples, scenario)
        md, ed = self.mutation_dim, self.expression_dim
        if hasattr(self.model, "vae"):
            # BiologyConstrainedVAE (utils/train.py:233's dispatch; load_trained_model's "cvae" branch): the reference's
            # generate() only ever calls model.sample(conditions, num_samples) and binarises on the host (:124-135)
            if seed is not None or row_offset or x_T is not None or noise is not None:
                raise ValueError("seed / row_offset / x_T / noise drive the diffusion sampler's Philox stream and are not "
                                 "accepted for a cVAE model (pass z= to model.sample directly)")
            samples = self.model.sample(conditions, num_samples=num_samples).cpu().numpy()
            mutations = (samples[:, :md] > 0.5).astype(float)
        else:
            samples, mask = self.model.sample(conditions, num_samples=num_samples, seed=seed, row_offset=row_offset,
                                              x_T=x_T, noise=noise, return_mutation_mask=True)
            samples = samples.cpu().numpy()
            # (mutations > 0.5).astype(float), evaluated by the last reverse step's epilogue on the device
            mutations = mask.cpu().numpy().astype(float)
        expression = samples[:, md:md + ed]
        pathways = samples[:, md + ed:]
        logger.info("Generation complete!")
        return {"mutations": mutations, "expression": expression, "pathways": pathways,
                "conditions": conditions.cpu().numpy()}

    def generate_scenarios(self, scenarios: List[Dict], samples_per_scenario: int, *, seed: Optional[int] = None,
                           batched: bool = True) -> Dict[str, Dict[str, np.ndarray]]:
        """utils/generate.py:146-175: one result dict per scenario name.

        The reference runs the scenarios one after the other, each a chain of T sequential steps.  Rows never interact and the
        conditions are per row, so here all scenarios form ONE batch (scenario k = rows k*N .. (k+1)*N-1) and the chain runs
        once: at the reference's default size (3 scenarios x 1000 patients, config.yaml:119-141) a reverse step is bound by
        launch latency, not by rows, and T steps over 3000 rows cost about what T steps over 1000 do.  ``batched=False`` restores
        the reference's loop (one chain, and one freshly drawn Philox seed, per scenario)."""
        if not batched or hasattr(self.model, "vae") or len(scenarios) < 2:
            out = {}
            for scenario in scenarios:
                name = scenario["name"]
                logger.info(f"\nGenerating scenario: {name}")
                out[name] = self.generate(num_samples=samples_per_scenario, scenario=scenario["conditions"])
            return out
        n = int(samples_per_scenario)
        for scenario in scenarios:
            logger.info(f"\nGenerating scenario: {scenario['name']}")
            logger.info(f"Scenario: {scenario['conditions']}")
        logger.info(f"Generating {len(scenarios)} x {n} synthetic patients in one batch...")
        conditions = torch.cat([self.create_conditions(n, sc["conditions"]) for sc in scenarios], dim=0)
        with torch.no_grad():
            samples, mask = self.model.sample(conditions, num_samples=conditions.shape[0], seed=seed, return_mutation_mask=True)
        samples, mask, cond_np = samples.cpu().numpy(), mask.cpu().numpy().astype(float), conditions.cpu().numpy()
        md, ed = self.mutation_dim, self.expression_dim
        out = {}
        for k, scenario in enumerate(scenarios):
            rows = slice(k * n, (k + 1) * n)
            out[scenario["name"]] = {"mutations": mask[rows], "expression": samples[rows, md:md + ed], "pathways": samples[rows, md + ed:],
                                     "conditions": cond_np[rows]}
        logger.info("Generation complete!")
        return out

    def save_synthetic_data(self, synthetic_data: Dict[str, np.ndarray], output_dir: Path,
                            gene_names: Dict[str, List[str]], prefix: str = "synthetic"):
        """Four CSVs per scenario (utils/generate.py:177-235)."""
        output_dir = Path(output_dir)
        output_dir.mkdir(parents=True, exist_ok=True)
        for key, cols_key, stem in (("mutations", "mutation_genes", "mutations"),
                                    ("expression", "expression_genes", "expression"),
                                    ("pathways", "pathway_names", "pathways")):
            if cols_key in gene_names:
                path = output_dir / f"{prefix}_{stem}.csv"
                pd.DataFrame(synthetic_data[key], columns=gene_names[cols_key]).to_csv(path, index=False)
                logger.info(f"Saved {stem} to {path}")
        cond_path = output_dir / f"{prefix}_conditions.csv"
        pd.DataFrame(synthetic_data["conditions"], columns=self.config["model"]["condition_on"]).to_csv(cond_path, index=False)
        logger.info(f"Saved conditions to {cond_path}")


def load_trained_model(checkpoint_path: Path, config: dict, device: str):
    """Checkpoint -> model (utils/generate.py:238-298): condition width from the saved
    ``condition_embed.mlp.0.weight``, feature dims from the processed CSV headers."""
    from .diffusion import BiologyAwareDiffusionModel
    logger.info(f"Loading model from {checkpoint_path}")
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    state_dict = checkpoint["model_state_dict"]
    arch = config["model"]["architecture"]
    if arch not in ("diffusion", "cvae"):
        raise ValueError(f"Unknown architecture: {arch}")
    processed = Path(config["data"]["processed_dir"])
    dims = []
    for fname in ("mutation_matrix_aligned.csv", "expression_matrix_aligned.csv", "pathway_scores.csv"):
        dims.append(pd.read_csv(processed / fname, index_col=0, nrows=1).shape[1])
    if arch == "diffusion":
        saved_cond_dim = state_dict["condition_embed.mlp.0.weight"].shape[1]
        model = BiologyAwareDiffusionModel(mutation_dim=dims[0], expression_dim=dims[1], pathway_dim=dims[2],
                                           condition_dim=saved_cond_dim, config=config)
    else:
        # the reference reads the diffusion key even for a cVAE checkpoint (utils/generate.py:250) and fails on it;
        # here the condition width comes from the encoder's first Linear: in_features = data_dim + condition_dim
        from .cvae import BiologyConstrainedVAE
        saved_cond_dim = state_dict["vae.encoder.mlp.0.weight"].shape[1] - sum(dims)
        model = BiologyConstrainedVAE(mutation_dim=dims[0], expression_dim=dims[1], pathway_dim=dims[2],
                                      condition_dim=saved_cond_dim, config=config)
    model.load_state_dict(state_dict)
    model.to(device)
    model.eval()
    logger.info("Model loaded successfully!")
    return model


def generate_patients(model_or_checkpoint, config: dict, num_samples: int, scenario: Optional[Dict] = None,
                      device: Optional[str] = None, **kw) -> Dict[str, np.ndarray]:
    """north_star name: load (if given a path) and run SyntheticPatientGenerator.generate."""
    device = device or "cuda"
    model = model_or_checkpoint
    if isinstance(model_or_checkpoint, (str, Path)):
        model = load_trained_model(Path(model_or_checkpoint), config, device)
    return SyntheticPatientGenerator(model, config, device).generate(num_samples, scenario, **kw)
