"""GPU: the reference's pipeline shape end to end on synthetic "TARGET-OS"-style CSVs
(QUICKSTART.md:206-248 recipe, SURVEY section 8d config 1 data): prepare_data -> Trainer.train ->
checkpoint -> load_trained_model -> generate_scenarios -> save_synthetic_data."""
import numpy as np
import pandas as pd
import pytest
import torch

from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, SyntheticPatientGenerator, load_trained_model
from osteosarcoma_diffusionmodel_amd.train import Trainer, prepare_data

pytestmark = pytest.mark.gpu


def write_dummy_processed(root, n=100, n_mut=50, n_expr=100, n_path=30, seed=42):
    rng = np.random.RandomState(seed)
    ids = [f"TARGET-40-{i:04d}" for i in range(n)]
    root.mkdir(parents=True, exist_ok=True)
    pd.DataFrame(rng.randint(0, 2, (n, n_mut)), index=ids, columns=[f"GENE{i}" for i in range(n_mut)]).to_csv(root / "mutation_matrix_aligned.csv")
    pd.DataFrame(rng.randn(n, n_expr), index=ids, columns=[f"EXPR{i}" for i in range(n_expr)]).to_csv(root / "expression_matrix_aligned.csv")
    pd.DataFrame(rng.randn(n, n_path), index=ids, columns=[f"HALLMARK_{i}" for i in range(n_path)]).to_csv(root / "pathway_scores.csv")
    pd.DataFrame({"submitter_id": ids, "survival_days": rng.randint(100, 2000, n), "event_occurred": rng.randint(0, 2, n),
                  "age_years": rng.uniform(10, 18, n), "metastasis_at_diagnosis": rng.randint(0, 2, n)}).to_csv(root / "clinical_aligned.csv", index=False)


def test_train_generate_pipeline(tmp_path):
    processed = tmp_path / "data" / "processed"
    write_dummy_processed(processed)
    config = {
        "data": {"processed_dir": str(processed), "pathway_database": "msigdb_hallmark"},
        "model": {"architecture": "diffusion", "latent_dim": 128, "hidden_dims": [256, 512, 256], "gnn": {"dropout": 0.2},
                  "diffusion": {"num_steps": 40, "beta_schedule": "cosine"},
                  "condition_on": ["survival_time", "event_occurred", "metastasis_at_diagnosis"]},
        "training": {"batch_size": 16, "num_epochs": 3, "learning_rate": 1e-4, "weight_decay": 1e-5, "patience": 100,
                     "min_delta": 1e-4, "augmentation": {"mixup_alpha": 0.2}, "val_split": 0.2, "random_seed": 42,
                     "save_dir": str(tmp_path / "ckpt"), "save_frequency": 10},
        "generation": {"scenarios": [{"name": "typical_patient", "conditions": {"survival_time": 800, "event_occurred": 0, "metastasis_at_diagnosis": 0}},
                                     {"name": "metastatic_poor_prognosis", "conditions": {"survival_time": 300, "event_occurred": 1, "metastasis_at_diagnosis": 1}}]},
    }
    train_loader, val_loader, config = prepare_data(config)
    m = config["model"]
    assert (m["n_genes_mutation"], m["n_genes_expression"], m["n_pathways"], m["n_conditions"]) == (50, 100, 30, 4)
    assert len(train_loader) == 5                                   # 80 rows, batch 16, drop_last
    torch.manual_seed(0)
    np.random.seed(0)
    model = BiologyAwareDiffusionModel(m["n_genes_mutation"], m["n_genes_expression"], m["n_pathways"], m["n_conditions"], config)
    hist = Trainer(model, train_loader, val_loader, config, device="cuda").train()
    assert len(hist["train_loss"]) == 3 and np.isfinite(hist["train_loss"]).all() and np.isfinite(hist["val_loss"]).all()
    ckpt = tmp_path / "ckpt" / "best_model.pt"
    assert ckpt.exists()
    loaded = load_trained_model(ckpt, config, "cuda")
    assert loaded.condition_dim == 4 and loaded.data_dim == 180
    for (k, a), (_, b) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert torch.equal(a.cpu(), b.cpu()), k
    gen = SyntheticPatientGenerator(loaded, config, device="cuda")
    out = gen.generate_scenarios(config["generation"]["scenarios"], 25)     # 3 condition names vs cond_dim 4: padded with a warning
    assert set(out) == {"typical_patient", "metastatic_poor_prognosis"}
    names = {"mutation_genes": [f"GENE{i}" for i in range(50)], "expression_genes": [f"EXPR{i}" for i in range(100)],
             "pathway_names": [f"HALLMARK_{i}" for i in range(30)]}
    config["model"]["condition_on"] = config["model"]["condition_on"] + ["pad"]      # 4 condition columns for the CSV header
    for scen, data in out.items():
        assert data["mutations"].shape == (25, 50) and set(np.unique(data["mutations"])) <= {0.0, 1.0}
        assert data["expression"].shape == (25, 100) and data["pathways"].shape == (25, 30) and data["conditions"].shape == (25, 4)
        assert np.isfinite(data["expression"]).all()
        gen.save_synthetic_data(data, tmp_path / "synthetic", names, prefix=scen)
        for part in ("mutations", "expression", "pathways", "conditions"):
            df = pd.read_csv(tmp_path / "synthetic" / f"{scen}_{part}.csv")
            assert len(df) == 25
