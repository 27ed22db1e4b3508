"""GPU: the on-device validation metrics (osd_val_*, osteosarcoma_diffusionmodel_amd/validation.py) against
the reference's BiologicalValidator outputs (tests/golden/g9_validation.npz) and the numpy oracle.
Tolerances: KS extremes are exact integers (statistic and p-value to 1e-12); MMD 1e-5 absolute
(fp32 Gram against float64 cdist); correlations 1e-6 absolute."""
import numpy as np
import pandas as pd
import pytest
import torch

from oracle import validation_oracle as V
from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator
from helpers import load_golden

pytestmark = pytest.mark.gpu

CONF = {"evaluation": {"driver_genes": ["TP53"], "mutually_exclusive_pairs": [],
                       "required_correlations": [{"mutation": "TP53", "pathway": "HALLMARK_P53_PATHWAY", "direction": "negative"},
                                                 {"mutation": "MYC", "pathway": "HALLMARK_MYC_TARGETS_V1", "direction": "positive"},
                                                 {"mutation": "ABSENT", "pathway": "HALLMARK_P53_PATHWAY", "direction": "positive"}]}}


def test_mmd_and_ks_vs_reference(golden_dir):
    g = load_golden(golden_dir, "g9_validation")
    val = BiologicalValidator(CONF)
    real, synth = g["real"], g["synth"]
    assert abs(val.compute_mmd(real, synth) - g["mmd"]) < 1e-5
    assert val.compute_mmd(real, real) < 2e-3            # sqrt of fp32 round-off around an exact zero
    d, p = val.ks_tests(real, synth)
    assert np.abs(d - g["ks_stat"]).max() < 1e-12 and np.abs(p - g["ks_pvalue"]).max() < 1e-12
    st = val.statistical_tests(real, synth)
    assert abs(st["ks_test_mean_pvalue"] - g["stat.ks_test_mean_pvalue"]) < 1e-12
    assert st["ks_test_fraction_significant"] == g["stat.ks_test_fraction_significant"]
    assert abs(st["mmd"] - g["stat.mmd"]) < 1e-5


def test_pathway_coherence_and_sign_rules_vs_reference(golden_dir):
    g = load_golden(golden_dir, "g9_validation")
    val = BiologicalValidator(CONF)
    genes = [f"G{i}" for i in range(40)]
    pgm = pd.DataFrame(g["coh_member"], index=[f"G{i}" for i in range(45)], columns=[f"P{i}" for i in range(12)])
    coh = val.validate_pathway_coherence(pd.DataFrame(g["coh_real"], columns=genes), pd.DataFrame(g["coh_synth"], columns=genes), pgm)
    assert set(coh) == {"real_pathway_coherence", "synthetic_pathway_coherence", "pathway_coherence_correlation"}
    for k, v in coh.items():
        assert abs(v - g["coh." + k]) < 1e-6, k
    mut = pd.DataFrame(g["me_mut"], columns=["TP53", "MYC", "RB1"])
    pw = pd.DataFrame(g["me_pw"], columns=["HALLMARK_P53_PATHWAY", "HALLMARK_MYC_TARGETS_V1"])
    me = val.validate_mutation_expression_correlation(mut, None, pw)
    assert me["mutation_expression_violation_rate"] == g["me.violation_rate"]


def test_metrics_at_scale_vs_oracle():
    """Sizes past one tile / one block, ties, and the asymptotic p-value branch (n > 10000)."""
    rs = np.random.RandomState(3)
    n1, n2, D = 12001, 9000, 200
    real = rs.randn(n1, D).astype(np.float32)
    synth = (rs.randn(n2, D) * 1.05 + 0.02).astype(np.float32)
    synth[:, 5] = np.round(synth[:, 5], 1)              # heavy ties
    real[:, 5] = np.round(real[:, 5], 1)
    val = BiologicalValidator(CONF)
    d, p = val.ks_tests(real, synth, max_features=12)
    for i in range(12):
        dmax, dmin = V.ks_count_extremes(real[:, i], synth[:, i])
        dd, pp = V.ks_pvalue(n1, n2, dmax, dmin)
        assert d[i] == dd and p[i] == pp
    sub_r, sub_s = real[:3000], synth[:2500]
    assert abs(val.compute_mmd(sub_r, sub_s) - V.mmd_rbf(sub_r, sub_s)) < 1e-5
    cols = list(range(3, 150, 2))
    x = torch.from_numpy(real).cuda()
    assert abs(val._mean_offdiag(x, cols) - V.mean_offdiag_correlation(real, cols)) < 1e-6
    with pytest.raises(ValueError):
        val._mean_offdiag(x, [0])


def test_ks_sort_edge_values():
    """The segmented radix sort under the KS statistic (csrc/segsort.h): segments shorter than one wave step, lengths that
    are not multiples of the 4096-key chunk, +-0, denormal-sized and huge magnitudes, 0/1 columns (all ties), constant
    columns.  The statistic is an exact integer function of the two sorted samples: any misplaced key changes it."""
    rs = np.random.RandomState(11)
    for n1, n2 in ((37, 50), (4096, 4097), (8191, 63)):
        D = 8
        real = rs.randn(n1, D).astype(np.float32)
        synth = rs.randn(n2, D).astype(np.float32)
        real[:, 0] = (rs.rand(n1) < 0.3); synth[:, 0] = (rs.rand(n2) < 0.4)                       # mutation-like
        real[:, 1] *= 1e30; synth[:, 1] *= 1e-30                                                   # exponent extremes
        real[:, 2] = np.where(rs.rand(n1) < 0.5, 0.0, -0.0); synth[:, 2] = np.where(rs.rand(n2) < 0.5, -0.0, 0.0)
        real[:, 3] = 2.5; synth[:, 3] = 2.5                                                         # constant
        real[:, 4] = -np.abs(real[:, 4]); synth[:, 4] = np.abs(synth[:, 4])                         # disjoint supports
        val = BiologicalValidator(CONF)
        d, p = val.ks_tests(real, synth, max_features=D)
        for i in range(D):
            dmax, dmin = V.ks_count_extremes(real[:, i], synth[:, i])
            dd, pp = V.ks_pvalue(n1, n2, dmax, dmin)
            assert d[i] == dd and p[i] == pp, (n1, n2, i)


REF_EVAL = {"evaluation": {"driver_genes": ["TP53", "RB1", "ATRX", "DLG2", "PTEN"], "mutually_exclusive_pairs": [["TP53", "MDM2"]],
                           "required_correlations": [{"mutation": "TP53", "pathway": "HALLMARK_P53_PATHWAY", "direction": "negative"},
                                                     {"mutation": "MYC", "pathway": "HALLMARK_MYC_TARGETS_V1", "direction": "positive"}]}}
CO_NAMES = ["TP53", "RB1", "ATRX", "PTEN", "MDM2", "MYC"] + [f"M{i}" for i in range(54)]


def test_mutation_cooccurrence_vs_reference(golden_dir):
    """Exact device counts -> the reference's numbers (frequencies, exclusivity, chi-square pattern correlation)."""
    g = load_golden(golden_dir, "g9_validation")
    val = BiologicalValidator(REF_EVAL)
    rm, sm = pd.DataFrame(g["co_real"], columns=CO_NAMES), pd.DataFrame(g["co_synth"], columns=CO_NAMES)
    np.random.seed(123)                                  # the seed the fixture was generated under (np.random.choice of 50 genes)
    co = val.validate_mutation_cooccurrence(rm, sm)
    assert set(co) == {k[3:] for k in g if k.startswith("co.")}
    for k, v in co.items():
        assert abs(v - g["co." + k]) < 1e-10, (k, v, g["co." + k])


def test_validate_all_vs_reference(golden_dir):
    g = load_golden(golden_dir, "g9_validation")
    val = BiologicalValidator(REF_EVAL)
    genes = [f"G{i}" for i in range(40)]
    pw_cols = ["HALLMARK_P53_PATHWAY", "HALLMARK_MYC_TARGETS_V1"]
    pgm = pd.DataFrame(g["coh_member"], index=[f"G{i}" for i in range(45)], columns=[f"P{i}" for i in range(12)])
    np.random.seed(123)
    res = val.validate_all(pd.DataFrame(g["co_real"], columns=CO_NAMES), pd.DataFrame(g["coh_real"], columns=genes),
                           pd.DataFrame(g["all_real_pw"], columns=pw_cols), pd.DataFrame(g["co_synth"], columns=CO_NAMES),
                           pd.DataFrame(g["coh_synth"], columns=genes), pd.DataFrame(g["me_pw"], columns=pw_cols), pgm)
    assert set(res) == {k[4:] for k in g if k.startswith("all.") and k != "all_real_pw"}
    tol = {"mmd": 1e-5, "wasserstein_distance_mean": 1e-4, "real_pathway_coherence": 1e-6, "synthetic_pathway_coherence": 1e-6,
           "pathway_coherence_correlation": 1e-5}
    for k, v in res.items():
        assert abs(v - g["all." + k]) < tol.get(k, 1e-10), (k, v, g["all." + k])


def test_gram_counts_at_scale():
    """Joint counts of 64 binary columns over 300 001 rows are exact integers."""
    rs = np.random.RandomState(5)
    x = (rs.rand(300001, 70) < 0.3).astype(np.float32)
    val = BiologicalValidator(REF_EVAL)
    t = torch.from_numpy(x).cuda()
    cols = list(range(3, 67))
    gram = val._gram(t, cols)
    ref = x[:, cols].astype(np.float64).T @ x[:, cols].astype(np.float64)
    assert np.array_equal(gram, ref)
    assert np.array_equal(val._column_sums(t), x.astype(np.float64).sum(0))
    with pytest.raises(ValueError):
        val._gram(t, list(range(65)))
