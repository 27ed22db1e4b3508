"""CPU oracle for the validation metrics -- TEST INFRASTRUCTURE ONLY.

numpy/float64 restatement of the four definitions of utils/validation.py that the HIP validation
kernels implement; pinned to the reference's own outputs by tests/golden/g9_validation.npz
(tests/test_oracle_golden.py).  Nothing in osteosarcoma_diffusionmodel_amd/ imports this file.
"""
from __future__ import annotations

from math import gcd
from typing import Dict, List, Sequence

import numpy as np


def mmd_rbf(X: np.ndarray, Y: np.ndarray, gamma: float = None) -> float:
    """sqrt(max(mean K_XX + mean K_YY - 2 mean K_XY, 0)), K = exp(-gamma |x-y|^2), gamma = 1/D,
    means over ALL pairs including the diagonal (utils/validation.py:273-298)."""
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    if gamma is None:
        gamma = 1.0 / X.shape[1]

    def kmean(A, B):
        d2 = (A * A).sum(1)[:, None] + (B * B).sum(1)[None, :] - 2.0 * A @ B.T
        return np.exp(-gamma * np.maximum(d2, 0.0)).mean()
    return float(np.sqrt(max(kmean(X, X) + kmean(Y, Y) - 2 * kmean(X, Y), 0.0)))


def ks_count_extremes(a: np.ndarray, b: np.ndarray):
    """max and min over all sample points v of  cnt(a <= v) * n2 - cnt(b <= v) * n1  (exact integers):
    the two-sample KS statistic is max(max_, -min_) / (n1 n2)  (scipy.stats.ks_2samp, as called at
    utils/validation.py:241)."""
    a, b = np.sort(a), np.sort(b)
    n1, n2 = len(a), len(b)
    allv = np.concatenate([a, b])
    c1 = np.searchsorted(a, allv, side="right").astype(np.int64)
    c2 = np.searchsorted(b, allv, side="right").astype(np.int64)
    diff = c1 * n2 - c2 * n1
    return int(diff.max()), int(diff.min())


def ks_pvalue(n1: int, n2: int, dmax: int, dmin: int):
    """(statistic, p-value) of the two-sided test from the integer extremes, following
    scipy.stats.ks_2samp(method='auto'): exact when max(n1, n2) <= 10000, else asymptotic."""
    from scipy.stats import distributions
    from scipy.stats._stats_py import _attempt_exact_2kssamp
    d = max(dmax, -dmin, 0) / (float(n1) * float(n2))
    if max(n1, n2) <= 10000:
        ok, d2, prob = _attempt_exact_2kssamp(n1, n2, gcd(n1, n2), d, "two-sided")
        if ok:
            return float(d2), float(np.clip(prob, 0, 1))
    m, n = sorted([float(n1), float(n2)], reverse=True)
    en = m * n / (m + n)
    return float(d), float(np.clip(distributions.kstwo.sf(d, np.round(en)), 0, 1))


def ks_summary(real: np.ndarray, synth: np.ndarray, max_features: int = 100) -> Dict[str, float]:
    """ks_test_mean_pvalue / ks_test_fraction_significant over the first min(D, 100) features
    (utils/validation.py:238-249)."""
    p = []
    for i in range(min(real.shape[1], max_features)):
        dmax, dmin = ks_count_extremes(real[:, i], synth[:, i])
        p.append(ks_pvalue(real.shape[0], synth.shape[0], dmax, dmin)[1])
    p = np.asarray(p)
    return {"ks_test_mean_pvalue": float(p.mean()), "ks_test_fraction_significant": float((p < 0.05).mean())}


def mean_offdiag_correlation(data: np.ndarray, cols: Sequence[int]) -> float:
    """Mean of the strict upper triangle of the Pearson matrix of data[:, cols]
    (utils/validation.py:156-161).  With z the column-standardised data (ddof = 1),
    sum_ij corr_ij = sum_n (sum_g z_ng)^2 / (N-1) and the diagonal is 1."""
    x = np.asarray(data, dtype=np.float64)[:, list(cols)]
    n, g = x.shape
    z = (x - x.mean(0)) / x.std(0, ddof=1)
    s = (z.sum(1) ** 2).sum() / (n - 1)
    return float((s - g) / (g * (g - 1)))


def pathway_coherence(real: np.ndarray, synth: np.ndarray, member: np.ndarray, n_data_genes: int) -> Dict[str, float]:
    """validate_pathway_coherence (utils/validation.py:144-173): first 10 pathways (columns of
    `member`, rows = listed genes of which the first n_data_genes exist in the data), >= 3 genes present."""
    rs, ss = [], []
    for p in range(min(member.shape[1], 10)):
        cols = [g for g in np.nonzero(member[:, p] == 1)[0] if g < n_data_genes]
        if len(cols) < 3:
            continue
        rs.append(mean_offdiag_correlation(real, cols))
        ss.append(mean_offdiag_correlation(synth, cols))
    if not rs:
        return {}
    return {"real_pathway_coherence": float(np.mean(rs)), "synthetic_pathway_coherence": float(np.mean(ss)),
            "pathway_coherence_correlation": float(np.corrcoef(rs, ss)[0, 1])}


def pearson(a: np.ndarray, b: np.ndarray) -> float:
    """Series.corr (utils/validation.py:205)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.corrcoef(a, b)[0, 1])


def violation_rate(corrs: Sequence[float], directions: Sequence[str]) -> float:
    """utils/validation.py:208-219."""
    v = sum(1 for c, d in zip(corrs, directions) if (d == "positive" and c < 0) or (d == "negative" and c > 0))
    return v / len(corrs)


def chi2_from_counts(n: int, n1: int, n2: int, n11: int) -> float:
    """chi2 statistic of scipy.stats.chi2_contingency(pd.crosstab(a, b)) for two 0/1 columns with n rows,
    n1 = #(a == 1), n2 = #(b == 1), n11 = #(a == 1 and b == 1) (utils/validation.py:98-108).  crosstab only lists the
    values that occur, so a constant column gives a one-row / one-column table (chi2 = 0)."""
    from scipy import stats
    table = np.array([[n - n1 - n2 + n11, n2 - n11], [n1 - n11, n11]], dtype=np.int64)
    table = table[table.sum(1) > 0][:, table.sum(0) > 0]
    return float(stats.chi2_contingency(table)[0])


def mutation_cooccurrence(real: np.ndarray, synth: np.ndarray, names: Sequence[str], driver_genes: Sequence[str],
                          exclusive_pairs: Sequence[Sequence[str]], picked: Sequence[int]) -> Dict[str, float]:
    """utils/validation.py:27-121 for 0/1 matrices with identical columns; ``picked`` = the genes drawn by
    np.random.choice at :91-93, in draw order."""
    real, synth = np.asarray(real, dtype=np.float64), np.asarray(synth, dtype=np.float64)
    idx = {g: i for i, g in enumerate(names)}
    out = {"mutation_frequency_correlation": float(np.corrcoef(real.mean(0), synth.mean(0))[0, 1])}
    drivers = [idx[g] for g in driver_genes if g in idx]
    if drivers:
        out["driver_gene_frequency_diff"] = float(np.abs(real[:, drivers].mean(0) - synth[:, drivers].mean(0)).mean())
    viol = pairs = 0
    for a, b in exclusive_pairs:
        if a in idx and b in idx:
            viol += int(((synth[:, idx[a]] == 1) & (synth[:, idx[b]] == 1)).sum())
            pairs += 1
    if pairs:
        out["mutual_exclusivity_violation_rate"] = viol / (len(synth) * pairs)
    cr, cs = [], []
    for i, a in enumerate(picked):
        for b in picked[i + 1:]:
            for data, dst in ((real, cr), (synth, cs)):
                x, y = data[:, a], data[:, b]
                dst.append(chi2_from_counts(len(data), int(x.sum()), int(y.sum()), int((x * y).sum())))
    if cr:
        out["cooccurrence_pattern_correlation"] = float(np.corrcoef(cr, cs)[0, 1])
    return out


def wasserstein_pca_mean(real: np.ndarray, synth: np.ndarray, n_components: int = 10) -> float:
    """utils/validation.py:256-269: PCA fitted on the real data, mean 1-D Wasserstein distance over the components."""
    from scipy import stats
    from sklearn.decomposition import PCA
    pca = PCA(n_components=n_components)
    rp = pca.fit_transform(real)
    sp = pca.transform(synth)
    return float(np.mean([stats.wasserstein_distance(rp[:, i], sp[:, i]) for i in range(n_components)]))
