"""Validation metrics on MI355X -- host-side mirror of the part of the reference's utils/validation.py
that consumes the sampler output at scale: RBF-MMD, per-feature two-sample KS,
pathway coherence and the mutation-expression sign check.  Same method names and result keys as
``BiologicalValidator``; the arithmetic runs in libosdiff.so (``osd_val_*``).

Not provided (outside the hot path): the chi-square co-occurrence test on a random
gene subset (utils/validation.py:94-121) and the Wasserstein distance on PCA components (:256-269).
"""
from __future__ import annotations

import ctypes as C
import logging
from math import gcd
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L

logger = logging.getLogger(__name__)


def _dev(a, device) -> torch.Tensor:
    """fp32 contiguous [n, d] tensor on the device from numpy / DataFrame / tensor."""
    if hasattr(a, "values") and not isinstance(a, torch.Tensor):
        a = a.values
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


def _ks_pvalue(n1: int, n2: int, dmax: int, dmin: int):
    """(statistic, p-value) from the integer extremes, following scipy.stats.ks_2samp(method='auto'):
    exact when max(n1, n2) <= 10000, Smirnov's asymptotic formula otherwise (a scalar per feature)."""
    from scipy.stats import distributions
    from scipy.stats._stats_py import _attempt_exact_2kssamp
    d = max(dmax, -dmin, 0) / (float(n1) * float(n2))
    if max(n1, n2) <= 10000:
        ok, d2, prob = _attempt_exact_2kssamp(n1, n2, gcd(n1, n2), d, "two-sided")
        if ok:
            return float(d2), float(np.clip(prob, 0, 1))
    m, n = sorted([float(n1), float(n2)], reverse=True)
    return float(d), float(np.clip(distributions.kstwo.sf(d, np.round(m * n / (m + n))), 0, 1))


class BiologicalValidator:
    """utils/validation.py:18 -- device versions of the metrics named in the module docstring."""

    def __init__(self, config: dict, device: str = "cuda"):
        self.config = config
        ev = config.get("evaluation", {})
        self.driver_genes = ev.get("driver_genes", [])
        self.mutually_exclusive_pairs = ev.get("mutually_exclusive_pairs", [])
        self.required_correlations = ev.get("required_correlations", [])
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the validation kernels run on a ROCm device; there is no CPU fallback")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- utils/validation.py:273-298 ----------------------------------------------------------
    def compute_mmd(self, X, Y, kernel: str = "rbf", gamma: Optional[float] = None) -> float:
        if kernel != "rbf":
            raise ValueError("only the rbf kernel exists in the reference")
        x, y = _dev(X, self.device), _dev(Y, self.device)
        if x.shape[1] != y.shape[1]:
            raise ValueError("X and Y must have the same number of features")
        out = C.c_double()
        L.check(L.lib().osd_val_mmd(self._stream(), self._dev_index, L.ptr(x), x.shape[0], L.ptr(y), y.shape[0], x.shape[1],
                                    float(gamma) if gamma else 0.0, C.byref(out)))
        return float(out.value)

    def ks_tests(self, real_data, synthetic_data, max_features: int = 100):
        """Per-feature (statistic, p-value) arrays for the first min(D, 100) features."""
        r, s = _dev(real_data, self.device), _dev(synthetic_data, self.device)
        nf = min(r.shape[1], max_features)
        dmax, dmin = (C.c_int64 * nf)(), (C.c_int64 * nf)()
        L.check(L.lib().osd_val_ks_extremes(self._stream(), self._dev_index, L.ptr(r), r.shape[0], L.ptr(s), s.shape[0], r.shape[1], nf,
                                            dmax, dmin))
        res = [_ks_pvalue(r.shape[0], s.shape[0], int(dmax[i]), int(dmin[i])) for i in range(nf)]
        return np.array([d for d, _ in res]), np.array([p for _, p in res])

    # -- utils/validation.py:225-271 (KS + MMD; the PCA/Wasserstein part is not provided) --------
    def statistical_tests(self, real_data, synthetic_data) -> Dict[str, float]:
        logger.info("Running statistical tests...")
        _, pvals = self.ks_tests(real_data, synthetic_data)
        results = {"ks_test_mean_pvalue": float(np.mean(pvals)), "ks_test_fraction_significant": float((pvals < 0.05).mean())}
        results["mmd"] = self.compute_mmd(real_data, synthetic_data)
        logger.info(f"KS test mean p-value: {results['ks_test_mean_pvalue']:.3f}")
        logger.info(f"KS test fraction significant: {results['ks_test_fraction_significant']:.3f}")
        logger.info(f"MMD: {results['mmd']:.4f}")
        return results

    # -- utils/validation.py:125-175 -------------------------------------------------------------
    def _mean_offdiag(self, data: torch.Tensor, cols) -> float:
        arr = (C.c_int32 * len(cols))(*cols)
        out = C.c_double()
        L.check(L.lib().osd_val_mean_offdiag_corr(self._stream(), self._dev_index, L.ptr(data), data.shape[0], data.shape[1], arr,
                                                  len(cols), C.byref(out)))
        return float(out.value)

    def validate_pathway_coherence(self, real_data, synthetic_data, pathway_gene_matrix) -> Dict[str, float]:
        """real_data / synthetic_data: DataFrames with gene columns; pathway_gene_matrix: genes x pathways 0/1."""
        logger.info("Validating pathway coherence...")
        col_of = {g: i for i, g in enumerate(real_data.columns)}
        syn_of = {g: i for i, g in enumerate(synthetic_data.columns)}
        r, s = _dev(real_data, self.device), _dev(synthetic_data, self.device)
        real_scores, synth_scores = [], []
        for pathway in pathway_gene_matrix.columns[:10]:
            genes = pathway_gene_matrix[pathway_gene_matrix[pathway] == 1].index
            genes = [g for g in genes if g in col_of]
            if len(genes) < 3:
                continue
            real_scores.append(self._mean_offdiag(r, [col_of[g] for g in genes]))
            synth_scores.append(self._mean_offdiag(s, [syn_of[g] for g in genes]))
        results = {}
        if real_scores:
            results["real_pathway_coherence"] = float(np.mean(real_scores))
            results["synthetic_pathway_coherence"] = float(np.mean(synth_scores))
            results["pathway_coherence_correlation"] = float(np.corrcoef(real_scores, synth_scores)[0, 1])
        return results

    # -- utils/validation.py:177-223 -------------------------------------------------------------
    def validate_mutation_expression_correlation(self, mutations, expression, pathway_scores) -> Dict[str, float]:
        logger.info("Validating mutation-expression correlations...")
        mut, pw = _dev(mutations, self.device), _dev(pathway_scores, self.device)
        violations = total = 0
        for rule in self.required_correlations:
            gene, pathway, expected = rule["mutation"], rule["pathway"], rule["direction"]
            if gene not in mutations.columns or pathway not in pathway_scores.columns:
                continue
            gi, pi = list(mutations.columns).index(gene), list(pathway_scores.columns).index(pathway)
            out = C.c_double()
            L.check(L.lib().osd_val_pearson(self._stream(), self._dev_index, C.c_void_p(mut.data_ptr() + 4 * gi), mut.shape[1],
                                            C.c_void_p(pw.data_ptr() + 4 * pi), pw.shape[1], mut.shape[0], C.byref(out)))
            corr = out.value
            if (expected == "positive" and corr < 0) or (expected == "negative" and corr > 0):
                violations += 1
            total += 1
            logger.info(f"{gene} vs {pathway}: corr={corr:.3f} (expected: {expected})")
        return {"mutation_expression_violation_rate": violations / total} if total else {}
