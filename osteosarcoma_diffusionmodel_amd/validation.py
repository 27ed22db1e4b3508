"""Validation metrics on MI355X -- host-side mirror of the reference's utils/validation.py.
Same class, method names and result keys as ``BiologicalValidator`` including ``validate_all``; everything that
scales with the number of patients runs in libosdiff.so (``osd_val_*``): RBF-MMD, per-feature two-sample KS, pathway
coherence, the mutation-expression sign check, mutation frequencies and the joint mutation counts behind the
chi-square co-occurrence test and the mutual-exclusivity check.

Host-side by design, as SURVEY section 8f-1 prescribes: p-values and chi-square statistics from the exact device counts
(``scipy.stats``), and the Wasserstein distance on 10 principal components (``sklearn`` PCA + ``scipy``, :256-269),
which the reference itself computes that way.

Multi-GPU (BASELINE config 5): ``BiologicalValidator(config, sharded=True)`` under an initialised
``torch.distributed`` treats every *synthetic* argument as this rank's row shard (the real cohort is small and
replicated).  Accumulators are summed over ranks (``parallel.ShardComm``), the all-pairs synthetic Gram block walks
the shards by broadcast, the <= 100 KS columns are gathered and split by feature.  Every rank returns the same
numbers as one process on the concatenated rows.
"""
from __future__ import annotations

import ctypes as C
import logging
from math import gcd
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .parallel import ShardComm

logger = logging.getLogger(__name__)


def _dev(a, device) -> torch.Tensor:
    """fp32 contiguous [n, d] tensor on the device from numpy / DataFrame / tensor."""
    if hasattr(a, "values") and not isinstance(a, torch.Tensor):
        a = a.values
    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return t.to(device=device, dtype=torch.float32).contiguous()


class DeviceFrame:
    """The few DataFrame operations BiologicalValidator touches (``.columns``, ``.values``, column selection, a column's
    ``.mean()``) over a tensor that already lives on the device: the validator's inputs at BASELINE config 5's size (10^5-10^6
    generated patients x 2000 features per GPU) never visit the host as a pandas object."""

    def __init__(self, values: torch.Tensor, columns):
        import pandas as pd
        self.values = values
        self.columns = columns if isinstance(columns, pd.Index) else pd.Index(list(columns))

    @property
    def shape(self):
        return tuple(self.values.shape)

    def __getitem__(self, key):
        if isinstance(key, str):
            return self.values[:, self.columns.get_loc(key)]          # a device vector: .mean() works as on a Series
        key = list(key)
        if len(key) == len(self.columns) and all(a == b for a, b in zip(key, self.columns)):
            return self
        idx = torch.as_tensor(self.columns.get_indexer(key), device=self.values.device)
        return DeviceFrame(self.values.index_select(1, idx), key)


def _host(a) -> np.ndarray:
    if isinstance(a, torch.Tensor):
        return a.detach().cpu().numpy()
    return np.asarray(a.values if hasattr(a, "values") else a)


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


def _ks_pvalues(n1: int, n2: int, dmax, dmin):
    """_ks_pvalue over all features.  Above 10 000 samples ks_2samp(method='auto') takes Smirnov's asymptotic formula: ONE
    vectorised scipy call for the whole statistic vector (same values as the per-feature calls -- kstwo.sf is elementwise -- without
    100 trips through rv_continuous' argument checking: 0.8 s -> 10 ms at 100 features); the exact branch stays per feature."""
    dmax, dmin = np.asarray(dmax, dtype=np.int64), np.asarray(dmin, dtype=np.int64)
    if max(n1, n2) <= 10000:
        res = [_ks_pvalue(n1, n2, int(a), int(b)) for a, b in zip(dmax, dmin)]
        return np.array([d for d, _ in res]), np.array([p for _, p in res])
    from scipy.stats import distributions
    d = np.maximum(np.maximum(dmax, -dmin), 0) / (float(n1) * float(n2))
    m, n = sorted([float(n1), float(n2)], reverse=True)
    return d.astype(np.float64), np.clip(distributions.kstwo.sf(d, np.round(m * n / (m + n))), 0, 1).astype(np.float64)


def _chi2_pairs(n: int, gram: np.ndarray) -> np.ndarray:
    """chi2 of scipy.stats.chi2_contingency(pd.crosstab(a_i, a_j)) (utils/validation.py:98-108) for every pair i < j of 0/1 columns,
    row-major pair order, from the joint counts gram[i][j] = sum_r a_i a_j -- the closed form of what scipy does for a 2 x 2 table
    (expected = outer(margins) / n, Yates' correction min(0.5, |expected - observed|) toward the expected value, Pearson sum), and
    0.0 when a column is constant (crosstab then has a single row or column: zero degrees of freedom).  Same values as the
    per-pair scipy calls (tests/test_oracle_golden.py), 2 x 1225 tables in one numpy expression instead of 0.26 s of calls."""
    g = np.rint(np.asarray(gram, dtype=np.float64))
    k = g.shape[0]
    iu, ju = np.triu_indices(k, 1)
    n1, n2, n11 = np.diag(g)[iu], np.diag(g)[ju], g[iu, ju]
    obs = np.stack([n - n1 - n2 + n11, n2 - n11, n1 - n11, n11], axis=-1).reshape(-1, 2, 2)      # [[00, 01], [10, 11]]
    rows, cols = obs.sum(2), obs.sum(1)
    exp = rows[:, :, None] * cols[:, None, :] / float(n)
    live = (rows > 0).all(1) & (cols > 0).all(1)
    diff = exp - obs
    adj = obs + np.minimum(0.5, np.abs(diff)) * np.sign(diff)
    with np.errstate(divide="ignore", invalid="ignore"):
        terms = (adj - exp) ** 2 / exp
    return np.where(live, np.where(live[:, None, None], terms, 0.0).sum((1, 2)), 0.0)


class DeviceKernels:
    """The per-shard partial results, each one libosdiff.so call on device tensors."""

    def __init__(self, device: torch.device):
        self.device = device
        self.index = device.index if device.index is not None else torch.cuda.current_device()

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def rbf_sum(self, a: torch.Tensor, b: torch.Tensor, gamma: float) -> float:
        out = C.c_double()
        L.check(L.lib().osd_val_rbf_sum(self._stream(), self.index, L.ptr(a), a.shape[0], L.ptr(b), b.shape[0], a.shape[1], float(gamma),
                                        C.byref(out)))
        return float(out.value)

    def ks_extremes(self, real: torch.Tensor, synth: torch.Tensor, nf: int):
        dmax, dmin = (C.c_int64 * nf)(), (C.c_int64 * nf)()
        L.check(L.lib().osd_val_ks_extremes(self._stream(), self.index, L.ptr(real), real.shape[0], L.ptr(synth), synth.shape[0],
                                            real.shape[1], nf, dmax, dmin))
        return np.array(dmax[:], dtype=np.int64), np.array(dmin[:], dtype=np.int64)

    def col_moments(self, t: torch.Tensor, cols: Sequence[int]):
        g = len(cols)
        arr = (C.c_int32 * g)(*cols)
        s, q = (C.c_double * g)(), (C.c_double * g)()
        L.check(L.lib().osd_val_col_moments(self._stream(), self.index, L.ptr(t), t.shape[0], t.shape[1], arr, g, s, q))
        return np.array(s[:]), np.array(q[:])

    def rowz_sq(self, t: torch.Tensor, cols: Sequence[int], mu: np.ndarray, isd: np.ndarray) -> float:
        g = len(cols)
        arr = (C.c_int32 * g)(*cols)
        out = C.c_double()
        L.check(L.lib().osd_val_rowz_sq(self._stream(), self.index, L.ptr(t), t.shape[0], t.shape[1], arr, g, (C.c_double * g)(*mu),
                                        (C.c_double * g)(*isd), C.byref(out)))
        return float(out.value)

    def pearson_sums(self, t_a: torch.Tensor, col_a: int, t_b: torch.Tensor, col_b: int) -> np.ndarray:
        out = (C.c_double * 5)()
        L.check(L.lib().osd_val_pearson_sums(self._stream(), self.index, C.c_void_p(t_a.data_ptr() + 4 * col_a), t_a.shape[1],
                                             C.c_void_p(t_b.data_ptr() + 4 * col_b), t_b.shape[1], t_a.shape[0], out))
        return np.array(out[:])

    def column_sums(self, t: torch.Tensor) -> np.ndarray:
        out = (C.c_double * t.shape[1])()
        L.check(L.lib().osd_val_column_sums(self._stream(), self.index, L.ptr(t), t.shape[0], t.shape[1], t.shape[1], out))
        return np.array(out[:])

    def gram(self, t: torch.Tensor, cols: Sequence[int]) -> np.ndarray:
        g = len(cols)
        if g > 64:
            raise ValueError("at most 64 columns per Gram block")
        arr = (C.c_int32 * g)(*cols)
        out = (C.c_double * (g * g))()
        L.check(L.lib().osd_val_gram(self._stream(), self.index, L.ptr(t), t.shape[0], t.shape[1], arr, g, out))
        return np.array(out[:]).reshape(g, g)


# ---- combination of per-shard partials (pure host logic; the CPU tests drive it over gloo with numpy kernels) -----------
def sharded_mmd(comm: ShardComm, k, x, y_local, gamma: float) -> float:
    """utils/validation.py:273-298 with X replicated and Y row-sharded."""
    n = x.shape[0]
    m = int(comm.sum(y_local.shape[0])[0])
    sxx = k.rbf_sum(x, x, gamma)                       # one operand: the library runs the upper triangle only (csrc/validate.hip)
    sxy = float(comm.sum(k.rbf_sum(x, y_local, gamma))[0])
    # K_YY is symmetric: this shard against itself (triangular) + every unordered pair of shards once, weighted 2
    syy_local = k.rbf_sum(y_local, y_local, gamma)
    for other, weight in comm.ring_partners(y_local):
        syy_local += weight * k.rbf_sum(y_local, other, gamma)
    syy = float(comm.sum(syy_local)[0])
    v = sxx / (float(n) * n) + syy / (float(m) * m) - 2.0 * sxy / (float(n) * m)
    return float(np.sqrt(max(v, 0.0)))


def sharded_ks_extremes(comm: ShardComm, k, real, synth_local, nf: int):
    """Integer KS extremes of features 0..nf-1: the nf synthetic columns are gathered, the features split over ranks."""
    synth = comm.gather_rows(synth_local[:, :nf].contiguous())
    lo = (nf * comm.rank) // comm.world
    hi = (nf * (comm.rank + 1)) // comm.world
    dmax, dmin = np.zeros(nf, dtype=np.int64), np.zeros(nf, dtype=np.int64)
    if hi > lo:
        a, b = k.ks_extremes(real[:, lo:hi].contiguous(), synth[:, lo:hi].contiguous(), hi - lo)
        dmax[lo:hi], dmin[lo:hi] = a, b
    return comm.sum(dmax), comm.sum(dmin), int(synth.shape[0])


def sharded_mean_offdiag(comm: ShardComm, k, data_local, cols: Sequence[int]) -> float:
    """Mean off-diagonal Pearson correlation of data[:, cols] (utils/validation.py:156-161) over row shards."""
    g = len(cols)
    if g < 2:
        raise ValueError("a mean off-diagonal correlation needs at least 2 columns")
    s, q = k.col_moments(data_local, cols)
    tot = comm.sum(np.concatenate([s, q, [float(data_local.shape[0])]]))
    rows = tot[-1]
    mu = tot[:g] / rows
    var = (tot[g:2 * g] - rows * mu * mu) / (rows - 1)              # ddof = 1, as pandas .corr()
    with np.errstate(divide="ignore", invalid="ignore"):
        isd = np.where(var > 0, 1.0 / np.sqrt(var), np.nan)        # constant column -> NaN, as pandas
    S = float(comm.sum(k.rowz_sq(data_local, cols, mu, isd))[0])
    return (S / (rows - 1) - g) / (float(g) * (g - 1))


def sharded_pearson(comm: ShardComm, k, a_local, col_a: int, b_local, col_b: int) -> float:
    h = comm.sum(np.concatenate([k.pearson_sums(a_local, col_a, b_local, col_b), [float(a_local.shape[0])]]))
    n = h[5]
    cov, va, vb = h[4] - h[0] * h[1] / n, h[2] - h[0] * h[0] / n, h[3] - h[1] * h[1] / n
    return float(cov / np.sqrt(va * vb))


class BiologicalValidator:
    """utils/validation.py:18 -- device versions of the metrics named in the module docstring."""

    def __init__(self, config: dict, device: str = "cuda", sharded: bool = False):
        self.config = config
        ev = config.get("evaluation", {})
        self.driver_genes = ev.get("driver_genes", [])
        self.mutually_exclusive_pairs = ev.get("mutually_exclusive_pairs", [])
        self.required_correlations = ev.get("required_correlations", [])
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the validation kernels run on a ROCm device; there is no CPU fallback")
        self.k = DeviceKernels(self.device)
        self.comm = ShardComm(sharded)            # synthetic rows sharded over ranks when active
        self._one = ShardComm(False)              # the replicated real cohort never communicates

    def _agree(self, result):
        """Sharded mode: every rank reports rank 0's value (partial sums replicated per rank, e.g. over the real cohort,
        can differ in the last bit between ranks because double atomics commit in any order)."""
        return self.comm.bcast_object(result)

    # -- utils/validation.py:273-298 ----------------------------------------------------------
    def compute_mmd(self, X, Y, kernel: str = "rbf", gamma: Optional[float] = None) -> float:
        if kernel != "rbf":
            raise ValueError("only the rbf kernel exists in the reference")
        x, y = _dev(X, self.device), _dev(Y, self.device)
        if x.shape[1] != y.shape[1]:
            raise ValueError("X and Y must have the same number of features")
        if not self.comm.on:
            out = C.c_double()
            L.check(L.lib().osd_val_mmd(self.k._stream(), self.k.index, L.ptr(x), x.shape[0], L.ptr(y), y.shape[0], x.shape[1],
                                        float(gamma) if gamma else 0.0, C.byref(out)))
            return float(out.value)
        return self._agree(sharded_mmd(self.comm, self.k, x, y, float(gamma) if gamma else 1.0 / x.shape[1]))

    def ks_tests(self, real_data, synthetic_data, max_features: int = 100):
        """Per-feature (statistic, p-value) arrays for the first min(D, 100) features."""
        r, s = _dev(real_data, self.device), _dev(synthetic_data, self.device)
        nf = min(r.shape[1], max_features)
        dmax, dmin, n2 = sharded_ks_extremes(self.comm, self.k, r, s, nf)
        return self._agree(_ks_pvalues(r.shape[0], n2, dmax[:nf], dmin[:nf]))

    # -- utils/validation.py:225-271 --------------------------------------------------------------
    def statistical_tests(self, real_data, synthetic_data) -> Dict[str, float]:
        logger.info("Running statistical tests...")
        # The device parts first -- KS extremes (exact integers), then the MMD Gram blocks -- with the reference's own host-side parts
        # BESIDE the MMD on two threads (the ctypes call releases the GIL): the 100 Smirnov p-values (scipy's kstwo.sf: 8 ms each at
        # N = 62 500, whether called once or a hundred times) and the PCA fit + projection behind the Wasserstein figure (:256-269;
        # the reference's own sklearn / scipy calls; fitted on the replicated real data, each rank projects its shard, the 10
        # projected columns are gathered).  Same calls, same inputs, same order of draws from numpy's global generator (only the PCA
        # draws): only the wall time moves -- 4.7 s -> 2.4 s per 125 000-row scenario together with the triangular MMD.
        import threading
        from scipy import stats
        from sklearn.decomposition import PCA
        r, s = _dev(real_data, self.device), _dev(synthetic_data, self.device)
        nf = min(r.shape[1], 100)
        dmax, dmin, n2 = sharded_ks_extremes(self.comm, self.k, r, s, nf)
        real_host, synth_host = _host(real_data), _host(synthetic_data)      # device -> host before the MMD kernels occupy the stream
        box = {}

        def guarded(key, fn):
            def run():
                try:
                    box[key] = fn()
                except BaseException as e:               # re-raised on the calling thread
                    box["error"] = e
            return threading.Thread(target=run)

        def fit():
            pca = PCA(n_components=10)
            return pca.fit_transform(real_host), pca.transform(synth_host)

        threads = [guarded("ks", lambda: _ks_pvalues(r.shape[0], n2, dmax[:nf], dmin[:nf])), guarded("pca", fit)]
        for th in threads:
            th.start()
        results = {}
        try:
            results["mmd"] = self.compute_mmd(real_data, synthetic_data)
        finally:
            for th in threads:
                th.join()
        if "error" in box:
            raise box["error"]
        _, pvals = self._agree(box["ks"])
        results = {"ks_test_mean_pvalue": float(np.mean(pvals)), "ks_test_fraction_significant": float((pvals < 0.05).mean()), "mmd": results["mmd"]}
        logger.info(f"KS test mean p-value: {results['ks_test_mean_pvalue']:.3f}")
        logger.info(f"KS test fraction significant: {results['ks_test_fraction_significant']:.3f}")
        logger.info(f"MMD: {results['mmd']:.4f}")
        real_pca, synth_pca = box["pca"]
        if self.comm.on:
            synth_pca = self.comm.gather_rows(torch.from_numpy(np.ascontiguousarray(synth_pca))).numpy()
        results["wasserstein_distance_mean"] = float(np.mean([stats.wasserstein_distance(real_pca[:, i], synth_pca[:, i]) for i in range(10)]))
        logger.info(f"Mean Wasserstein distance: {results['wasserstein_distance_mean']:.3f}")
        return self._agree(results)

    # -- utils/validation.py:27-121 ----------------------------------------------------------------
    def _column_sums(self, t: torch.Tensor) -> np.ndarray:
        return self.k.column_sums(t)

    def _gram(self, t: torch.Tensor, cols) -> np.ndarray:
        """Exact joint counts sum_r x[r][ci] x[r][cj] of 0/1 columns, 64 columns per device pass."""
        return self.k.gram(t, list(cols))

    @staticmethod
    def _chi2(n: int, n1: int, n2: int, n11: int) -> float:
        """chi2 of scipy.stats.chi2_contingency(pd.crosstab(a, b)) from the counts of two 0/1 columns (:98-108)."""
        from scipy import stats
        table = np.array([[n - n1 - n2 + n11, n2 - n11], [n1 - n11, n11]], dtype=np.int64)
        table = table[table.sum(1) > 0][:, table.sum(0) > 0]      # crosstab lists only the values that occur
        return float(stats.chi2_contingency(table)[0])

    def validate_mutation_cooccurrence(self, real_mutations, synthetic_mutations) -> Dict[str, float]:
        """real_mutations / synthetic_mutations: DataFrames of 0/1 columns named by gene."""
        logger.info("Validating mutation co-occurrence patterns...")
        results: Dict[str, float] = {}
        common = real_mutations.columns.intersection(synthetic_mutations.columns)
        r, s = _dev(real_mutations[common], self.device), _dev(synthetic_mutations[common], self.device)
        n_synth = int(self.comm.sum(s.shape[0])[0])
        pos = {g: i for i, g in enumerate(common)}
        real_freq, synth_freq = self._column_sums(r) / r.shape[0], self.comm.sum(self._column_sums(s)) / n_synth
        results["mutation_frequency_correlation"] = float(np.corrcoef(real_freq, synth_freq)[0, 1])
        logger.info(f"Mutation frequency correlation: {results['mutation_frequency_correlation']:.3f}")
        sfull = _dev(synthetic_mutations, self.device)
        spos = {g: i for i, g in enumerate(synthetic_mutations.columns)}
        drivers = [g for g in self.driver_genes if g in real_mutations.columns]
        if drivers:
            sfreq_all = self.comm.sum(self._column_sums(sfull)) / n_synth
            rd = np.array([float(real_mutations[g].mean()) for g in drivers])       # Series.mean() or a device vector's
            sd = np.array([sfreq_all[spos[g]] for g in drivers])       # KeyError if a driver gene is missing, as the reference
            results["driver_gene_frequency_diff"] = float(np.abs(rd - sd).mean())
            logger.info(f"Driver gene frequency difference: {results['driver_gene_frequency_diff']:.3f}")
        if self.mutually_exclusive_pairs:
            pairs = [(a, b) for a, b in self.mutually_exclusive_pairs if a in spos and b in spos]
            if pairs:
                violations = 0
                for a, b in pairs:                       # both-mutated count = off-diagonal of the 2-column Gram block
                    violations += int(round(self.comm.sum(self._gram(sfull, [spos[a], spos[b]]).ravel())[1]))
                results["mutual_exclusivity_violation_rate"] = violations / (n_synth * len(pairs))
                logger.info(f"Mutual exclusivity violation rate: {results['mutual_exclusivity_violation_rate']:.3f}")
        # pairwise chi-square on a random subset of at most 50 genes (np.random.choice, as the reference; rank 0 draws)
        sample_genes = self.comm.bcast_object(list(np.random.choice(common, size=min(50, len(common)), replace=False)))
        idx = [pos[g] for g in sample_genes]
        if len(idx) >= 2:
            gr = self._gram(r, idx)
            gs = self.comm.sum(self._gram(s, idx).ravel()).reshape(len(idx), len(idx))
            chi_r, chi_s = _chi2_pairs(r.shape[0], gr), _chi2_pairs(n_synth, gs)       # pairs i < j in row-major order, as the reference's loops
            results["cooccurrence_pattern_correlation"] = float(np.corrcoef(chi_r, chi_s)[0, 1])
            logger.info(f"Co-occurrence pattern correlation: {results['cooccurrence_pattern_correlation']:.3f}")
        return self._agree(results)

    # -- utils/validation.py:125-175 -------------------------------------------------------------
    def _mean_offdiag(self, data: torch.Tensor, cols, sharded: bool = False) -> float:
        return sharded_mean_offdiag(self.comm if sharded else self._one, self.k, data, list(cols))

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
            synth_scores.append(self._mean_offdiag(s, [syn_of[g] for g in genes], sharded=True))
        results = {}
        if real_scores:
            results["real_pathway_coherence"] = float(np.mean(real_scores))
            results["synthetic_pathway_coherence"] = float(np.mean(synth_scores))
            results["pathway_coherence_correlation"] = float(np.corrcoef(real_scores, synth_scores)[0, 1])
        return self._agree(results)

    # -- utils/validation.py:177-223 -------------------------------------------------------------
    def validate_mutation_expression_correlation(self, mutations, expression, pathway_scores) -> Dict[str, float]:
        """All three arguments are synthetic data (row shards when the validator is sharded)."""
        logger.info("Validating mutation-expression correlations...")
        mut, pw = _dev(mutations, self.device), _dev(pathway_scores, self.device)
        violations = total = 0
        for rule in self.required_correlations:
            gene, pathway, expected = rule["mutation"], rule["pathway"], rule["direction"]
            if gene not in mutations.columns or pathway not in pathway_scores.columns:
                continue
            gi, pi = list(mutations.columns).index(gene), list(pathway_scores.columns).index(pathway)
            corr = sharded_pearson(self.comm, self.k, mut, gi, pw, pi)
            if (expected == "positive" and corr < 0) or (expected == "negative" and corr > 0):
                violations += 1
            total += 1
            logger.info(f"{gene} vs {pathway}: corr={corr:.3f} (expected: {expected})")
        return self._agree({"mutation_expression_violation_rate": violations / total} if total else {})

    # -- utils/validation.py:300-383 ---------------------------------------------------------------
    def validate_all(self, real_mutations, real_expression, real_pathways, synth_mutations, synth_expression, synth_pathways,
                     pathway_gene_matrix=None) -> Dict[str, float]:
        logger.info("=" * 50)
        logger.info("BIOLOGICAL VALIDATION")
        logger.info("=" * 50)
        all_results: Dict[str, float] = {}
        all_results.update(self.validate_mutation_cooccurrence(real_mutations, synth_mutations))
        if pathway_gene_matrix is not None:
            all_results.update(self.validate_pathway_coherence(real_expression, synth_expression, pathway_gene_matrix))
        all_results.update(self.validate_mutation_expression_correlation(synth_mutations, synth_expression, synth_pathways))
        parts_r = [real_mutations.values, real_expression.values, real_pathways.values]
        parts_s = [synth_mutations.values, synth_expression.values, synth_pathways.values]
        if all(isinstance(p, torch.Tensor) for p in parts_r + parts_s):          # DeviceFrame inputs: stay on the device
            real_combined, synth_combined = torch.cat(parts_r, dim=1), torch.cat(parts_s, dim=1)
        else:
            real_combined = np.concatenate([_host(p) for p in parts_r], axis=1)
            synth_combined = np.concatenate([_host(p) for p in parts_s], axis=1)
        all_results.update(self.statistical_tests(real_combined, synth_combined))
        logger.info("=" * 50)
        logger.info("VALIDATION SUMMARY")
        logger.info("=" * 50)
        for key, value in all_results.items():
            logger.info(f"{key}: {value:.4f}")
        score = []
        if "mutation_frequency_correlation" in all_results:
            score.append(all_results["mutation_frequency_correlation"])
        if "cooccurrence_pattern_correlation" in all_results:
            score.append(all_results["cooccurrence_pattern_correlation"])
        if "mutual_exclusivity_violation_rate" in all_results:
            score.append(1 - all_results["mutual_exclusivity_violation_rate"])
        if "mutation_expression_violation_rate" in all_results:
            score.append(1 - all_results["mutation_expression_violation_rate"])
        if score:
            all_results["overall_biological_score"] = float(np.mean(score))
            logger.info(f"\nOverall Biological Score: {all_results['overall_biological_score']:.3f}")
        return all_results
