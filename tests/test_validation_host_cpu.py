"""CPU: the vectorised host statistics of validate_all (osteosarcoma_diffusionmodel_amd/validation.py) against the per-item scipy
calls the reference makes -- chi2_contingency on pd.crosstab tables (utils/validation.py:98-108) and ks_2samp's p-value branches
(:238-249).  Same values, not approximations: the closed forms restate what scipy computes for these shapes."""
import numpy as np
import pytest

from osteosarcoma_diffusionmodel_amd.validation import BiologicalValidator, _chi2_pairs, _ks_pvalue, _ks_pvalues


@pytest.mark.parametrize("n", [37, 1000, 125000])
def test_chi2_pairs_equals_scipy_chi2_contingency(n):
    rs = np.random.RandomState(n)
    x = (rs.rand(n, 14) < rs.rand(14)).astype(np.float64)
    x[:, 3] = 0.0            # constant columns: crosstab has one row / column, zero degrees of freedom
    x[:, 7] = 1.0
    x[:, 9] = x[:, 8]        # perfectly dependent pair (Yates' correction at work)
    g = x.T @ x
    ref = np.array([BiologicalValidator._chi2(n, int(g[i, i]), int(g[j, j]), int(g[i, j])) for i in range(14) for j in range(i + 1, 14)])
    got = _chi2_pairs(n, g)
    assert got.shape == ref.shape and np.all(np.isfinite(got))
    np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-13)


def test_chi2_pairs_matches_crosstab_route():
    pd = pytest.importorskip("pandas")
    from scipy import stats
    rs = np.random.RandomState(5)
    n = 400
    x = (rs.rand(n, 6) < 0.3).astype(np.int64)
    g = (x.T @ x).astype(np.float64)
    ref = [stats.chi2_contingency(pd.crosstab(x[:, i], x[:, j]))[0] for i in range(6) for j in range(i + 1, 6)]
    np.testing.assert_allclose(_chi2_pairs(n, g), ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("n1,n2", [(60, 101), (9000, 10000), (20000, 9000), (125000, 125000)])
def test_ks_pvalues_equal_the_per_feature_calls(n1, n2):
    rs = np.random.RandomState(n1 % 97)
    hi = max(n1 * n2 // 40, 2)
    dmax = rs.randint(0, hi, size=25)
    dmin = -rs.randint(0, hi, size=25)
    d, p = _ks_pvalues(n1, n2, dmax, dmin)
    ref = [_ks_pvalue(n1, n2, int(a), int(b)) for a, b in zip(dmax, dmin)]
    np.testing.assert_array_equal(d, np.array([r[0] for r in ref]))
    np.testing.assert_array_equal(p, np.array([r[1] for r in ref]))
