"""GPU: the small-batch reverse-chain kernel (csrc/chain_squad.h: eight workgroups per 32 patients, activations handed between
them with agent-scope loads and stores behind a per-squad barrier, the chain state private to the workgroup that updates it)
against the CPU oracle (models/diffusion.py:382-449) and against the per-layer kernels.  It splits K differently from the other
engines (four waves per output tile, eight workgroups for input_proj), so the statement against them is the chain tolerance of
the split-K tests, not bit equality; against itself (segmented launches, repeated runs) it is bit equality."""
import pytest
import torch

from oracle import diffusion_oracle as O
from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, _lib as L
from helpers import FULL, FULL_H, assert_close, config

pytestmark = pytest.mark.gpu


def _model(T, hidden=FULL_H, seed=0, **dims):
    torch.manual_seed(seed)
    d = dict(FULL)
    d.update(dims)
    m = BiologyAwareDiffusionModel(config=config(hidden, T=T), **d).cuda().eval()
    m.input_splitk = 0
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                  # non-trivial GroupNorm affine
        for k, p in m.named_parameters():
            if k.endswith((".1.weight", ".5.weight")):
                p.copy_((1 + 0.2 * torch.randn(p.shape, generator=gen)).cuda())
            if k.endswith((".1.bias", ".5.bias")):
                p.copy_((0.1 * torch.randn(p.shape, generator=gen)).cuda())
    return m


def _get_option(m, name):
    import ctypes as C
    v = C.c_int64(-1)
    L.check(L.lib().osd_get_option(m._engine().handle, name.encode(), C.byref(v)))
    return v.value


def _squad(m, cond, n, panel=None, **kw):
    m.sampler, m.chain_variant, m.squad_panel = "chain", "squad", panel
    out, mask = m.sample(cond, n, return_mutation_mask=True, **kw)
    assert (m.last_sampler, m.last_chain_variant) == ("chain", "squad")
    if panel:
        assert m.last_squad_panel == panel
    return out, mask


def _graph(m, cond, n, **kw):
    m.sampler, m.chain_variant = "graph", None
    out, mask = m.sample(cond, n, return_mutation_mask=True, **kw)
    assert m.last_sampler == "graph"
    return out, mask


def _masks_agree(mask, ref_mask, ref, tol):
    """Masks may differ only where the reference value sits within the chain tolerance of the 0.5 threshold."""
    near = (ref[:, :mask.shape[1]] - 0.5).abs() <= tol
    return ((mask != ref_mask) & ~near).sum().item() == 0


def test_squad_chain_vs_oracle_with_injected_draws():
    """The oracle's x_T and z injected, T = 20: 100 rows (three full panels + 4 rows) at the BASELINE dims, then the reference's real
    dims 62 / 5054 / 26 (D = 5142: D % 4 = 2, 161 state tiles -- one workgroup of each squad owns 21, the others 20, i.e. a K-split
    left-over tile) with 45 rows; both panel sizes (chain_squad.h: 32 patients, 32x32x2 MFMA; chain_squad16.h: 16 patients, 16x16x4)."""
    T = 20
    for dims, n, seed, panel in ((dict(), 100, 6, 32), (dict(mutation_dim=62, expression_dim=5054, pathway_dim=26), 45, 3, 32),
                                 (dict(), 100, 6, 16), (dict(mutation_dim=62, expression_dim=5054, pathway_dim=26), 45, 3, 16)):
        m = _model(T, seed=seed, **dims)
        D = m.data_dim
        gen = torch.Generator().manual_seed(11)
        cond = torch.randn(n, 3, generator=gen)
        x_T = torch.randn(n, D, generator=gen)
        zs = torch.randn(T - 1, n, D, generator=gen)
        sd = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(("condition_embed", "unet"))}
        ref = O.sample(sd, O.schedule_buffers("cosine", T), cond, x_T, lambda t: zs[T - 1 - t], 3, 128)
        out, mask = _squad(m, cond.cuda(), n, panel, x_T=x_T.cuda(), noise=zs.cuda())
        assert_close(out, ref, 5e-5, atol=1e-5, what=f"squad chain ({panel}-patient panels) vs oracle, D = {D}")
        md = m.mutation_dim
        refm = (ref[:, :md] > 0.5).float()
        assert _masks_agree(mask.cpu(), refm, ref, 5e-5 * ref.abs().max().item() + 1e-5)


@pytest.mark.parametrize("dims,n,panel", [
    (dict(), 32, 32),                                                             # one full panel
    (dict(), 37, 32),                                                             # ... and five rows of a second
    (dict(), 999, 32),                                                            # 3 x 333 (utils/generate.py's small default): one workgroup per CU
    (dict(), 1500, 32),                                                           # two workgroups per CU
    (dict(), 3000, 32),                                                           # 3 x 1000: three workgroups per CU
    (dict(mutation_dim=62, expression_dim=5054, pathway_dim=26), 999, 32),        # the reference's real dims
    (dict(mutation_dim=10, expression_dim=487, pathway_dim=5), 300, 32),          # D = 502: 16 state tiles, two per workgroup (both K-split)
    (dict(mutation_dim=3, expression_dim=250, pathway_dim=3), 64, 32),            # D = 256: exactly one tile per workgroup
    (dict(), 16, 16), (dict(), 21, 16), (dict(), 999, 16), (dict(), 1024, 16),    # 16-patient panels: one panel ... two workgroups on every CU
    (dict(mutation_dim=62, expression_dim=5054, pathway_dim=26), 999, 16),        # 322 tiles of 16 features: 40 or 41 per workgroup
    (dict(mutation_dim=10, expression_dim=487, pathway_dim=5), 300, 16),          # D = 502: 32 tiles, whole rounds only
    (dict(mutation_dim=3, expression_dim=250, pathway_dim=3), 64, 16),            # D = 256: two tiles per workgroup, both K-split
    (dict(mutation_dim=3, expression_dim=330, pathway_dim=3), 40, 16),            # D = 336: 21 tiles: 2 or 3 per workgroup
])
def test_squad_chain_agrees_with_the_per_layer_kernels(dims, n, panel):
    """Philox draws (same addressing in both engines), T = 8, row_offset != 0: chain tolerance against the per-layer kernels, masks
    equal away from the threshold, and bit equality with itself -- a second run, and the chain cut into launches of 3 steps (the
    state goes out row-major and comes back in between)."""
    T = 8
    m = _model(T, seed=5, **dims)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    ref, refm = _graph(m, cond, n, seed=77, row_offset=5)
    out, mask = _squad(m, cond, n, panel, seed=77, row_offset=5)
    assert torch.isfinite(out).all()
    assert_close(out, ref, 2e-5, atol=1e-6, what=f"squad chain vs per-layer kernels, n = {n}")
    assert _masks_agree(mask, refm, ref, 2e-5 * ref.abs().max().item() + 1e-6)
    again, mask2 = _squad(m, cond, n, panel, seed=77, row_offset=5)
    assert torch.equal(again, out) and torch.equal(mask2, mask)
    m.chain_steps_per_launch = 3
    cut, mask3 = _squad(m, cond, n, panel, seed=77, row_offset=5)
    assert torch.equal(cut, out) and torch.equal(mask3, mask)


def test_auto_picks_the_squad_chain_for_small_batches_only():
    """auto in the small-batch mode (input_splitk != 0: SyntheticPatientGenerator's default for its own model): batches whose
    squads are all resident at once (8 workgroups per 32 patients, three workgroups per CU: 3 072 rows on 256 CUs) run on the squad
    chain; larger ones keep their engines; the library default (input_splitk = 0: results independent of the batch size bit for
    bit) never picks it; models outside its decomposition (H0 != 256, a 128-wide block, fewer than 8 state tiles) never see it; an
    explicit sampler = "chain" keeps the bit-identical chain kernels."""
    T = 4
    m = _model(T, seed=1)
    cond = torch.randn(6144, 3, generator=torch.Generator().manual_seed(1)).cuda()
    m.sampler, m.chain_variant = "auto", None
    m.sample(cond[:333], 333, seed=2)                 # input_splitk = 0 (from _model)
    assert m.last_sampler == "graph"
    m.input_splitk = -1
    for n, want, panel in ((333, ("chain", "squad"), 16), (1024, ("chain", "squad"), 16), (1025, ("chain", "squad"), 32), (3000, ("chain", "squad"), 32),
                           (6144, ("graph", None), None)):
        m.sample(cond[:n], n, seed=2)
        assert (m.last_sampler, m.last_chain_variant) == want, n
        assert m.last_squad_panel == panel, n          # 16-patient panels while they are at most two workgroups per CU (256 CUs: 1 024 rows)
    assert _get_option(m, "squad_chain_supported") == 1
    m.sampler = "chain"
    m.sample(cond[:333], 333, seed=2)
    assert (m.last_sampler, m.last_chain_variant) == ("chain", "workspace")
    # asked for by name, batch too large: the other engines
    m.sampler, m.chain_variant = "auto", "squad"
    m.sample(cond, 6144, seed=2)
    assert m.last_sampler == "graph"
    for hidden, dims in (([512, 256, 512], dict()), ([256, 128, 256], dict()), (FULL_H, dict(mutation_dim=10, expression_dim=100, pathway_dim=10))):
        m2 = _model(T, hidden=hidden, seed=2, **dims)
        assert _get_option(m2, "squad_chain_supported") == 0
        m2.sampler, m2.chain_variant = "auto", "squad"
        m2.sample(cond[:100], 100, seed=2)
        assert m2.last_chain_variant != "squad"


def test_squad_chain_spin_timeout_is_recovered_on_the_per_layer_kernels():
    """models/diffusion.py:427-449 cannot fail.  A spin budget of one tick (10 ns) makes the first squad barrier that has to wait
    give up: the workgroup raises the status word and leaves, its partners see the word in their own polls and leave too, the kernel
    drains; the synchronous call re-runs the chain on the per-layer kernels from the same x_T / seed and warns."""
    T, n = 12, 200
    m = _model(T, seed=6)
    gen = torch.Generator().manual_seed(4)
    cond = torch.randn(n, 3, generator=gen).cuda()
    x_T = torch.randn(n, 2000, generator=gen).cuda()
    ref, ref_mask = _graph(m, cond, n, x_T=x_T, seed=31, row_offset=7)
    m.sampler, m.chain_variant, m.chain_spin_budget = "chain", "squad", 1
    with pytest.warns(UserWarning, match="re-run on the per-layer kernels"):
        out, mask = m.sample(cond, n, x_T=x_T, seed=31, row_offset=7, return_mutation_mask=True)
    assert m.last_sampler == "graph" and m.last_chain_variant == "squad"
    assert _get_option(m, "chain_fallbacks") == 1 and _get_option(m, "last_engine") == 0
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)          # the re-run IS the per-layer engine
    # with a sane budget the same call completes on the squad chain
    m.chain_spin_budget = 500_000_000
    out2, _ = _squad(m, cond, n, x_T=x_T, seed=31, row_offset=7)
    assert_close(out2, ref, 2e-5, atol=1e-6, what="squad chain after the recovered one")
    assert _get_option(m, "chain_fallbacks") == 1


def test_squad_chain_wall_clock_budget_aborts_and_recovers():
    """The host side of the same guarantee: budget 1 ms against a chain of ~100 ms; the host raises the abort word, the result of
    the chain is discarded whether or not its workgroups saw the word in time, and the per-layer kernels re-run it."""
    T, n = 400, 3000
    m = _model(T, seed=8)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).cuda()
    ref, ref_mask = _graph(m, cond, n, seed=5)
    m.sampler, m.chain_variant, m.chain_wall_budget_ms = "chain", "squad", 1
    with pytest.warns(UserWarning):
        out, mask = m.sample(cond, n, seed=5, return_mutation_mask=True)
    assert m.last_sampler == "graph" and m.last_chain_variant == "squad" and _get_option(m, "chain_fallbacks") == 1
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)


def test_squad_chain_follows_the_parameters():
    """The fragment-ordered weight copies are remade after anything that changes the parameters (load_state_dict, a training
    step): a chain after such a change agrees with the per-layer kernels on the NEW parameters."""
    T, n = 6, 96
    m = _model(T, seed=3)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(2)).cuda()
    _squad(m, cond, n, seed=1)
    other = _model(T, seed=9)
    m.load_state_dict(other.state_dict())
    ref, _ = _graph(m, cond, n, seed=1)
    out, _ = _squad(m, cond, n, seed=1)
    assert_close(out, ref, 2e-5, atol=1e-6, what="squad chain after load_state_dict")
    ref_other, _ = _graph(other, cond, n, seed=1)
    assert torch.equal(ref, ref_other)
