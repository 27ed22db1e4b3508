"""GPU: the persistent reverse-chain kernels (csrc/chain.h: 128-row tiles through a workspace; csrc/chain_panel.h: 64 patients
resident in LDS) -- every layer and every step of sample() in one launch, row
tiles handed between workgroups through agent-scope release / acquire -- against the per-layer kernels (bitwise: same
tile loop, same epilogues, same Philox addressing) and against the CPU oracle (models/diffusion.py:382-449).

The workgroup count is capped (``chain_grid``) below the tile count so that every step of a tile runs on a different
workgroup than the previous one, i.e. every x_t crosses a hand-off; row counts are ragged (last tile partly filled)."""
import numpy as np
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
    # bit equality between the two engines is a statement about the single-pass input_proj (the library default); the small-batch
    # split-K path of the per-layer engine (SyntheticPatientGenerator's default) is another fp32 summation order, covered by
    # test_input_splitk_small_batches below
    m.input_splitk = 0
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                  # non-trivial GroupNorm affine
        for k, p in m.named_parameters():
            if k.endswith((".1.weight", ".5.weight")):
                p.copy_((1 + 0.2 * torch.randn(p.shape, generator=gen)).cuda())
            if k.endswith((".1.bias", ".5.bias")):
                p.copy_((0.1 * torch.randn(p.shape, generator=gen)).cuda())
    return m


def _run(m, cond, n, sampler, variant=None, **kw):
    m.sampler, m.chain_variant = sampler, variant
    out, mask = m.sample(cond, n, return_mutation_mask=True, **kw)
    assert m.last_sampler == ("chain" if sampler == "chain" else "graph")
    if sampler == "chain":
        assert m.last_chain_variant == (variant or "workspace")
    return out, mask


@pytest.mark.parametrize("variant", ["workspace", "panel"])
@pytest.mark.parametrize("n,grid", [(1000, 3), (128, 1), (1337, 5), (4096, 0)])
def test_chain_kernel_equals_per_layer_kernels_bitwise(n, grid, variant):
    """Both chain kernels -- 128-row tiles through a workspace (csrc/chain.h) and 64 patients resident in LDS
    (csrc/chain_panel.h: other tile shape, weights streamed from a fragment-ordered copy, activations never in global
    memory) -- against the per-layer kernels, bit for bit."""
    T = 12
    m = _model(T)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(3)).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", seed=77, row_offset=5)
    m.chain_grid = grid
    out, mask = _run(m, cond, n, "chain", variant, seed=77, row_offset=5)
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref), f"max|d| = {(out - ref).abs().max().item():.3e} of {ref.abs().max().item():.3e}"
    assert torch.equal(mask, ref_mask)
    # segmented launches (progress carries over kernel boundaries) and no stagger: same bits
    m.chain_steps_per_launch, m.chain_stagger = 5, 0
    out2, mask2 = _run(m, cond, n, "chain", variant, seed=77, row_offset=5)
    assert torch.equal(out2, ref) and torch.equal(mask2, ref_mask)


def test_auto_takes_the_lds_resident_chain_for_mid_size_batches():
    """auto: 18 432 rows are 288 units of 64 patients (every CU busy) but 1.125 rounds of the per-layer kernels' tiles -- the
    LDS-resident chain runs, with the per-layer kernels' bits; 2 048 rows stay on the per-layer kernels (the library default keeps
    results independent of the batch size bit for bit; the small-batch mode that trades this for latency -- input_splitk, the
    squad chain -- is opt-in: tests/test_gpu_squad.py)."""
    T, n = 3, 18432
    m = _model(T, seed=8)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(5)).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", seed=31)
    m.sampler, m.chain_variant = "auto", None
    out, mask = m.sample(cond, n, return_mutation_mask=True, seed=31)
    assert (m.last_sampler, m.last_chain_variant) == ("chain", "panel")
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)
    m.sample(cond[:2048], 2048, seed=31)
    assert m.last_sampler == "graph"


@pytest.mark.parametrize("hidden,dims,n", [
    ([256, 256], dict(mutation_dim=30, expression_dim=560, pathway_dim=10, condition_dim=3), 150),               # one skip, kept in the panel
    ([256, 512, 512, 256], dict(mutation_dim=40, expression_dim=700, pathway_dim=28, condition_dim=5), 97),       # two spilled skips + one kept
    ([256, 512, 256], dict(mutation_dim=50, expression_dim=966, pathway_dim=8, condition_dim=3), 64),             # D = 1024: whole chunks, whole passes
])
def test_panel_chain_other_architectures_bitwise(hidden, dims, n):
    """The LDS-resident chain's panel planner (csrc/chain_panel.hip: make_plan) on other layer lists: which skip stays in panel
    columns [256, 512), which ones are spilled in fragment order and reloaded between K segments, chunk / pass counts."""
    m = _model(5, hidden=hidden, seed=12, **dims)
    cond = torch.randn(n, dims["condition_dim"], generator=torch.Generator().manual_seed(2)).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", seed=3)
    m.chain_grid = 2
    out, mask = _run(m, cond, n, "chain", "panel", seed=3)
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)


def test_panel_chain_vs_oracle_injected_draws_and_unaligned_dims():
    """The LDS-resident chain kernel against the CPU oracle (injected x_T / z: 200 rows = three 64-patient units + 8 rows, T = 20),
    then at the reference's real dims 62/5054/26 (D = 5142, D % 4 = 2: padded state, eleven output passes) against the per-layer
    kernels bit for bit, and on an architecture whose panels do not fit (it must run the workspace kernel instead)."""
    T, n = 20, 200
    m = _model(T, seed=6)
    gen = torch.Generator().manual_seed(11)
    cond = torch.randn(n, 3, generator=gen)
    x_T = torch.randn(n, 2000, generator=gen)
    zs = torch.randn(T - 1, n, 2000, generator=gen)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(("condition_embed", "unet"))}
    ref = O.sample(sd, O.schedule_buffers("cosine", T), cond, x_T, lambda t: zs[T - 1 - t], 3, 128)
    m.chain_grid = 3
    out, _ = _run(m, cond.cuda(), n, "chain", "panel", x_T=x_T.cuda(), noise=zs.cuda())
    assert_close(out, ref, 5e-5, atol=1e-5, what="LDS-resident chain kernel vs oracle")

    m2 = _model(6, seed=2, mutation_dim=62, expression_dim=5054, pathway_dim=26, condition_dim=4)
    n2 = 150
    cond2 = torch.randn(n2, 4, generator=gen).cuda()
    ref2, refm2 = _run(m2, cond2, n2, "graph", seed=9)
    m2.chain_grid = 2
    out2, mask2 = _run(m2, cond2, n2, "chain", "panel", seed=9)
    assert torch.equal(out2, ref2) and torch.equal(mask2, refm2)

    m3 = _model(4, hidden=[512, 512, 512], seed=3)          # H0 = 512: no LDS-resident layout
    cond3 = torch.randn(130, 3, generator=gen).cuda()
    ref3, _ = _run(m3, cond3, 130, "graph", seed=1)
    m3.sampler, m3.chain_variant = "chain", "panel"
    out3 = m3.sample(cond3, 130, seed=1)
    assert m3.last_sampler == "chain" and m3.last_chain_variant == "workspace"
    assert torch.equal(out3, ref3)


def test_chain_kernel_vs_oracle_with_injected_draws():
    """300 rows (two full tiles + 44 rows), T = 30, the oracle's x_T and z injected; two workgroups alternate over three
    tiles.  Chain tolerance 5e-5 * max|ref| (the un-clamped x0_hat amplifies the first steps)."""
    T, n = 30, 300
    m = _model(T, seed=4)
    gen = torch.Generator().manual_seed(9)
    cond = torch.randn(n, 3, generator=gen)
    x_T = torch.randn(n, 2000, generator=gen)
    zs = torch.randn(T - 1, n, 2000, generator=gen)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(("condition_embed", "unet"))}
    bufs = O.schedule_buffers("cosine", T)
    ref = O.sample(sd, bufs, cond, x_T, lambda t: zs[T - 1 - t], 3, 128)
    m.chain_grid = 2
    out, mask = _run(m, cond.cuda(), n, "chain", x_T=x_T.cuda(), noise=zs.cuda())
    assert_close(out, ref, 5e-5, atol=1e-5, what="chain kernel vs oracle")
    refm = (ref[:, :50] > 0.5).float()
    near = (ref[:, :50] - 0.5).abs() <= 5e-5 * ref.abs().max() + 1e-5
    assert ((mask.cpu() != refm) & ~near).sum().item() == 0


def test_chain_kernel_other_widths_and_fallbacks():
    """512-wide first layer, 512/256 blocks (both GroupNorm widths in other positions); architectures or modes outside the
    chain kernel fall back to the per-layer kernels without being asked."""
    T, n = 8, 700
    m = _model(T, hidden=[512, 256, 512], seed=2)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).cuda()
    ref, _ = _run(m, cond, n, "graph", seed=5)
    m.chain_grid = 4
    out, _ = _run(m, cond, n, "chain", seed=5)
    assert torch.equal(out, ref)
    # dropout active (train-mode sampling, as the reference's __main__ smoke test does): per-layer kernels
    m.train()
    m.sampler = "chain"
    m.sample(cond, n, seed=5)
    assert m.last_sampler == "graph"
    # a width the chain kernel does not cover
    m2 = _model(T, hidden=[256, 128, 256], seed=2)
    m2.sampler = "chain"
    m2.sample(cond, n, seed=5)
    assert m2.last_sampler == "graph"
    # auto: small batches stay on the per-layer kernels, config-3-sized ones take the chain kernel
    m3 = _model(T)
    eng = m3._engine()
    assert L.lib().osd_sample_engine(eng.handle, 4096, 0) == 0
    assert L.lib().osd_sample_engine(eng.handle, 100_000, 0) == 1
    assert L.lib().osd_sample_engine(eng.handle, 100_000, L.OSD_F_TRAIN_MODE) == 0


@pytest.mark.parametrize("dims,hidden,n", [
    (dict(mutation_dim=7, expression_dim=121, pathway_dim=4, condition_dim=4), [256, 256], 129),           # D = 132: one ragged 128-feature tile + 4
    (dict(mutation_dim=50, expression_dim=200, pathway_dim=10, condition_dim=3), [512, 512, 512], 300),    # D = 260; 512-wide everywhere
    (dict(mutation_dim=33, expression_dim=1, pathway_dim=2, condition_dim=2), [256, 512, 512, 256], 1),    # D = 36 < one tile, a single row
    (dict(mutation_dim=3, expression_dim=2040, pathway_dim=5, condition_dim=3), [256, 512, 256], 127),     # D = 2048: whole feature tiles
])
def test_chain_kernel_edge_shapes_bitwise(dims, hidden, n):
    """Feature counts that are not multiples of the 128-wide output tile (the posterior epilogue's parameter DMA, x_t row
    segments and stores are clamped / guarded there), mutation widths that are not multiples of 4 (mask at t = 0), row counts
    of 1 / 127 / 129, four- and eight-layer trunks: the chain kernel against the per-layer kernels, bit for bit."""
    T = 6
    m = _model(T, hidden=hidden, seed=7, **dims)
    cond = torch.randn(n, dims["condition_dim"], generator=torch.Generator().manual_seed(2)).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", seed=11, row_offset=3)
    m.chain_grid = 2
    out, mask = _run(m, cond, n, "chain", seed=11, row_offset=3)
    assert torch.isfinite(out).all()
    assert torch.equal(out, ref), f"max|d| = {(out - ref).abs().max().item():.3e}"
    assert torch.equal(mask, ref_mask)


def test_config3_full_size_engines_agree():
    """BASELINE config 3 at its stated size -- 100 000 conditional patients, T = 1000, D = 2000, in-kernel Philox -- once on the
    persistent chain kernel (782 row tiles over 512 workgroups, ~10^6 cross-workgroup hand-offs) and once on the per-layer
    kernels under hipGraph replay: the final states and mutation masks must be bit-identical (same tile loop, same epilogue
    arithmetic, draws addressed by (row, feature, step) only), finite, and independent of the engine's chunking."""
    n, T = 100_000, 1000
    m = _model(T, seed=3)
    gen = torch.Generator().manual_seed(5)
    cond = torch.randn(n, 3, generator=gen).cuda()
    out_c, mask_c = _run(m, cond, n, "chain", seed=2024, row_offset=0)
    assert torch.isfinite(out_c).all()
    out_g, mask_g = _run(m, cond, n, "graph", seed=2024, row_offset=0)
    assert torch.equal(out_c, out_g), f"max|d| = {(out_c - out_g).abs().max().item():.3e} of {out_g.abs().max().item():.3e}"
    assert torch.equal(mask_c, mask_g)
    assert set(mask_c.unique().tolist()) <= {0.0, 1.0}
    # ... and on the LDS-resident chain kernel (1 563 units of 64 patients over 256 workgroups, 1.5 x 10^6 hand-offs)
    out_p, mask_p = _run(m, cond, n, "chain", "panel", seed=2024, row_offset=0)
    assert torch.equal(out_p, out_g) and torch.equal(mask_p, mask_g)



def _get_option(m, name):
    import ctypes as C
    v = C.c_int64(-1)
    L.check(L.lib().osd_get_option(m._engine().handle, name.encode(), C.byref(v)))
    return v.value


@pytest.mark.parametrize("variant,n", [("workspace", 128), ("panel", 64)])
def test_chain_spin_timeout_is_recovered_on_the_per_layer_kernels(variant, n):
    """models/diffusion.py:427-449 cannot fail.  One row tile and TWO workgroups: the second workgroup takes step 1 of the tile
    while the first is still inside step 0, so its dependency wait is certain; a spin budget of one tick (10 ns) makes that
    wait give up (CHAIN_TIMEOUT).  The synchronous call must notice, re-run the chain on the per-layer kernels from the same
    x_T / draws and return their result, with a warning instead of an error."""
    T = 12
    m = _model(T, seed=6)
    gen = torch.Generator().manual_seed(4)
    cond = torch.randn(n, 3, generator=gen).cuda()
    x_T = torch.randn(n, 2000, generator=gen).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", x_T=x_T, seed=31, row_offset=7)
    m.sampler, m.chain_variant, m.chain_grid, m.chain_spin_budget = "chain", variant, 2, 1
    assert _get_option(m, "chain_fallbacks") == 0
    with pytest.warns(UserWarning, match="re-run on the per-layer kernels"):
        out, mask = m.sample(cond, n, x_T=x_T, seed=31, row_offset=7, return_mutation_mask=True)
    assert m.last_sampler == "graph" and m.last_chain_variant == variant
    assert _get_option(m, "chain_fallbacks") == 1 and _get_option(m, "last_engine") == 0
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)
    # Philox x_T (regenerated from the seed for the second run)
    ref2, _ = _run(m, cond, n, "graph", seed=32)
    m.sampler = "chain"
    with pytest.warns(UserWarning):
        out2 = m.sample(cond, n, seed=32)
    assert torch.equal(out2, ref2) and _get_option(m, "chain_fallbacks") == 2
    # with a sane budget the same geometry (surplus workgroup waiting for its turn) completes on the chain kernel
    m.chain_spin_budget = 500_000_000
    out3, _ = _run(m, cond, n, "chain", variant, seed=32)
    assert torch.equal(out3, ref2) and _get_option(m, "chain_fallbacks") == 2


@pytest.mark.parametrize("variant", ["workspace", "panel"])
def test_chain_wall_clock_budget_aborts_and_recovers(variant):
    """The host side of the same guarantee: a synchronous chain is polled (hipStreamQuery) against a wall-clock budget instead
    of a blind hipStreamSynchronize; on expiry the host raises the abort flag, the workgroups leave at their next unit
    boundary / dependency poll, and the chain is re-run on the per-layer kernels.  Budget 1 ms against a chain of ~50 ms."""
    T, n = 40, 20_000
    m = _model(T, seed=8)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).cuda()
    ref, ref_mask = _run(m, cond, n, "graph", seed=5)
    m.sampler, m.chain_variant, m.chain_wall_budget_ms = "chain", variant, 1
    with pytest.warns(UserWarning):
        out, mask = m.sample(cond, n, seed=5, return_mutation_mask=True)
    assert m.last_sampler == "graph" and m.last_chain_variant == variant and _get_option(m, "chain_fallbacks") == 1
    assert torch.equal(out, ref) and torch.equal(mask, ref_mask)
    m.chain_wall_budget_ms = 0                  # automatic budget (10 x the estimate + 2 s): the chain kernel finishes
    out2, _ = _run(m, cond, n, "chain", variant, seed=5)
    assert torch.equal(out2, ref) and _get_option(m, "chain_fallbacks") == 1


def test_chain_timeout_without_sync_is_reported_by_the_next_call():
    """C ABI without OSD_F_SYNC: the call that launched a chain which gives up returns OSD_OK (nothing is known yet); the next
    osd_sample_chain on the handle returns OSD_EHIP with the reason; the handle stays usable."""
    T, n = 6, 128
    m = _model(T, seed=9)
    cond = torch.randn(n, 3, generator=torch.Generator().manual_seed(1)).cuda()
    m.sampler, m.chain_grid, m.chain_spin_budget = "chain", 2, 1
    eng = m._engine()
    out = torch.empty(n, 2000, device="cuda")
    args = (eng.handle, L.ptr(cond), n, None, None, 3, 0, L.ptr(out), None)
    assert L.lib().osd_sample_chain(*args, 0) == L.OSD_OK
    assert L.lib().osd_sample_chain(*args, 0) == L.OSD_EHIP
    assert "gave up" in L.last_error()
    torch.cuda.synchronize()
    m.chain_spin_budget = 500_000_000
    ref, _ = _run(m, cond, n, "graph", seed=3)
    got, _ = _run(m, cond, n, "chain", seed=3)
    assert torch.equal(got, ref)


def test_input_splitk_small_batches():
    """The small-batch latency path of the per-layer engine (the reference's default generation workload: 1000 patients per
    scenario, config.yaml:119): input_proj's K range cut into slices over workgroups + a reduce/epilogue kernel.  Auto mode
    engages below 128 output tiles; against the single-pass kernels the chain agrees to the chain tolerance (summation order),
    against the oracle likewise; graph replay and eager launches stay bit-identical to each other."""
    T, n = 10, 1000
    m = _model(T, seed=12)
    gen = torch.Generator().manual_seed(2)
    cond = torch.randn(n, 3, generator=gen)
    x_T = torch.randn(n, 2000, generator=gen)
    zs = torch.randn(T - 1, n, 2000, generator=gen)
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(("condition_embed", "unet"))}
    ref = O.sample(sd, O.schedule_buffers("cosine", T), cond, x_T, lambda t: zs[T - 1 - t], 3, 128)
    kw = dict(x_T=x_T.cuda(), noise=zs.cuda())
    single, _ = _run(m, cond.cuda(), n, "graph", **kw)              # input_splitk = 0 (from _model)
    outs = {}
    for mode in (-1, 4, 16):
        m.input_splitk = mode
        outs[mode], _ = _run(m, cond.cuda(), n, "graph", **kw)
        assert_close(outs[mode], ref, 5e-5, atol=1e-5, what=f"split-K {mode} vs oracle")
        assert_close(outs[mode], single, 2e-5, atol=1e-6, what=f"split-K {mode} vs single pass")
    assert not torch.equal(outs[4], single)                          # it really is another summation order
    m.use_graph = False
    eager, _ = _run(m, cond.cuda(), n, "graph", **kw)
    assert torch.equal(eager, outs[16])
