"""GPU: boundary hygiene (round-1 review): resume of the fused optimizer, cVAE generation through the generator, the
osd_create failure path, the 32-bit row id space, timestep range errors, the library's RCCL entry points on a
single-rank communicator."""
import ctypes as C
import copy

import numpy as np
import pytest
import torch

from osteosarcoma_diffusionmodel_amd import BiologyAwareDiffusionModel, SyntheticPatientGenerator, _lib as L
from osteosarcoma_diffusionmodel_amd.train import Trainer
from helpers import SM, SM_H, RawHandle, assert_close, config, small_model

pytestmark = pytest.mark.gpu


def _train_conf(tmp_path, p=0.0):
    conf = config(SM_H, p=p)
    conf["training"] = {"learning_rate": 1e-3, "weight_decay": 1e-5, "patience": 100, "min_delta": 1e-4,
                        "augmentation": {"mixup_alpha": 0.0}, "save_dir": str(tmp_path), "num_epochs": 1,
                        "save_frequency": 10, "val_split": 0.2, "random_seed": 42, "batch_size": 16}
    return conf


def _steps(tr, batches, lo, hi):
    out = []
    for i in range(lo, hi):
        x, c, t, nz = batches[i]
        out.append(tr.train_step(x, c, t=t, noise=nz).item())
    return out


def test_fused_adamw_resumes_from_checkpoint(golden_dir, tmp_path):
    """utils/train.py:275-294 saves 'optimizer_state_dict'; a Trainer restored from it must continue exactly like the
    uninterrupted run (moments and bias-correction step), and its next checkpoint must carry the advanced state."""
    gen = torch.Generator().manual_seed(5)
    batches = [(torch.randn(16, 40, generator=gen).cuda(), torch.randn(16, 3, generator=gen).cuda(),
                torch.randint(0, 1000, (16,), generator=gen).cuda(), torch.randn(16, 40, generator=gen).cuda()) for _ in range(6)]
    conf = _train_conf(tmp_path)
    base = small_model(golden_dir, p=0.0).train()
    # uninterrupted: 6 steps
    m_a = copy.deepcopy(base)
    tr_a = Trainer(m_a, [], [], conf, device="cuda")
    la = _steps(tr_a, batches, 0, 6)
    # interrupted after 3 steps, checkpointed through the reference-shaped save path, resumed in a NEW Trainer
    m_b = copy.deepcopy(base)
    tr_b = Trainer(m_b, [], [], conf, device="cuda")
    lb = _steps(tr_b, batches, 0, 3)
    tr_b.save_checkpoint(0, 0.0, is_best=True)
    ck = torch.load(tmp_path / "best_model.pt", weights_only=True)
    m_c = BiologyAwareDiffusionModel(config=conf, **SM)
    m_c.load_state_dict(ck["model_state_dict"])
    tr_c = Trainer(m_c.cuda().train(), [], [], conf, device="cuda")
    tr_c.optimizer.load_state_dict(ck["optimizer_state_dict"])
    assert tr_c.optimizer._step == 3
    p0 = tr_c.flat.params[0]
    assert tr_c.optimizer.state[p0]["exp_avg"].data_ptr() == tr_c.optimizer.exp_avg.data_ptr()     # views of the flat buffers
    assert float(tr_c.optimizer.exp_avg.abs().sum()) > 0
    lb += _steps(tr_c, batches, 3, 6)
    assert np.allclose(la, lb, rtol=1e-6, atol=0)
    for (k, pa), pc in zip(m_a.named_parameters(), m_c.parameters()):
        assert_close(pc.detach().cpu(), pa.detach().cpu(), 1e-6, atol=1e-9, what=f"resumed param {k}")
    osd = tr_c.optimizer.state_dict()
    assert all(float(st["step"]) == 6.0 for st in osd["state"].values())
    assert_close(osd["state"][0]["exp_avg"].cpu(), tr_a.optimizer.state_dict()["state"][0]["exp_avg"].cpu(), 1e-6, atol=1e-12)
    # a stock torch AdamW state dict (the reference's optimizer) loads the same way
    stock = torch.optim.AdamW(m_b.parameters(), lr=1e-3, weight_decay=1e-5)
    stock.load_state_dict(ck["optimizer_state_dict"])
    tr_c.optimizer.load_state_dict(stock.state_dict())
    assert tr_c.optimizer._step == 3


def test_generator_runs_a_cvae_checkpoint(tmp_path):
    """load_trained_model builds the 'cvae' architecture (utils/generate.py:238-298) and SyntheticPatientGenerator.generate
    must sample from it: model.sample(conditions, num_samples) + host binarisation, as utils/generate.py:124-135."""
    import pandas as pd
    from osteosarcoma_diffusionmodel_amd.cvae import BiologyConstrainedVAE
    from osteosarcoma_diffusionmodel_amd.generate import generate_patients, load_trained_model
    conf = config(SM_H)
    conf["model"].update({"architecture": "cvae",
                          "constraints": {"pathway_coherence_weight": 0.5, "mutation_expression_weight": 0.3, "survival_prediction_weight": 0.3}})
    conf["data"] = {"processed_dir": str(tmp_path)}
    for fname, cols in (("mutation_matrix_aligned.csv", 8), ("expression_matrix_aligned.csv", 24), ("pathway_scores.csv", 8)):
        pd.DataFrame(np.zeros((2, cols)), index=["a", "b"]).to_csv(tmp_path / fname)
    torch.manual_seed(0)
    vae = BiologyConstrainedVAE(mutation_dim=8, expression_dim=24, pathway_dim=8, condition_dim=3, config=conf)
    ck = tmp_path / "best_model.pt"
    torch.save({"epoch": 0, "model_state_dict": vae.state_dict(), "optimizer_state_dict": {}, "val_loss": 0.0, "config": conf}, ck)
    model = load_trained_model(ck, conf, "cuda")
    assert hasattr(model, "vae")
    gen = SyntheticPatientGenerator(model, conf, device="cuda")
    scen = {"survival_time": 300, "event_occurred": 1, "metastasis_at_diagnosis": 1}
    out = gen.generate(33, scen)
    assert out["mutations"].shape == (33, 8) and out["expression"].shape == (33, 24) and out["pathways"].shape == (33, 8)
    assert set(np.unique(out["mutations"])) <= {0.0, 1.0} and out["mutations"].dtype == np.float64
    assert np.isfinite(out["expression"]).all() and out["conditions"].shape == (33, 3)
    # same draws through the model API: the generator adds nothing but the split and the threshold
    z = torch.randn(5, model.vae.latent_dim, device="cuda")
    cond = gen.create_conditions(5, scen)
    direct = model.sample(cond, 5, z=z).cpu().numpy()
    assert np.array_equal((direct[:, :8] > 0.5).astype(float), (model.sample(cond, 5, z=z).cpu().numpy()[:, :8] > 0.5).astype(float))
    with pytest.raises(ValueError, match="cVAE"):
        gen.generate(4, scen, seed=1)
    out2 = generate_patients(ck, conf, 7, scen, device="cuda")
    assert out2["mutations"].shape == (7, 8)


def test_osd_create_failure_releases_everything():
    """A failing allocation inside osd_create must not leak the handle's earlier device buffers: a schedule of 2e9 steps
    makes the fourth table (T x time_dim floats = 1 TB) fail after three allocations of 8 + 8 + 32 GB succeeded."""
    lib = L.lib()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cfg = L.OsdConfig()
    cfg.mutation_dim, cfg.expression_dim, cfg.pathway_dim, cfg.condition_dim = 8, 24, 8, 3
    cfg.time_dim, cfg.n_hidden, cfg.dropout_p, cfg.device = 128, 3, 0.0, torch.cuda.current_device()
    for i, v in enumerate(SM_H):
        cfg.hidden_dims[i] = v
    cfg.num_steps = 2_000_000_000
    h = C.c_void_p(0xdead)
    rc = lib.osd_create(C.byref(cfg), C.byref(h))
    assert rc in (L.OSD_EHIP, L.OSD_ENOMEM) and not h.value
    assert b"hipMalloc" in lib.osd_last_error()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < (256 << 20), f"osd_create leaked {(free0 - free1) / 2**30:.1f} GiB on its failure path"
    cfg.num_steps = 10                                   # and the process can still create a working handle
    assert lib.osd_create(C.byref(cfg), C.byref(h)) == L.OSD_OK and h.value
    assert lib.osd_destroy(h) == L.OSD_OK


def test_row_offset_outside_32_bit_id_space_is_einval(golden_dir):
    """Global row ids are 32-bit Philox counters: a shard that would wrap is rejected (not truncated)."""
    lib = L.lib()
    m = small_model(golden_dir, T=10)
    eng = m._engine()
    n = 8
    x = torch.zeros(n, 40, device="cuda")
    c = torch.zeros(n, 3, device="cuda")
    t = torch.zeros(n, dtype=torch.int32, device="cuda")
    for off in (-1, (1 << 32) - n + 1, 1 << 40):
        assert lib.osd_sample_chain(eng.handle, L.ptr(c), n, None, None, 1, off, L.ptr(x), None, 0) == L.OSD_EINVAL
        assert b"row" in lib.osd_last_error()
        assert lib.osd_p_sample_step(eng.handle, L.ptr(x), 3, L.ptr(c), None, n, 1, off, L.ptr(x), 0) == L.OSD_EINVAL
        assert lib.osd_q_sample(eng.handle, L.ptr(x), L.ptr(t), None, n, 1, off, L.ptr(x), L.ptr(x)) == L.OSD_EINVAL
        loss = torch.zeros(1, device="cuda")
        assert lib.osd_train_loss_fwd_bwd(eng.handle, L.ptr(x), L.ptr(c), n, None, None, None, 1, off, 0, L.ptr(loss), None, 1.0, None, 0) == L.OSD_EINVAL
    with pytest.raises(ValueError):
        m.sample(c, n, seed=1, row_offset=1 << 32)
    # the last valid shard still runs, and its draws differ from the first shard's
    a = m.sample(c, n, seed=1, row_offset=(1 << 32) - n)
    b = m.sample(c, n, seed=1, row_offset=0)
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_timestep_index_out_of_range(golden_dir):
    """models/diffusion.py:337 gathers sqrt_alphas_cumprod[t]: t outside [0, T) is an IndexError in the reference and in
    the Python mirror; at the C boundary the library clamps (no out-of-bounds table read), so the result equals t = T-1."""
    m = small_model(golden_dir, T=10)
    x = torch.randn(4, 40, device="cuda")
    c = torch.randn(4, 3, device="cuda")
    nz = torch.randn(4, 40, device="cuda")
    for bad in (torch.tensor([0, 1, 10, 2]), torch.tensor([0, -1, 3, 2])):
        with pytest.raises(IndexError):
            m.q_sample(x, bad.cuda(), nz)
        with pytest.raises(IndexError):
            m(x, c, t=bad.cuda(), noise=nz)
        with pytest.raises(IndexError):
            m.predict_noise(x, bad.cuda(), c)
    eng = m._engine()
    lib = L.lib()
    t_bad = torch.tensor([0, 1, 1 << 20, -7], dtype=torch.int32, device="cuda")
    t_ok = torch.tensor([0, 1, 9, 0], dtype=torch.int32, device="cuda")
    out_bad, out_ok = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.osd_q_sample(eng.handle, L.ptr(x), L.ptr(t_bad), L.ptr(nz), 4, 0, 0, L.ptr(out_bad), None))
    L.check(lib.osd_q_sample(eng.handle, L.ptr(x), L.ptr(t_ok), L.ptr(nz), 4, 0, 0, L.ptr(out_ok), None))
    assert torch.equal(out_bad, out_ok)
    e_bad, e_ok = torch.empty_like(x), torch.empty_like(x)
    L.check(lib.osd_denoiser_forward(eng.handle, L.ptr(x), L.ptr(t_bad), 0, L.ptr(c), 4, L.ptr(e_bad), 0, None, 0))
    L.check(lib.osd_denoiser_forward(eng.handle, L.ptr(x), L.ptr(t_ok), 0, L.ptr(c), 4, L.ptr(e_ok), 0, None, 0))
    assert torch.equal(e_bad, e_ok)


def test_rccl_entry_points_single_rank():
    """osd_comm_* / osd_allreduce_grads_begin/end (include/osdiff.h) on a one-rank communicator: RCCL binds at run time,
    the bucketed SUM over one rank is the identity, and the handle's stream is ordered behind the communicator's."""
    lib = L.lib()
    uid = (C.c_char * L.OSD_COMM_ID_BYTES)()
    L.check(lib.osd_comm_unique_id(uid))
    comm = C.c_void_p()
    L.check(lib.osd_comm_create(uid, 0, 1, torch.cuda.current_device(), C.byref(comm)))
    assert comm.value
    rh = RawHandle()
    try:
        g = torch.randn(10_000, device="cuda")
        want = g.clone()
        starts = (C.c_int64 * 3)(0, 4_000, 4_000)
        ends = (C.c_int64 * 3)(4_000, 4_000, 10_000)          # the middle bucket is empty
        evs = [torch.cuda.Event() for _ in range(3)]
        for e in evs:
            e.record()
        ev = (C.c_void_p * 3)(*[e.cuda_event for e in evs])
        L.check(lib.osd_allreduce_grads_begin(rh.h, comm, L.ptr(g), starts, ends, ev, 3))
        L.check(lib.osd_allreduce_grads_end(rh.h, comm))
        g.mul_(2.0)                                            # on the handle's stream: ordered behind the collectives
        torch.cuda.synchronize()
        assert torch.equal(g, 2.0 * want)
        L.check(lib.osd_allreduce_grads_begin(rh.h, comm, L.ptr(g), starts, ends, None, 3))     # no events: whole-stream order
        L.check(lib.osd_allreduce_grads_end(rh.h, comm))
        torch.cuda.synchronize()
        assert torch.equal(g, 2.0 * want)
        bad = (C.c_int64 * 1)(5)
        bad_e = (C.c_int64 * 1)(3)
        assert lib.osd_allreduce_grads_begin(rh.h, comm, L.ptr(g), bad, bad_e, None, 1) == L.OSD_EINVAL
        assert lib.osd_comm_create(uid, 2, 2, 0, C.byref(C.c_void_p())) == L.OSD_EINVAL
    finally:
        rh.close()
        assert lib.osd_comm_destroy(comm) == L.OSD_OK
