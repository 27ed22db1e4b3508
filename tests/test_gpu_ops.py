"""GPU: the MFMA GEMM building blocks of libosdiff.so against fp64 torch on the host.
Tolerance (fp32, stated): max|d| <= 1e-5 * max|ref| (BASELINE.md parity gate, per op)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from osteosarcoma_diffusionmodel_amd import _lib as L
from helpers import RawHandle, assert_close

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def rh():
    h = RawHandle()
    yield h
    h.close()


def _gemm(rh, A, a_kc, B, b_kc, F_, P, K, C0=None):
    out = torch.zeros(P, F_, device="cuda") if C0 is None else C0.clone()
    L.check(L.lib().osd_op_gemm(rh.h, L.ptr(A), A.shape[1], int(a_kc), L.ptr(B), B.shape[1], int(b_kc), F_, P, K,
                                L.ptr(out), F_, int(C0 is not None)))
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("F_,P,K", [(32, 32, 32), (64, 128, 64), (70, 45, 37), (256, 300, 2000), (2000, 257, 256),
                                    (512, 8192 + 40, 96), (24, 7, 3)])
@pytest.mark.parametrize("a_kc,b_kc", [(1, 1), (0, 1), (0, 0), (1, 0)])
def test_gemm_layouts(rh, F_, P, K, a_kc, b_kc):
    """out[p][f] = sum_k A(f,k) B(p,k) for every operand storage; asymmetric random data, ragged edges."""
    if (F_ % 4 or P % 4) and not (a_kc and b_kc):
        # row-contiguous (NKC) operands are loaded in 4-element chunks along the row index
        F_, P = (F_ + 3) // 4 * 4, (P + 3) // 4 * 4
    g = torch.Generator().manual_seed(F_ * 7 + P * 3 + K)
    Af = torch.randn(F_, K, generator=g, dtype=torch.float64)
    Bf = torch.randn(P, K, generator=g, dtype=torch.float64)
    ref = Bf @ Af.T
    A = (Af if a_kc else Af.T).contiguous().float().cuda()
    B = (Bf if b_kc else Bf.T).contiguous().float().cuda()
    out = _gemm(rh, A, a_kc, B, b_kc, F_, P, K)
    assert_close(out.cpu(), ref, RTOL, what=f"gemm {F_}x{P}x{K} kc=({a_kc},{b_kc})")


def test_gemm_identity_asymmetric(rh):
    """A = I with an asymmetric B: catches a transposed accumulator map (exact integers)."""
    n = 64
    A = torch.eye(n).cuda()
    B = (torch.arange(n * n, dtype=torch.float32).reshape(n, n) % 251).cuda()   # B[p][k]
    out = _gemm(rh, A, 1, B, 1, n, n, n)
    assert torch.equal(out, B)     # out[p][f] = B[p][f]


def test_gemm_accumulate(rh):
    g = torch.Generator().manual_seed(5)
    F_, P, K = 96, 200, 128
    Af = torch.randn(F_, K, generator=g, dtype=torch.float64)
    Bf = torch.randn(P, K, generator=g, dtype=torch.float64)
    C0 = torch.randn(P, F_, generator=g, dtype=torch.float64)
    Ad, Bd, Cd = Af.T.contiguous().float().cuda(), Bf.float().cuda(), C0.float().cuda()
    out = _gemm(rh, Ad, 0, Bd, 1, F_, P, K, Cd)
    assert_close(out.cpu(), C0 + Bf @ Af.T, RTOL, what="dgrad accumulate")


@pytest.mark.parametrize("n,K,N,silu", [(5, 3, 64, 1), (300, 64, 64, 0), (1000, 128, 256, 0), (130, 2000, 256, 0), (64, 256, 2000, 1)])
def test_linear(rh, n, K, N, silu):
    g = torch.Generator().manual_seed(n + K + N)
    x = torch.randn(n, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = F.linear(x.double(), w.double(), b.double())
    if silu:
        ref = F.silu(ref)
    y = torch.empty(n, N, device="cuda")
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()          # keep the device tensors alive across the async call
    L.check(L.lib().osd_op_linear(rh.h, L.ptr(xd), L.ptr(wd), L.ptr(bd), n, K, N, silu, L.ptr(y)))
    torch.cuda.synchronize()
    assert_close(y.cpu(), ref, RTOL, what=f"linear n={n} K={K} N={N}")


@pytest.mark.parametrize("N", [32, 64, 128, 256, 512, 1024])          # group widths 4 .. 128
@pytest.mark.parametrize("n,K1,K2", [(37, 64, 0), (300, 256, 0), (129, 64, 32), (9000, 32, 0)])
def test_linear_gn_silu(rh, N, n, K1, K2):
    """Fused Linear -> GroupNorm(8) -> SiLU (models/diffusion.py:200-203) incl. the concat-free
    two-panel input (:250), against torch fp64; large-mean rows stress the two-pass variance."""
    g = torch.Generator().manual_seed(N + n + K1)
    K = K1 + K2
    x = torch.randn(n, K, generator=g)
    x[::3] += 5.0
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    gamma = 1 + 0.3 * torch.randn(N, generator=g)
    beta = 0.2 * torch.randn(N, generator=g)
    ref = F.silu(F.group_norm(F.linear(x.double(), w.double(), b.double()), 8, gamma.double(), beta.double(), 1e-5))
    x1 = x[:, :K1].contiguous().cuda()
    x2 = x[:, K1:].contiguous().cuda() if K2 else None
    y = torch.empty(n, N, device="cuda")
    wd, bd, gd, bed = w.cuda(), b.cuda(), gamma.cuda(), beta.cuda()
    L.check(L.lib().osd_op_linear_gn_silu(rh.h, L.ptr(x1), K1, L.ptr(x2), K2, L.ptr(wd), L.ptr(bd),
                                          L.ptr(gd), L.ptr(bed), n, N, L.ptr(y)))
    torch.cuda.synchronize()
    assert_close(y.cpu(), ref, RTOL, atol=1e-6, what=f"linear_gn_silu N={N} n={n} K=({K1},{K2})")


def test_randn_stream(rh):
    """Philox normals: moments, determinism, and addressing by global row (sharding invariance)."""
    rows, cols = 4096, 2000
    a = torch.empty(rows, cols, device="cuda")
    L.check(L.lib().osd_op_randn(rh.h, L.ptr(a), rows, cols, 1234, 0, 7, 0))
    b = torch.empty(rows, cols, device="cuda")
    L.check(L.lib().osd_op_randn(rh.h, L.ptr(b), rows, cols, 1234, 0, 7, 0))
    assert torch.equal(a, b)
    # rows [1000, 1500) generated as their own shard
    c = torch.empty(500, cols, device="cuda")
    L.check(L.lib().osd_op_randn(rh.h, L.ptr(c), 500, cols, 1234, 1000, 7, 0))
    assert torch.equal(c, a[1000:1500])
    d = torch.empty(rows, cols, device="cuda")
    L.check(L.lib().osd_op_randn(rh.h, L.ptr(d), rows, cols, 1234, 0, 8, 0))     # another step
    x = a.double().cpu().numpy().ravel()
    n = x.size
    assert abs(x.mean()) < 5 / np.sqrt(n)
    assert abs(x.var() - 1) < 5 * np.sqrt(2 / n)
    assert abs((x ** 3).mean()) < 5 * np.sqrt(15 / n)
    assert abs((x ** 4).mean() - 3) < 5 * np.sqrt(96 / n)
    assert np.abs(x).max() < 7.0
    y = d.double().cpu().numpy().ravel()
    assert abs(np.corrcoef(x, y)[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 5 / np.sqrt(n)


def test_randn_matches_host_philox(rh):
    """Device normals == Box-Muller over a host restatement of Philox4x32-10 (Random123 KAT-checked)."""
    from helpers import TAG_POSTERIOR, TAG_QNOISE, philox4x32_10, philox_normals
    kat = philox4x32_10(np.uint32(0x243f6a88), np.uint32(0x85a308d3), np.uint32(0x13198a2e), np.uint32(0x03707344), 0xa4093822, 0x299f31d0)
    assert [int(v) for v in kat] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    rows, cols, seed = 257, 2000, (9 << 33) + 12345
    for step, kind, tag, off in ((1000, 0, TAG_POSTERIOR, 0), (17, 0, TAG_POSTERIOR, 4000), (0, 1, TAG_QNOISE, 123)):
        a = torch.empty(rows, cols, device="cuda")
        L.check(L.lib().osd_op_randn(rh.h, L.ptr(a), rows, cols, seed, off, step, kind))
        ref = philox_normals(seed, rows, cols, step, tag, off)
        assert np.abs(a.cpu().numpy() - ref).max() < 2e-5
