"""GPU parity of the dense building blocks (through the C ABI) against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from driftscan_amd._lib import Context

    c = Context(0, workspace_bytes=1 << 30)
    yield c
    c.close()


def crand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (37, 91, 53), (130, 70, 129), (5, 3, 2), (200, 200, 1)])
@pytest.mark.parametrize("mode", ["NN", "NC", "CN", "TN_strided"])
def test_zgemm(ctx, M, N, K, mode):
    rng = np.random.default_rng(M * 1000 + N * 10 + K)
    if mode == "NN":
        A, B = crand(rng, M, K), crand(rng, K, N)
        ref = A @ B
        args = dict(rsA=K, csA=1, rsB=N, csB=1)
    elif mode == "NC":  # A B^H with B stored (N x K)
        A, B = crand(rng, M, K), crand(rng, N, K)
        ref = A @ B.conj().T
        args = dict(rsA=K, csA=1, rsB=1, csB=K, conjB=True)
    elif mode == "CN":  # A^H B with A stored (K x M)
        A, B = crand(rng, K, M), crand(rng, K, N)
        ref = A.conj().T @ B
        args = dict(rsA=1, csA=M, rsB=N, csB=1, conjA=True)
    else:  # A^T B (no conj), A stored (K x M)
        A, B = crand(rng, K, M), crand(rng, K, N)
        ref = A.T @ B
        args = dict(rsA=1, csA=M, rsB=N, csB=1)
    C0 = crand(rng, M, N)
    dA, dB, dC = ctx.to_device(A), ctx.to_device(B), ctx.to_device(C0)
    ctx.zgemm(dA, dB, dC, M, N, K, ldc=N, alpha=0.5, beta=-2.0, **args)
    ctx.sync()
    out = dC.cpu().numpy()
    exp = 0.5 * ref - 2.0 * C0
    assert np.abs(out - exp).max() <= 1e-12 * max(1.0, np.abs(exp).max())


def test_zgemm_kscale_batched(ctx):
    rng = np.random.default_rng(7)
    nb, M, N, K = 5, 50, 40, 129
    A, B = crand(rng, nb, M, K), crand(rng, nb, N, K)
    s = rng.uniform(0.1, 2.0, (nb, K))
    ref = np.einsum("bmk,bk,bnk->bmn", A, s, B.conj())
    dA, dB, ds = ctx.to_device(A), ctx.to_device(B), ctx.to_device(s)
    dC = ctx.zeros((nb, M, N), np.complex128)
    ctx.zgemm(dA, dB, dC, M, N, K, rsA=K, csA=1, rsB=1, csB=K, conjB=True, ldc=N, kscale=ds, batch=nb,
              strideA=M * K, strideB=N * K, strideC=M * N, stride_kscale=K)
    ctx.sync()
    assert np.abs(dC.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()


@pytest.mark.parametrize("n", [1, 31, 32, 33, 100, 257])
def test_potrf_trsm(ctx, n):
    rng = np.random.default_rng(n)
    nb = 3
    X = crand(rng, nb, n, n + 3)
    A = X @ X.conj().transpose(0, 2, 1) + 0.1 * np.eye(n)
    dA = ctx.to_device(A)
    info = ctx.zpotrf(dA, n, n, stride=n * n, batch=nb)
    assert (info == 0).all()
    L = dA.cpu().numpy()
    assert np.abs(np.triu(L, 1)).max() == 0.0
    assert np.abs(L @ L.conj().transpose(0, 2, 1) - A).max() <= 1e-12 * np.abs(A).max()
    nrhs = 70
    Bm = crand(rng, nb, n, nrhs)
    for conjtrans in (False, True):
        dB = ctx.to_device(Bm)
        ctx.ztrsm(dA, dB, n, nrhs, n, nrhs, conjtrans=conjtrans, strideL=n * n, strideB=n * nrhs, batch=nb)
        ctx.sync()
        Xs = dB.cpu().numpy()
        Lop = L.conj().transpose(0, 2, 1) if conjtrans else L
        res = np.abs(Lop @ Xs - Bm).max() / np.abs(Bm).max()
        assert res <= 1e-10


def test_potrf_not_pd(ctx):
    rng = np.random.default_rng(3)
    n = 70
    X = crand(rng, n, n - 5)
    A = X @ X.conj().T - 1e-9 * np.eye(n)
    dA = ctx.to_device(A[None])
    info = ctx.zpotrf(dA, n, n, stride=n * n, batch=1)
    assert 0 < info[0] <= n


@pytest.mark.parametrize("rows,cols", [(20, 25), (92, 129), (33, 200), (104, 388 + 104), (70, 30)])
def test_jacobi_rows(ctx, rows, cols):
    rng = np.random.default_rng(rows * cols)
    nb = 4
    A = crand(rng, nb, rows, cols)
    A *= np.exp(-np.arange(cols) / 8.0)  # graded columns -> spectrum spans decades
    dZ = ctx.to_device(A)
    sigma, sweeps = ctx.jacobi_rows(dZ, rows, cols, 0, cols, cols, stride=rows * cols, batch=nb)
    Z = dZ.cpu().numpy()
    s = sigma.cpu().numpy()
    for b in range(nb):
        ref = np.linalg.svd(A[b], compute_uv=False)
        k = min(rows, cols)
        assert np.abs(s[b, :k] - ref).max() <= 1e-12 * ref[0], (b, sweeps)
        G = Z[b] @ Z[b].conj().T
        assert np.abs(G - np.diag(np.diag(G))).max() <= 1e-11 * ref[0] ** 2
        # same row space / Gram: Z^H Z == A^H A
        assert np.abs(Z[b].conj().T @ Z[b] - A[b].conj().T @ A[b]).max() <= 1e-11 * ref[0] ** 2
    assert sweeps < 30


def test_jacobi_rows_passengers(ctx):
    """Gram over a column subset; passenger columns carry the accumulated U^H."""
    rng = np.random.default_rng(11)
    rows, cols = 40, 60
    A = crand(rng, rows, cols)
    Z = np.concatenate([A, np.eye(rows)], axis=1)
    dZ = ctx.to_device(Z[None])
    sigma, sweeps = ctx.jacobi_rows(dZ, rows, cols + rows, 10, 50, cols + rows, stride=0, batch=1)
    out = dZ.cpu().numpy()[0]
    Y, W = out[:, :cols], out[:, cols:]
    assert np.abs(W @ W.conj().T - np.eye(rows)).max() < 1e-12
    assert np.abs(W @ A - Y).max() < 1e-11
    ref = np.linalg.svd(A[:, 10:50], compute_uv=False)
    assert np.abs(sigma.cpu().numpy()[0, :rows] - ref).max() <= 1e-12 * ref[0]


@pytest.mark.parametrize("n", [5, 64, 80, 200])
def test_jacobi_herm(ctx, n):
    rng = np.random.default_rng(n)
    nb = 3
    X = crand(rng, nb, n, n)
    lam = 10.0 ** rng.uniform(-12, 0, (nb, n))
    Qm = np.linalg.qr(X)[0]
    C = (Qm * lam[:, None, :]) @ Qm.conj().transpose(0, 2, 1)
    C = 0.5 * (C + C.conj().transpose(0, 2, 1))
    dC = ctx.to_device(C)
    ev, W, sweeps = ctx.jacobi_herm(dC, n, n, strideC=n * n, batch=nb)
    ev = ev.cpu().numpy()
    W = W.cpu().numpy()
    for b in range(nb):
        ref = np.linalg.eigvalsh(C[b])
        assert np.abs(np.sort(ev[b, :n]) - ref).max() <= 2e-12 * np.abs(ref).max(), sweeps
        assert np.abs(W[b] @ W[b].conj().T - np.eye(n)).max() < 1e-12
        D = W[b] @ C[b] @ W[b].conj().T
        assert np.abs(D - np.diag(ev[b, :n])).max() <= 2e-12 * np.abs(ref).max()
    assert sweeps < 30


@pytest.mark.parametrize("n", [1, 2, 5, 33, 64, 80, 200, 517])
def test_herm_eig_tridiag(ctx, n):
    """Householder tridiagonalisation + QL + back-transformation (the production eigensolver)."""
    rng = np.random.default_rng(n)
    nb = 3
    X = crand(rng, nb, n, n)
    lam = 10.0 ** rng.uniform(-12, 0, (nb, n)) * rng.choice([1.0, 1.0, -1.0], (nb, n))
    Qm = np.linalg.qr(X)[0]
    C = (Qm * lam[:, None, :]) @ Qm.conj().transpose(0, 2, 1)
    C = 0.5 * (C + C.conj().transpose(0, 2, 1))
    dC = ctx.to_device(C)
    ev, W = ctx.herm_eig(dC, n, n, strideC=n * n, batch=nb)
    ev = ev.cpu().numpy()
    W = W.cpu().numpy()
    for b in range(nb):
        ref = np.linalg.eigvalsh(C[b])
        scale = np.abs(ref).max()
        assert np.abs(np.sort(ev[b, :n]) - ref).max() <= 5e-14 * scale
        assert np.abs(W[b] @ W[b].conj().T - np.eye(n)).max() < 5e-13
        D = W[b] @ C[b] @ W[b].conj().T
        assert np.abs(D - np.diag(ev[b, :n])).max() <= 5e-13 * scale


@pytest.mark.parametrize("kind", ["graded", "clustered"])
def test_herm_eig_tridiag_beyond_lds(ctx, kind):
    """n > 4096: the divide & conquer merge nodes no longer fit in LDS and take the global-scratch
    setup / secular kernels (config 3 has ndof up to 32 832).  The spectrum is prescribed:
    C = H diag(lam) H^H with H a product of three Householder reflectors (dense, exactly unitary)."""
    n = 4500
    rng = np.random.default_rng(4500)
    if kind == "graded":
        lam = 10.0 ** rng.uniform(-12, 0, n) * rng.choice([1.0, 1.0, -1.0], n)
    else:
        lam = rng.choice(np.linspace(0.5, 3.0, 40), n)  # heavy deflation
    C = np.diag(lam).astype(np.complex128)
    for _ in range(3):
        u = crand(rng, n)
        u /= np.linalg.norm(u)
        C -= 2.0 * np.outer(u, u.conj() @ C)
        C -= 2.0 * np.outer(C @ u, u.conj())
    C = 0.5 * (C + C.conj().T)
    ev, W = ctx.herm_eig(ctx.to_device(C), n, n, strideC=n * n, batch=1)
    ev = ev.cpu().numpy()[0, :n]
    W = W.cpu().numpy()[0]
    scale = np.abs(lam).max()
    assert np.abs(np.sort(ev) - np.sort(lam)).max() <= 2e-13 * scale
    assert np.abs(W @ W.conj().T - np.eye(n)).max() < 2e-12
    D = W @ C @ W.conj().T
    assert np.abs(D - np.diag(ev)).max() <= 2e-12 * scale


@pytest.mark.parametrize("n", [20, 66, 159, 300])
@pytest.mark.parametrize("scale", [1.0, 1e-9, 1e-18, 1e12])
def test_herm_eig_is_scale_invariant(ctx, n, scale):
    """The eigenvalues of s A are s times those of A to a few ulp whatever the norm of the matrix: the divide & conquer
    works on the tridiagonal scaled to unit max-norm, as LAPACK's dstedc does (DLASCL) — dlaed2's deflation tolerance
    compares poles with components of unit vectors.  (Unscaled, a matrix of norm 1e-9 lost six digits: the S/N pencils of
    the configs[1] blocks m = 101 .. 104 have lambda_max ~ 1e-10.)"""
    rng = np.random.default_rng(n)
    X = crand(rng, n, n)
    A = (X @ X.conj().T) * scale
    ref = np.linalg.eigvalsh(A)
    ev, _ = ctx.herm_eig(ctx.to_device(A[None].copy()), n, n)
    got = np.sort(ev.cpu().numpy()[0][:n])
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()


@pytest.mark.parametrize("twostage", ["0", "1"])
def test_herm_eig_with_exactly_zero_rows(ctx, monkeypatch, twostage):
    """Gram matrices in which most rows and columns are exactly zero (the tall SVD chains of blocks with l < m columns
    dropped get there; so does any rank-deficient covariance): the rounding residue of a reflector over a zero column is
    reflected again by every later sweep of the band chase and falls through the underflow threshold — the Householder
    scalars of such a column were 0/0 (NaN tridiagonals, "QL did not converge", status 1000 + p) until columns below
    DM_REFL_TINY were left alone.  Both reductions, eigenvalues to rounding and unitary vectors."""
    monkeypatch.setenv("DM_TRD_TWOSTAGE", twostage)
    rng = np.random.default_rng(5)
    n, K, nb = 452, 864, 4
    G = np.zeros((nb, n, n), dtype=np.complex128)
    for b in range(nb):
        A = crand(rng, n, K)
        A[rng.permutation(n)[: int(0.95 * n)]] = 0.0
        G[b] = A @ A.conj().T
    ref = np.linalg.eigvalsh(G)
    ev, W = ctx.herm_eig(ctx.to_device(G.copy()), n, n, strideC=n * n, batch=nb)
    got = np.sort(ev.cpu().numpy()[:, :n], axis=1)
    assert np.isfinite(got).all()
    assert np.abs(got - ref).max() <= 1e-13 * np.abs(ref).max()
    Wh = W.cpu().numpy()
    for b in range(nb):
        assert np.abs(Wh[b] @ Wh[b].conj().T - np.eye(n)).max() < 1e-12


def test_herm_eig_mixed_sizes_via_eigh_gen(ctx):
    """Different n in one batch (the KL use: ndof varies with m)."""
    from driftscan_amd._lib import block_offsets
    import scipy.linalg as la

    rng = np.random.default_rng(5)
    ns = [70, 3, 129, 1, 40]
    off, tot = block_offsets(ns)
    As, Bs = [], []
    for n in ns:
        X, Y = crand(rng, n, n), crand(rng, n, n)
        As.append(X @ X.conj().T)
        Bs.append(Y @ Y.conj().T + n * np.eye(n))
    A = np.concatenate([a.ravel() for a in As])
    B = np.concatenate([b.ravel() for b in Bs])
    evals, evoff, evecs, ac, _ = ctx.eigh_gen(ctx.to_device(A), ctx.to_device(B), ns, off)
    ev = evals.cpu().numpy()
    E = evecs.cpu().numpy()
    for i, n in enumerate(ns):
        ref = la.eigh(As[i], Bs[i], eigvals_only=True)
        assert np.abs(ev[evoff[i]: evoff[i] + n] - ref).max() <= 1e-11 * ref.max()
        Ei = E[off[i]: off[i] + n * n].reshape(n, n)
        assert np.abs(Ei @ Bs[i] @ Ei.conj().T - np.eye(n)).max() < 1e-10


def test_workspace_reset(ctx):
    """dm_ctx_workspace_reset: the idle arena is handed back / re-reserved; work afterwards is unaffected."""
    rng = np.random.default_rng(11)
    n = 40
    C = crand(rng, 1, n, n)
    C = C @ C.conj().transpose(0, 2, 1)
    ref = np.linalg.eigvalsh(C[0])
    for nbytes in (0, 300 << 20):
        ctx.workspace_reset(nbytes)
        assert ctx.lib.dm_ctx_workspace_bytes(ctx.h) >= nbytes
        ev, _ = ctx.herm_eig(ctx.to_device(C), n, n, strideC=n * n, batch=1)
        assert np.abs(np.sort(ev.cpu().numpy()[0, :n]) - ref).max() <= 1e-12 * ref.max()


def test_driftcomm_single_rank(ctx):
    """The RCCL wrappers of include/driftcomm.h with one rank (all a one-GPU box can run): id, init, all-reduce and
    gather of doubles are the identity, rank / size are reported, sync returns."""
    import torch

    from driftscan_amd import comm

    c = comm.Communicator(1, 0, comm.unique_id(), device=0)
    assert c.rank == 0 and c.size == 1
    x = torch.arange(1000, dtype=torch.float64, device="cuda") * 0.5
    ref = x.clone()
    c.allreduce(x)
    out = torch.zeros(1000, dtype=torch.float64, device="cuda")
    c.gather(x, out, root=0)
    c.sync()
    assert torch.equal(x, ref) and torch.equal(out, ref)
    c.close()


def test_driftcomm_orders_against_the_producer_stream(ctx):
    """include/driftcomm.h, "Stream ordering": a collective on the communicator's private stream starts only after the
    kernels enqueued on the caller's stream have written the buffer, and later work on the caller's stream sees its
    result — checked with a producer that is still in flight when the collective is enqueued (a long chain of fills on
    a side stream), without any host synchronisation in between."""
    import torch

    from driftscan_amd import comm

    c = comm.Communicator(1, 0, comm.unique_id(), device=0)
    n = 1 << 24  # 128 MB of doubles
    side = torch.cuda.Stream()
    send = torch.zeros(n, dtype=torch.float64, device="cuda")
    recv = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(40):  # ~40 passes over 128 MB: still running when the gather below is enqueued
            send.add_(1.0)
        c.gather(send, recv, root=0)          # user stream = torch's current stream = `side`
        out = recv * 2.0                      # consumer on the same stream, enqueued right behind the collective
        c.allreduce(send)                     # in place, one rank: the identity, ordered behind the gather's read
        tail = send + 1.0
    side.synchronize()
    c.sync()
    assert float(recv.min()) == 40.0 and float(recv.max()) == 40.0
    assert float(out.min()) == 80.0 and float(out.max()) == 80.0
    assert float(tail.min()) == 41.0 and float(tail.max()) == 41.0
    c.close()
