"""GPU parity of the SVD chain, covariance projections and generalised eigensolver
(through the C ABI) against golden vectors produced by the unmodified reference."""
import os

import numpy as np
import pytest

from parity_util import assert_spectrum, pencil_sensitivity, pencil_tol, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from driftscan_amd._lib import Context

    c = Context(0, workspace_bytes=2 << 30)
    yield c
    c.close()


@pytest.fixture(scope="module", params=["unpol", "pol", "unpol_harsh"])
def gold(request, golden_dir):
    return np.load(os.path.join(golden_dir, "svdkl_%s.npz" % request.param))


def _run_svd(ctx, g):
    F, B, P, L = int(g["F"]), int(g["B"]), int(g["P"]), int(g["lmax"]) + 1
    T = 2 * B
    mlist = list(g["mlist"])
    beam = np.stack([g["m%d_beam_m" % m].reshape(F, T, P, L) for m in mlist])
    nw = np.concatenate([g["npower"], g["npower"]], axis=1) ** -0.5
    res = ctx.svd_chain(ctx.to_device(beam), ctx.to_device(nw), float(g["polsvcut"]))
    return mlist, res


def test_svd_chain(ctx, gold):
    g = gold
    from oracle import svdchain as osvd

    mlist, res = _run_svd(ctx, g)
    F = int(g["F"])
    sv = res["singularvalues"].cpu().numpy()
    bs = res["beam_svd"].cpu().numpy()
    ut = res["beam_ut"].cpu().numpy()
    ib = res["invbeam_svd"].cpu().numpy()
    assert max(res["sweeps"]) < 30
    for bi, mi in enumerate(mlist):
        pre = "m%d_" % mi
        ref = g[pre + "singularvalues"]
        assert_spectrum(sv[bi], ref, 1e-10, "singular values m=%d" % mi)
        svnum, svbounds = osvd.svd_num(sv[bi], float(g["svcut"]))
        assert (svnum == g[pre + "svnum"]).all()
        for f in range(F):
            n = svnum[f]
            b0 = g[pre + "beam_svd"][f, :n].reshape(n, -1)
            b1 = bs[bi, f, :n].reshape(n, -1)
            assert relerr(b1.T.conj() @ b1, b0.T.conj() @ b0) < 1e-9
            u0, u1 = g[pre + "beam_ut"][f, :n], ut[bi, f, :n]
            assert relerr(u1.T.conj() @ u1, u0.T.conj() @ u0) < 1e-9
            K = ib.shape[-1]
            i0 = g[pre + "invbeam_svd"][f].reshape(-1, K)[:, :n]
            i1 = ib[bi, f].reshape(-1, K)[:, :n]
            assert relerr(i1 @ b1, i0 @ b0) < 1e-7
            # rows past nmodes stay zero, like the reference's zero-initialised datasets
            nm = res["nmodes"][bi, f]
            assert np.abs(bs[bi, f, nm:]).max(initial=0.0) == 0.0


def _projections(ctx, g, mlist, bsvd, but, svnum_all):
    from driftscan_amd._lib import block_offsets

    F, P, L = int(g["F"]), int(g["P"]), int(g["lmax"]) + 1
    ndofs = svnum_all.sum(axis=1)
    off, tot = block_offsets(ndofs)
    return ndofs, off, tot


def test_projections_and_eigh(ctx, gold):
    g = gold
    from driftscan_amd._lib import block_offsets

    F, P, L = int(g["F"]), int(g["P"]), int(g["lmax"]) + 1
    mlist = list(g["mlist"])
    # feed the *reference's* SVD products so that this test isolates the projections
    bsvd = ctx.to_device(np.stack([g["m%d_beam_svd" % m] for m in mlist]))
    but = ctx.to_device(np.stack([g["m%d_beam_ut" % m] for m in mlist]))
    svnum = np.stack([g["m%d_svnum" % m] for m in mlist])
    ndofs = svnum.sum(axis=1)
    off, tot = block_offsets(ndofs)
    cl_sg = ctx.to_device(np.ascontiguousarray(g["cv_sg"].transpose(0, 1, 3, 4, 2)))
    cl_fg = ctx.to_device(np.ascontiguousarray(g["cv_fg"].transpose(0, 1, 3, 4, 2)))
    pm_sg = (np.abs(g["cv_sg"]).reshape(P, P, -1).max(axis=-1) > 0).astype(np.int32)
    pm_fg = (np.abs(g["cv_fg"]).reshape(P, P, -1).max(axis=-1) > 0).astype(np.int32)
    S = ctx.empty((tot,), np.complex128)
    N = ctx.empty((tot,), np.complex128)
    ctx.project_cov(bsvd, svnum, cl_sg, S, off, polmask=pm_sg, l0=np.array(mlist))
    ctx.project_cov(bsvd, svnum, cl_fg, N, off, polmask=pm_fg, l0=np.array(mlist))
    ctx.sync()
    Sh, Nh = S.cpu().numpy(), N.cpu().numpy()
    for bi, mi in enumerate(mlist):
        n = ndofs[bi]
        s = Sh[off[bi]: off[bi] + n * n].reshape(n, n)
        f = Nh[off[bi]: off[bi] + n * n].reshape(n, n)
        assert relerr(s, g["m%d_proj_sg" % mi]) < 1e-12
        assert relerr(f, g["m%d_proj_fg" % mi]) < 1e-12
    # N = fg + reg max + noise  -> compare with the reference's sn_covariance
    ctx.regularise(N, ndofs, off, 1e-14)
    npw = ctx.to_device(np.concatenate([g["npower"], g["npower"]], axis=1))
    ctx.project_diag(but, svnum, npw, N, off, alpha=1.0, accumulate=True)
    ctx.sync()
    Nh = N.cpu().numpy()
    for bi, mi in enumerate(mlist):
        n = ndofs[bi]
        assert relerr(Nh[off[bi]: off[bi] + n * n].reshape(n, n), g["m%d_kl_cn" % mi]) < 1e-12
    # generalised eigenproblem, all m at once
    evals, evoff, evecs, ac, sweeps = ctx.eigh_gen(S, N, ndofs, off)
    ev = evals.cpu().numpy()
    E = evecs.cpu().numpy()
    assert (ac == 0.0).all()
    for bi, mi in enumerate(mlist):
        n = ndofs[bi]
        ref = g["m%d_kl_evals" % mi]
        cs, cn = g["m%d_kl_cs" % mi], g["m%d_kl_cn" % mi]
        tol = pencil_tol(cn)
        if tol > 1e-6:
            # foreground-dominated pencil: use the measured sensitivity of LAPACK's own answer
            tol = max(1e-10, 10.0 * pencil_sensitivity(cs, cn))
        assert tol < 5e-2
        assert_spectrum(ev[evoff[bi]: evoff[bi] + n], ref, tol, "kl evals m=%d" % mi)
        Eb = E[off[bi]: off[bi] + n * n].reshape(n, n)
        eref = g["m%d_kl_evecs" % mi]
        rn_ref = relerr(eref @ cn @ eref.T.conj(), np.eye(n), 1.0)
        rs_ref = relerr(eref @ cs @ eref.T.conj(), np.diag(ref), np.abs(ref).max())
        assert relerr(Eb @ cn @ Eb.T.conj(), np.eye(n), 1.0) <= max(10 * rn_ref, 10 * tol, 1e-8)
        assert relerr(Eb @ cs @ Eb.T.conj(), np.diag(ev[evoff[bi]: evoff[bi] + n]), np.abs(ref).max()) <= max(10 * rs_ref, 10 * tol, 1e-8)


def test_eigh_gen_rescue_and_zero(ctx, golden_dir):
    from driftscan_amd._lib import block_offsets

    g = np.load(os.path.join(golden_dir, "eigh_gen.npz"))
    cases = ["pd", "npd", "zero"]
    ndofs = np.array([g[c + "_A"].shape[0] for c in cases])
    off, tot = block_offsets(ndofs)
    A = np.concatenate([g[c + "_A"].ravel() for c in cases])
    B = np.concatenate([g[c + "_B"].ravel() for c in cases])
    evals, evoff, evecs, ac, sweeps = ctx.eigh_gen(ctx.to_device(A), ctx.to_device(B), ndofs, off)
    ev = evals.cpu().numpy()
    for i, c in enumerate(cases):
        n = ndofs[i]
        Bc = g[c + "_B"] + float(g[c + "_ac"]) * np.eye(n)
        tol = pencil_tol(Bc) if c != "zero" else 1e-12
        assert_spectrum(ev[evoff[i]: evoff[i] + n], g[c + "_evals"], tol, c)
    assert ac[0] == 0.0 and ac[2] == 0.0
    # the shift is 1e-15 ev_max - 2 ev_min + 1e-60 with ev_min ~ -1e-9: well determined
    assert np.isclose(ac[1], float(g["npd_ac"]), rtol=1e-5)


def test_svd_chain_wide_dynamic_range(ctx):
    """A polarised block whose spectrum runs over 15 decades with a rank-deficient polarised part, as a
    config-3 beam block does (864 rows, several hundred of them below 1e-8 sigma_1): the rows that the first
    Gram preconditioner cannot resolve go through its second and third level.  Checked against the oracle
    (the restatement pinned on the reference) on the same input."""
    from oracle import svdchain as osvd

    rng = np.random.default_rng(77)
    F, B, P, L = 2, 100, 4, 60
    T = 2 * B
    beam = np.zeros((F, T, P, L), dtype=np.complex128)
    for f in range(F):
        u = np.linalg.qr(rng.standard_normal((T, T)) + 1j * rng.standard_normal((T, T)))[0]
        v = np.linalg.qr(rng.standard_normal((L, L)) + 1j * rng.standard_normal((L, L)))[0]
        s = 10.0 ** np.linspace(0, -15, L)
        beam[f, :, 0, :] = (u[:, :L] * s) @ v.conj().T
        r = T // 4
        a = rng.standard_normal((T, r)) + 1j * rng.standard_normal((T, r))
        c = rng.standard_normal((r, 3 * L)) + 1j * rng.standard_normal((r, 3 * L))
        beam[f, :, 1:, :] = 0.05 * (a @ c).reshape(T, 3, L) * 10.0 ** rng.uniform(-6, 0, (T, 1, 1))
    npower = rng.uniform(0.5, 2.0, (F, B))
    nw = np.concatenate([npower, npower], axis=1) ** -0.5
    polsvcut, svcut = 1e-4, 1e-6
    res = ctx.svd_chain(ctx.to_device(beam[None]), ctx.to_device(nw), polsvcut)
    ref = osvd.svd_m(beam.reshape(F, 2, B, P, L), npower ** -0.5, polsvcut=polsvcut)
    sv = res["singularvalues"].cpu().numpy()[0]
    assert_spectrum(sv, ref["singularvalues"], 1e-10, "singular values")
    svnum, _ = osvd.svd_num(sv, svcut)
    svnum_ref, _ = osvd.svd_num(ref["singularvalues"], svcut)
    assert (svnum == svnum_ref).all() and svnum.min() > 0
    assert max(res["sweeps"]) <= 6, res["sweeps"]  # 12-20 without the deeper preconditioner levels
    bs = res["beam_svd"].cpu().numpy()[0]
    ut = res["beam_ut"].cpu().numpy()[0]
    ib = res["invbeam_svd"].cpu().numpy()[0]
    for f in range(F):
        n = svnum[f]
        b0, b1 = ref["beam_svd"][f, :n].reshape(n, -1), bs[f, :n].reshape(n, -1)
        assert relerr(b1.T.conj() @ b1, b0.T.conj() @ b0) < 1e-9
        u1 = ut[f, :n] / nw[f][None, :]
        assert np.abs(u1 @ u1.conj().T - np.eye(n)).max() < 1e-12
        K = ib.shape[-1]
        i1 = ib[f].reshape(-1, K)[:, :n]
        assert np.abs(b1 @ i1 - np.eye(n)).max() < 1e-8  # kappa = 1 / svcut


def test_svd_chain_subspace_phases_match_converged_ones(ctx, monkeypatch):
    """SVD1 hands its image and SVD2 its null space to the next phase as SUBSPACES: the products do not depend on the
    basis inside them, so those two phases stop when the preconditioner level that holds the rows next to the cut has
    placed it, without Jacobi sweeps (`dm_jac_rows_opts::subspace_cut`; a configs[2] batch 4.47 -> 3.39 s).  Against the
    converged phases (DM_SVD_SUBSPACE=0) on a block with several hundred rows below 1e-8 sigma_1 and polarised singular
    values on both sides of `polsvcut`: the same spectrum to 1e-12 sigma_max — the north-star tolerance is 1e-10; the
    first version, which left SVD2's split to the first level, was at 5e-9 — and the same row spaces."""
    rng = np.random.default_rng(78)
    F, B, P, L = 2, 200, 4, 120
    T = 2 * B
    beam = np.zeros((F, T, P, L), dtype=np.complex128)
    for f in range(F):
        u = np.linalg.qr(rng.standard_normal((T, T)) + 1j * rng.standard_normal((T, T)))[0]
        v = np.linalg.qr(rng.standard_normal((L, L)) + 1j * rng.standard_normal((L, L)))[0]
        beam[f, :, 0, :] = (u[:, :L] * 10.0 ** np.linspace(0, -15, L)) @ v.conj().T
        r = T // 2
        a = np.linalg.qr(rng.standard_normal((T, r)) + 1j * rng.standard_normal((T, r)))[0]
        c = rng.standard_normal((r, 3 * L)) + 1j * rng.standard_normal((r, 3 * L))
        beam[f, :, 1:, :] = ((a * 10.0 ** np.linspace(0, -9, r)) @ c).reshape(T, 3, L) * 0.05   # polarised part: sigma over 9 decades
    nw = rng.uniform(0.5, 2.0, (F, T))
    polsvcut = 1e-4
    dev = ctx.to_device(beam[None])
    fast = ctx.svd_chain(dev, ctx.to_device(nw), polsvcut)
    monkeypatch.setenv("DM_SVD_SUBSPACE", "0")
    full = ctx.svd_chain(dev, ctx.to_device(nw), polsvcut)
    monkeypatch.delenv("DM_SVD_SUBSPACE")
    assert (fast["nmodes"] == full["nmodes"]).all() and fast["nmodes"].min() > 10
    s0, s1 = full["singularvalues"].cpu().numpy()[0], fast["singularvalues"].cpu().numpy()[0]
    assert np.abs(s0 - s1).max() <= 1e-12 * s0.max()
    keep = s0 > 1e-6 * s0.max(axis=1, keepdims=True)
    assert (np.abs(s0 - s1)[keep] <= 1e-10 * s0[keep]).all()
    assert fast["sweeps"][0] == 0 or fast["sweeps"][0] < full["sweeps"][0]   # SVD1 placed its cut without sweeps
    for f in range(F):
        n = int((s0[f] > 1e-6 * s0[f].max()).sum())
        b0 = full["beam_svd"].cpu().numpy()[0, f, :n].reshape(n, -1)
        b1 = fast["beam_svd"].cpu().numpy()[0, f, :n].reshape(n, -1)
        assert relerr(b1.T.conj() @ b1, b0.T.conj() @ b0) < 1e-9
        u1 = fast["beam_ut"].cpu().numpy()[0, f, :n] / nw[f][None, :]
        assert np.abs(u1 @ u1.conj().T - np.eye(n)).max() < 1e-12



@pytest.mark.parametrize("P", [1, 4])
def test_svd_chain_lmin_tall_and_narrow_routes_match_the_plain_chain(ctx, monkeypatch, P):
    """`dm_svd_chain_lmin` has four routes of its own: the chains compacted to the columns l >= lmin, the transposed "tall"
    SVD1 / SVD3 (P (L - m) <= 0.95 T), the narrow SVD2 with a separate Z3 for SVD3, the per-polarisation pinv scatter.  Each
    against the plain chain — lmin = None, DM_SVD_TALL = 0, DM_SVD_NARROW = 0, DM_SVD_NO_COMPACT — on blocks that are zero
    below l = m, wide (m = 0, 3) and tall (m = 14, 17: P (L - m) = 24 and 12 of T = 40) and rank-deficient in the polarised
    part, and against the oracle's restatement of beamtransfer.py:802-924: singular values, nmodes, the row space of
    beam_ut (whitened rows orthonormal to 1e-12), beam_svd and invbeam_svd beam_svd = 1 on the kept modes."""
    from oracle import svdchain as osvd

    rng = np.random.default_rng(100 + P)
    F, B, L = 3, 20, 20
    T = 2 * B
    ms = [0, 3, 14, 17]
    beam = np.zeros((len(ms), F, T, P, L), dtype=np.complex128)
    for i, m in enumerate(ms):
        for f in range(F):
            lc = L - m
            r = min(T, lc)
            u = np.linalg.qr(rng.standard_normal((T, T)) + 1j * rng.standard_normal((T, T)))[0]
            v = np.linalg.qr(rng.standard_normal((lc, lc)) + 1j * rng.standard_normal((lc, lc)))[0]
            beam[i, f, :, 0, m:] = (u[:, :r] * 10.0 ** np.linspace(0, -7, r)) @ v[:, :r].conj().T
            if P > 1:
                k = max(2, r // 2)   # polarised part of rank k sharing half of its row space with the temperature part
                a = np.concatenate([u[:, : k // 2], np.linalg.qr(rng.standard_normal((T, k - k // 2)) + 0j)[0]], axis=1)
                c = rng.standard_normal((k, 3 * lc)) + 1j * rng.standard_normal((k, 3 * lc))
                beam[i, f, :, 1:, m:] = ((a * 10.0 ** np.linspace(0, -6, k)) @ c).reshape(T, 3, lc) * 0.05
    nw = rng.uniform(0.5, 2.0, (F, T))
    polsvcut, svcut = 1e-4, 1e-6
    dev, dnw = ctx.to_device(beam), ctx.to_device(nw)

    def run(lmin=None, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        try:
            r = ctx.svd_chain(dev, dnw, polsvcut, lmin=lmin)
            ctx.sync()
        finally:
            for k in env:
                monkeypatch.delenv(k)
        return dict(sv=r["singularvalues"].cpu().numpy(), bs=r["beam_svd"].cpu().numpy(), ut=r["beam_ut"].cpu().numpy(),
                    ib=r["invbeam_svd"].cpu().numpy(), nmodes=np.array(r["nmodes"]))

    base = run(lmin=ms)
    variants = dict(no_lmin=run(lmin=None), no_compact=run(lmin=ms, DM_SVD_NO_COMPACT="1"), not_tall=run(lmin=ms, DM_SVD_TALL="0"),
                    not_narrow=run(lmin=ms, DM_SVD_NARROW="0"), plainest=run(lmin=None, DM_SVD_TALL="0", DM_SVD_NARROW="0"))
    for i, m in enumerate(ms):
        smax = base["sv"][i].max()
        for name, var in variants.items():
            assert (var["nmodes"][i] == base["nmodes"][i]).all(), (name, m)
            assert np.abs(var["sv"][i] - base["sv"][i]).max() <= 1e-11 * smax, (name, m)
        for f in range(F):
            n = int((base["sv"][i, f] > svcut * smax).sum())
            if n == 0:
                continue
            U = base["ut"][i, f, :n] / nw[f][None, :]
            assert np.abs(U @ U.conj().T - np.eye(n)).max() < 1e-12, (m, f)
            b = base["bs"][i, f, :n].reshape(n, P * L)
            assert not b.reshape(n, P, L)[:, :, :m].any()                       # l < m stays zero in the padded products
            Bw = beam[i, f].reshape(T, P * L) * nw[f][:, None]
            assert np.abs(U @ Bw - b).max() <= 1e-11 * smax
            ib = base["ib"][i, f].reshape(P * L, -1)[:, :n]
            assert np.abs(b @ ib - np.eye(n)).max() < 1e-7
            for name, var in variants.items():
                U2 = var["ut"][i, f, :n] / nw[f][None, :]
                assert np.abs(U2.conj().T @ U2 - U.conj().T @ U).max() < 1e-8, (name, m, f)       # the same row space
                b2 = var["bs"][i, f, :n].reshape(n, P * L)
                assert relerr(b2.conj().T @ b2, b.conj().T @ b) < 1e-8, (name, m, f)
                ib2 = var["ib"][i, f].reshape(P * L, -1)[:, :n]
                assert relerr(ib2 @ b2, ib @ b) < 1e-6, (name, m, f)
    # ... and the oracle on the same blocks (noise weights per baseline, duplicated for the two m signs inside)
    nwb = rng.uniform(0.5, 2.0, (F, B))
    nw2 = np.concatenate([nwb, nwb], axis=1)
    res = ctx.svd_chain(dev, ctx.to_device(nw2), polsvcut, lmin=ms)
    sv = res["singularvalues"].cpu().numpy()
    for i, m in enumerate(ms):
        o = osvd.svd_m(beam[i].reshape(F, 2, B, P, L), nwb, polsvcut=polsvcut)
        assert np.abs(o["singularvalues"] - sv[i]).max() <= 1e-10 * sv[i].max(), m
        # svnum is what defines the shapes downstream (beamtransfer.py:1116-1133).  nmodes itself is not comparable here: the
        # reference counts sigma > 0 (rtol = 0), and of the P m exactly-zero columns LAPACK returns some sigma ~ 1e-17 sigma_max
        # as positive where the compacted chain has none at all (19 against 17 at m = 3, P = 1)
        n_o = (o["singularvalues"] > svcut * o["singularvalues"].max()).sum(axis=1)
        n_g = (sv[i] > svcut * sv[i].max()).sum(axis=1)
        assert (n_o == n_g).all() and (np.asarray(res["nmodes"][i]) >= n_g).all(), m


def test_svd_chain_frequency_slices(ctx, gold):
    """The (m, frequency) chains are independent: pushing the frequencies through the library in slices
    (what a CHIME-sized block needs, 111 GB of augmented matrices otherwise) changes nothing."""
    g = gold
    F, B, P, L = int(g["F"]), int(g["B"]), int(g["P"]), int(g["lmax"]) + 1
    T = 2 * B
    mlist = list(g["mlist"])
    beam = ctx.to_device(np.stack([g["m%d_beam_m" % m].reshape(F, T, P, L) for m in mlist]))
    nw = ctx.to_device(np.concatenate([g["npower"], g["npower"]], axis=1) ** -0.5)
    whole = ctx.svd_chain(beam, nw, float(g["polsvcut"]))
    per_chain = 16.0 * (2.0 * T * (P * L + T) + 16.0 * T * T)
    sliced = ctx.svd_chain(beam, nw, float(g["polsvcut"]), max_bytes=per_chain * len(mlist) * 1.5)  # one frequency per call
    assert (whole["nmodes"] == sliced["nmodes"]).all()
    s0, s1 = whole["singularvalues"].cpu().numpy(), sliced["singularvalues"].cpu().numpy()
    assert np.abs(s0 - s1).max() <= 1e-12 * s0.max()
    for k in ("beam_svd", "beam_ut", "invbeam_svd"):
        a, b = whole[k].cpu().numpy(), sliced[k].cpu().numpy()
        # same row spaces; the rows themselves may differ by a phase
        if k == "beam_svd":
            ga = np.einsum("bfkpl,bfkpl->bfk", a.conj(), a).real
            gb = np.einsum("bfkpl,bfkpl->bfk", b.conj(), b).real
            assert np.abs(ga - gb).max() <= 1e-10 * ga.max()
        assert a.shape == b.shape


@pytest.mark.parametrize("P", [1, 4])
def test_svd_chain_degenerate_blocks(ctx, P):
    """All-zero blocks and blocks with a single non-zero multipole (m = lmax of a polarised telescope: every
    row is cut after SVD1 / SVD2): datasets stay blank as in the reference (beamtransfer.py:855-872)."""
    from oracle import svdchain as osvd

    rng = np.random.default_rng(3)
    F, B, L = 2, 10, 12
    T = 2 * B
    beam = np.zeros((2, F, T, P, L), dtype=np.complex128)     # block 0: zeros; block 1: only l = L - 1
    beam[1, :, :, :, L - 1] = rng.standard_normal((F, T, P)) + 1j * rng.standard_normal((F, T, P))
    npower = rng.uniform(0.5, 2.0, (F, B))
    nw = np.concatenate([npower, npower], axis=1) ** -0.5
    for blocks in (beam[:1], beam):
        res = ctx.svd_chain(ctx.to_device(blocks), ctx.to_device(nw), 1e-4)
        sv = res["singularvalues"].cpu().numpy()
        for b in range(blocks.shape[0]):
            ref = osvd.svd_m(blocks[b].reshape(F, 2, B, P, L), npower ** -0.5, polsvcut=1e-4)
            # `nmodes` counts s > 0.0 (rtol = 0, beamtransfer.py:863): for the rank-one block that is a count of
            # rounding residues, in LAPACK as here; what is defined is the spectrum and the count above svcut
            assert_spectrum(sv[b], ref["singularvalues"], 1e-10, "degenerate block %d" % b)
            assert (osvd.svd_num(sv[b], 1e-6)[0] == osvd.svd_num(ref["singularvalues"], 1e-6)[0]).all()
            if not blocks[b].any():
                assert (res["nmodes"][b] == 0).all() and (ref["nmodes"] == 0).all()
        assert np.abs(res["beam_svd"].cpu().numpy()[0]).max() == 0.0


def test_empty_kl_batches(ctx):
    """Batches whose blocks all have ndof = 0 (the highest m of a polarised telescope keep no mode): every entry
    point returns without launching anything (kltransform.py:322-331, the nside == 0 early-out)."""
    from driftscan_amd._lib import block_offsets

    ndofs = np.array([0, 0], dtype=np.int64)
    off, tot = block_offsets(ndofs)
    assert tot == 0
    S = ctx.empty((1,), np.complex128)
    N = ctx.empty((1,), np.complex128)
    ctx.regularise(N, ndofs, off, 1e-14)
    for cut in (None, ("upper", 0.1)):
        evals, evoff, evecs, ac, _ = ctx.eigh_gen(S, N, ndofs, off, cut=cut)
        assert list(evoff) == [0, 0, 0] and (ac == 0).all()
    # mixed: an empty block between two real ones
    rng = np.random.default_rng(0)
    ns = [5, 0, 7]
    off, tot = block_offsets(ns)
    As, Bs = [], []
    for n in ns:
        X = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        Y = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        As.append(X @ X.conj().T)
        Bs.append(Y @ Y.conj().T + n * np.eye(n))
    A = np.concatenate([a.ravel() for a in As]) if tot else np.zeros(1, complex)
    Bm = np.concatenate([b.ravel() for b in Bs])
    evals, evoff, evecs, ac, _ = ctx.eigh_gen(ctx.to_device(A), ctx.to_device(Bm), ns, off)
    import scipy.linalg as la

    ev = evals.cpu().numpy()
    for i, n in enumerate(ns):
        if n:
            ref = la.eigh(As[i], Bs[i], eigvals_only=True)
            assert np.abs(ev[evoff[i]: evoff[i] + n] - ref).max() <= 1e-11 * ref.max()
