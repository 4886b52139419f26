"""numpy prototype of the blocked one-sided (row) Jacobi engine with a Gram-column mask."""
import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo')

def round_robin_pairs(nb):
    """tournament schedule: list of rounds, each a list of (i,j) block pairs (nb even)."""
    idx = list(range(nb))
    rounds = []
    for r in range(nb - 1):
        pairs = [(idx[i], idx[nb - 1 - i]) for i in range(nb // 2)]
        rounds.append(pairs)
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return rounds

def block_jacobi_rows(Z, gcols, b=8, tol=1e-14, maxsweeps=30, verbose=False):
    """Orthogonalise the rows of Z[:, gcols] by unitary row mixing applied to all of Z.
    Returns rotated Z (rows sorted by descending norm of masked part), sigma."""
    Z = Z.copy()
    n = Z.shape[0]
    nb = -(-n // b)
    if nb % 2: nb += 1
    rounds = round_robin_pairs(nb) if nb > 1 else [[(0,0)]]
    for sweep in range(maxsweeps):
        maxoff = 0.0
        for pairs in rounds:
            for (bi, bj) in pairs:
                rows = np.r_[np.arange(bi*b, min((bi+1)*b, n)), np.arange(bj*b, min((bj+1)*b, n))] if bi != bj else np.arange(bi*b, min((bi+1)*b,n))
                rows = rows[rows < n]
                if rows.size < 2: continue
                X = Z[rows][:, gcols]
                G = X @ X.conj().T
                d = np.sqrt(np.abs(np.diag(G).real))
                dd = np.outer(d, d); dd[dd == 0] = 1.0
                off = np.abs(G - np.diag(np.diag(G))) / dd
                mo = off.max()
                maxoff = max(maxoff, mo)
                if mo <= tol: continue
                w, Q = np.linalg.eigh(G)   # G = Q diag(w) Q^H
                Q = Q[:, ::-1]
                Z[rows] = Q.conj().T @ Z[rows]
        if verbose: print("sweep", sweep, "maxoff", maxoff)
        if maxoff <= tol: break
    sig = np.linalg.norm(Z[:, gcols], axis=1)
    order = np.argsort(-sig, kind="stable")
    return Z[order], sig[order], sweep + 1

def svd_chain_jacobi(beam_f, nw, P, L, polsvcut, b=8):
    T = beam_f.shape[0]
    bfr = (beam_f.reshape(T, P, L) * nw[:, None, None]).reshape(T, P * L)
    Z = np.concatenate([bfr, np.eye(T)], axis=1)
    PL = P * L
    if P > 1:
        Z, s1, sw1 = block_jacobi_rows(Z, np.arange(PL), b)
        r1 = int((s1 > s1[0] * 1e-10).sum())
        Z = Z[:r1]
        Z, s2, sw2 = block_jacobi_rows(Z, np.arange(L, PL), b)
        cut = int((s2 >= s2[0] * polsvcut).sum())
        Z = Z[cut:]
    Z, s3, sw3 = block_jacobi_rows(Z, np.arange(L), b)
    nmodes = int((s3 > 0).sum())
    return Z[:nmodes, :PL], Z[:nmodes, PL:], s3[:nmodes]

if __name__ == "__main__":
    from oracle import svdchain
    for tag in ("unpol", "pol"):
        g = np.load(f"/root/repo/tests/golden/svdkl_{tag}.npz")
        F, B, P, L = int(g["F"]), int(g["B"]), int(g["P"]), int(g["lmax"]) + 1
        for m in g["mlist"]:
            bm = g[f"m{m}_beam_m"]
            for f in range(F):
                nw = np.concatenate([g["npower"][f]]*2) ** -0.5
                beam, ut, s3 = svd_chain_jacobi(bm[f].reshape(2*B, P, L), nw, P, L, float(g["polsvcut"]))
                ref = g[f"m{m}_singularvalues"][f]
                n = min(len(s3), (ref>0).sum())
                print(tag, m, f, "nmodes", len(s3), "ref nmodes", (ref > 0).sum(), "sig err", np.abs(s3[:n]-ref[:n]).max()/ref.max())
                # compare B^H B
                nk = g[f"m{m}_svnum"][f]
                b0 = g[f"m{m}_beam_svd"][f,:nk].reshape(nk,-1)
                print("   BhB err", np.abs(beam[:nk].conj().T@beam[:nk] - b0.conj().T@b0).max()/np.abs(b0.conj().T@b0).max())
