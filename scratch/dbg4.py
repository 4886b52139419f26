import numpy as np, scipy.linalg as la, sys
sys.path.insert(0,'/root/repo/scratch')
import proto_jacobi2 as pj
from proto_jacobi import round_robin_pairs
g = np.load("/root/repo/tests/golden/svdkl_unpol.npz")
cs, cn = g["m0_kl_cs"], g["m0_kl_cn"]
n = cs.shape[0]
L = la.cholesky(cn, lower=True)
X = la.solve_triangular(L, cs, lower=True)
C = la.solve_triangular(L, X.conj().T, lower=True).conj().T
C = 0.5*(C+C.conj().T)
Z = np.concatenate([C, np.eye(n)], axis=1)
for it in range(8):
    Z, s, sw = pj.block_jacobi_rows2(Z, np.arange(n), 8, tol=1e-13, maxsweeps=1)
def cosij(Z,i,j):
    a,b = Z[i,:n], Z[j,:n]
    return abs(a@b.conj())/np.linalg.norm(a)/np.linalg.norm(b)
b=8
print("start cos(6,41)", cosij(Z,6,41))
for pairs in round_robin_pairs(10):
    for (bi,bj) in pairs:
        rows = np.r_[np.arange(bi*b,(bi+1)*b), np.arange(bj*b,(bj+1)*b)]
        Xp = Z[rows][:, :n]; G = Xp@Xp.conj().T
        Q,w = pj.herm_jacobi_evd(G)
        order = np.argsort(-w, kind="stable"); 
        moved = not (order == np.arange(16)).all()
        Q = Q[:, order]
        Z[rows] = Q.conj().T@Z[rows]
        if bi in (0,5) or bj in (0,5):
            print((bi,bj), "cos(6,41)", cosij(Z,6,41), "moved" if moved else "", "norm41", np.linalg.norm(Z[41,:n]))
