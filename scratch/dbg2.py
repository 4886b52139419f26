import numpy as np, sys
sys.path.insert(0,'/root/repo/scratch')
from proto_jacobi2 import herm_jacobi_evd
rng=np.random.default_rng(1)
for trial in range(5):
    X = rng.standard_normal((6,9))+1j*rng.standard_normal((6,9))
    X *= 10.0**(-3*np.arange(6))[:,None]
    G = X@X.conj().T
    Q,w = herm_jacobi_evd(G)
    D = Q.conj().T@G@Q
    d = np.sqrt(np.abs(np.diag(D))); off = np.abs(D)/np.outer(d,d); np.fill_diagonal(off,0)
    print("unitary err", np.abs(Q.conj().T@Q-np.eye(6)).max(), "rel off", off.max(), "w", w)
