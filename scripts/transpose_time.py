"""Developer tool (GPU box): time of materialising a transpose view at 8192^2 (k_transpose) + a ragged-shape check."""
import sys; sys.path.insert(0,'.')
from runmat_amd import HipProvider
import numpy as np
p=HipProvider(0)
n=8192
x=p.fill_uniform(5,-1,1,(n,n))
def f():
    t=p.transpose(x); r=p.reshape(t,(n*n,1)); p.free(r)
for _ in range(3): f()
p.timer_begin()
for _ in range(20): f()
ms=p.timer_end()/20
print(f"transpose view materialised 8192^2: {ms*1e3:.1f} us {16*n*n/ms/1e9:.3f} TB/s")
X=np.random.default_rng(1).standard_normal((300,517))
h=p.upload(X); t=p.transpose(h)
assert np.array_equal(p.download_matrix(t), X.T)
print("ok")
