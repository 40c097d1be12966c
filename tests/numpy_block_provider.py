"""Test double for the block-level provider ops (`blk_*`): plain numpy on `.arr`. Used only by the
world_size-2 gloo test of the distributed solver's host logic."""
import numpy as np


class H:
    def __init__(self, arr):
        self.arr = np.array(arr, dtype=np.float64, order="C")  # stored TRANSPOSED: arr[c, r] == element (r, c)
        self.shape = (self.arr.shape[1], self.arr.shape[0])


class NumpyBlockProvider:
    """Column-major semantics are emulated by storing the transpose row-major, which is what the
    GPU memory looks like to torch (so broadcasts of `.arr` match the device path)."""

    def upload(self, a, shape=None):
        a = np.asarray(a, dtype=np.float64)
        if shape is not None:
            a = a.reshape(shape, order="F")
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        return H(a.T)

    def zeros(self, shape):
        return H(np.zeros((shape[1], shape[0])))

    def fill(self, shape, value):
        return H(np.full((shape[1], shape[0]), float(value)))

    def download(self, h):
        return h.arr.reshape(-1).copy()  # transposed row-major == column-major flat

    def free(self, h):
        pass

    def _v(self, v):
        h, r0, c0, rows, cols = v
        return h.arr[c0:c0 + cols, r0:r0 + rows]  # transposed view

    def blk_copy(self, v):
        return H(self._v(v).copy())

    def blk_assign(self, v, src):
        self._v(v)[...] = src.arr

    def blk_gemm(self, alpha, a, b, beta, c):
        A, B, Cv = self._v(a).T, self._v(b).T, self._v(c)
        Cv[...] = (alpha * (A @ B) + beta * Cv.T).T

    def blk_trsm(self, upper, t, b):
        import scipy.linalg as sl

        T, Bv = self._v(t).T, self._v(b)
        if upper in (2, "right"):  # B <- B U^-1
            Bv[...] = sl.solve_triangular(T, Bv, trans="T", lower=False, check_finite=False)  # Bv is B' : U' X' = B'
            return
        Bv[...] = sl.solve_triangular(T, Bv.T, lower=not upper, unit_diagonal=not upper, check_finite=False).T

    def blk_absmax(self, v):
        a = self._v(v)
        return float(np.max(np.abs(a))) if a.size else 0.0

    def mldivide(self, a, b):
        return H(np.linalg.solve(a.arr.T, b.arr.T).T)

    def blk_lu(self, a):
        Av = self._v(a)  # transposed
        M = Av.T.copy()
        rows, cols = M.shape
        ipiv, info = [], 0
        for k in range(min(rows, cols)):
            col = np.abs(M[k:, k])
            p = k + int(np.argmax(col)) if col.max() > 0 else k  # first max
            ipiv.append(p)
            if p != k:
                M[[k, p], :] = M[[p, k], :]
            if abs(M[k, k]) <= 1e-12:
                info += 1
                M[k + 1:, k] = 0.0
                continue
            M[k + 1:, k] /= M[k, k]
            M[k + 1:, k + 1:] -= np.outer(M[k + 1:, k], M[k, k + 1:])
        Av[...] = M.T
        return H(np.array(ipiv, dtype=np.float64).reshape(1, -1)), info

    def blk_swap_rows(self, a, ipiv):
        Av = self._v(a)
        for k, p in enumerate(ipiv.arr.reshape(-1).astype(int)):
            if p != k:
                Av[:, [k, p]] = Av[:, [p, k]]
