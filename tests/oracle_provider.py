"""Test double: an object with HipProvider's method names whose arithmetic is the CPU oracle.
Used ONLY by the CPU tests of the sharding host logic (world_size-2 gloo); never by the product."""
import numpy as np


class Handle:
    def __init__(self, arr):
        self.arr = np.asarray(arr, dtype=np.float64)
        self.shape = self.arr.shape


class OracleProvider:
    def __init__(self, oracle):
        self.o = oracle
        self.state = oracle.rng_default_seed()

    def upload(self, a, shape=None):
        a = np.asarray(a, dtype=np.float64)
        return Handle(a if shape is None else a.reshape(shape, order="F"))

    def download(self, h):
        return h.arr.reshape(-1, order="F")

    def free(self, h):
        pass

    def fill(self, shape, v):
        return Handle(np.full(shape, float(v)))

    def matmul(self, a, b):
        return Handle(self.o.matmul(a.arr, b.arr))

    def set_rng_state(self, s):
        self.state = int(s)

    def get_rng_state(self):
        return self.state

    def random_normal(self, shape):
        n = int(np.prod(shape))
        z, self.state = self.o.rng_normal(self.state, n)
        return Handle(z.reshape(shape, order="F"))

    def stochastic_evolution(self, h, drift, scale, steps, draws_per_step=0):
        # sharded form of include/rmhip.h: step t draws from state + t * draws_per_step
        x = h.arr.reshape(-1, order="F").copy()
        per = draws_per_step or 2 * ((x.size + 1) // 2)
        base = self.state
        for t in range(steps):
            z, _ = self.o.rng_normal(self.o.rng_advance(base, t * per), x.size)
            x = self.o.binary("mul", x.reshape(-1, 1), self.o.unary("exp", self.o.binary("add", np.array([[float(drift)]]),
                              self.o.binary("mul", np.array([[float(scale)]]), z.reshape(-1, 1))))).reshape(-1)
        self.state = self.o.rng_advance(base, steps * per)
        return Handle(x.reshape(h.arr.shape, order="F"))

    def scalar_mul(self, a, s):
        return Handle(self.o.binary("mul", a.arr, np.array([[float(s)]])))

    def scalar_add(self, a, s):
        return Handle(self.o.binary("add", a.arr, np.array([[float(s)]])))

    def scalar_sub(self, a, s):
        return Handle(self.o.binary("sub", a.arr, np.array([[float(s)]])))

    def scalar_max(self, a, s):
        return Handle(self.o.binary("max", a.arr, np.array([[float(s)]])))

    def unary_exp(self, a):
        return Handle(self.o.unary("exp", a.arr))

    def elem_mul(self, a, b):
        return Handle(self.o.binary("mul", a.arr, b.arr))

    def reduce_sum(self, a):
        return Handle(self.o.reduce_sum(a.arr, "all"))
