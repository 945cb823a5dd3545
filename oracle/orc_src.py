"""oracle/orc_src.py -- TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/ems_source_oracle.c: the CPU statement of the device-side frame source
(SURVEY.md 8(f) row f2).  Only tests/ and smoke() may import this module."""
import ctypes as C

import numpy as np

import orc
import orc_v2

PI, PF = C.POINTER(C.c_int), C.POINTER(C.c_float)


def philox(ctr, key):
    c, k, o = (C.c_uint32 * 4)(*ctr), (C.c_uint32 * 2)(*key), (C.c_uint32 * 4)()
    orc.lib().srco_philox(c, k, o)
    return [int(x) for x in o]


class SourceOracle:
    """Frames of the counter-based source for one code; matUT [M][N] / perm [N] are GaussianElimination's outputs."""

    def __init__(self, path, matUT, perm):
        self.o = orc.Oracle(path)
        o = self.o
        self.L = o.L
        self.N, self.M, self.K, self.q, self.logq = o.N, o.M, o.N - o.M, o.q, o.logq
        self.matUT = np.ascontiguousarray(matUT, np.int32).reshape(self.M, self.N)
        self.perm = np.ascontiguousarray(perm, np.int32)
        self.add, self.mul, self.div = orc_v2.gf_tables(o)
        self.i2b = np.ascontiguousarray(o.idx2bin, np.int32)
        self.b2i = np.ascontiguousarray(o.bin2idx, np.int32)

    def info(self, seed, frame):
        ks = np.zeros(self.K, np.int32)
        self.L.srco_info(C.c_uint64(seed), C.c_uint64(frame), self.K, self.logq, self.b2i.ctypes.data_as(PI), ks.ctypes.data_as(PI))
        return ks

    def encode(self, ksymb):
        ks = np.ascontiguousarray(ksymb, np.int32)
        cw, scratch = np.zeros(self.N, np.int32), np.zeros(self.N, np.int32)
        self.L.srco_encode(self.N, self.M, self.q, self.matUT.ctypes.data_as(PI), self.perm.ctypes.data_as(PI),
                           self.add.ctypes.data_as(PI), self.mul.ctypes.data_as(PI), self.div.ctypes.data_as(PI),
                           ks.ctypes.data_as(PI), scratch.ctypes.data_as(PI), cw.ctypes.data_as(PI))
        return cw

    def noise(self, seed, frame, cw, sigma):
        cw = np.ascontiguousarray(cw, np.int32)
        noisy = np.zeros((self.N, self.logq), np.float32)
        self.L.srco_noise(C.c_uint64(seed), C.c_uint64(frame), self.N, self.logq, self.i2b.ctypes.data_as(PI), cw.ctypes.data_as(PI),
                          C.c_float(sigma), noisy.ctypes.data_as(PF))
        return noisy

    def frame(self, seed, frame, sigma):
        cw = self.encode(self.info(seed, frame))
        return cw, self.noise(seed, frame, cw, sigma)
