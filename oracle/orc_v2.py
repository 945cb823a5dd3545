"""oracle/orc_v2.py -- TEST INFRASTRUCTURE ONLY.

ctypes front end of oracle/ems_v2_oracle.c (this repo's restatement of the v2 decode core, SURVEY.md 8(a)
row a11) and, when oracle/_ref/libref_v2.so has been built from /root/reference (oracle/Makefile `ref`),
of the real v2 functions through oracle/ref_v2_shim.c.  Only tests/, smoke() and bench.py's cpu_baseline
leg may import this module.
"""
import ctypes as C
import os

import numpy as np

import orc

HERE = os.path.dirname(os.path.abspath(__file__))
REF_V2 = os.path.join(HERE, "_ref", "libref_v2.so")
PF, PI = C.POINTER(C.c_float), C.POINTER(C.c_int)


def _f(a):
    return a.ctypes.data_as(PF)


def _i(a):
    return a.ctypes.data_as(PI)


def gf_tables(o):
    """(add, mul, div) q x q int32 index tables of the field of orc.Oracle `o` (div by zero = -1 as in the
    reference's Table_Div_GF)."""
    q = o.q
    i2b, b2i = o.idx2bin.astype(np.int64), o.bin2idx.astype(np.int64)
    a = np.arange(q)
    add = b2i[i2b[a][:, None] ^ i2b[a][None, :]].astype(np.int32)
    mul = np.zeros((q, q), np.int32)
    div = np.full((q, q), -1, np.int32)
    ea, eb = np.meshgrid(a[1:] - 1, a[1:] - 1, indexing="ij")
    mul[1:, 1:] = (ea + eb) % (q - 1) + 1
    div[1:, 1:] = (ea - eb) % (q - 1) + 1
    div[0, 1:] = 0
    return np.ascontiguousarray(add), np.ascontiguousarray(mul), np.ascontiguousarray(div)


def synthetic_likelihood(N, q, logq, idx2bin, sigma, seed):
    """Max-is-best log-likelihoods of an all-zero BPSK codeword over AWGN, per node the q values sorted in
    decreasing order with their symbols (the layout of decoder.likelihood / likelihood_index, v2.c:1483-1497:
    selection by repeated first strict maximum, i.e. ties -> lowest symbol first)."""
    rs = np.random.RandomState(seed)
    y = (1.0 + sigma * rs.randn(N, logq)).astype(np.float32)
    bits = ((idx2bin.astype(np.int64)[:, None] >> np.arange(logq)[None, :]) & 1).astype(np.float32)     # [q][logq]
    tx = 1.0 - 2.0 * bits
    d = (y[:, None, :] - tx[None, :, :]) ** 2
    llr = (-d.sum(axis=2) / np.float32(2.0 * sigma * sigma)).astype(np.float32)                         # [N][q]
    order = np.argsort(-llr, axis=1, kind="stable")
    like = np.take_along_axis(llr, order, axis=1).astype(np.float32)
    return np.ascontiguousarray(like), np.ascontiguousarray(order.astype(np.int32))


class V2Oracle:
    """The restatement (libems_oracle.so) bound to one code."""

    def __init__(self, path):
        self.o = orc.Oracle(path)
        self.L = self.o.L
        o = self.o
        self.N, self.M, self.E, self.q, self.logq = o.N, o.M, o.E, o.q, o.logq
        self.add, self.mul, self.div = gf_tables(o)
        self.row_deg = o.row_deg.astype(np.int32)
        self.col = o.col.astype(np.int32)
        self.coef = o.coef.astype(np.int32)
        L = self.L
        L.v2o_elementary_step.restype = C.c_int
        L.v2o_decode.restype = C.c_int
        L.v2o_syndrome.restype = C.c_int
        L.v2o_graph.restype = C.c_int

    def elementary_step(self, in1, in2, i1, i2, nbOper):
        nm = len(in1)
        out, iout = np.zeros(nm, np.float32), np.zeros(nm, np.int32)
        in1, in2 = np.ascontiguousarray(in1, np.float32), np.ascontiguousarray(in2, np.float32)
        i1, i2 = np.ascontiguousarray(i1, np.int32), np.ascontiguousarray(i2, np.int32)
        s = self.L.v2o_elementary_step(_f(in1), _f(in2), _i(i1), _i(i2), _f(out), _i(iout), _i(self.add), self.q, nm, int(nbOper))
        return out, iout, s

    def check_node(self, coef, vc, ivc, nbOper):
        dc, nm = vc.shape
        vc, ivc = np.ascontiguousarray(vc, np.float32), np.ascontiguousarray(ivc, np.int32)
        coef = np.ascontiguousarray(coef, np.int32)
        cv, icv = np.zeros((dc, nm), np.float32), np.zeros((dc, nm), np.int32)
        self.L.v2o_check_node(dc, _i(coef), _f(vc), _i(ivc), _f(cv), _i(icv), _i(self.add), _i(self.mul), _i(self.div), self.q, nm,
                              int(nbOper))
        return cv, icv

    def data_pass(self, like, like_idx, cv, icv, offset):
        nm = len(cv)
        like, like_idx = np.ascontiguousarray(like, np.float32), np.ascontiguousarray(like_idx, np.int32)
        cv, icv = np.ascontiguousarray(cv, np.float32), np.ascontiguousarray(icv, np.int32)
        vc, ivc = np.zeros(nm, np.float32), np.zeros(nm, np.int32)
        self.L.v2o_data_pass(_f(like), _i(like_idx), _f(cv), _i(icv), C.c_float(offset), _f(vc), _i(ivc), self.q, nm)
        return vc, ivc

    def graph(self):
        col_deg = np.zeros(self.N, np.int32)
        ntob = np.zeros((self.N, 2), np.int32)
        inter = np.zeros(self.E, np.int32)
        assert self.L.v2o_graph(self.N, self.M, _i(self.row_deg), _i(self.col), _i(col_deg), _i(ntob), 2, _i(inter)) == 0
        return col_deg, ntob, inter

    def decode(self, like, like_idx, nm, nbOper, offset, NbIterMax, early_exit=True):
        like, like_idx = np.ascontiguousarray(like, np.float32), np.ascontiguousarray(like_idx, np.int32)
        decide = np.zeros(self.N, np.int32)
        synd = C.c_int(0)
        st = {k: np.zeros((self.E, nm), t) for k, t in (("vtoc", np.float32), ("vtoci", np.int32), ("ctov", np.float32), ("ctovi", np.int32))}
        app = np.zeros((self.N, self.q), np.float32)
        it = self.L.v2o_decode(self.N, self.M, self.q, nm, _i(self.row_deg), _i(self.col), _i(self.coef), _f(like), _i(like_idx),
                               int(NbIterMax), int(nbOper), C.c_float(offset), 1 if early_exit else 0, _i(self.add), _i(self.mul),
                               _i(self.div), _i(decide), C.byref(synd), _f(st["vtoc"]), _i(st["vtoci"]), _f(st["ctov"]),
                               _i(st["ctovi"]), _f(app))
        return dict(decide=decide, synd=synd.value, iters=it, app=app, **st)


class RefV2:
    """The real v2 functions (oracle/_ref/libref_v2.so); only where /root/reference was available at build time."""

    def __init__(self, path, nbMax, tables=None):
        self.L = C.CDLL(REF_V2)
        L = self.L
        L.refv2_open.restype = C.c_void_p
        L.refv2_decode.restype = C.c_int
        self.h = C.c_void_p(L.refv2_open(os.fsencode(path), int(nbMax)))
        info = np.zeros(6, np.int32)
        L.refv2_info(self.h, _i(info))
        self.N, self.M, self.E, self.q, self.logq, self.dc0 = [int(x) for x in info]
        self.nm = int(nbMax)
        if tables is not None:
            add, mul, div = tables
            L.refv2_set_tables(self.h, _i(add), _i(mul), _i(div))

    def tables(self):
        q = self.q
        add, mul, div = (np.zeros((q, q), np.int32) for _ in range(3))
        binm = np.zeros((q, self.logq), np.int32)
        self.L.refv2_get_tables(self.h, _i(add), _i(mul), _i(div), _i(binm))
        return add, mul, div, binm

    def graph(self):
        inter, ntob = np.zeros(self.E, np.int32), np.zeros((self.N, 2), np.int32)
        self.L.refv2_get_graph(self.h, _i(inter), _i(ntob))
        return ntob, inter

    def elementary_step(self, in1, in2, i1, i2, nbOper, offset=0.3):
        nm = len(in1)
        a, b = np.array(in1, np.float32), np.array(in2, np.float32)
        ia, ib = np.array(i1, np.int32), np.array(i2, np.int32)
        out, iout = np.zeros(nm, np.float32), np.zeros(nm, np.int32)
        self.L.refv2_elementary_step(self.h, _f(a), _f(b), _i(ia), _i(ib), _f(out), _i(iout), nm, int(nbOper), C.c_float(offset))
        return out, iout

    def check_node(self, node, vc, ivc, nbOper, offset=0.3):
        vc, ivc = np.ascontiguousarray(vc, np.float32), np.ascontiguousarray(ivc, np.int32)
        cv, icv = np.zeros_like(vc), np.zeros_like(ivc)
        self.L.refv2_check_node(self.h, int(node), _f(vc), _i(ivc), _f(cv), _i(icv), C.c_float(offset), int(nbOper))
        return cv, icv

    def data_pass(self, like, like_idx, cv, icv, offset):
        like, like_idx = np.ascontiguousarray(like, np.float32), np.ascontiguousarray(like_idx, np.int32)
        cv, icv = np.ascontiguousarray(cv, np.float32), np.ascontiguousarray(icv, np.int32)
        vc, ivc = np.zeros(self.nm, np.float32), np.zeros(self.nm, np.int32)
        self.L.refv2_data_pass(self.h, _f(like), _i(like_idx), _f(cv), _i(icv), C.c_float(offset), _f(vc), _i(ivc))
        return vc, ivc

    def channel_zero_codeword(self, EbN, seed):
        like, idx = np.zeros((self.N, self.q), np.float32), np.zeros((self.N, self.q), np.int32)
        self.L.refv2_channel_zero_codeword(self.h, C.c_float(EbN), C.c_long(seed), _f(like), _i(idx))
        return like, idx

    def decode(self, like, like_idx, nbOper, offset, NbIterMax):
        like, like_idx = np.ascontiguousarray(like, np.float32), np.ascontiguousarray(like_idx, np.int32)
        self.L.refv2_set_likelihood(self.h, _f(like), _i(like_idx))
        decide, synd = np.zeros(self.N, np.int32), C.c_int(0)
        it = self.L.refv2_decode(self.h, int(NbIterMax), int(nbOper), C.c_float(offset), _i(decide), C.byref(synd))
        st = {k: np.zeros((self.E, self.nm), t) for k, t in (("vtoc", np.float32), ("vtoci", np.int32), ("ctov", np.float32), ("ctovi", np.int32))}
        app = np.zeros((self.N, self.q), np.float32)
        self.L.refv2_get_state(self.h, _f(st["vtoc"]), _i(st["vtoci"]), _f(st["ctov"]), _i(st["ctovi"]), _f(app))
        return dict(decide=decide, synd=synd.value, iters=it, app=app, **st)

    def close(self):
        if self.h:
            self.L.refv2_close(self.h)
            self.h = None
