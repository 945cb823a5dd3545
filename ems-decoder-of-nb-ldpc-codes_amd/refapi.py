"""ctypes view of the reference-shaped C boundary (include/nbldpc_ref_api.h): the struct layouts of
struct.h:48-108 and a small `RefSession` that drives LoadCode / LoadTables / AllocateDecoder /
GaussianElimination / RandomBinaryGenerator / Encoding / ModelChannel_AWGN_BPSK / Decoding exactly in
the order the reference's main() does (NB_LDPC.c:122-125, :255-261)."""
import ctypes as C
import os

import numpy as np

from . import lib

PI = C.POINTER(C.c_int)
PF = C.POINTER(C.c_float)


class code_t(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("GF", C.c_int), ("logGF", C.c_int),
                ("mat", C.POINTER(PI)), ("matValue", C.POINTER(PI)), ("rowDegree", PI), ("columnDegree", PI),
                ("nbBranch", C.c_int), ("rate", C.c_float), ("matUT", C.POINTER(PI)), ("Perm", PI)]


class table_t(C.Structure):
    _fields_ = [("BINGF", C.POINTER(PI)), ("ADDGF", C.POINTER(PI)), ("MULGF", C.POINTER(PI)), ("DIVGF", C.POINTER(PI))]


class decoder_t(C.Structure):
    _fields_ = [("N", C.c_int), ("nbMax", C.c_int), ("nbBranch", C.c_int), ("CtoV", C.POINTER(PF)),
                ("intrinsic_LLR", C.POINTER(PF)), ("intrinsic_GF", C.POINTER(PI)), ("APP", C.POINTER(PF)),
                ("M_VtoC_LLR", C.POINTER(PF)), ("M_VtoC_GF", C.POINTER(PI)), ("M_CtoV_LLR", C.POINTER(PF)),
                ("M_CtoV_GF", C.POINTER(PI))]


_libc = C.CDLL("libc.so.6")


def seed48(state48):
    """Position libc's drand48 stream (the reference never seeds it: X0 = 0, tools.c:71-73)."""
    x = int(state48) & 0xFFFFFFFFFFFF
    _libc.seed48((C.c_ushort * 3)(x & 0xFFFF, (x >> 16) & 0xFFFF, (x >> 32) & 0xFFFF))


def block(pp, rows, cols, dtype):
    """numpy view of a row-pointer-over-one-block 2-D array (X[0] is the block, init.c:310-415)."""
    ct = C.c_float if dtype == np.float32 else C.c_int
    return np.ctypeslib.as_array(C.cast(pp[0], C.POINTER(ct)), (rows, cols))


class RefSession:
    def __init__(self, matrix, nbMax, need_encoder=True):
        L = lib()
        L.My_drand48.restype = C.c_float
        L.ems_awgn_bpsk_noise.restype = C.c_float
        L.ems_awgn_bpsk_noise.argtypes = [C.POINTER(code_t), C.POINTER(PI), C.c_float, PI, C.c_void_p]
        L.Decoding.argtypes = [C.POINTER(code_t), C.POINTER(table_t), C.POINTER(decoder_t), C.c_int, C.c_int, C.c_float,
                               C.c_void_p, PI]
        self.L = L
        self.code, self.table, self.dec = code_t(), table_t(), decoder_t()
        L.LoadCode(os.fsencode(matrix), C.byref(self.code))
        L.LoadTables(C.byref(self.table), self.code.GF, self.code.logGF)
        self.dec.nbMax = nbMax
        L.AllocateDecoder(C.byref(self.code), C.byref(self.dec))
        if need_encoder:
            L.GaussianElimination(C.byref(self.code), C.byref(self.table))
        c = self.code
        self.N, self.M, self.K, self.GF, self.logGF = c.N, c.M, c.K, c.GF, c.logGF
        self._nbin_rows = [(C.c_int * c.logGF)() for _ in range(c.N)]
        self._kbin_rows = [(C.c_int * c.logGF)() for _ in range(c.K)]
        self.NBIN = (PI * c.N)(*[C.cast(r, PI) for r in self._nbin_rows])
        self.KBIN = (PI * c.K)(*[C.cast(r, PI) for r in self._kbin_rows])
        self.KSYMB = (C.c_int * c.K)()
        self.idum = C.c_int(-1)

    def draw_codeword(self):
        """RandomBinaryGenerator + Encoding (NB_LDPC.c:255-258); returns the codeword symbols."""
        c = self.code
        cw = np.zeros(c.N, np.int32)
        self.L.RandomBinaryGenerator(c.N, c.M, c.GF, c.logGF, self.KBIN, self.KSYMB, self.table.BINGF, C.byref(self.idum))
        self.L.Encoding(C.byref(c), C.byref(self.table), cw.ctypes.data_as(PI), self.NBIN, self.KSYMB)
        return cw

    def noise(self, EbN):
        """channel.c:51-62 only; returns (sigma, noisy[N][logGF])."""
        noisy = np.zeros((self.N, self.logGF), np.float32)
        sigma = self.L.ems_awgn_bpsk_noise(C.byref(self.code), self.NBIN, C.c_float(EbN), C.byref(self.idum),
                                           noisy.ctypes.data_as(C.c_void_p))
        return float(sigma), noisy

    def model_channel(self, EbN):
        """ModelChannel_AWGN_BPSK (channel.c:38): fills decoder.intrinsic_LLR / intrinsic_GF."""
        self.L.ModelChannel_AWGN_BPSK(C.byref(self.code), C.byref(self.dec), C.byref(self.table), self.NBIN, C.c_float(EbN),
                                      C.byref(self.idum))
        return (block(self.dec.intrinsic_LLR, self.N, self.GF, np.float32).copy(),
                block(self.dec.intrinsic_GF, self.N, self.GF, np.int32).copy())

    def model_channel_llr(self, obs, att, table, mode, sigma):
        """ModelChannelLLR (NEW: the LLR + sort part of the constellation front ends, channel.c:270-304 / :676-730 /
        :861-915): fills decoder.intrinsic_LLR / intrinsic_GF from one frame's observations."""
        obs = np.ascontiguousarray(obs, np.float32)
        table = np.ascontiguousarray(table, np.float32)
        att = None if att is None else np.ascontiguousarray(att, np.float32)
        fp = lambda a: a.ctypes.data_as(C.c_void_p) if a is not None else None
        self.L.ModelChannelLLR.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        self.L.ModelChannelLLR.restype = None
        self.L.ModelChannelLLR(C.byref(self.code), C.byref(self.dec), fp(obs), fp(att), fp(table), obs.shape[1], mode, C.c_float(sigma))
        return (block(self.dec.intrinsic_LLR, self.N, self.GF, np.float32).copy(),
                block(self.dec.intrinsic_GF, self.N, self.GF, np.int32).copy())

    def decoding(self, NbIterMax, NbOper, offset):
        """Decoding() (hoist of NB_LDPC.c:266-472); returns (iter+1, decide, synd)."""
        decide = np.zeros(self.N, np.int32)
        synd = C.c_int(0)
        it = self.L.Decoding(C.byref(self.code), C.byref(self.table), C.byref(self.dec), NbIterMax, NbOper, C.c_float(offset),
                             decide.ctypes.data_as(C.c_void_p), C.byref(synd))
        if it < 0:
            raise RuntimeError("Decoding failed: %d" % it)
        return it, decide, synd.value

    @property
    def APP(self):
        return block(self.dec.APP, self.N, self.GF, np.float32)

    @property
    def CtoV(self):
        return block(self.dec.CtoV, self.code.nbBranch, self.GF, np.float32)

    def close(self):
        if self.code.N:
            self.L.FreeDecoder(C.byref(self.dec))
            self.L.FreeTable(C.byref(self.table))
            self.L.FreeCode(C.byref(self.code))
