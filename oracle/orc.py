"""ctypes bindings for the CPU oracle (oracle/libems_oracle.so) and, when present, for the real
reference compiled into oracle/_ref/ by oracle/Makefile.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libems_oracle.so")
REF_DIR = os.path.join(HERE, "_ref")


def build(force=False):
    """Compile the restatement (and the real reference when /root/reference exists)."""
    src = [os.path.join(HERE, f) for f in ("ems_oracle.c", "ems_oracle.h", "ems_v2_oracle.c", "ems_source_oracle.c", "ems_channel_oracle.c")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "libems_oracle.so"])
    if os.path.exists("/root/reference/NB_LDPC.c"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class GF(C.Structure):
    _fields_ = [("q", C.c_int), ("logq", C.c_int), ("idx2bin", C.c_uint8 * 256), ("bin2idx", C.c_uint8 * 256)]


class Code(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("q", C.c_int), ("logq", C.c_int),
                ("E", C.c_int), ("dc_max", C.c_int), ("rate", C.c_float),
                ("row_deg", C.POINTER(C.c_int)), ("col_deg", C.POINTER(C.c_int)),
                ("row_ptr", C.POINTER(C.c_int)), ("col", C.POINTER(C.c_int)), ("coef", C.POINTER(C.c_int))]


class Rng(C.Structure):
    _fields_ = [("x", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_load_code.restype = C.POINTER(Code)
        L.orc_load_code.argtypes = [C.c_char_p, C.c_int]
        L.orc_encoder_create.restype = C.c_void_p
        L.orc_encoder_create.argtypes = [C.POINTER(Code), C.POINTER(GF)]
        L.orc_encoder_free.argtypes = [C.c_void_p]
        L.orc_drand48.restype = C.c_double
        L.orc_sigma.restype = C.c_float
        L.orc_sigma.argtypes = [C.c_float, C.c_float]
        L.orc_crc32.restype = C.c_uint32
        L.orc_crc32.argtypes = [C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


class Oracle:
    """One code + field + (lazily) encoder, with numpy-friendly wrappers."""

    def __init__(self, path, fmt=0):
        L = lib()
        self.L = L
        self.code_p = L.orc_load_code(os.fsencode(path), fmt)
        if not self.code_p:
            raise IOError("oracle: cannot load code " + path)
        c = self.code_p.contents
        self.N, self.M, self.K, self.q, self.logq, self.E, self.dc = c.N, c.M, c.K, c.q, c.logq, c.E, c.dc_max
        self.rate = c.rate
        self.gf = GF()
        assert L.orc_gf_init(C.byref(self.gf), self.q) == 0
        self.row_deg = np.ctypeslib.as_array(c.row_deg, (self.M,)).copy()
        self.col_deg = np.ctypeslib.as_array(c.col_deg, (self.N,)).copy()
        self.row_ptr = np.ctypeslib.as_array(c.row_ptr, (self.M + 1,)).copy()
        self.col = np.ctypeslib.as_array(c.col, (self.E,)).copy()
        self.coef = np.ctypeslib.as_array(c.coef, (self.E,)).copy()
        self.idx2bin = np.frombuffer(bytes(self.gf.idx2bin), dtype=np.uint8)[: self.q].copy()
        self.bin2idx = np.frombuffer(bytes(self.gf.bin2idx), dtype=np.uint8)[: self.q].copy()
        self._enc = None

    @property
    def enc(self):
        if self._enc is None:
            self._enc = self.L.orc_encoder_create(self.code_p, C.byref(self.gf))
            if not self._enc:
                raise RuntimeError("oracle: matrix is not full rank")
        return self._enc

    # ---- field ----
    def add(self, a, b): return self.L.orc_gf_add(C.byref(self.gf), int(a), int(b))
    def mul(self, a, b): return self.L.orc_gf_mul(C.byref(self.gf), int(a), int(b))
    def div(self, a, b): return self.L.orc_gf_div(C.byref(self.gf), int(a), int(b))

    # ---- channel ----
    def sigma(self, EbN): return float(self.L.orc_sigma(C.c_float(self.rate), C.c_float(EbN)))

    def draw_frame(self, rng, sigma):
        cw = np.zeros(self.N, np.int32)
        noisy = np.zeros((self.N, self.logq), np.float32)
        self.L.orc_draw_frame(self.code_p, C.byref(self.gf), C.c_void_p(self.enc), C.byref(rng),
                              C.c_float(sigma), _p(cw, C.c_int), _p(noisy, C.c_float))
        return cw, noisy

    def llr(self, noisy, sigma):
        noisy = np.ascontiguousarray(noisy, np.float32)
        out = np.zeros((self.N, self.q), np.float32)
        self.L.orc_llr(self.code_p, C.byref(self.gf), _p(noisy, C.c_float), C.c_float(sigma), _p(out, C.c_float))
        return out

    def sort_llr(self, llr):
        llr = np.ascontiguousarray(llr, np.float32)
        n = llr.shape[0]
        sl = np.zeros_like(llr)
        sg = np.zeros(llr.shape, np.int32)
        self.L.orc_sort_llr(_p(llr, C.c_float), n, self.q, _p(sl, C.c_float), _p(sg, C.c_int))
        return sl, sg

    # ---- check node ----
    def elementary_step(self, in1, in2, idx1, idx2, nm, nbOper):
        in1 = np.ascontiguousarray(in1, np.float32); in2 = np.ascontiguousarray(in2, np.float32)
        idx1 = np.ascontiguousarray(idx1, np.int32); idx2 = np.ascontiguousarray(idx2, np.int32)
        out = np.zeros(nm, np.float32); io = np.zeros(nm, np.int32)
        r = self.L.orc_elementary_step(_p(in1, C.c_float), _p(in2, C.c_float), _p(idx1, C.c_int), _p(idx2, C.c_int),
                                       _p(out, C.c_float), _p(io, C.c_int), C.byref(self.gf), nm, nbOper)
        assert r == 0
        return out, io

    def check_node(self, coef, vl, vg, nbOper, offset):
        coef = np.ascontiguousarray(coef, np.int32)
        vl = np.ascontiguousarray(vl, np.float32); vg = np.ascontiguousarray(vg, np.int32)
        dc, nm = vl.shape
        out = np.zeros((dc, self.q), np.float32)
        self.L.orc_check_node(dc, nm, C.byref(self.gf), _p(coef, C.c_int), _p(vl, C.c_float), _p(vg, C.c_int),
                              nbOper, C.c_float(offset), _p(out, C.c_float))
        return out

    # ---- decode ----
    def decision(self, app):
        app = np.ascontiguousarray(app, np.float32)
        d = np.zeros(app.shape[0], np.int32)
        self.L.orc_decision(_p(app, C.c_float), app.shape[0], self.q, _p(d, C.c_int))
        return d

    def syndrome(self, decide):
        decide = np.ascontiguousarray(decide, np.int32)
        return self.L.orc_syndrome(self.code_p, C.byref(self.gf), _p(decide, C.c_int))

    def decode(self, llr, nm, nbOper, offset, NbIterMax, order=None, early_exit=True, want_state=False,
               want_sweeps=False):
        llr = np.ascontiguousarray(llr, np.float32)
        decide = np.zeros(self.N, np.int32)
        synd = C.c_int(0)
        app = np.zeros((self.N, self.q), np.float32) if want_state else None
        ctov = np.zeros((self.E, self.q), np.float32) if want_state else None
        ns = max(NbIterMax - 1, 1)
        sd = np.zeros((ns, self.N), np.uint8) if want_sweeps else None
        sc = np.zeros(ns, np.uint32) if want_sweeps else None
        ss = np.zeros(ns, np.int32) if want_sweeps else None
        o = np.ascontiguousarray(order, np.int32) if order is not None else None
        it = self.L.orc_decode_frame(self.code_p, C.byref(self.gf), _p(llr, C.c_float), nm, nbOper,
                                     C.c_float(offset), NbIterMax, _p(o, C.c_int), int(bool(early_exit)),
                                     _p(decide, C.c_int), C.byref(synd), _p(app, C.c_float), _p(ctov, C.c_float),
                                     _p(sd, C.c_uint8), _p(sc, C.c_uint32), _p(ss, C.c_int))
        res = {"decide": decide, "synd": synd.value, "iters": it}
        if want_state:
            res["app"] = app; res["ctov"] = ctov
        if want_sweeps:
            n = min(it, ns)
            res["sweep_decide"] = sd[:n]; res["sweep_crc"] = sc[:n]; res["sweep_synd"] = ss[:n]
        return res

    def bit_errors(self, decide, cw):
        decide = np.ascontiguousarray(decide, np.int32); cw = np.ascontiguousarray(cw, np.int32)
        return self.L.orc_bit_errors(self.code_p, C.byref(self.gf), _p(decide, C.c_int), _p(cw, C.c_int))

    def sim(self, rng, EbN, nm, nbOper, offset, NbIterMax, nframes, early_exit=True):
        rec = np.zeros((nframes, 3), np.int32)
        self.L.orc_sim_frames(self.code_p, C.byref(self.gf), C.c_void_p(self.enc), C.byref(rng), C.c_float(EbN),
                              nm, nbOper, C.c_float(offset), NbIterMax, int(bool(early_exit)), nframes,
                              _p(rec, C.c_int))
        return rec


def crc32(a):
    a = np.ascontiguousarray(a)
    return int(lib().orc_crc32(a.ctypes.data_as(C.c_void_p), a.nbytes))


def trace_from_records(rec, K, logq):
    """Cumulative (erroneous frames, bit errors, avr_it) as the reference prints them (NB_LDPC.c:491-499)."""
    out, ef, be, si = [], 0, 0, 0
    for nb, (e, _s, it) in enumerate(rec, 1):
        be += int(e); si += int(it); ef += int(e != 0)
        out.append((ef, be, "%.2f" % (si / nb)))
    return out


# ---------------------------------------------------------------------------------------------------
# the REAL reference (oracle/_ref, built from /root/reference by oracle/Makefile)
# ---------------------------------------------------------------------------------------------------
class RefCode(C.Structure):
    _fields_ = [("N", C.c_int), ("M", C.c_int), ("K", C.c_int), ("GF", C.c_int), ("logGF", C.c_int),
                ("mat", C.POINTER(C.POINTER(C.c_int))), ("matValue", C.POINTER(C.POINTER(C.c_int))),
                ("rowDegree", C.POINTER(C.c_int)), ("columnDegree", C.POINTER(C.c_int)),
                ("nbBranch", C.c_int), ("rate", C.c_float),
                ("matUT", C.POINTER(C.POINTER(C.c_int))), ("Perm", C.POINTER(C.c_int))]


class RefTable(C.Structure):
    _fields_ = [("BINGF", C.POINTER(C.POINTER(C.c_int))), ("ADDGF", C.POINTER(C.POINTER(C.c_int))),
                ("MULGF", C.POINTER(C.POINTER(C.c_int))), ("DIVGF", C.POINTER(C.POINTER(C.c_int)))]


class RefDecoder(C.Structure):
    _fields_ = [("N", C.c_int), ("nbMax", C.c_int), ("nbBranch", C.c_int),
                ("CtoV", C.POINTER(C.POINTER(C.c_float))),
                ("intrinsic_LLR", C.POINTER(C.POINTER(C.c_float))),
                ("intrinsic_GF", C.POINTER(C.POINTER(C.c_int))),
                ("APP", C.POINTER(C.POINTER(C.c_float))),
                ("M_VtoC_LLR", C.POINTER(C.POINTER(C.c_float))), ("M_VtoC_GF", C.POINTER(C.POINTER(C.c_int))),
                ("M_CtoV_LLR", C.POINTER(C.POINTER(C.c_float))), ("M_CtoV_GF", C.POINTER(C.POINTER(C.c_int)))]


def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "libref_ubs.so"))


def ref_lib(fmt="ubs"):
    L = C.CDLL(os.path.join(REF_DIR, "libref_%s.so" % fmt))
    L.My_drand48.restype = C.c_float
    return L


def ref_binary(fmt="ubs"):
    return os.path.join(REF_DIR, "essai_%s" % fmt)


def run_reference_binary(fmt, nframes, NbIterMax, matrix, EbN, nm, offset, nbOper, timeout=3000):
    """Run the reference's own executable; return the list of (errFrames, bitErr, avr_it) it printed."""
    import re
    run = os.path.join(REF_DIR, "run")
    os.makedirs(os.path.join(run, "data"), exist_ok=True)
    out = subprocess.run([ref_binary(fmt), str(nframes), str(NbIterMax), matrix, str(EbN), str(nm), str(offset),
                          str(nbOper)], input=b"\n", cwd=run, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         timeout=timeout).stdout.decode("latin-1")
    tr = []
    for m in re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", out):
        tr.append((int(m.group(1)), int(m.group(3)), m.group(4)))
    return tr
