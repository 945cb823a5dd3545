#!/usr/bin/env python3
"""Generates tests/golden/channel_const.npz from the REAL reference (run in the build container, where /root/reference
and oracle/_ref exist; the GPU box only sees the fixture).

  * raw constellation tables of the reference's include/channel.h (qam64, qam256, qam256_4d_16qam_r), dumped by a
    throw-away C program that #includes the header where it lies, and the (radius, numerator, denominator) triples of
    the 64-APSK table that channel.c:132-199 writes as radius*cos(PI*num/den) literals (parsed from the text);
  * outputs of the three channel front ends the reference compiles -- ModelChannel_AWGN_64, ModelChannel (rotated QAM,
    fading per axis, erasures: the shipped switches of channel.c:20-26) and ModelChannel_AWGN_256QAM_4D -- for seeded
    codewords: decoder.intrinsic_LLR / intrinsic_GF (head rows in full, CRC-32 of the whole blocks);
  * the same for ModelChannel's GF(16) branch and for ModelChannel compiled with the switches the reference ships disabled
    (APSK + rayleigh_fading, plain QAM without fading, rotated QAM + SSD without erasure): libraries built by oracle/Makefile
    (CHAN_VARIANTS) from channel.c streamed through sed into the compiler -- plus the literals those variants need (the 16-QAM
    table of include/channel.h, ModelChannel's own 64-APSK triples, the 8 ring radii of its GF(256) APSK).
"""
import ctypes as C
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

REF = "/root/reference"
MAT = os.path.join(HERE, "matrices")


def dump_tables():
    src = r'''
#include <stdio.h>
#include "channel.h"
int main(void) {
    int i, d;
    for (i = 0; i < 16; i++) printf("qam16 %.9g %.9g\n", table_16QAM[i][0], table_16QAM[i][1]);
    for (i = 0; i < 64; i++) printf("qam64 %.9g %.9g\n", table_64QAM[i][0], table_64QAM[i][1]);
    for (i = 0; i < 256; i++) printf("qam256 %.9g %.9g\n", table_256QAM[i][0], table_256QAM[i][1]);
    for (i = 0; i < 256; i++) { printf("qam4d"); for (d = 0; d < 4; d++) printf(" %.9g", table_256QAM_4D_16QAM_R[i][d]); printf("\n"); }
    return 0;
}
'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "dump.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "dump")
        subprocess.check_call(["gcc", "-w", "-I" + os.path.join(REF, "include"), "-o", exe, c, "-lm"])
        out = subprocess.check_output([exe]).decode().split("\n")
    t = {"qam16": [], "qam64": [], "qam256": [], "qam4d": []}
    for l in out:
        if l.strip():
            k, *v = l.split()
            t[k].append([float(x) for x in v])
    return (np.array(t["qam16"], np.float32), np.array(t["qam64"], np.float32), np.array(t["qam256"], np.float32), np.array(t["qam4d"], np.float32))


def parse_apsk256_radii():
    """channel.c:365: the 8 ring radii of ModelChannel's GF(256) APSK variant."""
    txt = open(os.path.join(REF, "channel.c")).read()
    m = re.search(r"float constellation_radius\[8\]=\s*\{([^}]*)\}", txt)
    r = np.array([float(x) for x in m.group(1).split(",")], np.float32)
    assert r.shape == (8,)
    return r


def parse_apsk64(decl="float table_64APSK[64][2]="):
    txt = open(os.path.join(REF, "channel.c")).read()
    body = txt[txt.index(decl):]
    body = body[:body.index("};")]
    rad, num, den = [], [], []
    for m in re.finditer(r"\{\s*(?:([\d.]+)\s*\*\s*)?cos\(PI\*(\d+)/(\d+)\s*\)", body):
        rad.append(float(m.group(1) or 1.0)); num.append(int(m.group(2))); den.append(int(m.group(3)))
    assert len(rad) == 64
    return np.array(rad), np.array(num, np.int32), np.array(den, np.int32)


def run_reference(fn, matrix, EbN, seed, variant=""):
    """decoder.intrinsic_LLR / intrinsic_GF of one frame from the real function, drand48 reset to X0 = 0.
    variant: one of oracle/Makefile's CHAN_VARIANTS (ModelChannel compiled with other switches) or "" = as shipped."""
    ref = orc.ref_lib("ubs_" + variant if variant else "ubs")
    code, table, dec = orc.RefCode(), orc.RefTable(), orc.RefDecoder()
    ref.LoadCode(os.fsencode(os.path.join(MAT, matrix)), C.byref(code))
    ref.LoadTables(C.byref(table), code.GF, code.logGF)
    dec.nbMax = 8
    ref.AllocateDecoder(C.byref(code), C.byref(dec))
    o = orc.Oracle(os.path.join(MAT, matrix))
    cw = np.random.RandomState(seed).randint(0, code.GF, code.N).astype(np.int32)
    PI = C.POINTER(C.c_int)
    rows = [(C.c_int * code.logGF)(*[(int(o.idx2bin[s]) >> l) & 1 for l in range(code.logGF)]) for s in cw]
    NBIN = (PI * code.N)(*[C.cast(r, PI) for r in rows])
    C.CDLL("libc.so.6").seed48((C.c_ushort * 3)(0, 0, 0))
    idum = C.c_int(-1)
    getattr(ref, fn)(C.byref(code), C.byref(dec), NBIN, C.c_float(EbN), C.byref(idum))
    sl = np.ctypeslib.as_array(dec.intrinsic_LLR[0], (code.N, code.GF)).copy()
    sg = np.ctypeslib.as_array(dec.intrinsic_GF[0], (code.N, code.GF)).copy()
    return cw, sl, sg


CASES = [  # tag, function, matrix, EbN, seed, variant
    ("apsk64_n96", "ModelChannel_AWGN_64", "N96_K48_GF64", 12.0, 1, ""),
    ("qamr64_n96", "ModelChannel", "N96_K48_GF64", 14.0, 2, ""),
    ("qamr256_big", "ModelChannel", "Ahmed_64800_R12_GF256", 20.0, 3, ""),
    ("qam4d_big", "ModelChannel_AWGN_256QAM_4D", "Ahmed_64800_R12_GF256", 12.0, 4, ""),
    ("qamr16_big", "ModelChannel", "Ahmed_64800_R34_GF16", 9.0, 5, ""),                  # the GF(16) branch, channel.c:520-557
    # ModelChannel's other compile-time variants (oracle/Makefile: CHAN_VARIANTS)
    ("apskrf256_big", "ModelChannel", "Ahmed_64800_R12_GF256", 22.0, 6, "apsk_rf"),
    ("apskrf64_n96", "ModelChannel", "N96_K48_GF64", 15.0, 7, "apsk_rf"),
    ("qamawgn64_n96", "ModelChannel", "N96_K48_GF64", 13.0, 8, "qam_awgn"),
    ("qamawgn16_big", "ModelChannel", "Ahmed_64800_R34_GF16", 8.0, 9, "qam_awgn"),
    ("qamrssdne256_big", "ModelChannel", "Ahmed_64800_R12_GF256", 19.0, 10, "qamr_ssd_ne"),
]

if __name__ == "__main__":
    q16, q64, q256, q4d = dump_tables()
    rad, num, den = parse_apsk64()
    rad2, num2, den2 = parse_apsk64("float table_APSK[64][2]=")          # ModelChannel's own 64-APSK (channel.c:423)
    out = {"qam16": q16, "qam64": q64, "qam256": q256, "qam256_4d_16qam_r": q4d, "apsk64_radius": rad, "apsk64_num": num, "apsk64_den": den,
           "apsk64mc_radius": rad2, "apsk64mc_num": num2, "apsk64mc_den": den2, "apsk256_radius": parse_apsk256_radii()}
    for tag, fn, matrix, ebn, seed, variant in CASES:
        cw, sl, sg = run_reference(fn, matrix, ebn, seed, variant)
        head = 16 if sl.shape[0] > 64 else sl.shape[0]
        out[tag + "_cw"] = cw.astype(np.int16)
        out[tag + "_ebn"] = np.float32(ebn)
        out[tag + "_llr_head"] = sl[:head]
        out[tag + "_gf_head"] = sg[:head].astype(np.int16)
        out[tag + "_crc"] = np.array([orc.crc32(sl), orc.crc32(sg.astype(np.int32))], np.uint32)
        print(tag, sl.shape, "crc", out[tag + "_crc"])
    # the older generation's own channel (EMS_HS_L-BubbleCheck_UBS_decoder_v2.c:1167) on the all-zero codeword, srand48(5),
    # through the hook of oracle/ref_v2_shim.c: decoder.likelihood / likelihood_index
    import orc_v2
    r = orc_v2.RefV2(os.path.join(MAT, "Ahmed_64800_R12_GF256"), 20)
    like, idx = r.channel_zero_codeword(11.0, 5)
    r.close()
    out["v2chan_ebn"] = np.float32(11.0)
    out["v2chan_seed"] = np.int32(5)
    out["v2chan_like_head"] = like[:16]
    out["v2chan_idx_head"] = idx[:16].astype(np.int16)
    out["v2chan_crc"] = np.array([orc.crc32(like), orc.crc32(idx.astype(np.int32))], np.uint32)
    print("v2chan", like.shape, out["v2chan_crc"])
    np.savez_compressed(os.path.join(HERE, "channel_const.npz"), **out)
