"""The drop-in boundary against the reference's OWN translation unit and headers.

CPU part (-m "not gpu"): `make -C oracle abi_check` compiles oracle/abi_layout_check.c (sizeof / offsetof / field types of code_t,
table_t, decoder_t in include/nbldpc_ref_api.h == the reference's include/struct.h:48-108) and oracle/abi_proto_check.c (every
prototype both sides declare, in one translation unit); and oracle/_ref/essai_gpu -- the reference's NB_LDPC.c with its inline
decode block replaced by Decoding(), compiled against the reference's headers -- links: every symbol it imports is exported by
libems_hip.so.  GPU part (-m gpu): that executable prints the traces the unmodified reference printed (tests/golden/kat_traces.json)."""
import os
import re
import subprocess

import pytest

from conftest import MAT, ROOT

REF_DIR = os.path.join(ROOT, "oracle", "_ref")
ESSAI_GPU = os.path.join(REF_DIR, "essai_gpu")
HAVE_REF_TREE = os.path.exists("/root/reference/include/struct.h")


@pytest.mark.skipif(not HAVE_REF_TREE, reason="needs the reference's headers under /root/reference (absent on the GPU box)")
def test_struct_layouts_and_prototypes_match_the_reference_headers():
    for obj in ("abi_layout_check.o", "abi_proto_check.o"):           # force a fresh compile: compiling IS the test
        p = os.path.join(REF_DIR, obj)
        if os.path.exists(p):
            os.remove(p)
    out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "abi_check"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert out.returncode == 0, out.stdout.decode()[-3000:]
    assert os.path.exists(os.path.join(REF_DIR, "abi_layout_check.o")) and os.path.exists(os.path.join(REF_DIR, "abi_proto_check.o"))


def _dynamic_symbols(path, undefined):
    out = subprocess.run(["nm", "-D", "--undefined-only" if undefined else "--defined-only", path], stdout=subprocess.PIPE, check=True).stdout.decode()
    return {ln.split()[-1].split("@")[0] for ln in out.splitlines() if ln.strip()}


def test_reference_main_links_against_the_library(pkg):
    if not os.path.exists(ESSAI_GPU):
        pytest.skip("oracle/_ref/essai_gpu is not built (make -C oracle ref, needs /root/reference)")
    wanted = {s for s in _dynamic_symbols(ESSAI_GPU, True) if not s.startswith("_") and s[0].isupper()}   # the reference's CamelCase API
    assert {"LoadCode", "LoadTables", "AllocateDecoder", "GaussianElimination", "RandomBinaryGenerator", "Encoding",
            "ModelChannel_AWGN_BPSK", "Decoding", "FreeCode", "FreeDecoder", "FreeTable"} <= wanted
    assert wanted <= _dynamic_symbols(pkg.LIB_PATH, False)


CHAN_SYMBOL = {"apsk": "ModelChannel_AWGN_64", "qam": "ModelChannel", "4d": "ModelChannel_AWGN_256QAM_4D"}


@pytest.mark.parametrize("chan", ["apsk", "qam", "4d"])
def test_reference_main_with_another_channel_links_against_the_library_alone(pkg, chan):
    """NB_LDPC.c with line 262 / 263 / 264 un-commented instead of the BPSK call (oracle/Makefile: essai_gpu_apsk / _qam / _4d): the library
    exports the reference's other channel front ends under their own names (channel.h:20-26), so nothing but libems_hip.so is linked."""
    exe = ESSAI_GPU + "_" + chan
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/essai_gpu_%s is not built (make -C oracle ref, needs /root/reference)" % chan)
    wanted = {s for s in _dynamic_symbols(exe, True) if not s.startswith("_") and s[0].isupper()}
    assert CHAN_SYMBOL[chan] in wanted and "ModelChannel_AWGN_BPSK" not in wanted and "Decoding" in wanted
    assert wanted <= _dynamic_symbols(pkg.LIB_PATH, False)
    assert "build_QAM_table" in _dynamic_symbols(pkg.LIB_PATH, False)                     # channel.h:26


def _run(exe, args, cwd):
    os.makedirs(os.path.join(cwd, "data"), exist_ok=True)
    out = subprocess.run([exe] + [str(a) for a in args], cwd=cwd, input=b"\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    return [[int(m.group(1)), int(m.group(3)), m.group(4)] for m in
            re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", out.stdout.decode("latin-1"))]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["K1", "K1_ubs", "K2", "K3", "K4", "K5", "I1", "N40", "N64", "D20b", "D10", "W2", "R1", "R3"])
def test_reference_main_on_the_gpu_library_prints_the_reference_traces(kat, tmp_path, name):
    """NB_LDPC.c's own main() -- argv parsing, LoadCode ... ModelChannel_AWGN_BPSK, BER/FER counting, progress line, 40-error stop --
    driving libems_hip.so through Decoding(): the trace of the unmodified reference executable, frame for frame."""
    if not os.path.exists(ESSAI_GPU):
        pytest.skip("oracle/_ref/essai_gpu is not built (make -C oracle ref, needs /root/reference)")
    k = kat[name]
    matrix = ("KN_" + k["matrix"]) if k.get("fmt") == "kn" and not k["matrix"].startswith("KN_") else k["matrix"]     # (the library's loader detects the layout itself)
    tr = _run(ESSAI_GPU, [k["frames"], k["NbIterMax"], os.path.join(MAT, matrix), k["EbN"], k["nm"], k["offset"], k["NbOper"]], str(tmp_path))
    assert tr == [list(t) for t in k["trace"]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["Q1", "Q2", "Q3", "Q4", "Q5"])
def test_reference_main_with_another_channel_prints_the_reference_traces(kat, tmp_path, name):
    """The same main() behind ModelChannel_AWGN_64 (64-APSK), ModelChannel (rotated QAM, fading and erasures per axis; GF(64) / GF(256) /
    GF(16)) and ModelChannel_AWGN_256QAM_4D, all three now exported by the library (host-side draws in the reference's drand48 order, LLR
    rows + sort on the GPU): the traces of the REAL executable with that call switched in (kat_traces.json Q1..Q5, from
    oracle/_ref/essai_ubs_apsk / _qam / _4d)."""
    k = kat[name]
    exe = ESSAI_GPU + "_" + k["chan"]
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/essai_gpu_%s is not built (make -C oracle ref, needs /root/reference)" % k["chan"])
    tr = _run(exe, [k["frames"], k["NbIterMax"], os.path.join(MAT, k["matrix"]), k["EbN"], k["nm"], k["offset"], k["NbOper"]], str(tmp_path))
    assert tr == [list(t) for t in k["trace"]]


@pytest.mark.gpu
@pytest.mark.parametrize("block", range(int(os.environ.get("EMS_FUZZ_BOUNDARY", "2"))))
def test_reference_main_on_the_gpu_library_on_random_codes(orc, tmp_path, block):
    """The same executable on the seeded random codes of tests/fuzz_codes.py (mixed check degrees 2..20, three fields, repeated variable
    nodes): the trace of the oracle's simulation, which tests/test_oracle_fuzz_vs_reference.py holds to the unmodified executable on the same
    codes.  A rank-deficient matrix stops both before the first frame (tools.c:181-185)."""
    if not os.path.exists(ESSAI_GPU):
        pytest.skip("oracle/_ref/essai_gpu is not built (make -C oracle ref, needs /root/reference)")
    from fuzz_codes import simulation_case
    ran = 0
    for seed in range(block * 10, block * 10 + 10):
        path = str(tmp_path / ("rnd_%d" % seed))
        frames, it, eb, nm, off, nop = simulation_case(seed, path)
        os.makedirs(os.path.join(str(tmp_path), "data"), exist_ok=True)
        out = subprocess.run([ESSAI_GPU] + [str(a) for a in (frames, it, path, eb, nm, off, nop)], cwd=str(tmp_path), input=b"\n",
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
        txt = out.stdout.decode("latin-1")
        tr = [[int(m.group(1)), int(m.group(3)), m.group(4)] for m in
              re.finditer(r"<\d+> FER= (\d+) / (\d+) = [\d.]+ BER= (\d+) / x = [\d.]+\s+avr_it=([\d.]+)", txt)]
        o = orc.Oracle(path)
        try:
            rec = o.sim(orc.Rng(0), eb, nm, nop, off, it, frames)
        except RuntimeError as e:
            assert "not full rank" in str(e) and "not full rank" in txt and not tr, (seed, str(e), txt[-300:])
            continue
        assert out.returncode == 0, (seed, out.stderr.decode()[-1000:])
        assert tr == [list(t) for t in orc.trace_from_records(rec, o.K, o.logq)], (seed, nm, nop, off, it, eb)
        ran += 1
    assert ran >= 7, ran
