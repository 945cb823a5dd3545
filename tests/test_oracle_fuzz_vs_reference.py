"""Live differential test of the oracle's whole simulation loop (frame source, encoder, channel, decode, statistics) against the REAL
reference executable (oracle/_ref/essai_*, built by `make -C oracle ref` from /root/reference where it lies) on seeded random codes the
reference does not ship: mixed check degrees 2..20, three fields, check nodes that list a variable node twice.  Skipped where the executable
has not been built (the GPU box: /root/reference does not travel)."""
import os

import pytest

from fuzz_codes import simulation_case


@pytest.mark.parametrize("block", range(4))
def test_oracle_traces_equal_the_real_executable_on_random_codes(orc, tmp_path, block):
    if not os.path.exists(orc.ref_binary("ubs")):
        pytest.skip("reference executable not built here")
    ran = 0
    for seed in range(block * 10, block * 10 + 10):
        path = str(tmp_path / ("rnd_%d" % seed))
        frames, it, eb, nm, off, nop = simulation_case(seed, path)
        tr = orc.run_reference_binary("ubs", frames, it, path, eb, nm, off, nop, timeout=60)
        o = orc.Oracle(path)
        if not tr:                                       # "The matrix is not full rank" (tools.c:181-185): the executable stops before the first frame
            with pytest.raises(RuntimeError, match="not full rank"):
                o.sim(orc.Rng(0), eb, nm, nop, off, it, 1)
            continue
        ran += 1
        rec = o.sim(orc.Rng(0), eb, nm, nop, off, it, len(tr))
        mine = orc.trace_from_records(rec, o.K, o.logq)
        assert [list(t) for t in mine] == [list(t) for t in tr], (seed, nm, nop, off, it, eb)
    assert ran >= 7, ran
