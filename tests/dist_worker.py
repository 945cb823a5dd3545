"""Worker of the world_size-2 gloo test: runs the sharded Monte-Carlo driver with a CPU stand-in decoder
(the oracle -- test infrastructure) so the sharding / counter all-reduce / ordered truncation logic is
exercised without a GPU.  Prints rank 0's trace as JSON."""
import importlib
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    matrix, nframes, itmax, ebn, nm, off, nop, batch, max_err = sys.argv[1:10]
    inflight = int(sys.argv[10]) if len(sys.argv) > 10 else 3
    nframes, itmax, nm, nop, batch, max_err = int(nframes), int(itmax), int(nm), int(nop), int(batch), int(max_err)
    ebn, off = float(ebn), float(off)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group("gloo")
    ge.load_package()
    harness = importlib.import_module(ge.PKG_NAME + ".harness")
    orc = ge.load_oracle()
    o = orc.Oracle(matrix)

    def decode_fn(noisy, sigma, cws):
        nberr, synd, iters = [], [], []
        for b in range(len(noisy)):
            r = o.decode(o.llr(noisy[b], sigma), nm, nop, off, itmax)
            nberr.append(o.bit_errors(r["decide"], cws[b])); synd.append(r["synd"]); iters.append(r["iters"])
        return np.array(nberr), np.array(synd), np.array(iters)

    cnt, rounds = harness.monte_carlo(matrix, nframes, itmax, ebn, nm, off, nop, batch=batch, decode_fn=decode_fn,
                                      max_err_frames=max_err, inflight=inflight)
    if not dist.is_initialized() or dist.get_rank() == 0:
        print("TRACE " + json.dumps({"trace": cnt.trace, "rounds": rounds, "nb": cnt.nb}))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
