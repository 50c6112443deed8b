#!/usr/bin/env python3
"""Regenerates tests/golden/bc_assign_v1.json from the ORACLE (oracle/sor_bc.c).

The reference cannot run in the build image (bytecode only, no JVM), so these vectors are NOT reference outputs:
they freeze the oracle's behaviour (pinned by the README known answers and the independent Python model) so that a
later edit of either the oracle or the kernels cannot drift silently.  Inputs are seeded synthetic reads.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402

import __graft_entry__ as graft  # noqa: E402


def main():
    graft.load_package()
    synth = importlib.import_module(graft.PKG_NAME + ".synth")
    sor = graft.load_oracle()
    sor.build()
    wl = synth.make_whitelist(20_000, seed=101)
    used = synth.pick_used(wl, 200, seed=102)
    cases = []
    for five_prime in (False, True):
        for max_ed in (0, 1, 2):
            reg = synth.gen_bc_region(150, used, seed=103 + five_prime, five_prime=five_prime, n_rate=0.01)
            codes, ae = reg["codes"].numpy(), reg["ae"].numpy()
            st, exp = sor.assign_batch(sor.BarcodeSet(wl.numpy()), codes, ae, max_ed=max_ed, five_prime=five_prime)
            lut = np.frombuffer(b"AGCTN", dtype=np.uint8)
            cases.append({
                "five_prime": five_prime, "max_ed": max_ed,
                "reads": [bytes(lut[r]).decode() for r in codes], "ae": ae.tolist(), "status": st.tolist(),
                "found": exp["found"].tolist(), "bc": [int(x) for x in exp["bc"]], "ed": exp["ed"].tolist(),
                "ed_sec": exp["ed_sec"].tolist(), "offset": exp["offset"].tolist(),
                "ins_minus_del": exp["ins_minus_del"].tolist(), "bc_start": exp["bc_start"].tolist(),
                "bc_end": exp["bc_end"].tolist(),
            })
    out = {"whitelist": [int(x) for x in wl.numpy()], "cases": cases, "generator": synth.GENERATOR_VERSION}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bc_assign_v1.json")
    json.dump(out, open(path, "w"))
    print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
