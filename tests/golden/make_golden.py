#!/usr/bin/env python3
"""tests/golden/make_golden.py -- regenerate the golden vectors.

Each fixture is the OUTPUT (f, flux) of the reference Fortran program itself
(built from /root/reference by oracle/build_ref.py, `amdflang -O3
-ffp-contract=off`) on inputs produced by this repo's seeded generator
(oracle/oracle.py:make_inputs).  Inputs are not stored: they are regenerated
from (shape, seed, dist).  Only data is committed (npz + manifest.json); no
reference source text.  Runs in the build container only (needs
/root/reference + amdflang).

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import build_ref  # noqa: E402
from oracle import oracle as O  # noqa: E402
from oracle import bwk as B  # noqa: E402
from oracle import nlk as N  # noqa: E402

# (ncrms, nx, nz, seed, dist)
CASES = [
    (3, 8, 6, 100, O.DIST_CONDITIONED),
    (3, 8, 6, 100, O.DIST_RAW),
    (3, 8, 6, 100, O.DIST_RAW_SIGNED),
    (8, 32, 28, 100, O.DIST_CONDITIONED),
    (8, 32, 28, 7, O.DIST_RAW_SIGNED),
    (64, 32, 28, 100, O.DIST_CONDITIONED),   # BASELINE.json configs[0]
    (64, 32, 28, 100, O.DIST_RAW),           # ... with the reference's U[0,1) law
    (48, 32, 58, 100, O.DIST_RAW),           # the reference's shipped size (:7-9)
]


# fp32 build of the reference (rp = IEEE single; see oracle/build_ref.py, edit 4): pins the
# fp32 build of the oracle (mpdata_oracle.c with -DMPDATA_ORACLE_F32)
CASES_F32 = [
    (3, 8, 6, 100, O.DIST_CONDITIONED),
    (3, 8, 6, 100, O.DIST_RAW_SIGNED),
    (8, 32, 28, 100, O.DIST_CONDITIONED),
    (64, 32, 28, 100, O.DIST_RAW),
]


def case_name(ncrms, nx, nz, seed, dist, f32=False):
    return f"ref_{ncrms}x{nx}x{nz}_seed{seed}_dist{dist}" + ("_f32" if f32 else "")


def main():
    manifest = {"generator": "tests/golden/make_golden.py",
                "reference": "mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90 "
                             "(advect_scalar2D_cpu, :477-642)",
                "compiler": "amdflang -O3 -ffp-contract=off",
                "cases": []}
    for (ncrms, nx, nz, seed, dist, f32) in [c + (False,) for c in CASES] + [c + (True,) for c in CASES_F32]:
        build_ref.build(ncrms, nx, nz, f32=f32)
        inp = O.make_inputs(ncrms, nx, nz, seed=seed, dist=dist,
                            dtype=np.float32 if f32 else np.float64)
        f, flux, _ = O.run_reference(inp)
        name = case_name(ncrms, nx, nz, seed, dist, f32)
        path = os.path.join(HERE, name + ".npz")
        np.savez(path, f=f, flux=flux)
        manifest["cases"].append({
            "name": name, "ncrms": ncrms, "nx": nx, "nz": nz, "seed": seed, "dist": dist,
            "dtype": "f32" if f32 else "f64",
            "f_sha256": hashlib.sha256(f.tobytes(order="F")).hexdigest(),
            "flux_sha256": hashlib.sha256(flux.tobytes(order="F")).hexdigest(),
            "inputs_sha256": hashlib.sha256(b"".join(
                inp[k].tobytes(order="F") for k in ("adz", "f", "u", "w", "rho", "rhow", "flux")
            )).hexdigest(),
            "f_min": float(f.min()), "f_max": float(f.max()),
        })
        print(name, f.shape, flux.shape)
    # second mini-app (atmosphere/biharmonic_wk_kernel.F90): the reference program generates
    # its own inputs (a portable LCG); fixtures = the qtens its CPU routine returns.  One
    # element in full, the shipped size (16 elements) as a sha256.
    manifest["bwk"] = {"reference": "atmosphere/biharmonic_wk_kernel.F90 (biharmonic_wk_scalar CPU, :186-200; "
                                    "inputs: initialize_data :48-58)", "cases": []}
    for nelemd, keep in ((1, True), (16, False)):
        build_ref.build_bwk(nelemd)
        inp, out, _ = B.run_reference(nelemd)
        name = f"bwk_ref_ne{nelemd}"
        if keep:
            np.savez(os.path.join(HERE, name + ".npz"), qtens=out)
        manifest["bwk"]["cases"].append({
            "name": name, "nelemd": nelemd, "nlev": B.NLEV, "qsize": B.QSIZE, "stored": keep,
            "qtens_out_sha256": hashlib.sha256(out.tobytes(order="F")).hexdigest(),
            "inputs_sha256": hashlib.sha256(b"".join(inp[k].tobytes(order="F") for k in ("dvv", "elem", "qtens"))).hexdigest(),
            "out_min": float(out.min()), "out_max": float(out.max())})
        print(name, out.shape)
    # third mini-app (nested_loops/nested.F90): the program's inputs depend on the compiler's
    # random_number, so the reference is fed a seeded stream (oracle/build_ref.py --nlk) and the
    # fixture keeps the inputs the PROGRAM built from it together with its refFlx.
    build_ref.build_nlk()
    manifest["nlk"] = {"reference": "nested_loops/nested.F90 (CPU reference loop :123-157 -> refFlx)", "cases": []}
    for name, shape, seed, keep in (("nlk_ref_small", (64, 20, 12, 4), 5, True),
                                    ("nlk_ref_nml", (25600, 2800, 100, 10), 5, False)):
        inp, ref = N.run_reference(*shape, seed=seed)
        if keep:
            np.savez(os.path.join(HERE, name + ".npz"), refFlx=ref, **{k: v for k, v in inp.items()})
        manifest["nlk"]["cases"].append({
            "name": name, "nEdges": shape[0], "nCells": shape[1], "nVertLevels": shape[2], "nAdv": shape[3],
            "seed": seed, "stored": keep, "refFlx_sha256": hashlib.sha256(ref.tobytes(order="F")).hexdigest()})
        print(name, ref.shape)
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)


if __name__ == "__main__":
    main()
