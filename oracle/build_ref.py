#!/usr/bin/env python3
"""oracle/build_ref.py -- TEST INFRASTRUCTURE: build the reference itself.

Compiles the reference Fortran program
  /root/reference/mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90
from the source where it lies, into per-shape executables under oracle/_ref/
(git-ignored; only binaries ever land there).  The reference is a `program`
with compile-time sizes (:7-9), compiler-RNG inputs (:654-660) and no output
of field values, so three line-anchored edits are applied IN MEMORY to a
scratch copy that lives in a mkdtemp() directory and is deleted after the
compile -- no reference source text is written into this repository:

  1. :7-9    nslices / nz / nx            -> the requested shape
  2. :654-660 call random_number(X) x7     -> stream-binary reads of X from
                                              ./mpdata_in.bin (same order:
                                              adz,f,u,w,rho,rhow,flux)
  3. after :49 (call advect_scalar2D_cpu)  -> stream-binary dump of f, flux
                                              to ./mpdata_out.bin
  4. (fp32 builds only) :12-13             -> the reference's precision switch:
                                              the (13) line is commented and rp
                                              becomes selected_real_kind(6).
                                              NOTE: the reference's own "single
                                              precision" line asks for
                                              selected_real_kind(7), which is
                                              kind 8 (fp64) under flang and
                                              every compiler whose fp32 has
                                              precision() = 6 -- flipping the
                                              switch as written reproduces the
                                              fp64 build.  (6) is what yields
                                              IEEE fp32, the evident intent.

The arithmetic body (:477-642) is untouched.  Flags: `amdflang -O3
-ffp-contract=off` (flang does not re-associate; -O0 and -O3 agree bitwise,
SURVEY.md section 8c).  The executable still prints the reference's own
`CPU Timing:` line (:640), which bench.py's cpu_baseline leg parses.

Usage:  python oracle/build_ref.py [--f32] NCRMS NX NZ [NCRMS NX NZ ...]
Needs /root/reference and amdflang; on the GPU box neither is used -- the
prebuilt oracle/_ref binaries travel with the snapshot.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/mmf-mpdata-tracer/advect_scalar2D_pushncols_openacc.F90"
OUT_DIR = os.path.join(HERE, "_ref")
FC = shutil.which("amdflang") or "/opt/rocm/bin/amdflang"


def ref_exe_path(ncrms, nx, nz, f32=False):
    return os.path.join(OUT_DIR, f"advect_ref_{ncrms}x{nx}x{nz}" + ("_f32" if f32 else ""))


def _patched_lines(ncrms, nx, nz, f32=False):
    with open(REF_SRC) as fh:
        lines = fh.read().split("\n")

    def expect(lineno, pattern):
        if not re.search(pattern, lines[lineno - 1]):
            raise RuntimeError(f"reference line {lineno} is not what build_ref.py expects: "
                               f"{lines[lineno - 1]!r}")

    # 1. sizes (:7-9)
    expect(7, r"parameter\s*::\s*nslices\s*=")
    expect(8, r"parameter\s*::\s*nz\s*=")
    expect(9, r"parameter\s*::\s*nx\s*=")
    lines[6] = f"  integer, parameter :: nslices = {ncrms}"
    lines[7] = f"  integer, parameter :: nz      = {nz}"
    lines[8] = f"  integer, parameter :: nx      = {nx}"
    # 4. precision (:11-12)
    if f32:
        expect(12, r"^\s*!\s*integer, parameter :: rp\s*=\s*selected_real_kind\(7\)")
        expect(13, r"^\s*integer, parameter :: rp\s*=\s*selected_real_kind\(13\)")
        lines[11] = "  integer, parameter :: rp      = selected_real_kind(6)"
        lines[12] = " !integer, parameter :: rp      = selected_real_kind(13)"
    # 2. inputs (:654-660): same arrays, same order, read instead of drawn
    names = ["adz", "f", "u", "w", "rho", "rhow", "flux"]
    for off, name in enumerate(names):
        expect(654 + off, rf"call random_number\({name}\s*\)")
        lines[653 + off] = f"    read(91) {name}"
    lines[653] = ("    open(unit=91, file='mpdata_in.bin', access='stream', form='unformatted', "
                  "status='old')\n" + lines[653])
    lines[659] = lines[659] + "\n    close(91)"
    # 3. outputs: dump after the CPU call (:49)
    expect(49, r"call advect_scalar2D_cpu\(f,u,w,rho,rhow,flux\)")
    lines[48] = (lines[48] + "\n"
                 "  open(unit=92, file='mpdata_out.bin', access='stream', form='unformatted', "
                 "status='replace')\n"
                 "  write(92) f\n  write(92) flux\n  close(92)")
    return "\n".join(lines)


def build(ncrms, nx, nz, force=False, f32=False):
    """Build oracle/_ref/advect_ref_<shape>[_f32]; returns its path."""
    exe = ref_exe_path(ncrms, nx, nz, f32)
    if os.path.exists(exe) and not force:
        return exe
    if not os.path.exists(REF_SRC):
        raise FileNotFoundError(f"{REF_SRC} not present (the reference does not travel); "
                                "use the prebuilt binary")
    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="mpdata_ref_")
    try:
        src = os.path.join(tmp, "ref_patched.F90")
        with open(src, "w") as fh:
            fh.write(_patched_lines(ncrms, nx, nz, f32))
        flags = ["-O3", "-ffp-contract=off"]
        # statics (f,u,w + their _save copies) beyond 2 GB need the medium model
        static_bytes = 2 * (4 if f32 else 8) * ncrms * ((nx + 6) + (nx + 5)) * (nz - 1) + 2 * (4 if f32 else 8) * ncrms * (nx + 4) * nz
        if static_bytes > 1.5e9:
            flags.append("-mcmodel=medium")
        subprocess.run([FC, *flags, "-o", exe, src], check=True, cwd=tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return exe


# ---------------------------------------------------------------------------------------
# second mini-app: atmosphere/biharmonic_wk_kernel.F90
BWK_SRC = "/root/reference/atmosphere/biharmonic_wk_kernel.F90"


def bwk_exe_path(nelemd):
    return os.path.join(OUT_DIR, f"bwk_ref_ne{nelemd}")


def build_bwk(nelemd, force=False):
    """Build oracle/_ref/bwk_ref_ne<nelemd> from the reference program.  In-memory edits to a
    scratch copy (deleted after the compile): :15,:17 nelemd / nete -> the requested number of
    elements; after :559 (save_qtens) -> stream-binary dump of the inputs the program
    generated itself (re-drawn by initialize_data, its LCG is deterministic: dvv, the
    elements, qtens) to ./bwk_in.bin and of the CPU routine's qtens to ./bwk_out.bin.  The
    arithmetic (:109-200) and the generator (:48-58, :77-91) are untouched."""
    exe = bwk_exe_path(nelemd)
    if os.path.exists(exe) and not force:
        return exe
    if not os.path.exists(BWK_SRC):
        raise FileNotFoundError(f"{BWK_SRC} not present (the reference does not travel)")
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(BWK_SRC) as fh:
        lines = fh.read().split("\n")

    def expect(lineno, pattern):
        if not re.search(pattern, lines[lineno - 1]):
            raise RuntimeError(f"reference line {lineno} is not what build_ref.py expects: {lines[lineno - 1]!r}")

    expect(15, r"integer :: nelemd = 16")
    expect(17, r"integer :: nete = 16")
    expect(559, r"call save_qtens\(\)")
    lines[14] = f"  integer :: nelemd = {nelemd}"
    lines[16] = f"  integer :: nete = {nelemd}"
    lines[558] = (lines[558] + "\n"
                  "  open(unit=92, file='bwk_out.bin', access='stream', form='unformatted', status='replace')\n"
                  "  write(92) qtens\n  close(92)\n"
                  "  call initialize_data()\n"
                  "  open(unit=91, file='bwk_in.bin', access='stream', form='unformatted', status='replace')\n"
                  "  write(91) deriv%dvv\n"
                  "  block\n    integer :: ie_dump\n    do ie_dump = 1, nelemd\n"
                  "      write(91) elem(ie_dump)%Dinv, elem(ie_dump)%spheremp, elem(ie_dump)%tensorVisc\n"
                  "    end do\n  end block\n"
                  "  write(91) qtens\n  close(91)\n"
                  "  qtens = qtens_save")
    tmp = tempfile.mkdtemp(prefix="bwk_ref_")
    try:
        src = os.path.join(tmp, "ref_patched.F90")
        with open(src, "w") as fh:
            fh.write("\n".join(lines))
        subprocess.run([FC, "-O3", "-ffp-contract=off", "-o", exe, src], check=True, cwd=tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return exe


# ---------------------------------------------------------------------------------------
# third mini-app: nested_loops/nested.F90 ("after that")
NLK_DIR = "/root/reference/nested_loops"


def nlk_exe_path():
    return os.path.join(OUT_DIR, "nlk_ref")


def build_nlk(force=False):
    """Build oracle/_ref/nlk_ref from nested_loops/{timerMod.f90,nested_vars.F90,nested.F90}
    with -DNO_MPI (plain CPU build: no OpenACC/OpenMP-offload/YAKL/CKE).  In-memory edits to a
    scratch copy of nested.F90 (deleted after the compile): the six `call random_number(randNum)`
    (:65,:75,:90,:92,:94,:103) -> stream reads of randNum from ./nlk_rand.bin (the compiler's
    RNG is not reproducible across compilers); after the program's own CPU reference loop
    (:157) -> stream dump of the sizes, every input array and refFlx to ./nlk_out.bin, then
    `stop`.  Sizes come from ./nested.nml as in the reference.  The loop (:123-157) is untouched."""
    exe = nlk_exe_path()
    if os.path.exists(exe) and not force:
        return exe
    src_main = os.path.join(NLK_DIR, "nested.F90")
    if not os.path.exists(src_main):
        raise FileNotFoundError(f"{src_main} not present (the reference does not travel)")
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(src_main) as fh:
        lines = fh.read().split("\n")

    def expect(lineno, pattern):
        if not re.search(pattern, lines[lineno - 1]):
            raise RuntimeError(f"reference line {lineno} is not what build_ref.py expects: {lines[lineno - 1]!r}")

    for ln in (65, 75, 90, 92, 94, 103):
        expect(ln, r"call random_number\(randNum\)")
        lines[ln - 1] = lines[ln - 1].replace("call random_number(randNum)", "read(91) randNum")
    expect(48, r"call alloc_vars")
    lines[47] = (lines[47] + "\n   open(unit=91, file='nlk_rand.bin', access='stream', form='unformatted', status='old')")
    expect(157, r"end do ! edge loop")
    lines[156] = (lines[156] + "\n"
                  "   close(91)\n"
                  "   open(unit=92, file='nlk_out.bin', access='stream', form='unformatted', status='replace')\n"
                  "   write(92) nEdges, nCells, nVertLevels, nvldim, nAdv\n"
                  "   write(92) nAdvCellsForEdge, advCellsForEdge, minLevelCell, maxLevelCell\n"
                  "   write(92) tracerCur, normalThicknessFlux, advMaskHighOrder, advCoefs, advCoefs3rd\n"
                  "   write(92) coef3rdOrder\n"
                  "   write(92) refFlx\n"
                  "   close(92)\n"
                  "   stop")
    tmp = tempfile.mkdtemp(prefix="nlk_ref_")
    try:
        src = os.path.join(tmp, "nested_patched.F90")
        with open(src, "w") as fh:
            fh.write("\n".join(lines))
        subprocess.run([FC, "-O3", "-ffp-contract=off", "-DNO_MPI", "-o", exe,
                        os.path.join(NLK_DIR, "timerMod.f90"), os.path.join(NLK_DIR, "nested_vars.F90"), src],
                       check=True, cwd=tmp)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return exe


if __name__ == "__main__":
    if "--nlk" in sys.argv[1:]:
        print(build_nlk(force=True))
        sys.exit(0)
    if "--bwk" in sys.argv[1:]:
        for a in sys.argv[1:]:
            if a != "--bwk":
                print(build_bwk(int(a), force=True))
        sys.exit(0)
    f32 = "--f32" in sys.argv[1:]
    args = [int(a) for a in sys.argv[1:] if a != "--f32"]
    if not args or len(args) % 3:
        sys.exit(__doc__)
    for j in range(0, len(args), 3):
        print(build(*args[j:j + 3], force=True, f32=f32))
