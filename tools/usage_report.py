#!/usr/bin/env python3
"""Register / LDS / scratch report of the compiled kernels (the *.usage.txt files the csrc Makefile
writes from -Rpass-analysis=kernel-resource-usage).  usage: python tools/usage_report.py [substring]"""
import os, re, sys
here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "codesign-kernels_amd", "csrc")
pat = sys.argv[1] if len(sys.argv) > 1 else "wm_kernel"
for f in sorted(os.listdir(here)):
    if not f.endswith(".usage.txt"):
        continue
    t = open(os.path.join(here, f)).read()
    for b in re.split(r"remark: [^\n]*Function Name: ", t)[1:]:
        name = b.split("\n")[0]
        if pat not in name:
            continue
        g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
        short = re.sub(r"^_ZN\d+", "", name)[:70]
        print(f"{f[:22]:22s} {short:70s} VGPR {g('VGPRs'):>4s} AGPR {g('AGPRs'):>3s} SGPR {g('SGPRs'):>4s} "
              f"scratch {g('ScratchSize [^:]*'):>4s} LDS {g('LDS Size [^:]*'):>6s} occ {g('Occupancy [^:]*')}")
