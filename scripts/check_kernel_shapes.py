#!/usr/bin/env python3
"""Compile every (KL, NB) instantiation of mtp_wave_kernel on its own and report registers,
spills and compiler errors (hipcc 7.2 rejects some shapes with a machine-verifier error)."""
import concurrent.futures
import re
import subprocess
import sys

SRC = "/root/repo/lammps_mtp_kokkos_amd/csrc/mtp_kernels.hip"
s = open(SRC).read()
cases = [tuple(map(int, m)) for m in re.findall(r"MTP_CASE\((\d+), (\d+)\)\n", s)]


def run(c):
    kl, kb = c
    a = re.sub(r"  MTP_CASE\((?!%d, %d\)).*\n" % (kl, kb), "", s)
    f = "/tmp/shape_%d_%d.hip" % c
    open(f, "w").write(a)
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                        "-munsafe-fp-atomics", "-I/root/repo/lammps_mtp_kokkos_amd/csrc",
                        "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c", f, "-o", f + ".o"],
                       capture_output=True, text=True)
    out = r.stderr
    res = []
    for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?VGPRs Spill: (\d+)", out, re.S):
        if "mtp_wave_kernel" in m.group(1):
            deg = re.search(r"ELb[01]ELi(\d+)E", m.group(1))
            res.append((("grade" if "ELb1E" in m.group(1) else "force") + "/deg" + (deg.group(1) if deg else "?"),
                        int(m.group(2)), int(m.group(3))))
    return c, ("ERR" if "error" in out else "ok"), res


with concurrent.futures.ThreadPoolExecutor(6) as ex:
    bad = 0
    for r in ex.map(run, cases):
        print(r)
        bad += r[1] != "ok"
sys.exit(1 if bad else 0)
