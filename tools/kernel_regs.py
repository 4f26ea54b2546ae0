"""Register / scratch / LDS use per kernel of a hipcc `--save-temps` assembly file (quick check that a change did not spill).
usage: python tools/kernel_regs.py <file>-hip-amdgcn-amd-amdhsa-gfx950.s"""
import re
import subprocess
import sys


def main(path):
    s = open(path).read()
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, flags=re.S):
        name, body = m.group(1), m.group(2)
        g = lambda k: re.search(k + r" (\d+)", body).group(1)
        rows.append((name, g("next_free_vgpr"), g("next_free_sgpr"), g("private_segment_fixed_size")))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for (n, vg, sg, pv), dn in zip(rows, names):
        dn = re.sub(r"^void ", "", dn)
        print(f"{dn[:110]:110s} vgpr {vg:>4s} sgpr {sg:>4s} scratch {pv}")


if __name__ == "__main__":
    main(sys.argv[1])
