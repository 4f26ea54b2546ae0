"""Instruction mix of one kernel of a hipcc `--save-temps` assembly file, per basic-block range: VALU (by opcode), MFMA, LDS, scalar,
global loads.  Used to see what a loop body really issues beside its MFMAs (e.g. the flash-attention tile loop: 32 v_exp_f32, 16
v_cvt_pk_bf16_f32, 16 v_dot2c_f32_bf16, 26 max per 16 MFMAs; the 32 v_pk_mul_f32 / 48 v_sub_f32 of its listing sit in the rare rebase branch).
usage: python tools/loop_mix.py <file>-hip-amdgcn-amd-amdhsa-gfx950.s <substring of the mangled kernel name> [first_line last_line]
Without a line range it lists the kernel's labels and branches (line numbers relative to the kernel) so that a range can be picked."""
import collections
import re
import sys


def main(path, needle, rng):
    s = open(path).read()
    m = next((m for m in re.finditer(r"^(\S+): ; @", s, flags=re.M) if needle in m.group(1)), None)
    if m is None:
        sys.exit(f"no kernel matching {needle!r}")
    body = s[m.end():s.index("s_endpgm", m.end())].split("\n")
    print(m.group(1), len(body), "lines")
    if rng is None:
        labels = {mm.group(1): i for i, line in enumerate(body) if (mm := re.match(r"^(\.LBB\d+_\d+):", line))}
        for i, line in enumerate(body):
            mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", line)
            if mm:
                t = labels.get(mm.group(1))
                print(f"{i:6d}  {line.strip():40s} -> {t}{'   (backward)' if t is not None and t < i else ''}")
        return
    a, b = rng
    ops = collections.Counter(line.strip().split()[0] for line in body[a:b] if line.startswith("\t") and line.strip() and not line.strip().startswith(";"))
    valu = {k: v for k, v in ops.items() if k.startswith("v_") and not k.startswith("v_mfma")}
    print(f"lines {a}..{b}: VALU {sum(valu.values())}, MFMA {sum(v for k, v in ops.items() if k.startswith('v_mfma'))}, "
          f"LDS {sum(v for k, v in ops.items() if k.startswith('ds_'))}, scalar {sum(v for k, v in ops.items() if k.startswith('s_'))}, "
          f"global {sum(v for k, v in ops.items() if k.startswith('global_') or k.startswith('buffer_'))}")
    for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
        print(f"   {v:5d}  {k}")


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    main(sys.argv[1], sys.argv[2], (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) >= 5 else None)
