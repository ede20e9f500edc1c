#!/usr/bin/env python3
"""Build guard: a DPP instruction must not read a VGPR that a VALU instruction wrote less than two wait states earlier
(gfx90a+ data hazard, 2 wait states; the DPP fmacs of factor_diag_tile_fast / factor_diag_tile_f32 are inline assembly, which
the compiler's hazard recogniser does not look into, while register copies the allocator inserts in front of them are its own).

usage: check_dpp_hazard.py file.s     exit 1 and a report when a hazard is found"""
import re
import sys

VALU = re.compile(r"^(v_[a-z0-9_]+)\s+(.*)$")
REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.search(tok)
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def check(path):
    bad = []
    kernel = "?"
    window = []  # (wait states this instruction provides, set of VGPRs it writes by VALU)
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            continue
        if line.endswith(":"):
            if not line.startswith(".L") and not line.startswith("BB"):
                kernel = line[:-1]
            window = []  # a label: control flow may join here; the compiler's own recogniser handles those edges conservatively
            continue
        if line.startswith("s_nop"):
            n = int(line.split()[1], 0) + 1
            window.append((n, set()))
            continue
        m = VALU.match(line)
        writes = set()
        if m:
            ops = [o.strip() for o in m.group(2).split(",")]
            if "_dpp" in m.group(1):
                src = regs(ops[1].lstrip("-|"))
                # wait states since each earlier VALU write: instructions in between count one each, s_nop N counts N + 1
                dist = 0
                for ws, wr in reversed(window):
                    if wr & src and dist < 2:
                        bad.append((kernel, ln, line, dist))
                        break
                    dist += ws
                    if dist >= 2:
                        break
            if not m.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                writes = regs(ops[0])
        window.append((1, writes))
        window = window[-4:]
    return bad


if __name__ == "__main__":
    bad = check(sys.argv[1])
    for k, ln, line, dist in bad:
        print("DPP hazard: %s line %d: `%s` reads a VGPR written %d wait state(s) earlier" % (k, ln, line, dist))
    if bad:
        sys.exit(1)
    print("check_dpp_hazard: clean (%s)" % sys.argv[1])
