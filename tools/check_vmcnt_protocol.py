"""Build-time guard for the hand-counted `s_waitcnt vmcnt` protocol of the bf16-pipe sweeps (csrc/ekf_large_trsm16.h: large_trsm_bf16,
large_chol_bf16).

Inside a sweep every vector-memory instruction is counted by hand: the LDS-DMA pieces of the blocks of L and of the row slices are issued
from inline assembly, and `Pipe::end<N>` / the closing block wait with `s_waitcnt vmcnt(12 + NST)` etc., where NST is the number of
vector-memory STORES the compiler emits for a closing block (the binary32 stores of the solved columns, plus the plane stores of the
Cholesky).  Those stores are C++ (`*reinterpret_cast<f4 *>(...) = x`, `__builtin_amdgcn_raw_buffer_store_b128`): one more vector-memory
instruction from the compiler -- a scratch spill or reload, a split store, a hoisted load -- would make every wait one less strict and bring
back the load-dependent LDS-DMA race of round 3 (data read one block early; passed every test on an idle chip).  A compiler-generated
`s_waitcnt vmcnt` inside a sweep would be the opposite failure (it drains the pieces in flight: slow, not wrong) and is flagged too.

The sweep's BEGIN marker carries what the counts assume:

    ; ASLAM_STRIP_LIVE_BEGIN vmem: global_store_dwordx4=4 buffer_store_dwordx4=6

and this script fails the build unless the compiler-generated code (outside `;;#ASMSTART` .. `;;#ASMEND`) between that marker and the next
ASLAM_STRIP_LIVE_END contains exactly those vector-memory instructions, no other global_/buffer_/flat_/scratch_ instruction, and no
`s_waitcnt` with a vmcnt field.  Kernels whose BEGIN marker has no `vmem:` clause (the fp32-MFMA sweeps, whose loads are the compiler's)
are not checked.  REQUIRED lists the kernels that must carry the clause, so the guard cannot go vacuous.

    python tools/check_vmcnt_protocol.py file.s [file.s ...]
"""
import re
import sys
from collections import Counter

KERNEL = re.compile(r"^([A-Za-z_][\w.$]*):\s*;\s*@")
VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_|image_)")
REQUIRED = ("large_trsm_bf16", "large_chol_bf16")


def scan(path):
    findings, checked = [], set()
    kernel, in_asm, expect, got, begin_line = None, False, None, None, 0
    for i, ln in enumerate(open(path, errors="replace").read().splitlines(), 1):
        m = KERNEL.match(ln)
        if m:
            kernel, in_asm, expect = m.group(1), False, None
            continue
        if kernel is None:
            continue
        if "ASLAM_STRIP_LIVE_BEGIN" in ln:
            expect = None
            if "vmem:" in ln:
                expect = Counter({k: int(v) for k, v in re.findall(r"(\w+)=(\d+)", ln.split("vmem:")[1])})
                got, begin_line = Counter(), i
            continue
        if "ASLAM_STRIP_LIVE_END" in ln:
            if expect is not None:
                checked.add(kernel)
                if got != expect:
                    findings.append((kernel, begin_line, f"compiler-generated vector-memory instructions {dict(got)} differ from what the vmcnt counts assume {dict(expect)}", ""))
            expect = None
            continue
        if ";;#ASMSTART" in ln:
            in_asm = True
            continue
        if ";;#ASMEND" in ln:
            in_asm = False
            continue
        if in_asm or expect is None:
            continue
        code = ln.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        op = code.split()[0]
        if VMEM.match(op):
            got[op] += 1
            if op not in expect:
                findings.append((kernel, i, "unexpected compiler-generated vector-memory instruction inside a hand-counted sweep", ln.strip()))
        elif op == "s_waitcnt" and "vmcnt" in code:
            findings.append((kernel, i, "compiler-generated s_waitcnt vmcnt inside a hand-counted sweep (drains the LDS-DMA pieces in flight)", ln.strip()))
    return findings, checked


def main():
    bad, checked = 0, set()
    for p in sys.argv[1:]:
        f, c = scan(p)
        checked |= c
        for kernel, line, why, text in f:
            print(f"{p}:{line}: [{kernel}] {why}: {text}")
            bad += 1
    for r in REQUIRED:
        if not any(r in k for k in checked):
            print(f"[{r}] no ASLAM_STRIP_LIVE_BEGIN marker with a vmem: clause found: the guard would be vacuous")
            bad += 1
    print(f"check_vmcnt_protocol: {bad} finding(s)" if bad else f"check_vmcnt_protocol: clean ({len(checked)} kernels)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
