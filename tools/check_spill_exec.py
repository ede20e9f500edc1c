"""Build-time guard against a hipcc (ROCm 7.2, gfx950) code-generation bug found in round 2 ("the UKF chol(P) hazard"):

the register allocator may put VGPR spill code (`scratch_store_* ; Folded Spill`, `scratch_load_* ; Folded Reload`,
`v_accvgpr_write/read` spills) at the top of a divergent loop's EXIT block, AHEAD of the `s_or_b64 exec, exec, <saved>` that
re-activates the lanes.  A loop of the form `for (i = tid; i < n; i += blockDim) ...` leaves EXEC = 0 at its exit (lanes drop
out one by one through `s_andn2_b64 exec, exec, <done>`), so such a spill store writes NOTHING, and the matching reload later
reads uninitialised scratch in every lane.  Which values are hit depends on what the allocator spills there, i.e. on
unrelated code: in the failing build it was the libm sin/cos polynomial coefficients that LICM had hoisted out of the callback
loop, and the first visible effect was a 1e-6-class error in the seeds of promoted landmarks from the second callback of a
launch on.

This script reads `hipcc -S` output (or -save-temps .s files) and reports every VGPR spill / reload that sits in a basic block
AHEAD of the instruction that re-activates lanes in that block (`s_or_b64 exec, exec, x`): the exit
block of a divergent loop (EXEC == 0) or the join block of a divergent if (EXEC a subset).  Exit code 1 if anything is found.

    python tools/check_spill_exec.py file.s [file.s ...]
"""
import re
import sys

RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")  # always a widening of EXEC (s_mov_b64 / s_xor_b64 exec can go either way)
SPILL = re.compile(r"((scratch|buffer)_(store|load)\S*\s.*;\s*\d+-byte Folded (Spill|Reload))")
LABEL = re.compile(r"^\.?[A-Za-z_][\w.$]*:")
KERNEL = re.compile(r"^([A-Za-z_][\w.$]*):\s*;\s*@")
BRANCH = re.compile(r"^\s*(s_branch|s_cbranch_\w+|s_endpgm|s_setpc_b64)\b")


def scan(path):
    """Basic blocks are cut at labels and branches.  Within a block, an instruction that re-activates lanes (`s_or_b64 exec, exec, x`) means the block was ENTERED with fewer lanes than the code after it runs with -- EXEC == 0 when the
    block is the exit of a divergent loop (the case found), a subset at the join of a divergent if.  Spill code ahead of that
    instruction in the same block is executed with the narrow mask and is reported."""
    findings = []
    kernel = "?"
    block = []  # (line number, text) of the current basic block
    after_execnz = False

    def flush(block, loop_exit):
        first_restore = next((k for k, (_, t) in enumerate(block) if RESTORE.match(t.split(";")[0])), None)
        if first_restore is None:
            return
        for k in range(first_restore):
            ln_no, text = block[k]
            if SPILL.search(text):
                why = ("exit block of a divergent loop: EXEC == 0" if loop_exit else "join block of a divergent region: EXEC is a subset") + \
                      f", lanes are re-activated only at line {block[first_restore][0]}"
                findings.append((kernel, ln_no, why, text.strip()))

    lines = open(path, errors="replace").read().splitlines()
    pending_exit = False  # the previous instruction was the back-edge `s_cbranch_execnz`: what follows is the loop's exit block
    for i, ln in enumerate(lines, 1):
        m = KERNEL.match(ln)
        if m:
            flush(block, after_execnz)
            block, after_execnz, pending_exit = [], False, False
            kernel = m.group(1)
            continue
        if LABEL.match(ln):
            flush(block, after_execnz)
            block, after_execnz, pending_exit = [], pending_exit, False
            continue
        code = ln.split(";")[0]
        if not code.strip():
            continue
        block.append((i, ln))
        if BRANCH.match(code):
            flush(block[:-1], after_execnz)
            pending_exit = re.match(r"^\s*s_cbranch_execnz\b", code) is not None
            block, after_execnz = [], pending_exit  # an unlabelled fall-through block behind the back-edge is the exit block too
    flush(block, after_execnz)
    return findings


def main():
    bad = 0
    for p in sys.argv[1:]:
        f = scan(p)
        for kernel, line, why, text in f:
            print(f"{p}:{line}: [{kernel}] VGPR spill code executed with EXEC == 0 ({why}): {text}")
        bad += len(f)
    print(f"check_spill_exec: {bad} spill instruction(s) in EXEC-empty loop-exit blocks" if bad else "check_spill_exec: clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
