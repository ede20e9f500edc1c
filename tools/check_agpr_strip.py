"""Build-time guard for the hand-allocated AGPR strip of large_trsm_pipe / large_chol_resident (csrc/ekf_large_trsm.h).

Those kernels keep a wave's solved row strip in the accumulator registers a0 .. a255 and name them in inline assembly only; the
compiler's register allocator does not know they are live.  The one reservation (`asm volatile("" ::: "a0", "a255")`) sizes the
allocation, it does not stop the compiler from using AGPRs for values of its own (fp64 MFMA accumulators, VGPR -> AGPR spill
copies).  Such a use INSIDE a sweep would silently corrupt V / L.  The sweep brackets itself with the assembly comments
`ASLAM_STRIP_LIVE_BEGIN` / `ASLAM_STRIP_LIVE_END`; this script reads the device assembly of the product compilation
(-save-temps) and fails the build when

  * a kernel that contains the markers references an AGPR (a<N>, a[<N>:<M>], v_accvgpr_*) in compiler-generated code -- i.e.
    outside `;;#ASMSTART` .. `;;#ASMEND` -- between a BEGIN and the next END marker (linear order: the sweep is one loop nest
    whose blocks the compiler lays out between the two), or
  * `large_trsm_pipe`, whose strip is live from its first to its last instruction, references one anywhere outside inline
    assembly, or
  * BEGIN / END markers do not pair up, or a branch crosses a BEGIN .. END range in either direction (a cold block -- the taken side of a
    `__builtin_expect(..., 0)`, a spin loop -- laid out behind the END marker would run with the strip live and not be scanned: the
    linear-order reading would be unsound), or
  * one of the four strip kernels (REQUIRED) has no marker pair at all (the guard would be vacuous for it).

    python tools/check_agpr_strip.py file.s [file.s ...]
"""
import re
import sys

KERNEL = re.compile(r"^([A-Za-z_][\w.$]*):\s*;\s*@")
AGPR = re.compile(r"(?<![\w.])a(\d+|\[\d+:\d+\])(?![\w])|v_accvgpr_")
WHOLE_KERNEL = ("large_trsm_pipe", "large_trsm_bf16")  # one sweep per workgroup: nothing of the compiler's may touch an AGPR anywhere in the kernel
REQUIRED = ("large_trsm_pipe", "large_chol_resident", "large_trsm_bf16", "large_chol_bf16")  # every kernel that keeps a strip
LABEL = re.compile(r"^(\.LBB\w+):")
BRANCH = re.compile(r"^\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\w+)")


def scan(path):
    findings = []
    kernel, in_asm, live, begins, ends = None, False, False, 0, 0
    seen_markers = {}
    labels, branches = {}, []  # (kernel, label) -> (line, inside a live range); (kernel, line, inside, target, text)
    for i, ln in enumerate(open(path, errors="replace").read().splitlines(), 1):
        m = KERNEL.match(ln)
        if m:
            if kernel and live:
                findings.append((kernel, i, "ASLAM_STRIP_LIVE_BEGIN without a matching END before the next kernel", ""))
            kernel, in_asm, live = m.group(1), False, False
            continue
        if kernel is None:
            continue
        if ";;#ASMSTART" in ln:
            in_asm = True
            continue
        if ";;#ASMEND" in ln:
            in_asm = False
            continue
        if "ASLAM_STRIP_LIVE_BEGIN" in ln:
            if live:
                findings.append((kernel, i, "nested ASLAM_STRIP_LIVE_BEGIN: the block layout no longer reads linearly", ln.strip()))
            live = True
            seen_markers[kernel] = seen_markers.get(kernel, 0) + 1
            continue
        if "ASLAM_STRIP_LIVE_END" in ln:
            if not live:
                findings.append((kernel, i, "ASLAM_STRIP_LIVE_END without a BEGIN in front of it", ln.strip()))
            live = False
            continue
        if in_asm:
            continue
        m = LABEL.match(ln)
        if m:
            labels[(kernel, m.group(1))] = (i, live)
        m = BRANCH.match(ln)
        if m:
            branches.append((kernel, i, live, m.group(2), ln.strip()))
        code = ln.split(";")[0]
        if not code.strip() or code.lstrip().startswith("."):
            continue
        whole = any(w in kernel for w in WHOLE_KERNEL)
        if (live or whole) and AGPR.search(code):
            findings.append((kernel, i, "compiler-generated AGPR use while the hand-allocated strip is live", ln.strip()))
    for k, i, inside, target, text in branches:
        if k in seen_markers and (k, target) in labels and labels[(k, target)][1] != inside:
            findings.append((k, i, "branch %s an ASLAM_STRIP_LIVE range (target at line %d): the linear-order reading is unsound" % ("out of" if inside else "into", labels[(k, target)][0]), text))
    for r in REQUIRED:
        if not any(r in k for k in seen_markers):
            findings.append((r, 0, "no ASLAM_STRIP_LIVE_BEGIN / END markers found in this kernel: the guard would be vacuous", ""))
    return findings


def main():
    bad = 0
    for p in sys.argv[1:]:
        for kernel, line, why, text in scan(p):
            print(f"{p}:{line}: [{kernel}] {why}: {text}")
            bad += 1
    print(f"check_agpr_strip: {bad} finding(s)" if bad else "check_agpr_strip: clean")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
