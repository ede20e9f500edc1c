"""Generator of awesomeslam_amd/csrc/ekf_large_trsm16_regions.inc: the hand-scheduled half-block regions of the bf16-pipe sweep
(ekf_large_trsm16.h).

A region = the 24 v_mfma_f32_16x16x32_bf16 of one half block (four row tiles x six split products) with, INSIDE their issue gaps, everything the
NEXT half block needs: the eight strip registers of its B operand read from the AGPRs and split into three bf16 pieces (52 VALU), its twelve
16-byte operand rows read from LDS, and (first halves) the previous block's sums added to the column's running sum (16 VALU).  One wave per SIMD:
an MFMA occupies the matrix pipe for 16 cycles and the vector issue for 8 of them, so about two 4-cycle VALU instructions fit a gap for free
(MI355X_MICROARCH.md, cycle constants); hipcc would not produce this interleave from builtins (profiles/r03_experiments.md).

Every operand lives in a FIXED physical register tuple, bound through "{v[a:b]}" constraints: the assembly text can then name single registers of
a tuple (the split writes one packed register of a four-register MFMA operand at a time), and the compiler still tracks the liveness of every
tuple -- unlike the AGPR strip nothing here is hidden from it.  Two operand sets alternate (a region consumes one and fills the other):

    set P   A rows v[96:143]  (plane p, row tile t -> v[96 + 16 p + 4 t ..+3])    B pieces h v[144:147]  m v[148:151]  l v[152:155]
    set Q   A rows v[160:207]                                                     B pieces h v[208:211]  m v[212:215]  l v[216:219]
    sums    even blocks v[32:47], odd blocks v[48:63], running sum of the column v[64:79]     scratch v[224:239]

    python tools/gen_trsm16_regions.py            (re)writes the .inc; the build fails if the checked-in file differs from the generator's output
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "awesomeslam_amd", "csrc", "ekf_large_trsm16_regions.inc")

PLD, LB = 64, 64  # unpadded rows (LDS-DMA); the chunk swizzle and the half are part of the lane's base address
PLANE = LB * PLD
SETS = {"P": dict(A=96, B=144), "Q": dict(A=160, B=208)}
ACC = {"E": 32, "O": 48}
RUN = 64
X, TMP = 224, 232
PRODUCTS = [(0, "l"), (1, "m"), (2, "h"), (0, "m"), (1, "h"), (0, "h")]  # (plane of A, piece of B): small terms first
BOFF = {"h": 0, "m": 4, "l": 8}


def region(src, dst, acc, first, prev, half_next, strip=True, tiles=4, rows=4, diag="", dma=(), csplit=False):
    """src/dst: operand sets; acc: 'E' / 'O'; first: the sums start here; prev: set whose sums are added to the running sum (or None);
    half_next: which half (0 / 1) of the staged block the next operand rows come from; strip: read + split the next B operand from the strip;
    csplit: the next B operand is split from registers 8 .. 15 of the running-sum tuple instead (the closing block: C tiles 2, 3);
    tiles: row tiles t = 4 - tiles .. 3 are multiplied (the triangular closing block); rows: operand row tiles loaded for the next region;
    dma: which of the six LDS-DMA pieces of the block three ahead this region issues (piece i: plane i >> 1, eight-row group 4 (i & 1) + wave)"""
    a_src, b_src, a_dst, b_dst, cbase = SETS[src]["A"], SETS[src]["B"], SETS[dst]["A"], SETS[dst]["B"], ACC[acc]
    t0 = 4 - tiles
    mf = []
    for s, (p, piece) in enumerate(PRODUCTS):
        for t in range(t0, 4):
            d = f"v[{cbase + 4 * t}:{cbase + 4 * t + 3}]"
            a = f"v[{a_src + 16 * p + 4 * t}:{a_src + 16 * p + 4 * t + 3}]"
            b = f"v[{b_src + BOFF[piece]}:{b_src + BOFF[piece] + 3}]"
            mf.append(f"v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {'0' if first and s == 0 else d}")
    valu = []
    if strip:
        valu += [f"v_accvgpr_read_b32 v{X + i}, a[%c[r0]+{i}]" for i in range(8)]
    adds = [f"v_add_f32 v{RUN + i}, v{RUN + i}, v{ACC[prev] + i}" for i in range(16)] if prev else []
    split = []
    if strip or csplit:
        steps = []
        xb = RUN + 8 if csplit else X
        for e in range(4):
            x0, x1, ta, tb = xb + 2 * e, xb + 2 * e + 1, TMP + 2 * e, TMP + 2 * e + 1
            h, m, l = b_dst + e, b_dst + 4 + e, b_dst + 8 + e
            steps.append([
                f"v_cvt_pk_bf16_f32 v{h}, v{x0}, v{x1}",
                f"v_lshlrev_b32 v{ta}, 16, v{h}",
                f"v_and_b32 v{tb}, %[msk], v{h}",
                f"v_sub_f32 v{x0}, v{x0}, v{ta}",
                f"v_sub_f32 v{x1}, v{x1}, v{tb}",
                f"v_cvt_pk_bf16_f32 v{m}, v{x0}, v{x1}",
                f"v_lshlrev_b32 v{ta}, 16, v{m}",
                f"v_and_b32 v{tb}, %[msk], v{m}",
                f"v_sub_f32 v{x0}, v{x0}, v{ta}",
                f"v_sub_f32 v{x1}, v{x1}, v{tb}",
                f"v_cvt_pk_bf16_f32 v{l}, v{x0}, v{x1}",
            ])
        for s in range(11):
            for e in range(4):
                split.append(steps[e][s])
    # the additions are independent of everything else in the region: they fill the gaps while the strip reads land, then alternate with the split
    queue = list(valu)
    while adds or split:
        if split:
            queue.append(split.pop(0))
        if split:
            queue.append(split.pop(0))
        if adds:
            queue.append(adds.pop(0))
    ds = []
    for p in range(3):
        for t in range(4 - rows, 4):
            off = (16 * t * PLD + p * PLANE) * 2
            ds.append(f"ds_read_b128 v[{a_dst + 16 * p + 4 * t}:{a_dst + 16 * p + 4 * t + 3}], %[lds] offset:{off}")
    if "novalu" in diag:
        queue = []
    if "nods" in diag:
        ds = []
    n = len(mf)
    lines = []
    vq, dq = list(queue), list(ds)
    # LDS-DMA pieces: M0 = the piece's LDS address one gap ahead of the load that uses it
    dma_at = {}
    for q, piece in enumerate(dma):
        g = 2 + (q * (n - 4)) // max(1, len(dma))
        dma_at[g] = f"s_add_u32 m0, %[ldsw], {(piece >> 1) * 8192 + (piece & 1) * 4096}"
        dma_at[g + 1] = f"buffer_load_dwordx4 %[vo{q}], %[rsrc], %[so] offen lds"
    for i, m in enumerate(mf):
        lines.append(m)
        if i in dma_at:
            lines.append(dma_at[i])
        take = (len(queue) * (i + 1)) // n - (len(queue) * i) // n
        for _ in range(take):
            lines.append(vq.pop(0))
        if dq:
            lines.append(dq.pop(0))  # the operand rows of the next region: one LDS read per gap from the first gap on (the last has landed long before the wait)
    assert not vq and not dq
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_nop 1")
    return lines


def rregion(src, dst, dma):
    """Right-looking history half (large_trsm_bf16r): the 24 MFMAs of one half block accumulate STRAIGHT INTO THE STRIP -- tile t of block column K
    = a[r0 + 4 t .. + 3], r0 = 16 K a compile-time operand -- with the pieces of the (negated) solved column as B operand; behind them only the
    twelve operand-row reads of the next half and three DMA pieces: no VALU instruction at all."""
    a_src, b_src, a_dst = SETS[src]["A"], SETS[src]["B"], SETS[dst]["A"]
    mf = []
    for s, (p, piece) in enumerate(PRODUCTS):
        for t in range(4):
            d = f"a[%c[r0]+{4 * t}:%c[r0]+{4 * t + 3}]"
            a = f"v[{a_src + 16 * p + 4 * t}:{a_src + 16 * p + 4 * t + 3}]"
            b = f"v[{b_src + BOFF[piece]}:{b_src + BOFF[piece] + 3}]"
            mf.append(f"v_mfma_f32_16x16x32_bf16 {d}, {a}, {b}, {d}")
    ds = []
    for p in range(3):
        for t in range(4):
            off = (16 * t * PLD + p * PLANE) * 2
            ds.append(f"ds_read_b128 v[{a_dst + 16 * p + 4 * t}:{a_dst + 16 * p + 4 * t + 3}], %[lds] offset:{off}")
    n = len(mf)
    dma_at = {}
    for q, piece in enumerate(dma):
        g = 2 + (q * (n - 4)) // max(1, len(dma))
        dma_at[g] = f"s_add_u32 m0, %[ldsw], {(piece >> 1) * 8192 + (piece & 1) * 4096}"
        dma_at[g + 1] = f"buffer_load_dwordx4 %[vo{q}], %[rsrc], %[so] offen lds"
    lines = []
    for i, m in enumerate(mf):
        lines.append(m)
        if i in dma_at:
            lines.append(dma_at[i])
        if ds:
            lines.append(ds.pop(0))
    lines.append("s_waitcnt lgkmcnt(0)")
    lines.append("s_nop 1")
    return lines


def strip_write_dyn():
    """x (v[32:47], the closing block's accumulators) -> strip registers a[16 k .. 16 k + 15] for a RUN-TIME block column k (0 .. 15, in a scalar
    register): a computed jump into a table of sixteen 136-byte entries (sixteen 8-byte v_accvgpr_write + s_branch + s_nop).  hipcc compiles the
    equivalent switch into a tree of compares and branches that costs ~ 600 cycles per closing block (trsm_bench phase stamps)."""
    lines = ["s_getpc_b64 s[96:97]", ".Lt16_pc_%=:", "s_mul_i32 s98, %[k], 136", "s_add_u32 s96, s96, s98", "s_addc_u32 s97, s97, 0",
             "s_add_u32 s96, s96, .Lt16_tab_%=-.Lt16_pc_%=", "s_addc_u32 s97, s97, 0", "s_setpc_b64 s[96:97]", ".Lt16_tab_%=:"]
    for k in range(16):
        lines += [f"v_accvgpr_write_b32 a{16 * k + i}, v{ACC['E'] + i}" for i in range(16)]
        lines += ["s_branch .Lt16_end_%=", "s_nop 0"]
    lines += [".Lt16_end_%=:", "s_nop 3"]
    return lines


def cstring(lines):
    return "\n".join(f'        "{ln}\\n\\t"' for ln in lines[:-1]) + f'\n        "{lines[-1]}"'


def main():
    variants = {
        # history blocks: first half (sums start), second half
        "H0_E": region("P", "Q", "E", True, None, 1, dma=(0, 1, 2)),
        "H0_E_ADD": region("P", "Q", "E", True, "O", 1, dma=(0, 1, 2)),
        "H0_O_ADD": region("P", "Q", "O", True, "E", 1, dma=(0, 1, 2)),
        "H1_E": region("Q", "P", "E", False, None, 0, dma=(3, 4, 5)),
        "H1_O": region("Q", "P", "O", False, None, 0, dma=(3, 4, 5)),
        # the closing block X = C Linv^T (sums in the E registers): first half (C tiles 2, 3 are split from the running-sum registers behind it),
        # second half (row tiles 2, 3 only: Linv is lower triangular; the first operand of the next block column is prepared behind it)
        "C0": region("P", "Q", "E", True, None, 1, strip=False, csplit=True, rows=2, dma=(0, 1, 2)),
        "C1": region("Q", "P", "E", False, None, 0, tiles=2, dma=(3, 4, 5)),
        # timing diagnostics (tools/ubench/trsm_bench.hip): wrong results
        "H1_E_NOVALU": region("Q", "P", "E", False, None, 0, diag="novalu", dma=(3, 4, 5)),
        "H1_E_NODS": region("Q", "P", "E", False, None, 0, diag="nods", dma=(3, 4, 5)),
        "H1_E_BARE": region("Q", "P", "E", False, None, 0, diag="novalu nods"),
        "H1_E_NODMA": region("Q", "P", "E", False, None, 0),
        # right-looking sweep (large_trsm_bf16r): history halves into the strip; the closing block's second half without a strip operand
        "R0": rregion("P", "Q", (0, 1, 2)),
        "R1": rregion("Q", "P", (3, 4, 5)),
        "C1R": region("Q", "P", "E", False, None, 0, strip=False, tiles=2, dma=(3, 4, 5)),
        "STRIP_WRITE_DYN": strip_write_dyn(),
    }
    out = ["// GENERATED by tools/gen_trsm16_regions.py -- do not edit (the register map and the interleave are described there)"]
    for name, lines in variants.items():
        out.append(f"#define ASLAM_T16_{name} \\")
        body = [f'        "{ln}\\n\\t" \\' for ln in lines[:-1]] + [f'        "{lines[-1]}"']
        out += body
        out.append("")
    text = "\n".join(out) + "\n"
    if len(sys.argv) > 1 and sys.argv[1] == "--check":
        cur = open(OUT).read() if os.path.exists(OUT) else ""
        if cur != text:
            print("ekf_large_trsm16_regions.inc is stale: run python tools/gen_trsm16_regions.py")
            return 1
        print("gen_trsm16_regions: up to date")
        return 0
    with open(OUT, "w") as f:
        f.write(text)
    for name, lines in variants.items():
        print(f"{name}: {sum(l.startswith('v_mfma') for l in lines)} MFMA, {sum(l.startswith('v_') and not l.startswith('v_mfma') for l in lines)} VALU, "
              f"{sum(l.startswith('ds_') for l in lines)} LDS reads")
    return 0


if __name__ == "__main__":
    sys.exit(main())
