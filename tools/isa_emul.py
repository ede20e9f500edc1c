"""A small interpreter for straight-line gfx950 assembly (the subset hipcc emits for factor_diag_tile_fast), used to check
whether a compiled instance of the routine is LOGICALLY right on a concrete tile: if it is, a wrong result on the GPU can
only come from timing (a hardware hazard the instruction stream does not cover).  One wave, 64 lanes, LDS as a byte array.

    python tools/isa_emul.py <file.s> <first line> <last line> [--set v59=row ...]

Registers the region reads but does not define are supplied with --init (see main).  Not a general emulator.
"""
import re
import struct
import sys

import numpy as np

MASK64 = (1 << 64) - 1


class Wave:
    def __init__(self, lds_bytes=65536):
        self.v = np.zeros((512, 64), np.uint32)
        self.s = np.zeros(128, np.uint32)
        self.vcc = 0
        self.exec = MASK64
        self.lds = np.zeros(lds_bytes, np.uint8)
        self.undef_reads = set()
        self.vdef = np.zeros(512, bool)
        self.sdef = np.zeros(128, bool)
        self.lgkm = []  # in-order queue of outstanding LDS operations: sets of destination VGPRs (empty for stores)
        self.early_use = []  # (line text, register) read before the s_waitcnt that retires its ds_read

    # ---- helpers
    def lanes(self):
        return np.array([(self.exec >> i) & 1 for i in range(64)], bool)

    def get_s64(self, i):
        return int(self.s[i]) | (int(self.s[i + 1]) << 32)

    def set_s64(self, i, x):
        self.s[i] = x & 0xFFFFFFFF
        self.s[i + 1] = (x >> 32) & 0xFFFFFFFF
        self.sdef[i] = self.sdef[i + 1] = True

    def v64(self, i):
        return (self.v[i].astype(np.uint64) | (self.v[i + 1].astype(np.uint64) << np.uint64(32))).view(np.float64)

    def set_v64(self, i, x, m):
        u = np.asarray(x, np.float64).view(np.uint64)
        self.v[i][m] = (u & np.uint64(0xFFFFFFFF)).astype(np.uint32)[m]
        self.v[i + 1][m] = (u >> np.uint64(32)).astype(np.uint32)[m]
        self.vdef[i] = self.vdef[i + 1] = True

    def set_v32(self, i, x, m):
        self.v[i][m] = np.asarray(x, np.uint32)[m] if np.ndim(x) else np.uint32(x)
        self.vdef[i] = True


def parse_reg(tok):
    m = re.fullmatch(r"([vs])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3)) - int(m.group(2)) + 1
    m = re.fullmatch(r"([vs])(\d+)", tok)
    if m:
        return m.group(1), int(m.group(2)), 1
    return None


def src32(w, tok):
    """32-bit source operand as a [64] uint32 array"""
    r = parse_reg(tok)
    if r:
        k, i, _ = r
        if k == "v":
            if not w.vdef[i]:
                w.undef_reads.add(tok)
            return w.v[i].copy()
        if not w.sdef[i]:
            w.undef_reads.add(tok)
        return np.full(64, w.s[i], np.uint32)
    if tok.startswith("0x"):
        return np.full(64, int(tok, 16), np.uint32)
    return np.full(64, int(tok) & 0xFFFFFFFF, np.uint32)


def src64(w, tok):
    """64-bit floating source operand as [64] float64 (with optional neg modifier)"""
    neg = tok.startswith("-")
    if neg:
        tok = tok[1:]
    r = parse_reg(tok)
    if r:
        k, i, _ = r
        if k == "v":
            if not (w.vdef[i] and w.vdef[i + 1]):
                w.undef_reads.add(tok)
            x = w.v64(i).copy()
        else:
            if not (w.sdef[i] and w.sdef[i + 1]):
                w.undef_reads.add(tok)
            x = np.full(64, struct.unpack("<d", struct.pack("<Q", w.get_s64(i)))[0])
    else:
        x = np.full(64, float(tok))
    return -x if neg else x


def mask_of(w, tok):
    if tok == "vcc":
        return w.vcc
    if tok == "exec":
        return w.exec
    k, i, n = parse_reg(tok)
    assert k == "s" and n == 2
    return w.get_s64(i)


def set_mask(w, tok, x):
    if tok == "vcc":
        w.vcc = x & MASK64
    elif tok == "exec":
        w.exec = x & MASK64
    else:
        k, i, n = parse_reg(tok)
        w.set_s64(i, x & MASK64)


def bits(arr):
    x = 0
    for i, b in enumerate(arr):
        if b:
            x |= 1 << i
    return x


def run(lines, w, trace=False):
    """returns the index of the line where execution stopped (s_barrier / end)"""
    pc = 0
    labels = {ln.split(":")[0]: i for i, ln in enumerate(lines) if re.match(r"^\.?\w+:", ln)}
    while pc < len(lines):
        ln = lines[pc].split(";")[0].strip()
        pc += 1
        if not ln or ln.endswith(":") or ln.startswith("."):
            continue
        parts = ln.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        # ds offsets ride in the last argument
        mods = {}
        if args and (" offset" in args[-1] or args[-1].startswith("offset")):
            last = args[-1].split()
            args[-1] = last[0] if not last[0].startswith("offset") else ""
            for mtok in last[(0 if last[0].startswith("offset") else 1):]:
                k, val = mtok.split(":")
                mods[k] = int(val)
            if args[-1] == "":
                args.pop()
        m = w.lanes()
        # a VGPR that is the destination of a still-outstanding ds_read must not be touched
        if w.lgkm and not op.startswith("s_waitcnt"):
            pending = set().union(*w.lgkm)
            for tok in re.findall(r"v\[(\d+):(\d+)\]|v(\d+)", ln):
                regs = range(int(tok[0]), int(tok[1]) + 1) if tok[0] else [int(tok[2])]
                for rg in regs:
                    if rg in pending and not (op.startswith("ds_read") and ln.split()[1].startswith("v")):
                        w.early_use.append((ln, rg))
        if op == "s_waitcnt":
            mm = re.search(r"lgkmcnt\((\d+)\)", ln)
            if mm:
                keep = int(mm.group(1))
                w.lgkm = w.lgkm[len(w.lgkm) - keep:] if keep else []
            continue
        if op in ("s_nop", "scratch_store_dwordx2", "s_branch", "s_cmp_lt_i32"):
            if op == "s_branch":
                pc = labels[args[0]]
            continue
        if op == "s_barrier":
            return pc
        if op == "v_readlane_b32":
            k, d, _ = parse_reg(args[0])
            _, vs, _ = parse_reg(args[1])
            if not w.vdef[vs]:
                w.undef_reads.add(args[1])
            w.s[d] = w.v[vs][int(args[2])]
            w.sdef[d] = True
        elif op == "v_readfirstlane_b32":
            k, d, _ = parse_reg(args[0])
            first = next(i for i in range(64) if (w.exec >> i) & 1)
            w.s[d] = src32(w, args[1])[first]
            w.sdef[d] = True
        elif op == "v_writelane_b32":
            _, d, _ = parse_reg(args[0])
            w.v[d][int(args[2])] = src32(w, args[1])[0]
            w.vdef[d] = True
        elif op in ("v_fma_f64", "v_mul_f64", "v_fmac_f64_e32", "v_add_f64"):
            _, d, _ = parse_reg(args[0])
            if op == "v_fma_f64":
                a, b, c = (src64(w, t) for t in args[1:4])
                # fused multiply-add in extended precision, then one rounding
                r = (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(np.float64)
            elif op == "v_mul_f64":
                r = src64(w, args[1]) * src64(w, args[2])
            elif op == "v_add_f64":
                r = src64(w, args[1]) + src64(w, args[2])
            else:
                a, b = src64(w, args[1]), src64(w, args[2])
                r = (a.astype(np.longdouble) * b.astype(np.longdouble) + w.v64(d).astype(np.longdouble)).astype(np.float64)
            w.set_v64(d, r, m)
        elif op == "v_rsq_f64_e32":
            _, d, _ = parse_reg(args[0])
            x = src64(w, args[1])
            with np.errstate(all="ignore"):
                r = 1.0 / np.sqrt(x)
            # the hardware seed is good to about 2^-26: perturb so that the Newton step matters
            r = (r.view(np.uint64) & np.uint64(0xFFFFFFFFF8000000)).view(np.float64)
            w.set_v64(d, r, m)
        elif op == "v_mov_b32_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, src32(w, args[1]), m)
        elif op == "v_cndmask_b32_e64":
            _, d, _ = parse_reg(args[0])
            a, b = src32(w, args[1]), src32(w, args[2])
            mk = mask_of(w, args[3])
            sel = np.array([(mk >> i) & 1 for i in range(64)], bool)
            w.set_v32(d, np.where(sel, b, a), m)
        elif op in ("v_cmp_eq_u32_e64", "v_cmp_lt_u32_e64", "v_cmp_gt_u32_e64"):
            a, b = src32(w, args[1]), src32(w, args[2])
            r = {"eq": a == b, "lt": a < b, "gt": a > b}[op[6:8]]
            set_mask(w, args[0], bits(r & m))
        elif op == "v_cmp_gt_u32_e32":
            a, b = src32(w, args[1]), src32(w, args[2])
            w.vcc = bits((a > b) & m)
        elif op == "v_cmp_gt_f64_e64":
            a, b = src64(w, args[1]), src64(w, args[2])
            set_mask(w, args[0], bits((a > b) & m))
        elif op == "v_and_b32_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, src32(w, args[1]) & src32(w, args[2]), m)
        elif op == "v_mul_u32_u24_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, (src32(w, args[1]) & 0xFFFFFF) * (src32(w, args[2]) & 0xFFFFFF), m)
        elif op == "v_lshl_add_u32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, (src32(w, args[1]) << src32(w, args[2])) + src32(w, args[3]), m)
        elif op == "v_lshlrev_b32_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, src32(w, args[2]) << src32(w, args[1]), m)
        elif op == "v_sub_u32_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, src32(w, args[1]) - src32(w, args[2]), m)
        elif op == "v_add_u32_e32":
            _, d, _ = parse_reg(args[0])
            w.set_v32(d, src32(w, args[1]) + src32(w, args[2]), m)
        elif op == "s_mov_b32":
            _, d, _ = parse_reg(args[0])
            w.s[d] = src32(w, args[1])[0]
            w.sdef[d] = True
        elif op in ("s_and_b64", "s_or_b64", "s_andn2_b64"):
            a, b = mask_of(w, args[1]), mask_of(w, args[2])
            set_mask(w, args[0], {"s_and_b64": a & b, "s_or_b64": a | b, "s_andn2_b64": a & ~b}[op])
        elif op == "s_and_saveexec_b64":
            old = w.exec
            w.exec = old & mask_of(w, args[1])
            set_mask(w, args[0], old)
        elif op == "s_cbranch_execz":
            if w.exec == 0:
                pc = labels[args[0]]
        elif op == "s_cbranch_vccnz":
            if w.vcc != 0:
                pc = labels[args[0]]
        elif op == "ds_read2_b64":
            _, d, _ = parse_reg(args[0])
            addr = src32(w, args[1]).astype(np.int64)
            w.lgkm.append(set(range(d, d + 4)))
            for q, key in enumerate(("offset0", "offset1")):
                off = mods.get(key, 0) * 8
                val = np.array([w.lds[a + off:a + off + 8].view(np.float64)[0] for a in addr])
                w.set_v64(d + 2 * q, val, m)
        elif op == "ds_write_b64":
            addr = src32(w, args[0]).astype(np.int64)
            _, sreg, _ = parse_reg(args[1])
            val = w.v64(sreg)
            off = mods.get("offset", 0)
            w.lgkm.append(set())
            for lane in range(64):
                if m[lane]:
                    w.lds[addr[lane] + off:addr[lane] + off + 8] = np.array([val[lane]]).view(np.uint8)
        else:
            raise SystemExit(f"unhandled instruction at region line {pc}: {ln}")
    return pc


def main():
    path, first, last = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    inits = dict(a.split("=") for a in sys.argv[4:])
    lines = open(path).read().splitlines()[first - 1:last]
    TLD, NT = 17, 2
    ntiles = NT * (NT + 1) // 2
    rng = np.random.default_rng(3)
    A = rng.normal(size=(16, 16))
    S = A @ A.T / 16 + np.eye(16) * 0.3
    w = Wave()
    T = np.zeros((16, TLD))
    T[:, :16] = S
    w.lds[: 16 * TLD * 8] = T.reshape(-1).view(np.uint8)
    lane = np.arange(64, dtype=np.uint32)
    tid = lane + 64 * NT  # the diagonal wave of the NT = 2 instance
    allm = np.ones(64, bool)
    for reg, what in inits.items():
        k, i, n = parse_reg(reg)
        val = {"tid": tid, "lane": lane, "row": lane & 15, "zero": np.zeros(64, np.uint32), "one_hi": np.full(64, 0x3FF00000, np.uint32),
               "row17x8": (lane & 15) * 17 * 8}.get(what)
        if val is None:
            if what.startswith("ident"):  # ident<c>: the double (row == c) ? 1.0 : 0.0
                c = int(what[5:])
                w.set_v64(i, np.where((lane & 15) == c, 1.0, 0.0), allm)
                continue
            val = np.full(64, int(what, 0), np.uint32)
        if k == "v":
            w.set_v32(i, val, allm)
        else:
            w.s[i] = val[0]
            w.sdef[i] = True
    stop = run(lines, w)
    out = w.lds[: 16 * TLD * 8].view(np.float64).reshape(16, TLD)[:, :16]
    dinv_off = ntiles * 16 * TLD * 8
    inv = w.lds[dinv_off: dinv_off + 16 * TLD * 8].view(np.float64).reshape(16, TLD)[:, :16]
    Lref = np.linalg.cholesky(S)
    print("stopped at file line", first + stop - 1, " undefined reads:", sorted(w.undef_reads))
    print("registers used before the s_waitcnt that retires their ds_read:", w.early_use[:8] if w.early_use else "none")
    print("max |L - chol|      =", np.abs(out - Lref).max())
    print("max |Linv - L^-1|   =", np.abs(inv - np.linalg.inv(Lref)).max())
    print("max |Linv L - I|    =", np.abs(inv @ out - np.eye(16)).max())


if __name__ == "__main__":
    main()
