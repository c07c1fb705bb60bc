"""A small executable model of the gfx950 instructions the generated interpreter stream uses,
for ONE lane of a wavefront (scalar registers are shared by construction, so one lane at a
time sees everything the stream computes for it).

It is strict where the hardware is forgiving: a read of a register whose load has not been
waited for, a buffer access beyond its descriptor's byte count, an access outside every
mapped region, a write to an operand of the asm statement before the results are due, or a
write to a register outside the clobber list all raise.  That is what makes a green run
here evidence that the stream stays inside its buffers on the device.
"""
import re

M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF


class ModelError(AssertionError):
    pass


class Region:
    def __init__(self, name, base, size, writable):
        self.name, self.base, self.size, self.writable = name, base, size, writable
        self.words = {}


class Memory:
    def __init__(self):
        self.regions = []

    def map(self, name, base, size, writable=False, words=None):
        r = Region(name, base, size, writable)
        if words:
            for i, w in enumerate(words):
                r.words[base + 4 * i] = w & M32
        self.regions.append(r)
        return r

    def find(self, addr, nbytes, write):
        for r in self.regions:
            if r.base <= addr and addr + nbytes <= r.base + r.size:
                if write and not r.writable:
                    raise ModelError("store into read-only region %s at %#x" % (r.name, addr))
                return r
        raise ModelError("%s of %d bytes at %#x is outside every mapped region" % ("store" if write else "load", nbytes, addr))

    def load(self, addr, ndw):
        r = self.find(addr, 4 * ndw, False)
        out = []
        for i in range(ndw):
            a = addr + 4 * i
            if a not in r.words:
                raise ModelError("load of never-written word at %#x (%s)" % (a, r.name))
            out.append(r.words[a])
        return out

    def store(self, addr, words):
        r = self.find(addr, 4 * len(words), True)
        for i, w in enumerate(words):
            r.words[addr + 4 * i] = w & M32


def parse_program(lines):
    """-> (instruction list [(mnemonic, [operands], modifiers)], label -> index)."""
    prog, labels = [], {}
    for raw in lines:
        s = raw.strip()
        if not s:
            continue
        if s.endswith(":"):
            labels[s[:-1]] = len(prog)
            continue
        op, _, rest = s.partition(" ")
        ops = [x.strip() for x in rest.split(",")] if rest else []
        mods = []
        if ops:  # trailing modifiers ride on the last operand
            parts = ops[-1].split()
            ops[-1], mods = parts[0], parts[1:]
        prog.append((op, ops, mods))
    return prog, labels


class Lane:
    """One lane (`lane` of 64) of one wavefront running a parsed program."""

    def __init__(self, mem, lds_bytes, lane, operands, clobber_v, clobber_s, result_operands=()):
        self.mem, self.lds, self.lds_bytes, self.lane = mem, {}, lds_bytes, lane
        self.bit = 1 << lane
        self.operands = dict(operands)          # "%[name]" -> value (32- or 64-bit int)
        self.operand_written = {}
        self.result_operands = set(result_operands)
        self.clobber_v, self.clobber_s = clobber_v, clobber_s
        self.v, self.s = {}, {}
        self.vcc, self.scc, self.exec = 0, 0, M64
        self.pending = {}  # register -> "vm" | "lgkm": a load not yet waited for
        self.stores_waited = True  # no store (LDS or device memory) issued since the last full wait
        self.steps = 0
        self.prio, self.prio_log, self.clock = 0, [], 0  # s_setprio history; the value s_memrealtime returns
        self.counts = {}
        self.wrap_ok = set()  # destination registers whose v_add_u32 may wrap (signed limb arithmetic), by name

    # ---- register access ---------------------------------------------------
    def _check_ready(self, name):
        if name in self.pending:
            raise ModelError("%s read before the wait for its %s load" % (name, self.pending[name]))

    def sread(self, x):
        """32-bit scalar source: sN, operand, vcc_lo, literal."""
        if x.startswith("%["):
            v = self.operands[x]
            if v > M32:
                raise ModelError("64-bit operand %s read as 32 bits" % x)
            return v
        if re.fullmatch(r"s\d+", x):
            self._check_ready(x)
            if x not in self.s:
                raise ModelError("read of unwritten " + x)
            return self.s[x]
        return self.imm(x)

    def sread64(self, x):
        if x.startswith("%["):
            return self.operands[x]
        if x == "vcc":
            return self.vcc
        if x == "exec":
            return self.exec
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", x)
        if m:
            lo, hi = int(m.group(1)), int(m.group(2))
            if hi != lo + 1 or lo % 2:
                raise ModelError("misaligned scalar pair " + x)
            return self.sread("s%d" % lo) | (self.sread("s%d" % hi) << 32)
        if re.fullmatch(r"-\d+", x):
            return int(x) & M64  # inline constants are sign-extended
        return self.imm(x)

    def swrite(self, x, v):
        if x.startswith("%["):
            if x not in self.result_operands:
                raise ModelError("write to input operand " + x)
            self.operand_written[x] = v
            return
        if not re.fullmatch(r"s\d+", x):
            raise ModelError("bad scalar destination " + x)
        if int(x[1:]) not in self.clobber_s:
            raise ModelError(x + " written but not in the clobber list")
        if x in self.pending:
            raise ModelError(x + " overwritten while its load is in flight")
        self.s[x] = v & M32

    def swrite64(self, x, v):
        if x == "vcc":
            self.vcc = v & M64
            return
        if x == "exec":
            self.exec = v & M64
            return
        if x.startswith("%["):
            self.swrite(x, v)
            return
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", x)
        lo, hi = int(m.group(1)), int(m.group(2))
        if hi != lo + 1 or lo % 2:
            raise ModelError("misaligned scalar pair " + x)
        self.swrite("s%d" % lo, v & M32)
        self.swrite("s%d" % hi, (v >> 32) & M32)

    def imm(self, x):
        if re.fullmatch(r"-?\d+", x):
            return int(x) & M32
        if re.fullmatch(r"0x[0-9a-fA-F]+", x):
            return int(x, 16)
        raise ModelError("cannot read operand " + x)

    def vread(self, x):
        """32-bit vector-ALU source: vN, sN, operand, literal."""
        if re.fullmatch(r"v\d+", x):
            self._check_ready(x)
            if x not in self.v:
                raise ModelError("read of unwritten " + x)
            return self.v[x]
        if x.startswith("%["):
            return self.operands[x] & M32 if self.operands[x] <= M32 else self._bad64(x)
        return self.sread(x)

    def _bad64(self, x):
        raise ModelError("64-bit operand %s read as 32 bits" % x)

    def vwrite(self, x, v):
        if x.startswith("%["):
            if x not in self.result_operands:
                raise ModelError("write to input operand " + x)
            self.operand_written[x] = v & M32
            return
        if not re.fullmatch(r"v\d+", x):
            raise ModelError("bad vector destination " + x)
        if int(x[1:]) not in self.clobber_v:
            raise ModelError(x + " written but not in the clobber list")
        if x in self.pending:
            raise ModelError(x + " overwritten while its load is in flight")
        if self.exec & self.bit:
            self.v[x] = v & M32

    def vrange(self, x, n, align=2):
        m = re.fullmatch(r"v\[(\d+):(\d+)\]", x)
        if not m:
            if n == 1 and re.fullmatch(r"v\d+", x):
                return [x]
            raise ModelError("bad vector range " + x)
        lo, hi = int(m.group(1)), int(m.group(2))
        if hi - lo + 1 != n or lo % align:
            raise ModelError("vector range %s: want %d registers, aligned to %d" % (x, n, align))
        return ["v%d" % k for k in range(lo, hi + 1)]

    def srange(self, x, n):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", x)
        lo, hi = int(m.group(1)), int(m.group(2))
        if hi - lo + 1 != n or lo % min(n, 4):
            raise ModelError("scalar range %s: want %d registers, aligned" % (x, n))
        return ["s%d" % k for k in range(lo, hi + 1)]

    def carry_read(self, x):
        return 1 if (self.sread64(x) & self.bit) else 0

    def carry_write(self, x, c):
        cur = self.sread64(x) if (x == "vcc" or self._pair_written(x)) else 0
        self.swrite64(x, (cur & ~self.bit) | (self.bit if c else 0))

    def _pair_written(self, x):
        m = re.fullmatch(r"s\[(\d+):(\d+)\]", x)
        return m and ("s" + m.group(1)) in self.s and ("s" + m.group(2)) in self.s

    # ---- buffers -------------------------------------------------------------
    def descriptor(self, x):
        w = [self.sread(r) for r in self.srange(x, 4)]
        if w[1] >> 16:
            raise ModelError("descriptor %s: stride/swizzle bits set (%#x)" % (x, w[1]))
        if w[3] != 0x00020000:
            raise ModelError("descriptor %s: word 3 is %#x" % (x, w[3]))
        return w[0] | (w[1] << 32), w[2]

    def buffer_addr(self, rsrc, voffset, soffset, inst_off, nbytes):
        base, records = self.descriptor(rsrc)
        off = voffset + inst_off
        if off + nbytes > records or off + soffset + nbytes > records:
            raise ModelError("buffer access at offset %d+%d (+%d bytes) beyond the %d-byte descriptor %s"
                             % (off, soffset, nbytes, records, rsrc))
        return base + soffset + off

    # ---- execution -------------------------------------------------------------
    def run(self, prog, labels, max_steps=50_000_000, pc=0, stop_at_barrier=False):
        """Run from `pc` to the end of the program.  With stop_at_barrier the run also ends after
        an s_barrier and returns the index of the next instruction (None at the end of the
        program): the caller interleaves the wavefronts of a workgroup that way."""
        while pc < len(prog):
            op, a, mods = prog[pc]
            self.steps += 1
            if self.steps > max_steps:
                raise ModelError("step limit reached: the stream does not terminate")
            self.counts[op] = self.counts.get(op, 0) + 1
            pc = self.step(op, a, mods, pc, labels)
            if op == "s_barrier" and stop_at_barrier:
                return pc
        if self.pending:
            raise ModelError("loads still in flight at the end: %s" % sorted(self.pending))
        return None

    def step(self, op, a, mods, pc, labels):
        nxt = pc + 1
        inst_off = 0
        for m in mods:
            if m.startswith("offset:"):
                inst_off = int(m[7:])
        # ---- scalar ALU
        if op == "s_mov_b32":
            self.swrite(a[0], self.sread(a[1]))
        elif op == "s_mov_b64":
            self.swrite64(a[0], self.sread64(a[1]))
        elif op in ("s_and_b32", "s_lshr_b32", "s_lshl_b32", "s_add_u32", "s_addc_u32", "s_sub_u32", "s_mul_i32",
                    "s_mul_hi_u32"):
            x, y = self.sread(a[1]), self.sread(a[2])
            if op == "s_and_b32":
                r = x & y
                self.scc = int(r != 0)
            elif op == "s_lshr_b32":
                r = x >> (y & 31)
                self.scc = int(r != 0)
            elif op == "s_lshl_b32":
                full = x << (y & 31)
                if full > M32:
                    raise ModelError("s_lshl_b32 shifted bits out (%#x << %d)" % (x, y))
                r = full
                self.scc = int(r != 0)
            elif op == "s_add_u32":
                t = x + y
                r, self.scc = t & M32, t >> 32
            elif op == "s_addc_u32":
                t = x + y + self.scc
                r, self.scc = t & M32, t >> 32
            elif op == "s_sub_u32":
                t = x - y
                r, self.scc = t & M32, int(t < 0)
            elif op == "s_mul_i32":
                r = (x * y) & M32
            else:
                r = (x * y) >> 32
            self.swrite(a[0], r)
        elif op == "s_cselect_b32":
            self.swrite(a[0], self.sread(a[1]) if self.scc else self.sread(a[2]))
        elif op == "s_and_b64":
            r = self.sread64(a[1]) & self.sread64(a[2])
            self.swrite64(a[0], r)
            self.scc = int(r != 0)
        elif op == "s_bcnt1_i32_b64":
            r = bin(self.sread64(a[1])).count("1")
            self.swrite(a[0], r)
            self.scc = int(r != 0)
        elif op == "s_cmp_eq_u64":
            self.scc = int(self.sread64(a[0]) == self.sread64(a[1]))
        elif op in ("s_cmp_eq_u32", "s_cmp_lg_u32", "s_cmp_lt_u32"):
            x, y = self.sread(a[0]), self.sread(a[1])
            self.scc = int({"s_cmp_eq_u32": x == y, "s_cmp_lg_u32": x != y, "s_cmp_lt_u32": x < y}[op])
        elif op == "s_branch":
            nxt = labels[a[0]]
        elif op == "s_cbranch_scc1":
            if self.scc:
                nxt = labels[a[0]]
        elif op == "s_cbranch_scc0":
            if not self.scc:
                nxt = labels[a[0]]
        elif op == "s_nop":
            pass
        elif op == "s_setprio":
            v = self.imm(a[0])
            if v > 3:
                raise ModelError("s_setprio %d: priorities are 0..3" % v)
            self.prio = v
            self.prio_log.append(v)
        elif op == "s_memrealtime":
            # the constant-rate clock: the test sets self.clock; the value arrives like a scalar load
            lo, hi = self.srange(a[0], 2)
            self.swrite(lo, self.clock & M32)
            self.swrite(hi, self.clock >> 32)
            self.pending[lo] = self.pending[hi] = "lgkm"
        elif op == "s_barrier":
            # what another wavefront is to see must have left this one: no memory operation in flight
            if self.pending:
                raise ModelError("s_barrier with loads in flight: %s" % sorted(self.pending))
            if not self.stores_waited:
                raise ModelError("s_barrier with stores not waited for")
        elif op == "s_waitcnt":
            kinds = []
            for tok in a + mods:
                m = re.fullmatch(r"(vmcnt|lgkmcnt)\(0\)", tok)
                if not m:
                    raise ModelError("only zero-count waits are modelled: " + tok)
                kinds.append("vm" if m.group(1) == "vmcnt" else "lgkm")
            self.pending = {r: k for r, k in self.pending.items() if k not in kinds}
            if "vm" in kinds and "lgkm" in kinds:
                self.stores_waited = True
        # ---- scalar memory
        elif op in ("s_buffer_load_dword", "s_buffer_load_dwordx4", "s_buffer_load_dwordx8"):
            n = 4 if op.endswith("x4") else 8 if op.endswith("x8") else 1
            base, records = self.descriptor(a[1])
            off = self.sread(a[2])
            if off % 4 or off + 4 * n > records:
                raise ModelError("scalar buffer load at offset %d (+%d bytes) beyond %d" % (off, 4 * n, records))
            words = self.mem.load(base + off, n)
            for r, w in zip(self.srange(a[0], n) if n > 1 else [a[0]], words):
                self.swrite(r, w)
                self.pending[r] = "lgkm"
        # ---- vector ALU
        elif op == "v_mov_b32":
            self.vwrite(a[0], self.vread(a[1]))
        elif op in ("v_add_u32", "v_xor_b32", "v_or_b32", "v_min_u32", "v_mul_lo_u32"):
            x, y = self.vread(a[1]), self.vread(a[2])
            r = {"v_add_u32": x + y, "v_xor_b32": x ^ y, "v_or_b32": x | y, "v_min_u32": min(x, y),
                 "v_mul_lo_u32": x * y}[op]
            if op == "v_add_u32" and r > M32 and a[0] not in self.wrap_ok:
                raise ModelError("v_add_u32 wrapped: an address computation overflowed")
            self.vwrite(a[0], r)
        elif op in ("v_and_b32", "v_sub_u32", "v_mul_hi_u32"):
            x, y = self.vread(a[1]), self.vread(a[2])
            if op == "v_sub_u32" and x < y:
                raise ModelError("v_sub_u32 went negative (%#x - %#x)" % (x, y))
            self.vwrite(a[0], {"v_and_b32": x & y, "v_sub_u32": x - y, "v_mul_hi_u32": (x * y) >> 32}[op])
        elif op == "v_subrev_u32":  # src1 - src0, wrapping: signed limb arithmetic
            self.vwrite(a[0], (self.vread(a[2]) - self.vread(a[1])) & M32)
        elif op == "v_mad_u32_u24":
            x, y = self.vread(a[1]), self.vread(a[2])
            if x >> 24 or y >> 24:
                raise ModelError("v_mad_u32_u24 operand beyond 24 bits")
            r = x * y + self.vread(a[3])
            if r > M32:
                raise ModelError("v_mad_u32_u24 wrapped")
            self.vwrite(a[0], r)
        elif op in ("v_lshrrev_b32", "v_ashrrev_i32"):
            sh, x = self.vread(a[1]) & 31, self.vread(a[2])
            if op == "v_ashrrev_i32" and x >> 31:
                x -= 1 << 32
            self.vwrite(a[0], (x >> sh) & M32)
        elif op == "v_lshlrev_b32":
            self.vwrite(a[0], (self.vread(a[2]) << (self.vread(a[1]) & 31)) & M32)
        elif op == "v_lshrrev_b64":
            lo, hi = self.vrange(a[2], 2)
            v = (self.vread(lo) | (self.vread(hi) << 32)) >> (self.vread(a[1]) & 63)
            lo, hi = self.vrange(a[0], 2)
            self.vwrite(lo, v & M32)
            self.vwrite(hi, v >> 32)
        elif op == "v_alignbit_b32":  # low 32 bits of {hi, lo} >> shift
            v = ((self.vread(a[1]) << 32) | self.vread(a[2])) >> (self.vread(a[3]) & 31)
            self.vwrite(a[0], v & M32)
        elif op == "v_lshl_or_b32":
            self.vwrite(a[0], ((self.vread(a[1]) << (self.vread(a[2]) & 31)) | self.vread(a[3])) & M32)
        elif op == "v_mad_u64_u32":
            add = 0
            if a[4] != "0":
                lo, hi = self.vrange(a[4], 2)
                add = self.vread(lo) | (self.vread(hi) << 32)
            r = self.vread(a[2]) * self.vread(a[3]) + add
            if r > M64:
                raise ModelError("multiply-add overflowed 64 bits")
            lo, hi = self.vrange(a[0], 2)
            self.vwrite(lo, r & M32)
            self.vwrite(hi, r >> 32)
            self.carry_write(a[1], 0)
        elif op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32"):
            cin = self.carry_read(a[4]) if op in ("v_addc_co_u32", "v_subb_co_u32") else 0
            x, y = self.vread(a[2]), self.vread(a[3])
            if op.startswith("v_add"):
                t = x + y + cin
                c = t >> 32
            else:
                t = x - y - cin
                c = int(t < 0)
            self.vwrite(a[0], t)
            self.carry_write(a[1], c)
        elif op in ("v_cndmask_b32", "v_cndmask_b32_e64"):
            sel = self.carry_read(a[3])
            self.vwrite(a[0], self.vread(a[2]) if sel else self.vread(a[1]))
        elif op == "v_cmp_ne_u32":
            self.carry_write(a[0], int(self.vread(a[1]) != self.vread(a[2])))
        # ---- LDS
        elif op in ("ds_read_b128", "ds_write_b128", "ds_read_b32", "ds_write_b32"):
            rd = op.startswith("ds_read")
            nb = 16 if op.endswith("b128") else 4
            addr = self.vread(a[1] if rd else a[0]) + inst_off
            if addr % nb or addr + nb > self.lds_bytes:
                raise ModelError("LDS access at %d (+%d) outside the %d-byte allocation" % (addr, nb, self.lds_bytes))
            regs = self.vrange(a[0] if rd else a[1], nb // 4)
            for k, r in enumerate(regs):
                if rd:
                    key = addr + 4 * k
                    if key not in self.lds:
                        raise ModelError("LDS read of never-written word %d" % key)
                    self.vwrite(r, self.lds[key])
                    self.pending[r] = "lgkm"
                else:
                    self.lds[addr + 4 * k] = self.vread(r)
                    self.stores_waited = False
        # ---- vector memory
        elif op in ("buffer_load_dwordx4", "buffer_load_dword", "buffer_store_dwordx4", "buffer_store_dword"):
            n = 4 if op.endswith("x4") else 1
            if "offen" not in mods:
                raise ModelError("buffer access without offen")
            regs = self.vrange(a[0], n)
            soff = self.sread(a[3])
            if not (self.exec & self.bit):
                return nxt
            addr = self.buffer_addr(a[2], self.vread(a[1]), soff, inst_off, 4 * n)
            if addr % (4 * n):
                raise ModelError("misaligned buffer access at %#x" % addr)
            if "load" in op:
                for r, w in zip(regs, self.mem.load(addr, n)):
                    self.vwrite(r, w)
                    self.pending[r] = "vm"
            else:
                self.mem.store(addr, [self.vread(r) for r in regs])
                self.stores_waited = False
        else:
            raise ModelError("instruction not modelled: %s %s" % (op, ", ".join(a)))
        return nxt
