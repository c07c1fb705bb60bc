"""The 29-bit-limb building blocks (tools/fr29.py: Montgomery product with R = 2^261, reduction alone, limb-wise
add / subtract, carry steps, 32-bit packing) executed instruction by instruction in the gfx950 model
against big integers.  No GPU needed."""
import os
import random
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fr29  # noqa: E402
import isa_model as M  # noqa: E402

P = fr29.P
A = ["v%d" % (10 + j) for j in range(9)]
B = ["v%d" % (20 + j) for j in range(9)]
Mr = ["v%d" % (30 + j) for j in range(9)]
SP = ["s%d" % (40 + j) for j in range(9)]
SNP, SINK, ACC = "s50", "s[52:53]", 40


def lane(regs):
    ln = M.Lane(M.Memory(), 0, 5, {}, set(range(0, 64)), set(range(0, 64)))
    for j in range(9):
        ln.s[SP[j]] = fr29.P_LIMBS[j]
    ln.s[SNP] = fr29.NP
    ln.s["s52"] = ln.s["s53"] = 0
    for name, limbs in regs.items():
        for r, v in zip(name, limbs):
            ln.v[r] = v
    return ln


def run(lines, regs):
    ln = lane(regs)
    prog, labels = M.parse_program(lines)
    ln.run(prog, labels)
    return ln


def rand_limbs(rnd, limb_max, top_max):
    return [rnd.randrange(limb_max) for _ in range(8)] + [rnd.randrange(top_max)]


def test_constants():
    assert fr29.NP == 0x0FFFFFFF and (fr29.P * fr29.NP + 1) % (1 << 29) == 0
    assert fr29.value29(fr29.P_LIMBS) == P and fr29.RBITS == 261 and (1 << 261) > 128 * P


def test_product_matches_big_integers():
    out = []
    fr29.mul29(out, A, B, Mr, Mr, ACC, SP, SNP, SINK)
    assert sum(1 for l in out if l.startswith("v_mad_u64_u32")) == 162
    assert sum(1 for l in out if l.startswith(("v_mad", "v_mul", "v_lshrrev_b64"))) == 188 and len(out) == 206
    rinv = pow(1 << 261, -1, P)
    rnd = random.Random(29)
    edge = [0, 1, P - 1, P, 2 * P - 1, fr29.R, fr29.R2, (1 << 254) - 1, 13 * P - 1]
    cases = [(fr29.limbs29(x), fr29.limbs29(y)) for x in edge for y in edge]
    cases += [(fr29.limbs29(rnd.randrange(2 * P)), fr29.limbs29(rnd.randrange(2 * P))) for _ in range(200)]
    # limbs beyond 2^29 (what lazy additions leave): 9 * 2^60 + 9 * 2^58 + a carry stays inside 64 bits
    cases += [(rand_limbs(rnd, 1 << 30, 1 << 24), rand_limbs(rnd, 1 << 30, 1 << 24)) for _ in range(100)]
    cases += [(rand_limbs(rnd, 1 << 31, 1 << 24), rand_limbs(rnd, 1 << 29, 1 << 24)) for _ in range(100)]
    cases += [([(1 << 30) - 1] * 8 + [(1 << 24) - 1], [(1 << 30) - 1] * 8 + [(1 << 24) - 1]),
              ([(1 << 31) - 1] * 8 + [(1 << 24) - 1], [(1 << 29) - 1] * 8 + [(1 << 24) - 1])]
    for al, bl in cases:
        ln = run(out, {tuple(A): al, tuple(B): bl})
        r = [ln.v[x] for x in Mr]
        a, b = fr29.value29(al), fr29.value29(bl)
        got = fr29.value29(r)
        assert got % P == (a * b * rinv) % P
        assert all(v < (1 << 29) for v in r[:8])
        assert got * (1 << 261) < a * b + (1 << 261) * P  # r < a*b/R + p
        if a < 13 * P and b < 13 * P:
            assert got < 2 * P


def test_reduction_alone():
    out = []
    fr29.mul29(out, A, None, Mr, Mr, ACC, SP, SNP, SINK, redc_only=True)
    rinv = pow(1 << 261, -1, P)
    rnd = random.Random(30)
    for _ in range(200):
        al = rand_limbs(rnd, 1 << 32, 1 << 32)
        ln = run(out, {tuple(A): al})
        r = [ln.v[x] for x in Mr]
        got, a = fr29.value29(r), fr29.value29(al)
        assert got % P == (a * rinv) % P and got * (1 << 261) < a + (1 << 261) * P and all(v < (1 << 29) for v in r[:8])


def test_lazy_add_sub_and_carry_steps():
    rnd = random.Random(31)
    T = "v60"
    for _ in range(200):
        al, bl = rand_limbs(rnd, (1 << 29) + 8, 1 << 25), rand_limbs(rnd, (1 << 29) + 8, 1 << 25)
        a, b = fr29.value29(al), fr29.value29(bl)
        # a multiple of p whose limbs dominate b's
        c = dominating_multiple((1 << 29) + 8, 1 << 25)
        out = []
        fr29.add29(out, Mr, A, B)
        ln = run(out, {tuple(A): al, tuple(B): bl})
        assert fr29.value29([ln.v[x] for x in Mr]) == a + b
        out = []
        fr29.sub29(out, Mr, A, B, c)
        fr29.softnorm29(out, Mr, T)
        ln = run(out, {tuple(A): al, tuple(B): bl})
        r = [ln.v[x] for x in Mr]
        assert fr29.value29(r) == a - b + fr29.value29(c) and all(v < (1 << 29) + 8 for v in r[:8])
        out = []
        xl = rand_limbs(rnd, 1 << 32, 1 << 28)
        fr29.fullnorm29(out, A, T)
        ln = run(out, {tuple(A): xl})
        r = [ln.v[x] for x in A]
        assert fr29.value29(r) == fr29.value29(xl) and all(v < (1 << 29) for v in r[:8])


def dominating_multiple(limb_bound, top_bound):
    """limbs of k*p, every limb 0..7 at least limb_bound and limb 8 at least top_bound (borrowing from above)."""
    k = 1
    while True:
        n = fr29.limbs29(k * P)
        lend = 1
        while (lend << 29) < limb_bound + (1 << 29):  # units lent from limb j+1 to limb j
            lend += 1
        c = [n[0] + (lend << 29)] + [n[j] + (lend << 29) - lend for j in range(1, 8)] + [n[8] - lend]
        if c[8] >= top_bound and all(v >= limb_bound for v in c[:8]):
            assert fr29.value29(c) == k * P and all(v < (1 << 32) for v in c)
            return c
        k += 1


def test_pack_and_unpack():
    rnd = random.Random(32)
    W = ["v%d" % (50 + j) for j in range(8)]
    for _ in range(200):
        x = rnd.randrange(1 << 256)
        out = []
        fr29.unpack29(out, A, W)
        fr29.pack29(out, B[:8], A, "v60")
        ln = run(out, {tuple(W): [(x >> (32 * j)) & 0xFFFFFFFF for j in range(8)]})
        assert fr29.value29([ln.v[r] for r in A]) == x and all(ln.v[r] < (1 << 29) for r in A[:8])
        assert sum(ln.v[r] << (32 * j) for j, r in enumerate(B[:8])) == x
