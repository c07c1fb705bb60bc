#!/usr/bin/env python3
"""Device-slot reads and writes per input vector of a tape, as the generated-stream kernels perform them: a model of
the operand registers' tags and the victim row (tools/gen_interp_asm.py: fetch, store) applied to the tape's ops.
Usage: python tools/slot_traffic.py [--signal-window W] [--linear-terms T] [--no-victim]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

K_GSLOT, K_LSLOT, K_CONST, K_INPUT, K_ACC, K_NONE, K_XSLOT = range(7)
OP_INPUT, OP_OUTPUT, OP_COPY, OP_NEG, OP_BARRIER, OP_CHECK = 1, 2, 3, 4, 7, 48


def slot_traffic(ops, victim=True, one_word=-1):
    """(device-slot reads, device-slot writes, LDS window reads) for one vector.
    one_word: operand word of the tape's constant 1 (a CHECK against it fetches no b)."""
    tag = {"a": -1, "b": -1, "v": -1}
    reads = writes = lds = 0

    def fetch(word, reg):
        nonlocal reads, lds
        kind = word >> 29
        if reg is not None and word == tag[reg]:
            return
        if reg is not None and word == tag["v"]:
            tag[reg] = word
            lds += 1
            return
        if kind == K_ACC:
            if reg == "b":
                tag[reg] = -1
            return
        if reg is not None and victim and 0 <= tag[reg] < (1 << 29):
            tag["v"] = tag[reg]
        if kind == K_GSLOT:
            reads += 1
        elif kind == K_LSLOT:
            lds += 1
        if reg is not None:
            tag[reg] = word if kind in (K_GSLOT, K_LSLOT, K_CONST) else -1

    for code, dst, a, b in ops:
        code &= 0xFF
        if code in (0, OP_BARRIER):
            continue
        if code == OP_INPUT:
            tag["a"] = tag["b"] = -1
        elif code == OP_OUTPUT:
            fetch(a, "a")
            continue
        elif code == OP_COPY:
            fetch(a, None)
        elif code == OP_NEG:
            fetch(a, "b")
            tag["a"] = -1
        elif code == OP_CHECK:
            if b != one_word:
                fetch(b, "b")
            fetch(a, "a")  # the product goes to scratch registers: a and b keep their values
            fetch(dst, "b")
            continue
        else:
            fetch(b, "b")
            fetch(a, "a")
        for r in ("a", "b", "v"):
            if tag[r] == dst:
                tag[r] = -1
        if dst >> 29 == K_GSLOT:
            writes += 1
    return reads, writes, lds


if __name__ == "__main__":
    import argparse
    import bench
    import zkir_amd as z
    ap = argparse.ArgumentParser()
    ap.add_argument("--signal-window", type=int, default=0)
    ap.add_argument("--linear-terms", type=int, default=1)
    ap.add_argument("--no-victim", action="store_true")
    args = ap.parse_args()
    t = bench.workload_tape(z, args.signal_window, args.linear_terms)
    h = t.hdr
    one = next((i for i, c in enumerate(t.consts()) if c == 1), None)
    r, w, l = slot_traffic(t.ops(), victim=not args.no_victim and h.n_lslots + 1 <= 4,
                           one_word=-1 if one is None else (K_CONST << 29) | one)
    per_vec = 32 * (r + w + h.n_inputs + h.n_outputs)
    print("ops %d, device slots %d, window slots %d: %d device-slot reads, %d writes, %d LDS reads per vector; "
          "%.1f KB per vector with inputs and outputs (%.1f GB per 2^20 vectors)"
          % (h.n_ops, h.n_gslots, h.n_lslots, r, w, l, per_vec / 1e3, per_vec * 2**20 / 1e9))
