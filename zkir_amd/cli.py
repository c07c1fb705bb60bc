"""`cargo run` of the reference (src/main.rs:81-92) with a fixed seed: generate one circuit,
transpile it to Circom, write ./circuit.circom, print the instruction tree and the source.

    python -m zkir_amd.cli --seed 12345 [--out circuit.circom] [--noir circuit.nr]

The reference seeds from the timestamp counter (src/rng.rs:14-16); a seed makes the run
repeatable.  Output is byte for byte what the Rust binary prints for the same random stream.
"""
import argparse
import sys

from . import api


def render(seed):
    """(stdout text, circom source) of one run."""
    c = api.Generator(seed).generate()
    requested, generated = c.generation_counts()
    source = api.CircomTranspiler(c).transpile_circuit()
    out = "Generating %d instructions...\n" % requested  # src/generator.rs:148
    out += "Generated %d/%d instructions\n" % (generated, requested)  # :158
    out += c.debug() + "\n"  # println!("{:#?}", circuit.instructions), src/main.rs:91
    out += source + "\n"  # println!("{}", output), src/main.rs:92
    return out, source


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--seed", type=lambda s: int(s, 0), required=True)
    ap.add_argument("--out", default="./circuit.circom")  # src/main.rs:88
    ap.add_argument("--noir", default=None, help="also write the Noir text of the same circuit here "
                    "(fails if the circuit uses an operator Noir's Field type lacks)")
    args = ap.parse_args(argv)
    text, source = render(args.seed)
    with open(args.out, "wb") as f:
        f.write(source.encode())
    if args.noir:
        with open(args.noir, "wb") as f:
            f.write(api.NoirTranspiler(api.Generator(args.seed).generate()).transpile_circuit().encode())
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
