#!/bin/sh
# needs circom (>= 2.0) and snarkjs on PATH; run inside this directory
circom circuit.circom --r1cs --wasm --sym > circom.log 2>&1 || { echo "circom refused the circuit (see circom.log)"; exit 0; }
set -e
for f in input_*.json; do
  k=${f#input_}; k=${k%.json}
  node circuit_js/generate_witness.js circuit_js/circuit.wasm "$f" "witness_$k.wtns"
  snarkjs wtns check circuit.r1cs "witness_$k.wtns"
done
