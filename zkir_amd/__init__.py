"""MI355X-native batched evaluator for zkIR circuits (see include/zkir_hip.h)."""
from .api import (Circuit, CircomTranspiler, NoirTranspiler, Context, Generator, GeneratorConfig, Lowerer, LowerOptions,  # noqa: F401
                  LowerRejected, Tape, TapeHdr, ZkirError, default_generator_config, default_lower_options,
                  generate_lower_many, pack_inputs, poseidon_style_gadget, rng_stream, unpack_outputs)
from .api import E_INVALID, E_LIMIT, E_LOWER  # noqa: F401
from .api import Reducer  # noqa: F401
