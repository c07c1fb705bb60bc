#include "../../include/zkir_hip.h"
#include <cstdio>
#include <cstdlib>
int main(){
  zkir_lower_options o; zkir_lower_options_default(&o); o.strands=4;
  size_t strand_tapes=0, total=0;
  for (int w=1; w<=6; w++) for (int r=1; r<=40; r+=13) {
    zkir_circuit* c=nullptr; if (zkir_gadget_poseidon_style(w, r, 7+w, &c)) return 1;
    zkir_tape* t=nullptr; int32_t cls=0; if (zkir_lower(c,&o,&t,&cls)) { printf("lower fail\n"); return 1; }
    zkir_tape_hdr h; zkir_tape_header(t,&h); strand_tapes += h.flags&1; total++;
    const void* blob; size_t bytes; zkir_tape_blob(t,&blob,&bytes);
    zkir_tape* t2=nullptr; if (zkir_tape_from_blob(&h, blob, bytes, &t2)) { printf("validation fail: %s\n", zkir_host_last_error()); return 1; }
    zkir_tape_free(t2); zkir_tape_free(t); zkir_circuit_free(c);
  }
  for (int seed=1; seed<=120; seed++) {
    zkir_generator_config g; zkir_generator_config_default(&g);
    g.min_instructions=g.max_instructions=10+seed*3; g.forced_inputs=4; g.inputs_readable=1; g.arithmetic_only=1; g.quadratic_aware=1; g.no_loops=seed%2; g.safe_array_index=1;
    zkir_circuit* c=nullptr; if (zkir_generate(seed,&g,&c)) return 1;
    zkir_tape* t=nullptr; int32_t cls=0; int rc=zkir_lower(c,&o,&t,&cls);
    if (rc==0) { zkir_tape_hdr h; zkir_tape_header(t,&h); strand_tapes += h.flags&1; total++;
      const void* blob; size_t bytes; zkir_tape_blob(t,&blob,&bytes);
      zkir_tape* t2=nullptr; if (zkir_tape_from_blob(&h, blob, bytes, &t2)) { printf("validation fail: %s\n", zkir_host_last_error()); return 1; }
      zkir_tape_free(t2); zkir_tape_free(t); }
    zkir_circuit_free(c);
  }
  printf("lowered %zu circuits, %zu as strand tapes, all validate\n", total, strand_tapes);
}
