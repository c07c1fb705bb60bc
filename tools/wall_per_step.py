import sys, time, os
sys.path.insert(0, '/root/repo')
import torch, bench
import zkir_amd as z
tape = bench.workload_tape(z)
ctx = z.Context(0); tid = ctx.upload(tape)
h = tape.hdr; n = 1 << 20
dev = torch.device("cuda:0")
inp = bench.synthetic_inputs(torch, h.n_inputs, n, dev, 1)
out = torch.empty((h.n_outputs, 8, n), dtype=torch.int32, device=dev)
res = torch.zeros(4, dtype=torch.int64, device=dev)
s = torch.cuda.current_stream().cuda_stream
for _ in range(3): ctx.eval_batch_dev(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), s)
torch.cuda.synchronize()
for steps in (5, 20):
    t0 = time.perf_counter()
    for _ in range(steps): ctx.eval_batch_dev(tid, inp.data_ptr(), n, out.data_ptr(), res.data_ptr(), s)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(os.environ.get("ZKIR_NO_CHAIN"), steps, "launch calls %.3f ms, total %.3f ms per step" % ((t1 - t0) * 1e3 / steps, (t2 - t0) * 1e3 / steps))
