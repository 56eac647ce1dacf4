#!/usr/bin/env python3
"""profiles/<tag>_ddp_rehearsal.md from gpurun_out/ddp1/summary.txt (tools/run_ddp1.sh | tee), gpurun_out/ddptl_final/{queues,step}.txt
(tools/run_ddp_trace.sh final + tools/queue_map.py):   python tools/write_ddp_rehearsal.py r02"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
summ = open(f"{R}/gpurun_out/ddp1/summary.txt").read().strip()
queues = open(f"{R}/gpurun_out/ddptl_final/queues.txt").read().strip()
step = open(f"{R}/gpurun_out/ddptl_final/step.txt").read().strip().splitlines()
big = [l for l in step if l.startswith("step") or (len(l.split()) > 3 and l.split()[3].replace(".", "").isdigit() and float(l.split()[3]) >= 150)]
nl = "\n"
doc = f"""# Round {int(tag[1:3])} — one-GPU rehearsal of the data-parallel step (MI355X, bf16, batch 16, preset-1 small)

No multi-GPU node is reachable from the build; this is what can be measured on one card.  `SGV_FORCE_COLLECTIVE=1` makes a one-rank
group issue every collective it would issue with more ranks (pack -> `ncclAllReduce(ncclAvg)` -> update); RCCL then runs its one-rank
kernel (`oneRankReduce<FuncPreMulSum<bf16>>`, a local HBM-bound pass at 0.3-0.6 TB/s) where a multi-rank run would run a link-bound
ring kernel, so the numbers bound the SCHEDULE overhead of the step, not the 8-GPU step time (DESIGN.md section 6).

## Step time (`tools/run_ddp1.sh`, bench.py --steps 30 --warmup 5, one box, in this order)

```
{summ}
```

"forced ddp" = a one-rank group without collectives (the torchrun process: its streams land on other hardware queues than in the
plain process -- equal to the plain step since the auxiliary streams are probed); "forced collectives" = every bucket through RCCL.
Default path = engine-issued collectives on the engine's own, probed communication stream (`modules/train.py::make_allreduce`),
weight buckets updated under backward, the last bucket exchanged in two chunks of its weight-gradient GEMM.

## Stream -> hardware queue map of the default path (`tools/run_ddp_trace.sh final` + `tools/queue_map.py`, rocprofv3 --kernel-trace)

```
{queues}
```

Stream 0 = the engine (main) stream; the streams with only `probe_nop_kernel` launches are candidates the probe rejected because they
shared a queue with a stream they must avoid (or probes of kept streams).  Kept: second lane, side (weight gradients), the
communication stream (pack + collectives) and the optimizer stream on the communication stream's queue.

## Kernels >= 150 us of one step of the default path (queue, start, duration)

```
{nl.join(big)}
```
"""
open(f"{R}/profiles/{tag}_ddp_rehearsal.md", "w").write(doc)
print(f"wrote profiles/{tag}_ddp_rehearsal.md ({len(doc.splitlines())} lines)")
