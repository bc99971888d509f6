"""Dev tool: host-side time of every step of makisu_amd.distributed.global_dedup on one rank
(world_size 1 over RCCL), while other batches keep the GPU busy -- shows where the exchange's
per-step cost goes (collectives are trivial at world 1; syncs, copies and launches are not)."""
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402
from makisu_amd import distributed as mdist  # noqa: E402


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    eng = makisu_amd.Engine(flags=makisu_amd.FLAG_NO_DEDUP)
    inflight, files, steps = 3, 100000, 24
    batches = []
    for i in range(inflight):
        b = eng.batch()
        b.add_synthetic([65536] * files, list(range(i * files, (i + 1) * files)))
        b.run()
        batches.append(b)
    acc = {}

    def tick(name, t0):
        t1 = time.perf_counter()
        acc[name] = acc.get(name, 0.0) + (t1 - t0)
        return t1

    def exchange(b):
        t = time.perf_counter()
        local = mdist.digests_tensor(b, dev)
        t = tick("view", t)
        counts = mdist.all_gather_counts(local.shape[0])
        t = tick("counts on the host group", t)
        st = mdist._exchange_stream(dev)
        m = max(counts)
        with torch.cuda.stream(st):
            slab = torch.zeros((m, 32), dtype=torch.uint8, device=dev)
            slab[: local.shape[0]] = local
            gathered = torch.empty((m, 32), dtype=torch.uint8, device=dev)
            dist.all_gather_into_tensor(gathered, slab)
            dup = torch.empty(m, dtype=torch.int64, device=dev)
        t = tick("slab + all-gather (enqueue)", t)
        st.synchronize()
        t = tick("exchange stream sync", t)
        nf = eng.dedup_mark_range(gathered.data_ptr(), m, 0, m, dup.data_ptr())
        t = tick("mark_range (sync)", t)
        b.set_global_dedup(dup.data_ptr(), 0)
        t = tick("set_global_dedup", t)
        tt = torch.tensor([nf], dtype=torch.int64)
        dist.all_reduce(tt, group=mdist._host_group(None))
        tt.item()
        t = tick("count all-reduce on the host group", t)

    def loop(with_ex):
        pending = []
        t0 = time.perf_counter()
        for k in range(steps):
            if len(pending) == inflight:
                b = pending.pop(0)
                tw = time.perf_counter()
                b.wait()
                tick("wait", tw)
                if with_ex:
                    exchange(b)
            b = batches[k % inflight]
            ts = time.perf_counter()
            b.submit()
            tick("submit", ts)
            pending.append(b)
        for b in pending:
            b.wait()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    for with_ex in (False, True, False, True):
        acc.clear()
        ms = loop(with_ex)
        print("exchange=%s: %.3f ms/step" % (with_ex, ms))
        for k, v in acc.items():
            print("    %-32s %.3f ms/step" % (k, v / steps * 1e3))
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
