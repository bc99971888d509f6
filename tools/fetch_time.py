"""Dev timing: what it costs a consumer to get the result rows of a C2 batch (721 k chunk rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
import makisu_amd  # noqa: E402

for flags, name in ((0, "on demand"), (makisu_amd.FLAG_PREFETCH_ROWS, "MI_FLAG_PREFETCH_ROWS")):
    with makisu_amd.Engine(flags=flags) as e:
        with e.batch() as b:
            b.add_synthetic([65536] * 100000, list(range(100000)))
            b.run()
            b.chunks_view()
            for rep in range(3):
                t0 = time.perf_counter(); b.rerun(); t1 = time.perf_counter()
                v = b.chunks_view(); t2 = time.perf_counter()
                ch = b.chunks(); t3 = time.perf_counter()
                fl = b.files(); t4 = time.perf_counter()
                print("%-22s rerun %.2f ms | chunks_view() %.2f ms | chunks() copy %.2f ms (%d rows) | files() %.2f ms"
                      % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(ch), (t4 - t3) * 1e3))
