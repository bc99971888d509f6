"""Round 4 probe: what does a memory-bound co-runner cost the hashing pass?  C2 steps one batch at a time while a side
stream of the same process copies device memory to device memory in a loop (a blit kernel: loads and stores, next to no
VALU work).  If the hashing pass stretches, its own loads are what another batch's marking slows down."""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import makisu_amd  # noqa: E402

hip = C.CDLL("libamdhip64.so.7")


def chk(rc, what):
    if rc != 0:
        raise RuntimeError("%s -> %d" % (what, rc))


def main():
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    n = int(gib * (1 << 30))
    with makisu_amd.Engine() as e:
        b = e.batch()
        b.add_synthetic([65536] * 100000, list(range(100000)))
        b.run()
        src, dst, stream = C.c_void_p(), C.c_void_p(), C.c_void_p()
        chk(hip.hipMalloc(C.byref(src), C.c_size_t(n)), "hipMalloc")
        chk(hip.hipMalloc(C.byref(dst), C.c_size_t(n)), "hipMalloc")
        chk(hip.hipMemset(src, 1, C.c_size_t(n)), "hipMemset")
        chk(hip.hipStreamCreateWithFlags(C.byref(stream), 1), "hipStreamCreate")
        stop, copies = threading.Event(), [0]

        def copier():
            chk(hip.hipSetDevice(0), "hipSetDevice")
            while not stop.is_set():
                for _ in range(4):
                    chk(hip.hipMemcpyAsync(dst, src, C.c_size_t(n), 3, stream), "hipMemcpyAsync")
                chk(hip.hipStreamSynchronize(stream), "sync")
                copies[0] += 4

        def steps(label, k=24):
            c0, t0 = copies[0], time.perf_counter()
            sha, cdc = [], []
            for _ in range(k):
                b.submit()
                b.wait()
                st = e.stats()
                sha.append(st["ms_sha_chunks"])
                cdc.append(st["ms_cdc"])
            dt = time.perf_counter() - t0
            print("%-28s %.3f ms/step | marking %.3f (min %.3f) hashing %.3f (min %.3f) | copy %.2f TB/s read+write" % (
                label, dt / k * 1e3, sum(cdc) / k, min(cdc), sum(sha) / k, min(sha), 2 * (copies[0] - c0) * n / dt / 1e12))

        steps("alone")
        t = threading.Thread(target=copier)
        t.start()
        time.sleep(0.05)
        steps("beside a device copy")
        steps("beside a device copy")
        stop.set()
        t.join()
        steps("alone again")
        b.free()


if __name__ == "__main__":
    main()
