# -*- coding: utf-8 -*-
"""What the host copies of train_words cost on this box: 2 000 templates (62 MB) concatenated into a kept / fresh buffer,
single-threaded and over sublists in threads, and the gather of the regrouped frames into a fresh / kept output."""
import threading
import time

import numpy as np

rng = np.random.default_rng(0)
lens = rng.integers(50, 150, size=2000)
N, D = int(lens.sum()), 39
off = np.concatenate([[0], np.cumsum(lens)])
sep = [rng.normal(size=(int(n), D)) for n in lens]
out = np.empty((N, D))


def best(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append((time.perf_counter() - t0) * 1e3)
    return "%.2f ms (median %.2f)" % (min(ts), sorted(ts)[len(ts) // 2])


print("concatenate, fresh output:", best(lambda: np.concatenate(sep)))
print("concatenate, kept output:", best(lambda: np.concatenate(sep, out=out)))
for nt in (2, 4, 8):
    cuts = np.linspace(0, len(sep), nt + 1).astype(int)

    def run():
        th = [threading.Thread(target=lambda a, b: np.concatenate(sep[a:b], out=out[off[a]:off[b]]), args=(cuts[i], cuts[i + 1])) for i in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
    print("concatenate, kept output, %d threads over sublists:" % nt, best(run))
big = out.copy()
order = rng.permutation(N)
o = np.empty((N, D))
for nt in (1, 2, 4, 8, 16):
    for fresh in (True, False):
        cuts = np.linspace(0, N, nt + 1).astype(int)

        def run():
            dst = np.empty((N, D)) if fresh else o
            th = [threading.Thread(target=np.take, args=(big, order[cuts[i]:cuts[i + 1]]), kwargs=dict(axis=0, out=dst[cuts[i]:cuts[i + 1]])) for i in range(nt)]
            [t.start() for t in th]
            [t.join() for t in th]
        print("gather, %s output, %d threads:" % ("fresh" if fresh else "kept", nt), best(run))
print("memcpy 62 MB kept -> kept:", best(lambda: np.copyto(o, big)))
print("fresh 62 MB + fill:", best(lambda: np.empty((N, D)).fill(0.0)))
# fresh output on transparent huge pages asked for explicitly
import mmap
print("numpy madvise hugepage:", np._core.multiarray._get_madvise_hugepage(),
      "| THP enabled:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(),
      "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())


def huge_empty(shape):
    nbytes = int(np.prod(shape)) * 8
    m = mmap.mmap(-1, (nbytes + (2 << 20) - 1) & ~((2 << 20) - 1), flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS)
    m.madvise(mmap.MADV_HUGEPAGE)
    return np.frombuffer(m, dtype=np.float64, count=nbytes // 8).reshape(shape)


print("fresh 62 MB (mmap + MADV_HUGEPAGE) + fill:", best(lambda: huge_empty((N, D)).fill(0.0)))
for nt in (4, 8):
    cuts = np.linspace(0, N, nt + 1).astype(int)

    def run():
        dst = huge_empty((N, D))
        th = [threading.Thread(target=np.take, args=(big, order[cuts[i]:cuts[i + 1]]), kwargs=dict(axis=0, out=dst[cuts[i]:cuts[i + 1]])) for i in range(nt)]
        [t.start() for t in th]
        [t.join() for t in th]
    print("gather, fresh huge-page output, %d threads:" % nt, best(run))
print(open("/proc/meminfo").read().split("AnonHugePages")[1].split("\n")[0])
