import os, time, mmap, numpy as np, tempfile
from concurrent.futures import ThreadPoolExecutor
d = tempfile.mkdtemp(dir="/tmp"); n = 60_000_000
data = np.random.default_rng(0).integers(32, 127, n, dtype=np.uint8)
def pw(th):
    p = os.path.join(d, "a.csv"); fd = os.open(p, os.O_WRONLY|os.O_CREAT|os.O_TRUNC, 0o666)
    def put(k):
        lo, hi = n*k//th, n*(k+1)//th; v = memoryview(data)[lo:hi]
        while len(v): w = os.pwrite(fd, v, lo); v = v[w:]; lo += w
    with ThreadPoolExecutor(th) as pool: list(pool.map(put, range(th)))
    os.close(fd)
def mm(th):
    p = os.path.join(d, "b.csv"); fd = os.open(p, os.O_RDWR|os.O_CREAT|os.O_TRUNC, 0o666); os.ftruncate(fd, n)
    m = mmap.mmap(fd, n); a = np.frombuffer(m, np.uint8)
    def put(k):
        lo, hi = n*k//th, n*(k+1)//th; a[lo:hi] = data[lo:hi]
    with ThreadPoolExecutor(th) as pool: list(pool.map(put, range(th)))
    del a; m.close(); os.close(fd)
def falloc(th):
    p = os.path.join(d, "c.csv"); fd = os.open(p, os.O_WRONLY|os.O_CREAT|os.O_TRUNC, 0o666); os.posix_fallocate(fd, 0, n)
    def put(k):
        lo, hi = n*k//th, n*(k+1)//th; v = memoryview(data)[lo:hi]
        while len(v): w = os.pwrite(fd, v, lo); v = v[w:]; lo += w
    with ThreadPoolExecutor(th) as pool: list(pool.map(put, range(th)))
    os.close(fd)
for rep in range(3):
    for name, f, th in (("pwrite", pw, 1), ("pwrite", pw, 8), ("mmap", mm, 1), ("mmap", mm, 8), ("mmap", mm, 16), ("fallocate+pwrite", falloc, 8)):
        t0 = time.perf_counter(); f(th); print(f"{name} x{th}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
