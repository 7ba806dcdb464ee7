"""The hand-over legs of bench.py at cfg 2 on their own: resident forward on the prepared graph, collate_to_device(csr=True)
per batch on this thread, data.DeviceLoader (native worker thread).  usage: python tools/exp/loader_rate.py [reps=5]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import yolat_vectorgraphicsrecognition_amd as yv
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
item, _, optkw, _ = yv.config("2")
for k in ("roots",):
    if hasattr(item, k):
        delattr(item, k)
model = yv.SparseCADGCN(yv.Opt(**optkw)).cuda().eval()


def med(f, n):
    f(40)
    r = []
    for _ in range(reps):
        t = time.perf_counter()
        f(n)
        r.append(n / (time.perf_counter() - t))
    return sorted(r)[len(r) // 2]


with torch.no_grad():
    b, sl = yv.collate_to_device([item], csr=True)

    def resident(n):
        for _ in range(n):
            model(b, sl)
        torch.cuda.synchronize()

    def sync_handover(n):
        for _ in range(n):
            bb, ss = yv.collate_to_device([item], csr=True)
            model(bb, ss)
        torch.cuda.synchronize()

    def loader(n):
        ld = yv.DeviceLoader(([item] for _ in range(n)), slots=3)
        for bb, ss in ld:
            model(bb, ss)
        torch.cuda.synchronize()
        ld.close()

    r = med(resident, 400)
    s = med(sync_handover, 200)
    l = med(loader, 400)
    print("resident %.0f graphs/s | collate_to_device per batch %.0f (%.2f) | DeviceLoader %.0f (%.2f of resident)" % (r, s, s / r, l, l / r))

# split of one loader loop: host time inside next(loader) and inside model(...)
with torch.no_grad():
    n = 1000
    ld = yv.DeviceLoader(([item] for _ in range(n)), slots=int(os.environ.get("SLOTS", "3")))
    tn = tf = 0.0
    t_all = time.perf_counter()
    it = iter(ld)
    while True:
        t0 = time.perf_counter()
        try:
            bb, ss = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        model(bb, ss)
        t2 = time.perf_counter()
        tn += t1 - t0
        tf += t2 - t1
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_tot = time.perf_counter() - t_all
    ld.close()
    print("loop of %d: next %.1f us, forward enqueue %.1f us per batch; enqueue done after %.1f ms, GPU drained after %.1f ms" % (n, tn / n * 1e6, tf / n * 1e6, t_enq * 1e3, t_tot * 1e3))

# where next() spends its time: the three native calls timed through a proxy
class _Proxy(object):
    def __init__(self, lib):
        self._lib, self.t, self.n = lib, {}, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)

        def timed(*a):
            t0 = time.perf_counter()
            r = fn(*a)
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
            self.n[name] = self.n.get(name, 0) + 1
            return r
        return timed


with torch.no_grad():
    n = 1000
    ld = yv.DeviceLoader(([item] for _ in range(n)), slots=3)
    px = ld._lib = _Proxy(ld._lib)
    t0 = time.perf_counter()
    for bb, ss in ld:
        model(bb, ss)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("per batch %.1f us; native calls inside next(): %s" % (dt / n * 1e6, ", ".join("%s %.1f us" % (k, v / n * 1e6) for k, v in px.t.items())))
    ld._lib = px._lib
    ld.close()

# the loader feeding k streams round-robin (forwards of consecutive batches overlap on the GPU)
with torch.no_grad():
    for k in (1, 2, 3, 4):
        streams = [torch.cuda.Stream() for _ in range(k)]
        for s in streams:                      # one plan per stream: warm them
            torch.cuda.set_stream(s)
            model(b, sl)
        torch.cuda.synchronize()

        def loader_k(n):
            ld = yv.DeviceLoader(([item] for _ in range(n)), slots=k + 3)
            i = 0
            for bb, ss in ld:
                model(bb, ss)
                i += 1
                torch.cuda.set_stream(streams[i % k])
            torch.cuda.synchronize()
            ld.close()
        print("DeviceLoader over %d stream(s): %.0f graphs/s" % (k, med(loader_k, 600)))
        torch.cuda.set_stream(torch.cuda.default_stream())
