"""Power / clock while the hot path runs: is the sustained clock set by the power cap?  (lab tool)
Samples amdsmi / sysfs in a thread while the main thread loops a workload."""
import glob, os, sys, threading, time
import torch

def find_sensors():
    out = {}
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        for name in ("power1_average", "power1_input", "power1_cap", "freq1_input", "temp1_input", "temp2_input", "temp3_input"):
            p = os.path.join(h, name)
            if os.path.exists(p):
                out.setdefault(h, {})[name] = p
    return out

def read(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None

class Sampler(threading.Thread):
    def __init__(self, files, period=0.02):
        super().__init__(daemon=True)
        self.files, self.period, self.stop_, self.rows = files, period, False, []
    def run(self):
        while not self.stop_:
            self.rows.append({k: read(p) for k, p in self.files.items()})
            time.sleep(self.period)

def summarize(tag, rows):
    keys = rows[0].keys() if rows else []
    s = {}
    for k in keys:
        v = [r[k] for r in rows if r[k] is not None]
        if v:
            s[k] = (sum(v) / len(v), max(v))
    # the busiest card only
    pw = {k: v for k, v in s.items() if k.endswith("power1_input")}
    best = max(pw, key=lambda k: pw[k][0]).split(":")[0] if pw else None
    print(tag, {k: (round(a / 1e6, 1), round(m / 1e6, 1)) for k, (a, m) in s.items() if k.startswith(str(best))}, "samples", len(rows), flush=True)

def main():
    sens = find_sensors()
    print({h: list(v) for h, v in sens.items()})
    if not sens:
        return
    # every card's power and clock (the box shows all eight; the busy one is ours)
    files = {}
    for h in sorted(sens):
        tag = h.split("/")[4]
        for k in ("power1_input", "freq1_input"):
            if k in sens[h]:
                files[tag + ":" + k] = sens[h][k]
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from cimrgp_amd import device as dev
    import numpy as np
    def loop(tag, fn, seconds=3.0):
        fn(); torch.cuda.synchronize()
        s = Sampler(files); s.start()
        t0 = time.time(); it = 0
        while time.time() - t0 < seconds:
            for _ in range(10): fn()
            torch.cuda.synchronize(); it += 10
        el = time.time() - t0
        s.stop_ = True; s.join()
        summarize(tag + " (%.3f ms/iter)" % (1e3 * el / it), s.rows)
    s = Sampler(files); s.start(); time.sleep(1.0); s.stop_ = True; s.join(); summarize("idle", s.rows)
    n = 8192
    x = torch.linspace(0, 1, n, dtype=torch.float64, device="cuda").reshape(n, 1)
    k0 = dev.rbf_gram(x, 0.1, 1.0, 0.01, lower_only=True)
    kb = k0.clone()
    def potrf():
        kb.copy_(k0); dev.potrf(kb, n)
    loop("potrf n=8192", potrf)
    a = torch.randn(7936, 256, dtype=torch.float64, device="cuda")
    c = torch.randn(7936, 7936, dtype=torch.float64, device="cuda")
    loop("update M=7936 K=256", lambda: dev.syrk_lower(c, a, 7936, 256))
    b = torch.randn(4096, 4096, dtype=torch.float64, device="cuda")
    def tmm():
        torch.mm(b, b)
    loop("torch.mm f64 4096 (rocBLAS)", tmm)
    for kk in (256, 512, 1024):
        a2 = torch.randn(7936, kk, dtype=torch.float64, device="cuda")
        loop("rocBLAS addmm C(7936x7936) -= A A^T, K=%d (%.1f Gflop)" % (kk, 2 * 7936 * 7936 * kk / 1e9), lambda: torch.addmm(c, a2, a2.t(), alpha=-1.0, out=c))
        loop("cimrgp syrk_lower M=7936 K=%d (%.1f Gflop)" % (kk, 7936 * 7936 * kk / 1e9), lambda: dev.syrk_lower(c, a2, 7936, kk))
    bf = torch.randn(8192, 8192, dtype=torch.bfloat16, device="cuda")
    loop("torch.mm bf16 8192 (hipBLASLt)", lambda: torch.mm(bf, bf))

main()
