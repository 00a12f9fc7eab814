import os, re, statistics, subprocess, sys, threading, time
import torch
def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout.strip().splitlines()
    d = dict(zip(out[0].split(","), out[1].split(",")))
    return int(re.sub(r"\D", "", d["sclk clock speed:"])), float([v for k, v in d.items() if "Power" in k][0])
def watched(fn, seconds=5):
    seen, stop = [], []
    def watch():
        while not stop:
            seen.append(sample()); time.sleep(0.2)
    th = threading.Thread(target=watch); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < seconds:
        fn(); torch.cuda.synchronize(); n += 1
    dt = time.time() - t0
    stop.append(1); th.join()
    seen = seen[2:]
    return n / dt, statistics.median(s[0] for s in seen), statistics.median(s[1] for s in seen)
time.sleep(3)
print("idle:", [sample() for _ in range(3)], flush=True)
for mb in (64, 2048):
    a = torch.empty(mb * 1024 * 1024 // 8, dtype=torch.float64, device="cuda"); b = torch.empty_like(a); a.normal_()
    def copy20():
        for _ in range(20): b.copy_(a)
    r, sclk, p = watched(copy20)
    print(f"copy {mb} MB x20: {r * 20 * 2 * mb / 1024 / 1024 * 1.048576:.2f} TB/s read+write, sclk {sclk} MHz, power {p} W", flush=True)
    def fma20():
        for _ in range(20): torch.addcmul(b, a, a, out=b)
    r, sclk, p = watched(fma20)
    print(f"addcmul {mb} MB x20: {r * 20 * 3 * mb / 1024 / 1024 * 1.048576:.2f} TB/s, sclk {sclk} MHz, power {p} W", flush=True)
n = 8192
x = torch.randn(n, n, dtype=torch.float64, device="cuda"); y = torch.randn(n, n, dtype=torch.float64, device="cuda")
r, sclk, p = watched(lambda: torch.mm(x, y))
print(f"fp64 GEMM {n}: {r * 2 * n**3 / 1e12:.1f} TFLOP/s, sclk {sclk} MHz, power {p} W", flush=True)
