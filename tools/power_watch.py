"""Clock and socket power beside the kernels (rocm-smi sampled from a thread while the kernel runs back to back): the fused
pass, its data movement alone, the gradient kernel, the flux kernel -- 64^3 (working set inside the Infinity Cache) and 128^3.
What the chip sustains is set by its 1400 W cap, not by its 2400 MHz: see DESIGN 8 / profiles/r05_power_watch.log.
    python tools/power_watch.py [64 128]"""
import os, re, statistics, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
m = load_package()
SECONDS = float(os.environ.get("WATCH_S", "6"))
print(subprocess.run(["rocm-smi", "--showmaxpower", "--showsclkrange"], capture_output=True, text=True, timeout=30).stdout.strip(), flush=True)

def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=20).stdout.strip().splitlines()
    hdr, val = out[0].split(","), out[1].split(",")
    d = dict(zip(hdr, val))
    sclk = int(re.sub(r"\D", "", d["sclk clock speed:"]))
    power = float([v for k, v in d.items() if "Power" in k][0])
    return sclk, power

def watched(run_chunk):
    """run_chunk() -> microseconds per launch of one chunk; repeated for SECONDS with the device sampled beside it"""
    seen, stop = [], []
    def watch():
        while not stop:
            try: seen.append(sample())
            except Exception as e: seen.append((0, 0.0))
            time.sleep(0.2)
    th = threading.Thread(target=watch); th.start()
    t0, us = time.time(), []
    while time.time() - t0 < SECONDS:
        us.append(run_chunk())
    stop.append(1); th.join()
    seen = [s for s in seen[2:] if s[0]] or [(0, 0.0)]  # (the first samples: the clock and the power still ramp)
    return statistics.median(us), statistics.median(s[0] for s in seen), statistics.median(s[1] for s in seen), max(s[1] for s in seen), len(seen)

for n in [int(a) for a in sys.argv[1:]] or [64, 128]:
    dom = m.gen_domain(m.gen_params(n, ndomains=1), 0); m.fill_var(dom, None, m.VAR_HASH)
    part = m.GpuPartition(dom); part.set_fusion(True)
    it = 2000 if n <= 64 else 300
    def launches(fn):
        def chunk():
            part.sync(); t = time.perf_counter()
            for _ in range(it): fn()
            part.sync()
            return (time.perf_counter() - t) / it * 1e6
        return chunk
    modes = [("fused pass (flux + gradients)", lambda: part.time_fused(it) * 1e3),
             ("its data movement alone", lambda: part.time_fused_movement(it) * 1e3),
             ("gradient kernel", launches(part.gradients)),
             ("flux kernel", launches(part.flux))]
    for name, fn in modes:
        fn()
        us, sclk, p_med, p_max, ns = watched(fn)
        print(f"n {n:3d}  {name:32s} {us:8.2f} us per launch   sclk {sclk:5.0f} MHz   socket power {p_med:5.0f} W (max {p_max:5.0f}, {ns} samples)", flush=True)
    part.close(); dom.free()
