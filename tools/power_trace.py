"""Sample rocm-smi (power, clocks, temperature) while the bench frame loop runs; prints a short table.
Diagnostic only: shows whether the render kernel runs power-capped (clock below the 2.4 GHz the roofline peak assumes)."""
import json, subprocess, sys, threading, time

samples = []
stop = False

def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
            d = json.loads(out)
            c = d.get("card0", {})
            samples.append((time.time(), {k: v for k, v in c.items() if any(s in k.lower() for s in ("power", "sclk", "mclk", "junction", "edge"))}))
        except Exception as e:   # noqa: BLE001
            samples.append((time.time(), {"error": str(e)}))
        time.sleep(0.25)

th = threading.Thread(target=sampler, daemon=True)
th.start()
t0 = time.time()
p = subprocess.run([sys.executable, "bench.py", "--steps", "12", "--warmup", "2", "--no-cpu-baseline"], capture_output=True, text=True)
stop = True
th.join(timeout=5)
print(p.stdout[-400:])
for t, s in samples:
    print(f"{t - t0:6.2f}s", s)
