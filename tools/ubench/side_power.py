"""Run tools/ubench/bin/side_power mode by mode while sampling rocm-smi (socket power, sclk): what each stream of the render
kernel's side traffic draws.  Usage on the GPU box: python3 tools/ubench/side_power.py > gpurun_out/side_power.txt"""
import json, os, subprocess, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
samples, stop = [], False
def sampler():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)["card0"]
            samples.append((time.time(), float(d.get("Current Socket Graphics Package Power (W)")), d.get("sclk clock speed:")))
        except Exception as e:   # noqa: BLE001
            pass
        time.sleep(0.25)
threading.Thread(target=sampler, daemon=True).start()
time.sleep(2.0)
idle = [s[1] for s in samples]
print(f"idle before the runs: {sum(idle) / max(len(idle), 1):.0f} W", flush=True)
for mode in (0, 1, 2, 3, 4, 5, 6, 7, 0):
    t0 = time.time()
    p = subprocess.run([os.path.join(HERE, "bin", "side_power"), str(mode), "4"], capture_output=True, text=True)
    t1 = time.time()
    w = [s[1] for s in samples if t0 + 1.5 < s[0] < t1 - 0.3]
    clk = [s[2] for s in samples if t0 + 1.5 < s[0] < t1 - 0.3][-1:]
    print((p.stdout.strip().splitlines() or ["?"])[-1], flush=True)
    print(f"   rocm-smi: {sum(w) / max(len(w), 1):.0f} W mean over {len(w)} samples (min {min(w or [0]):.0f}, max {max(w or [0]):.0f}), sclk {clk}", flush=True)
stop = True
