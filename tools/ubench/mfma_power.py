"""Run tools/ubench/mfma_power (built to /tmp on the GPU box) with zero and with random operands while sampling rocm-smi."""
import json, subprocess, sys, threading, time
samples, stop = [], False
def sampler():
    while not stop:
        try:
            d = json.loads(subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout)["card0"]
            samples.append((time.time(), d.get("Current Socket Graphics Package Power (W)"), d.get("sclk clock speed:")))
        except Exception as e:   # noqa: BLE001
            samples.append((time.time(), "err", str(e)))
        time.sleep(0.3)
th = threading.Thread(target=sampler, daemon=True); th.start()
runs = [["0"], ["1"]] + [["1"] + a for a in sys.argv[1:] and [x.split(",") for x in sys.argv[1:]]]
for mode in runs:
    t0 = time.time()
    p = subprocess.run(["/tmp/mfma_power"] + mode, capture_output=True, text=True)
    t1 = time.time()
    print(p.stdout.strip().splitlines()[-1])
    print("   rocm-smi:", [(s[1], s[2]) for s in samples if t0 + 1.0 < s[0] < t1][-6:])
stop = True
