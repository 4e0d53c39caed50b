"""Context for the roofline: what a library GEMM sustains on this socket (torch.matmul -> hipBLASLt/rocBLAS, fp16 and bf16,
random operands, ~3 s each) with rocm-smi sampled next to it.  Compare with tools/ubench/mfma_power.hip."""
import json, subprocess, threading, time
import torch

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
for dtype in (torch.float16, torch.bfloat16):
    for n in (4096, 8192):
        a = (torch.randn(n, n, device="cuda") * 0.05).to(dtype)
        b = (torch.randn(n, n, device="cuda") * 0.05).to(dtype)
        for _ in range(5):
            a @ b
        torch.cuda.synchronize()
        t0 = time.time(); it = 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        while time.time() - t0 < 3.0:
            for _ in range(20):
                a @ b
            it += 20
            torch.cuda.synchronize()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tf = 2.0 * n ** 3 * it / (ms * 1e-3) / 1e12
        recent = [(s[1], s[2]) for s in samples if s[0] > t0 + 1.0][-3:]
        print(f"{str(dtype):16s} {n}^3: {tf:7.1f} TFLOP/s sustained over {ms / 1e3:.1f} s   rocm-smi {recent}", flush=True)
stop = True
