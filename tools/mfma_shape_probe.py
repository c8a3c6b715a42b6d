"""tools/ubench/mfma_shape_mix (32x32x16 vs 16x16x32 MFMA on the GEMM's own K-loop instruction mix, random operands) with socket power and shader
clock polled from rocm-smi beside it: which MFMA shape gives more FLOP/s at the part's power / current limit?   python tools/mfma_shape_probe.py [secs]"""
import json, os, subprocess, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
exe = os.path.join(HERE, "ubench", "mfma_shape_mix")
if not os.path.exists(exe):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", exe + ".hip", "-o", exe])
secs = sys.argv[1] if len(sys.argv) > 1 else "3"
samples, stop = [], False
def sampler():
    import re
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "-c", "-P", "--json"], capture_output=True, text=True, timeout=5).stdout
            c = next(iter(json.loads(o).values()))
            row = {}
            for k, v in c.items():
                m = re.search(r"[-+]?\d+(\.\d+)?", str(v))
                if m and "sclk clock speed" in k.lower(): row["sclk"] = float(m.group())
                elif m and "power (w)" in k.lower() and "max" not in k.lower(): row["w"] = float(m.group())
            samples.append((time.time(), row))
        except Exception:      # noqa
            pass
        time.sleep(0.2)
th = threading.Thread(target=sampler, daemon=True); th.start()
p = subprocess.Popen([exe, secs], stdout=subprocess.PIPE, text=True)
t_prev = time.time()
for line in p.stdout:
    now = time.time()
    if line.startswith("PHASE"):
        win = [r for t, r in samples if now - float(secs) * 0.8 <= t <= now - 0.2]      # the steady part of the phase that just ended
        avg = lambda k: sum(r[k] for r in win if k in r) / max(1, sum(1 for r in win if k in r))
        tf = float(line.split("TF/s")[0].split()[-1])
        clk = avg("sclk")
        print(line.rstrip()[6:] + f"   | {avg('w'):6.0f} W  {clk:5.0f} MHz  {tf / clk * 1000 if clk else 0:6.0f} TF/s per GHz  ({len(win)} samples)")
    else:
        print(line.rstrip())
    sys.stdout.flush()
stop = True
