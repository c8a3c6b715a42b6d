"""Memory-side traffic of the benchmark's largest GEMM launch, from rocprofv3 PMC counters (GPU box only).

    python tools/pmc_traffic.py fp16 [out.json] [variant word]     (gemm_kernel_v12 — the 256x256 tile with the ring K loop, persistent walk —, the shape
                                                                    bench.py's roofline names; variant word 25 | 64 << 8 = walk groups of 4 row tiles, 25 | 128 << 8 = 2)
    python tools/pmc_traffic.py fp8  [out.json]        (gemm_mx8_kernel, same shape, GEGLU epilogue)

Runs `rocprofv3 --kernel-trace --pmc FETCH_SIZE` and, in a SEPARATE pass, `--pmc WRITE_SIZE` (the two do not fit one pass:
MI355X_MICROARCH.md, PMC slots) around tools/gemm_one.py / tools/mx8_one.py, keeps the dispatches of the GEMM kernel, and writes
the per-launch bytes with the guide's gfx950 correction (FETCH_SIZE counts 64 B per 128-B request: doubled).  The launch is
the GEGLU projection of the 1280-wide transformer blocks at 8 requests per step: M 65536, N 10240, K 1280, GEGLU epilogue
(output 65536 x 5120), i.e. what `bench.py` runs, not a plain-epilogue stand-in.
bench.py reads profiles/r06_pmc_traffic_{fp16,fp8}.json (falling back to r05, r04, r03) for `roofline.traffic`.
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
M, N, K = 65536, 10240, 1280


def one_pass(counter, cmd, match):
    d = tempfile.mkdtemp(prefix="pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--"] + cmd,
                   check=True, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    vals, names = {}, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if match in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    vals[int(row["Dispatch_Id"])] = vals.get(int(row["Dispatch_Id"]), 0.0) + float(row["Counter_Value"])
                    names.add(row["Kernel_Name"][:160])
    return [vals[k] for k in sorted(vals)], sorted(names)


def main():
    mode = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", f"r06_pmc_traffic_{mode}.json")
    word = sys.argv[3] if len(sys.argv) > 3 else "0"
    it = 4
    if mode == "fp16":
        cmd, match = [sys.executable, "tools/gemm_one.py", str(M), str(N), str(K), word, str(it), "geglu"], "gemm_kernel_v12"
        a_bytes, w_bytes = M * K * 2, N * K * 2
    else:
        cmd, match = [sys.executable, "tools/mx8_one.py", str(M), str(N), str(K), str(it), "geglu"], "gemm_mx8_kernel"
        a_bytes, w_bytes = M * K + M * (K // 32), N * K + N * (K // 32)          # e4m3 bytes + one E8M0 scale byte per 32
    fetch, names = one_pass("FETCH_SIZE", cmd, match)
    write, _ = one_pass("WRITE_SIZE", cmd, match)
    if not fetch or not write:
        raise SystemExit(f"no dispatch of a kernel matching {match!r} found in the counter collection")
    # steady state: drop the first launch (cold L2 / Infinity Cache)
    f_kb = sum(fetch[1:]) / max(1, len(fetch) - 1)
    w_kb = sum(write[1:]) / max(1, len(write) - 1)
    c_bytes = M * (N // 2) * 2
    rec = {"kernel": names, "shape": f"M {M}, N {N}, K {K}, GEGLU epilogue (C = {M} x {N // 2})", "command": " ".join(cmd[1:]),
           "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE; counter values are KB per dispatch; "
                     "FETCH_SIZE doubled for gfx950 (MI355X_MICROARCH.md, HBM section: wide coalesced reads are tallied at half); WRITE_SIZE "
                     "uncalibrated; first launch dropped",
           "fetch_size_kb_per_launch": fetch, "write_size_kb_per_launch": write,
           "read_bytes_per_launch_corrected": 2.0 * f_kb * 1024.0, "write_bytes_per_launch": w_kb * 1024.0,
           "algorithmic_read_bytes": a_bytes + w_bytes, "algorithmic_write_bytes": c_bytes,
           "flop_per_launch": 2.0 * M * N * K}
    rec["traffic_bytes_per_launch"] = rec["read_bytes_per_launch_corrected"] + rec["write_bytes_per_launch"]
    rec["traffic_over_algorithmic"] = rec["traffic_bytes_per_launch"] / (a_bytes + w_bytes + c_bytes)
    rec["note"] = ("counted at the L2's memory side (fabric), Infinity-Cache hits included: reads above the algorithmic bytes are L2 re-fetches of "
                   "A / W panels shared by the tiles of an XCD, served by the 256 MB Infinity Cache (A + W = %.0f MB fit it), not HBM traffic"
                   % ((a_bytes + w_bytes) / 1e6))
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps({k: rec[k] for k in ("shape", "read_bytes_per_launch_corrected", "write_bytes_per_launch", "traffic_over_algorithmic")}))


if __name__ == "__main__":
    main()
