"""PMC view of the two attention kernels at the benchmark's shapes (GPU box only):  python tools/pmc_attention.py [out.json]

  self   attn_fwd_kernel7, (64, 10, 4096, 4096):  SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE (+ SQ wait / active counters) in one pass
  cross  attn_fwd_kernel6, (64, 20, 1024, 77):    the same pass, then FETCH_SIZE and WRITE_SIZE in two further passes (they do not share one)

MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); a 32x32x16 MFMA is 32 busy cycles.  FETCH_SIZE is
doubled for gfx950 (MI355X_MICROARCH.md).  Counter runs clock lower than un-profiled ones: ratios, not times, are the result.
Round 6: the self-attention kernel's softmax denominator is eight 16x16x32 MFMAs per wave and key tile beside the 32 algorithmic 32x32x16 ones; `mfma_instructions`
(= busy cycles / 32) therefore counts 32x32x16 EQUIVALENTS (round 5: 40 per tile, now 32 + 8 x the smaller shape's busy cycles / 32).
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counters, cmd, match):
    d = tempfile.mkdtemp(prefix="pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "--output-format", "csv", "--"] + cmd,
                   check=True, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if match in row["Kernel_Name"]:
                    per.setdefault(int(row["Dispatch_Id"]), {}).setdefault(row["Counter_Name"], 0.0)
                    per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
    return [per[k] for k in sorted(per)]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_pmc_attention.json")
    sq = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_INSTS_LDS"]
    rec = {"method": __doc__}
    for name, shape, match in (("self", (64, 10, 4096, 4096), "attn_fwd_kernel7"), ("cross", (64, 20, 1024, 77), "attn_fwd_kernel6")):
        B, H, Nq, Nkv = shape
        cmd = [sys.executable, "tools/attn_one.py", str(B), str(H), str(Nq), str(Nkv), "0", "4"]
        d = one_pass(sq, cmd, match)
        if not d:
            raise SystemExit(f"no dispatch matching {match}")
        last = d[-1]
        waves = B * H * ((Nq + 255) // 256 if name == "self" else (Nq + 511) // 512) * 4
        tiles = (Nkv + 63) // 64
        r = {"shape": shape, "kernel": match, "last_dispatch": last,
             "mfma_util_all_mfmas": last["SQ_VALU_MFMA_BUSY_CYCLES"] / (last["GRBM_GUI_ACTIVE"] * 128.0),
             "mfma_instructions": last["SQ_VALU_MFMA_BUSY_CYCLES"] / 32.0,
             "algorithmic_mfma_instructions": 4.0 * B * H * Nq * Nkv * 64 / (2.0 * 32 * 32 * 16)}
        r["mfma_util_algorithmic"] = r["mfma_util_all_mfmas"] * r["algorithmic_mfma_instructions"] / r["mfma_instructions"]
        if name == "self":
            r["valu_instructions_per_wave_tile"] = last["SQ_INSTS_VALU"] / (waves * tiles)
        else:
            f = one_pass(["FETCH_SIZE"], cmd, match)
            w = one_pass(["WRITE_SIZE"], cmd, match)
            r["read_bytes_per_launch_corrected"] = 2.0 * 1024.0 * f[-1]["FETCH_SIZE"]
            r["write_bytes_per_launch"] = 1024.0 * w[-1]["WRITE_SIZE"]
            r["algorithmic_bytes"] = {"Q": B * Nq * H * 64 * 2, "O": B * Nq * H * 64 * 2, "K_and_Vt": 2 * B * H * 64 * ((Nkv + 63) // 64 * 64) * 2}
        rec[name] = r
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk.startswith(("mfma_util", "valu_", "read_", "write_"))} for k, v in rec.items() if k != "method"}))


if __name__ == "__main__":
    main()
