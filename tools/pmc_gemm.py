"""PMC view of the 16-bit GEMM at the benchmark's shapes (GPU box only):   python tools/pmc_gemm.py [out.json]
                                                                         python tools/pmc_gemm.py mx8 [out.json]     (round 6: gemm_mx8_kernel, fresh)

One rocprofv3 pass per shape with the SQ counters that fit together (MI355X_MICROARCH.md, PMC slots): SQ_VALU_MFMA_BUSY_CYCLES,
GRBM_GUI_ACTIVE, SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY, SQ_WAIT_ANY, SQ_ACTIVE_INST_ANY, SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE — for the product
kernel (variant 25 = gemm_kernel_v12, ring K loop, persistent walk) on the GEGLU projection 65536 x 10240 x 1280, the FF-out projection 65536 x 1280 x 5120 and
8192^3.  MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128) (8 XCDs count GUI_ACTIVE each; 1024 SIMDs; a 32x32x16 MFMA is 32 busy
cycles); the SQ_WAIT_* / SQ_ACTIVE_* counters are quad-cycles summed over waves, reported as shares of SQ_WAVE_CYCLES.  Counter runs clock
lower than un-profiled ones: ratios, not times, are the result."""
import csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ = ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]


def one_pass(counters, cmd, match):
    d = tempfile.mkdtemp(prefix="pmc_", dir=os.environ.get("TMPDIR", "/tmp"))
    subprocess.run(["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "--output-format", "csv", "--"] + cmd,
                   check=True, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    per, names = {}, set()
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if match in row["Kernel_Name"]:
                    per.setdefault(int(row["Dispatch_Id"]), {}).setdefault(row["Counter_Name"], 0.0)
                    per[int(row["Dispatch_Id"])][row["Counter_Name"]] += float(row["Counter_Value"])
                    names.add(row["Kernel_Name"][:120])
    return [per[k] for k in sorted(per)], sorted(names)


def main():
    argv = sys.argv[1:]
    mx8 = bool(argv) and argv[0] == "mx8"
    if mx8:
        argv = argv[1:]
    out = argv[0] if argv else os.path.join(ROOT, "profiles", "r06_pmc_mx8.json" if mx8 else "r06_pmc_gemm.json")
    rec = {"method": __doc__, "shapes": {}}
    # gemm_mx8_kernel: v_mfma_scale_f32_32x32x64_f8f6f4 is 64 busy cycles per instruction and 2 x 32 x 32 x 64 FLOP
    kname, busy, kdepth = ("gemm_mx8_kernel", 64.0, 64) if mx8 else ("gemm_kernel_v12", 32.0, 16)
    if mx8:
        rec["method"] += ("\nmx8 mode: tools/mx8_one.py; SQ_VALU_MFMA_BUSY_CYCLES = 64 per v_mfma_scale_f32_32x32x64_f8f6f4, so MfmaUtil is against the fp8 "
                          "matrix peak (5 PFLOP/s at 2.4 GHz).")
    for M, N, K, geglu in ((65536, 10240, 1280, True), (65536, 1280, 5120, False), (8192, 8192, 8192, False)):
        cmd = ([sys.executable, "tools/mx8_one.py", str(M), str(N), str(K), "4"] if mx8 else
               [sys.executable, "tools/gemm_one.py", str(M), str(N), str(K), "25", "4"]) + (["geglu"] if geglu else [])
        d, names = one_pass(SQ, cmd, kname)
        if not d:
            raise SystemExit(f"no dispatch of {kname} in the counter collection")
        c = d[-1]
        wc = c["SQ_WAVE_CYCLES"]
        rec["shapes"][f"{M}x{N}x{K}" + (" geglu" if geglu else "")] = {
            "kernel": names, "counters_last_dispatch": c,
            "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0),
            "mfma_instructions": c["SQ_VALU_MFMA_BUSY_CYCLES"] / busy, "algorithmic_mfma_instructions": 2.0 * M * N * K / (2.0 * 32 * 32 * kdepth),
            "wait_inst_any_share": c["SQ_WAIT_INST_ANY"] / wc, "wait_any_share": c["SQ_WAIT_ANY"] / wc, "active_inst_any_share": c["SQ_ACTIVE_INST_ANY"] / wc,
            "lds_bank_conflict_share_of_lds_cycles": c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"])}
    with open(out, "w") as fh:
        json.dump(rec, fh, indent=1)
    print(json.dumps({k: {kk: round(vv, 4) for kk, vv in v.items() if isinstance(vv, float)} for k, v in rec["shapes"].items()}))


if __name__ == "__main__":
    main()
