"""profiles/<TAG>_pmc_sweep.json from the three counter passes of tools/pmc_sweep.sh: per sweep kernel the fp64-MFMA pipe
utilisation (SQ_VALU_MFMA_BUSY_CYCLES over all SIMD cycles), the clock the chip held, and the HBM bytes per launch
(2 x FETCH_SIZE + WRITE_SIZE, counter unit KiB, gfx950 correction of MI355X_MICROARCH.md's HBM section) against the
algorithmic bytes of SURVEY 8(d).  Usage: python tools/pmc_sweep_summary.py [TAG] [dir]"""
import csv, json, os, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
d = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out"
csv.field_size_limit(1 << 30)
SHAPES = {"benzene_gga_def2svp": (143556, 114, 21, 4), "anthracene_b3lyp_def2tzvp": (294868, 494, 47, 4)}
KERNELS = {  # pattern -> (label, workload, role)
    "k_rho_ws<": ("k_rho_ws", "benzene_gga_def2svp", "rho"), "k_vxc_ws<": ("k_vxc_ws", "benzene_gga_def2svp", "vxc"),
    "k_rho_occ_rs<": ("k_rho_occ_rs", "benzene_gga_def2svp", "rho_occ"),
    "k_rho_big64<": ("k_rho_big64", "anthracene_b3lyp_def2tzvp", "rho"), "k_vxc_big<": ("k_vxc_big", "anthracene_b3lyp_def2tzvp", "vxc"),
    "k_rho_occ<": ("k_rho_occ", "anthracene_b3lyp_def2tzvp", "rho_occ")}


def label(name):
    for pat, v in KERNELS.items():
        if "qcdft::" + pat in name:
            return v
    return None


def collect(path):
    acc, dur, seen = defaultdict(lambda: defaultdict(list)), defaultdict(list), set()
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            lab = label(r["Kernel_Name"])
            if lab is None:
                continue
            acc[lab][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if (r["Dispatch_Id"]) not in seen:
                seen.add(r["Dispatch_Id"]); dur[lab].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return acc, dur


def alg(role, shape):
    ngrid, nao, nocc, c = shape
    planes = 8.0 * ngrid * nao * c
    if role in ("rho", "rho_occ"):
        flops = 2.0 * ngrid * nao * nao + 2.0 * c * ngrid * nao if role == "rho" else 4.0 * ngrid * nao * nocc + 2.0 * (c - 1) * ngrid * nao
        return planes + 8.0 * nao * nao + 8.0 * ngrid * c, flops
    return planes + 8.0 * ngrid * c + 8.0 * nao * nao, 2.0 * ngrid * nao * nao + 2.0 * c * ngrid * nao


mf, mdur = collect(os.path.join(d, f"{tag}_pmc_sweep_mfma_counter_collection.csv"))
fe, _ = collect(os.path.join(d, f"{tag}_pmc_sweep_fetch_counter_collection.csv"))
wr, _ = collect(os.path.join(d, f"{tag}_pmc_sweep_write_counter_collection.csv"))
mean = lambda v: sum(v) / len(v)
out = {"_how": "tools/pmc_sweep.sh: rocprofv3 --pmc <set> --kernel-trace -- python3 tools/pmc_sweep_driver.py, three separate passes "
               "(SQ/GRBM counters; FETCH_SIZE; WRITE_SIZE), no other trace domain. GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles = "
               "GUI_ACTIVE/8; mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs); hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE "
               "(KiB; FETCH_SIZE counts half the bytes of 16-byte-per-lane reads on gfx950, MI355X_MICROARCH.md HBM section). Profiled "
               "passes run at a lower clock than unprofiled ones (same guide): durations here are the counter pass's own.",
       "round": 3, "workloads": {}}
for lab in sorted(set(mf) | set(fe)):
    name, wl, role = lab
    b_alg, f_alg = alg(role, SHAPES[wl])
    e = {"role": role}
    if lab in mf:
        c = mf[lab]
        cyc = mean(c["GRBM_GUI_ACTIVE"]) / 8.0
        us = mean(mdur[lab])
        e.update({"calls": len(mdur[lab]), "avg_duration_us": us, "cycles": cyc, "clock_ghz": cyc / us / 1e3,
                  "mfma_utilisation": mean(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (cyc * 1024),
                  "wave_cycles_share": {k: mean(c[k]) / mean(c["SQ_WAVE_CYCLES"]) for k in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU") if k in c},
                  "alg_flops": f_alg, "tflops_in_this_pass": f_alg / us / 1e6})
    if lab in fe and lab in wr:
        hb = 2.0 * mean(fe[lab]["FETCH_SIZE"]) * 1024 + mean(wr[lab]["WRITE_SIZE"]) * 1024
        e.update({"FETCH_SIZE_KiB": mean(fe[lab]["FETCH_SIZE"]), "WRITE_SIZE_KiB": mean(wr[lab]["WRITE_SIZE"]), "hbm_bytes": hb,
                  "alg_bytes": b_alg, "traffic_over_algorithmic": hb / b_alg})
    out["workloads"].setdefault(wl, {})[name] = e
json.dump(out, open(os.path.join(d, f"{tag}_pmc_sweep.json"), "w"), indent=1)
for wl, ks in out["workloads"].items():
    for k, e in ks.items():
        print(f"{wl:28s} {k:14s} mfma {100 * e.get('mfma_utilisation', float('nan')):5.1f} %  clock {e.get('clock_ghz', float('nan')):.2f} GHz  "
              f"{e.get('avg_duration_us', float('nan')):8.1f} us  traffic/alg {e.get('traffic_over_algorithmic', float('nan')):.2f}")
